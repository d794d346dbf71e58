#!/bin/bash
# Round-2 profile collection on the GPU box (run from the repo root): kernel stats of the bench, PMC traffic (separate passes),
# summaries into gpurun_out/r02/ -- the raw traces stay on the box (gpurun copies back at most 64 MiB).
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
mkdir -p gpurun_out/r02
BENCH="python3 bench.py --steps 2 --warmup 1 --cpu-seconds 0"
if [ "${SKIP_STATS:-0}" != "1" ]; then
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/stats_r02 -- $BENCH > gpurun_out/r02/stats_bench.log 2>&1
python tools/summarize_rocprof.py stats /tmp/stats_r02 gpurun_out/r02/stats_bench.md > /dev/null
cp $(find /tmp/stats_r02 -name '*kernel_stats.csv' | head -1) gpurun_out/r02/kernel_stats.csv
fi
# (the PMC passes run a reduced command -- one timed prune + the phase-timer prune, no warm-up, no kernel pass: counter
#  collection over the default command's ~150 k dispatches crashed rocprofv3 (SIGSEGV in a tool thread) three times out of four)
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_$c -- python3 bench.py --steps 1 --warmup 0 --cpu-seconds 0 --kernel-pass 0 > gpurun_out/r02/pmc_$c.log 2>&1 || echo "rocprofv3 --pmc $c exited with $?" >> gpurun_out/r02/pmc_$c.log
done
python tools/traffic_from_pmc.py /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE gpurun_out/r02/traffic.json > /dev/null 2> gpurun_out/r02/traffic.err || true
python bench.py > gpurun_out/r02/bench_default.json 2> gpurun_out/r02/bench_default.err
ls -la gpurun_out/r02
