#!/bin/bash
# Round-4 profile collection on the GPU box (run from the repo root): kernel stats + GPU timeline of the bench, summaries into
# gpurun_out/r05/ -- the raw traces stay on the box (gpurun copies back at most 64 MiB).
set -o pipefail
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/r05
BENCH="python3 $R/bench.py --steps 3 --warmup 1 --cpu-seconds 0 --kernel-pass 0 --reference-ops 0"
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/stats_r05 -- $BENCH > $R/gpurun_out/r05/stats_bench.log 2>&1
python3 $R/tools/summarize_rocprof.py stats /tmp/stats_r05 $R/gpurun_out/r05/stats_bench.md > /dev/null
cp $(find /tmp/stats_r05 -name '*kernel_stats.csv' | head -1) $R/gpurun_out/r05/kernel_stats.csv
python3 $R/tools/gpu_timeline.py /tmp/stats_r05 $R/gpurun_out/r05/gpu_timeline.md > /dev/null 2>&1
if [ "${WITH_PMC:-0}" = "1" ]; then
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_$c -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-seconds 0 --kernel-pass 0 --reference-ops 0 > $R/gpurun_out/r05/pmc_$c.log 2>&1 || echo "rocprofv3 --pmc $c exited with $?" >> $R/gpurun_out/r05/pmc_$c.log
done
python3 $R/tools/traffic_from_pmc.py /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE $R/gpurun_out/r05/traffic.json > /dev/null 2> $R/gpurun_out/r05/traffic.err || true
fi
cd $R
ls -la gpurun_out/r05
cd /tmp
cd $R
ls -la gpurun_out/r05
