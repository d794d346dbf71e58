#!/bin/bash
# Round-4 profile collection on the GPU box (run from the repo root): kernel stats + GPU timeline of the bench, summaries into
# gpurun_out/r04/ -- the raw traces stay on the box (gpurun copies back at most 64 MiB).
set -o pipefail
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/r04
BENCH="python3 $R/bench.py --steps 3 --warmup 1 --cpu-seconds 0 --kernel-pass 0 --reference-ops 0"
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/stats_r04 -- $BENCH > $R/gpurun_out/r04/stats_bench.log 2>&1
python3 $R/tools/summarize_rocprof.py stats /tmp/stats_r04 $R/gpurun_out/r04/stats_bench.md > /dev/null
cp $(find /tmp/stats_r04 -name '*kernel_stats.csv' | head -1) $R/gpurun_out/r04/kernel_stats.csv
python3 $R/tools/gpu_timeline.py /tmp/stats_r04 $R/gpurun_out/r04/gpu_timeline.md > /dev/null 2>&1
if [ "${WITH_PMC:-0}" = "1" ]; then
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_$c -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-seconds 0 --kernel-pass 0 --reference-ops 0 > $R/gpurun_out/r04/pmc_$c.log 2>&1 || echo "rocprofv3 --pmc $c exited with $?" >> $R/gpurun_out/r04/pmc_$c.log
done
python3 $R/tools/traffic_from_pmc.py /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE $R/gpurun_out/r04/traffic.json > /dev/null 2> $R/gpurun_out/r04/traffic.err || true
fi
cd $R
ls -la gpurun_out/r04
# round 4: the same stats for the reference-op stand-in with ragged text (the attention products' kernel) and for a SparseGPT 2:4 prune
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/stats_refops -- python3 $R/tools/refops_probe.py 1 1 > $R/gpurun_out/r04/stats_refops.log 2>&1
python3 $R/tools/summarize_rocprof.py stats /tmp/stats_refops $R/gpurun_out/r04/stats_refops.md > /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/stats_refops_eq -- python3 $R/tools/refops_probe.py 0 1 > $R/gpurun_out/r04/stats_refops_equal.log 2>&1
python3 $R/tools/summarize_rocprof.py stats /tmp/stats_refops_eq $R/gpurun_out/r04/stats_refops_equal.md > /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/stats_sgpt -- python3 $R/tools/sgpt_profile.py 2:4 > $R/gpurun_out/r04/stats_sgpt.log 2>&1
python3 $R/tools/summarize_rocprof.py stats /tmp/stats_sgpt $R/gpurun_out/r04/stats_sgpt.md > /dev/null
cd $R
ls -la gpurun_out/r04
