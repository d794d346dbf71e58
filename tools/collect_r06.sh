#!/bin/bash
# Round-6 profile collection on the GPU box (run from the repo root): kernel stats of the reference-op / ragged prune (the bench
# headline's workload since round 6) with every kernel listed; summaries into gpurun_out/r06/ -- raw traces stay on the box.
#   tools/collect_r06.sh <tag> [pmc]
set -o pipefail
TAG="${1:-a}"
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/r06
rm -rf /tmp/stats_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/stats_$TAG -- python3 $R/tools/ragged_prof.py 8 > $R/gpurun_out/r06/rag_$TAG.log 2>&1
python3 $R/tools/summarize_rocprof.py stats_all /tmp/stats_$TAG $R/gpurun_out/r06/rag_stats_$TAG.md > /dev/null
if [ "${2:-}" = "pmc" ]; then
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_$c -- python3 $R/tools/ragged_prof.py 1 > $R/gpurun_out/r06/pmc_${c}_$TAG.log 2>&1 || echo "rocprofv3 --pmc $c exited with $?" >> $R/gpurun_out/r06/pmc_${c}_$TAG.log
done
python3 $R/tools/traffic_from_pmc.py /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE $R/gpurun_out/r06/traffic_$TAG.json > /dev/null 2> $R/gpurun_out/r06/traffic_$TAG.err || true
fi
cd $R
