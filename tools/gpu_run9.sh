set -e
cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
for t in 1 2; do
timeout -k 10 300 python tools/rank_timeline.py 1 2>&1 | grep prune_ms | tail -3 | cut -c1-700
VLMC_TOWER_TRACES=$t timeout -k 10 300 python tools/rank_timeline.py 1 2>&1 | grep prune_ms | tail -2 | cut -c1-200
done
timeout -k 10 600 python bench.py --steps 10 --warmup 2 --cpu-seconds 0 2>&1 | tail -1 > gpurun_out/bench_predict.json
