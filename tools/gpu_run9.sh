set -e
cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/t_all.log 2>&1 || { tail -60 gpurun_out/t_all.log; exit 1; }
tail -2 gpurun_out/t_all.log
for i in 1 2; do
RANK_TIMELINE_ITERS=6 timeout -k 10 300 python tools/rank_timeline.py 8 2>&1 | grep prune_ms | tail -2 | cut -c1-200
done
