set -e
cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_fastpath_gpu.py tests/test_sdpa_gpu.py tests/test_rms_norm_gpu.py -m gpu -x -q > gpurun_out/t_fast.log 2>&1 || { tail -60 gpurun_out/t_fast.log; exit 1; }
tail -3 gpurun_out/t_fast.log
for f in 0 1; do
echo "== VLMC_FAST=$f"
VLMC_FAST=$f RANK_TIMELINE_ITERS=6 timeout -k 10 300 python tools/rank_timeline.py 8 2>&1 | grep prune_ms | tail -3 | cut -c1-200
done
