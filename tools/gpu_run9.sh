set -e
cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_pruner_gpu.py tests/test_replay_invariance_gpu.py tests/test_multirank_gpu.py tests/test_rms_norm_gpu.py -m gpu -x -q > gpurun_out/t_last.log 2>&1 || { tail -60 gpurun_out/t_last.log; exit 1; }
tail -2 gpurun_out/t_last.log
RANK_TIMELINE_ITERS=7 timeout -k 10 300 python tools/rank_timeline.py 1 2>&1 | grep prune_ms | tail -3 | cut -c1-560
RANK_TIMELINE_ITERS=6 timeout -k 10 300 python tools/rank_timeline.py 8 2>&1 | grep prune_ms | tail -2 | cut -c1-200
