set -e
cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests/test_pruner_gpu.py tests/test_multirank_gpu.py tests/test_replay_invariance_gpu.py -m gpu -x -q > gpurun_out/t_pruner.log 2>&1
