set -e
cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_sdpa_gpu.py tests/test_rms_norm_gpu.py -m gpu -x -q > gpurun_out/t_edge.log 2>&1 || { tail -50 gpurun_out/t_edge.log; exit 1; }
tail -2 gpurun_out/t_edge.log
