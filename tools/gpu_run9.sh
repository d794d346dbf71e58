set -e
cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/t_all.log 2>&1 || { tail -60 gpurun_out/t_all.log; exit 1; }
tail -2 gpurun_out/t_all.log
