set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests/test_sparsegpt_gpu.py tests/test_sparsegpt_fullsize_gpu.py -x -q > gpurun_out/t_sgpt.log 2>&1 || { tail -60 gpurun_out/t_sgpt.log; exit 1; }
tail -2 gpurun_out/t_sgpt.log
timeout -k 10 300 python tools/sgpt_profile.py 2:4 2>&1 | grep -v amdgpu | head -4
VLMC_SGPT_PERSISTENT=0 timeout -k 10 300 python tools/sgpt_profile.py 2:4 2>&1 | grep -v amdgpu | head -4
