set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_sparsegpt_gpu.py tests/test_sparsegpt_fullsize_gpu.py tests/test_pruner_gpu.py -x -q -k "sparsegpt or sgpt" > gpurun_out/t_sgpt2.log 2>&1 || { tail -60 gpurun_out/t_sgpt2.log; exit 1; }
tail -2 gpurun_out/t_sgpt2.log
timeout -k 10 300 python tools/sgpt_profile.py 2:4 2>&1 | grep -v amdgpu | head -3
VLMC_SGPT_SWEEP_STREAMS=1 timeout -k 10 300 python tools/sgpt_profile.py 2:4 2>&1 | grep -v amdgpu | head -3
timeout -k 10 300 python tools/sgpt_profile.py 2>&1 | grep -v amdgpu | head -3
VLMC_SGPT_SWEEP_STREAMS=1 timeout -k 10 300 python tools/sgpt_profile.py 2>&1 | grep -v amdgpu | head -3
