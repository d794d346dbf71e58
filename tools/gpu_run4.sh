set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_fastpath_gpu.py tests/test_nm_ties.py tests/test_attn_matmul_gpu.py tests/test_replay_invariance_gpu.py tests/test_gemm_gpu.py -x -q > gpurun_out/t_fast.log 2>&1 || { tail -40 gpurun_out/t_fast.log; exit 1; }
tail -2 gpurun_out/t_fast.log
python tools/refops_probe.py 1 1 2>&1 | tail -2 | cut -c1-120
python tools/refops_probe.py 0 1 2>&1 | tail -2 | cut -c1-120
python tools/tower_times.py 1 8 2>&1 | grep -v amdgpu.ids | head -10
VLMC_FAST=0 python tools/tower_times.py 8 2>&1 | grep -v amdgpu.ids | head -5
