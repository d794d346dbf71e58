set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
true

timeout -k 10 900 python -m pytest tests/test_replay_invariance_gpu.py tests/test_gemm_gpu.py tests/test_multirank_gpu.py -x -q > gpurun_out/t_inv.log 2>&1 || { tail -40 gpurun_out/t_inv.log; exit 1; }
tail -3 gpurun_out/t_inv.log
timeout -k 10 300 python tools/refops_probe.py 0 1 > gpurun_out/refops_01.log 2>&1; tail -2 gpurun_out/refops_01.log
timeout -k 10 300 python tools/refops_probe.py 1 1 > gpurun_out/refops_11.log 2>&1; tail -2 gpurun_out/refops_11.log
timeout -k 10 300 python tools/ragged_host_profile.py 1 1 > gpurun_out/ragged_host.log 2>&1; head -12 gpurun_out/ragged_host.log
