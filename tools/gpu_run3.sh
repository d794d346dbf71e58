set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_attn_matmul_gpu.py tests/test_replay_invariance_gpu.py -x -q > gpurun_out/t_attn.log 2>&1 || { tail -40 gpurun_out/t_attn.log; exit 1; }
tail -2 gpurun_out/t_attn.log
bash tools/gpu_run2.sh 0 1 | grep -E "attn_matmul|softmax_warp|\"it\": 2"
