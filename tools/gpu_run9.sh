set -e
cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_sparse_lora_gpu.py tests/test_ressa.py -m gpu -x -q > gpurun_out/t_lora.log 2>&1 || { tail -40 gpurun_out/t_lora.log; exit 1; }
tail -2 gpurun_out/t_lora.log
for t in 1 0 2 4 8; do
echo "== VLMC_LORA_TPW=$t"
VLMC_LORA_TPW=$t timeout -k 10 300 python tools/bench_methods.py --only lora 2>&1 | grep -v amdgpu | grep -i "lora\|weff\|eff" | head -8
done
