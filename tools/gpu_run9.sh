set -e
cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gelu_gpu.py -m gpu -x -q > gpurun_out/t_gelu.log 2>&1 || { tail -50 gpurun_out/t_gelu.log; exit 1; }
tail -2 gpurun_out/t_gelu.log
python - <<'PY'
import sys, statistics
sys.path.insert(0, "/root/repo/vlm-compression_amd")
import torch, torch.nn.functional as F
from vlmc import ops
for shape, dt in (((128, 257, 6144), torch.float16), ((128 * 64, 5120), torch.bfloat16), ((16, 257, 6144), torch.float16)):
    x = torch.randn(*shape, device="cuda:0").to(dt)
    def t(fn):
        ts = []
        for _ in range(5):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(10): fn(x)
            b.record(); torch.cuda.synchronize()
            ts.append(a.elapsed_time(b) / 10)
        return statistics.median(ts) * 1e3
    print(shape, dt, f"vlmc_gelu {t(ops.gelu):.1f} us   F.gelu {t(F.gelu):.1f} us   ({x.numel() * 4 / 1e6:.0f} MB)")
PY
for f in 0 1; do
echo "== VLMC_GELU=$f"
VLMC_GELU=$f RANK_TIMELINE_ITERS=6 timeout -k 10 300 python tools/rank_timeline.py 1 2>&1 | grep prune_ms | tail -2 | cut -c1-200
done
