"""Developer micro-benchmark: per-kernel time and algorithmic GB/s on one GPU.
   python tools/bench_kernels.py [--reps 20]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vlm-compression_amd"))
import torch
from vlmc import ops

ap = argparse.ArgumentParser(); ap.add_argument("--reps", type=int, default=20); ap.add_argument("--only", default="")
args = ap.parse_args()
dev = "cuda:0"

def timeit(fn, reps, setup=None):
    ts = []
    for i in range(reps + 3):
        if setup: setup()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        if i >= 3: ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    return ts[len(ts) // 2], ts[0]

def bench_select(out_f, in_f, dtype, mode, **kw):
    W0 = (torch.randn(out_f, in_f, device=dev) * 0.02).to(dtype)
    s = ops.sqrt_scaler(torch.rand(in_f, device=dev) * 4 + 0.01)
    W = W0.clone(); mask = torch.empty(out_f, in_f, dtype=torch.bool, device=dev); ss = torch.empty(ops.select_partials(mode, out_f, in_f), dtype=torch.float64, device=dev)
    med, mn = timeit(lambda: ops.wanda_select(W, s, mode, mask=mask, partials=ss, **kw), args.reps, setup=lambda: W.copy_(W0))
    B = out_f * in_f * (2 * W.element_size() + 1) + 4 * in_f
    print(f"select {mode:6s} {out_f:5d}x{in_f:5d} {str(dtype)[6:]:8s} med {med:8.1f} us  min {mn:8.1f} us  {B/med/1e3:7.1f} GB/s (alg)  sparsity {1-mask.float().mean().item():.4f}")

def bench_sqnorm(calls, T, in_f, dtype):
    x = (torch.randn(calls, T, in_f, device=dev) + 0.1).to(dtype)
    out = torch.empty(calls, in_f, device=dev)
    med, mn = timeit(lambda: ops.act_sqnorm(x, out=out), args.reps)
    B = x.numel() * x.element_size() + out.numel() * 4
    print(f"sqnorm [{calls},{T},{in_f}] {str(dtype)[6:]:8s} med {med:8.1f} us  min {mn:8.1f} us  {B/med/1e3:7.1f} GB/s (alg)  VEC={os.environ.get('VLMC_SQNORM_VEC','auto')}")

if "select" in args.only or not args.only:
    for shp in [(5120, 2048), (2048, 5120), (2048, 2048), (4096, 4096), (11008, 4096), (4096, 11008)]:
        bench_select(*shp, torch.bfloat16, "row", k=shp[1] // 2)
    for shp in [(4224, 1408), (1408, 1408), (6144, 1408), (1408, 6144)]:
        bench_select(*shp, torch.float16, "matrix", k=shp[0] * shp[1] // 2)
    for shp in [(5120, 2048), (6144, 1408)]:
        bench_select(*shp, torch.bfloat16, "nm", n=2, m=4)
if "sqnorm" in args.only or not args.only:
    for shp in [(128, 64, 2048), (128, 64, 5120), (128, 257, 1408), (128, 257, 6144), (128, 16, 2048), (1, 64, 2048), (1, 257, 1408)]:
        bench_sqnorm(*shp, torch.bfloat16)
