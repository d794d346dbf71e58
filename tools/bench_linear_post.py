"""vlmc_linear_fwd_post against vlmc_linear_fwd + the separate torch op, at the prune's shapes (us per call, warm weights)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vlm-compression_amd"))
import torch, torch.nn.functional as F
from vlmc import ops
dev = "cuda:0"


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


print("| linear | M x N x K | plain GEMM | + separate op | folded | op alone |")
print("|---|---|---|---|---|---|")
for name, dt, M, N, K, kind in [("vit.fc1 + GELU", torch.float16, 32896, 6144, 1408, "gelu"), ("vit.fc2 + residual", torch.float16, 32896, 1408, 6144, "res"),
                                ("vit.proj + residual", torch.float16, 32896, 1408, 1408, "res"), ("vit.qkv + bias", torch.float16, 32896, 4224, 1408, "bias"),
                                ("t5enc.wo + residual", torch.bfloat16, 8192, 2048, 5120, "res"), ("t5enc.o + residual", torch.bfloat16, 8192, 2048, 2048, "res"),
                                ("t5dec.wo + residual", torch.bfloat16, 2048, 2048, 5120, "res")]:
    x = torch.randn(M, K, device=dev).to(dt)
    w = (torch.randn(N, K, device=dev) * 0.05).to(dt)
    r = torch.randn(M, N, device=dev).to(dt)
    pb = torch.randn(N, device=dev).to(dt)
    y = ops.linear_fwd(x, w)
    t_plain = timeit(lambda: ops.linear_fwd(x, w))
    if kind == "gelu":
        sep, fold, alone = (lambda: F.gelu(ops.linear_fwd(x, w))), (lambda: ops.linear_fwd_post(x, w, act=1)), (lambda: F.gelu(y))
    elif kind == "res":
        sep, fold, alone = (lambda: r + ops.linear_fwd(x, w)), (lambda: ops.linear_fwd_post(x, w, residual=r)), (lambda: r + y)
    else:
        sep, fold, alone = (lambda: ops.linear_fwd(x, w) + pb), (lambda: ops.linear_fwd_post(x, w, post_bias=pb)), (lambda: y + pb)
    print(f"| {name} | {M} x {N} x {K} | {t_plain:.1f} | {timeit(sep):.1f} | {timeit(fold):.1f} | {timeit(alone):.1f} |", flush=True)
