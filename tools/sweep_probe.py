"""One 128-column block of the SparseGPT sweep: `vlmc_sparsegpt_sweep` with the mask handed in against the one-launch
threshold + sweep (`vlmc_sparsegpt_select_sweep`), and the trailing update's library GEMM.   python tools/sweep_probe.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vlm-compression_amd"))
import torch  # noqa: E402
from vlmc import sparsegpt as SG  # noqa: E402

dev = torch.device("cuda:0")


def timed(fn, reps=30):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


print("| rows (scopes) | cols | sweep, mask given us | threshold + sweep in one launch us | trailing addmm_ us |")
print("|---|---|---|---|---|")
for scopes, cols in (((2048,), 2048), ((1408,), 6144), ((4224,), 1408), ((6144,), 1408), ((2048, 2048, 2048), 2048), ((5120, 5120), 2048), ((2048,), 5120)):
    rows = sum(scopes)
    W = torch.randn(rows, cols, device=dev) * 0.05
    A = torch.randn(cols, 2 * cols, device=dev)
    U = torch.linalg.cholesky(A @ A.t() / cols + 0.1 * torch.eye(cols, device=dev), upper=True).contiguous()
    err = torch.empty(rows, 128, device=dev)
    mask1 = torch.rand(rows, 128, device=dev) < 0.5
    ranks = [r * 64 for r in scopes]
    W0 = W.clone()
    t_old = timed(lambda: SG.sweep_block(W, 0, 128, U, mask1, 0, 0, err))
    W.copy_(W0)
    t_new = timed(lambda: SG.select_sweep_block(W, 0, 128, U, list(scopes), ranks, err))
    t_mm = timed(lambda: W[:, 128:].addmm_(err, U[0:128, 128:], beta=1.0, alpha=-1.0))
    print(f"| {rows} {scopes} | {cols} | {t_old:.1f} | {t_new:.1f} | {t_mm:.1f} |", flush=True)
