"""Blocked Cholesky (vlmc_chol_block + library GEMM/TRSM) vs torch.linalg.cholesky_ex on Hessian-like matrices."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "vlm-compression_amd"))
import torch
from vlmc import sparsegpt as SG
dev = "cuda:0"
def t(fn, reps=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
for n in (1408, 2048, 5120, 6144):
    X = (torch.randn(8192, n, device=dev) * torch.linspace(0.05, 2.0, n, device=dev)) + 0.3       # correlated, badly scaled columns
    H = (X.t() @ X) * (2 / 8192)
    ref = torch.linalg.cholesky(H.double())
    Lt, _ = torch.linalg.cholesky_ex(H)
    res = [f"n={n} cond~{float(torch.linalg.cond(H.double())):.1e}: torch {t(lambda: torch.linalg.cholesky_ex(H)):.2f} ms (err {float((Lt.double()-ref).abs().max()/ref.abs().max()):.1e})"]
    for gemm in (False, True):
        SG._CHOL_PANEL_GEMM = gemm
        SG._chol_graphs.clear()                      # the sweep is captured per matrix size with the variant in force
        Lm, info = SG.blocked_cholesky(H)
        res.append(f"blocked[{'gemm' if gemm else 'trsm'}] {t(lambda: SG.blocked_cholesky(H)):.2f} ms (err {float((Lm.double()-ref).abs().max()/ref.abs().max()):.1e}, info {int(info)})")
    print(" | ".join(res))
