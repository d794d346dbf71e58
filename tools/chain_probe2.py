"""vlmc_chol_inverse (one persistent launch) against the chain of launches per 128 columns: time per factor, agreement of
the factors, accuracy against fp64.   python tools/chain_probe2.py [n ...]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vlm-compression_amd"))
import torch  # noqa: E402
from vlmc import sparsegpt  # noqa: E402

dev = torch.device("cuda:0")
sizes = [int(a) for a in sys.argv[1:]] or [256, 1408, 2048, 5120, 6144]
reps = 5
for n in sizes:
    X = torch.randn(4 * n, n, device=dev)
    H = (X.t() @ X) / (4 * n) + 0.01 * torch.eye(n, device=dev)
    out = {}
    for mode, wgs in (("chain", None), ("persistent", 256), ("persistent", 128), ("persistent", 64), ("persistent", 32)):
        sparsegpt._PERSISTENT = mode == "persistent"
        for _ in range(2):
            U, info = sparsegpt.inverse_upper_factor(H, max_workgroups=wgs)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            U, info = sparsegpt.inverse_upper_factor(H, max_workgroups=wgs)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        steps = n // 128
        out[(mode, wgs)] = U
        line = f"n = {n:5d}  {mode:10s} wgs {wgs}: {dt * 1e3:7.2f} ms per factor = {dt * 1e6 / max(1, steps):6.1f} us per 128 columns, info {int(info)}"
        if mode == "persistent":
            ref = out[("chain", None)]
            line += f"  |U - U_chain| / |U_chain| = {float((U - ref).norm() / ref.norm()):.2e}"
            if wgs != 256:
                line += f"  same bits as 256 workgroups: {bool(torch.equal(U, out[('persistent', 256)]))}"
        print(line, flush=True)
    if n <= 2048:
        Hd = H.double()
        for k, U in out.items():
            Ud = U.double()
            err = float((Ud.t() @ Ud @ Hd - torch.eye(n, device=dev, dtype=torch.float64)).abs().max())
            print(f"      {k}: max |U^T U H - I| = {err:.2e}, upper triangular {bool((torch.tril(U, -1) == 0).all())}")
