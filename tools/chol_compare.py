"""Factor and inverse of the blocked Cholesky for a few sizes, saved to a file (to compare two builds of the library bit for bit):
    VLMC_LIB=<build A> python tools/chol_compare.py a.pt;  python tools/chol_compare.py b.pt;  python tools/chol_compare.py a.pt b.pt"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "vlm-compression_amd"))
import torch
if len(sys.argv) == 3:
    a, b = torch.load(sys.argv[1]), torch.load(sys.argv[2])
    for k in a:
        print(k, "identical" if torch.equal(a[k], b[k]) else f"DIFFERENT max |d| {float((a[k] - b[k]).abs().max()):.3e}")
    sys.exit(0)
from vlmc import sparsegpt
out = {}
for n in (128, 200, 2048, 5120):
    g = torch.Generator(device="cuda:0").manual_seed(n)
    X = torch.randn(3 * n, n, device="cuda:0", generator=g)
    H = (X.t() @ X) / (3 * n) + 0.01 * torch.eye(n, device="cuda:0")
    L, info = sparsegpt.blocked_cholesky(H.clone())
    U, info2 = sparsegpt.inverse_upper_factor(H.clone())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        sparsegpt.inverse_upper_factor(H.clone())
    torch.cuda.synchronize()
    print(n, "info", int(info), int(info2), f"inverse_upper_factor {1e3 * (time.perf_counter() - t0) / 5:.2f} ms", flush=True)
    out[f"L{n}"], out[f"U{n}"] = L.cpu(), U.cpu()
torch.save(out, sys.argv[1])
