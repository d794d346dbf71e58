"""One factorization chain alone (vlmc.sparsegpt.inverse_upper_factor): wall time per call for the graph and the eager
route, per matrix size.  Under `rocprofv3 --kernel-trace --stats` the per-kernel durations say what a 128-column step is made of.
   python tools/chain_probe.py [n ...]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vlm-compression_amd"))
import torch  # noqa: E402
from vlmc import sparsegpt  # noqa: E402

dev = torch.device("cuda:0")
sizes = [int(a) for a in sys.argv[1:]] or [1408, 2048, 5120, 6144]
reps = int(os.environ.get("CHAIN_REPS", "5"))
for n in sizes:
    X = torch.randn(4 * n, n, device=dev)
    H = (X.t() @ X) / (4 * n) + 0.01 * torch.eye(n, device=dev)
    for graph in (True, False):
        sparsegpt._CHOL_GRAPH = graph
        sparsegpt.release_caches()
        for _ in range(2):
            U, info = sparsegpt.inverse_upper_factor(H)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            U, info = sparsegpt.inverse_upper_factor(H)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        steps = (n + 127) // 128
        print(f"n = {n:5d}  {'graph' if graph else 'eager'}: {dt * 1e3:7.2f} ms per factor = {dt * 1e6 / steps:6.1f} us per 128-column step"
              f" ({steps} steps), info {int(info)}", flush=True)
