"""Launch time of sparsegpt_sweep_kernel vs rows / columns per block / n:m (HIP events)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vlm-compression_amd"))
import torch
from vlmc import sparsegpt as SG
dev = "cuda:0"
for rows in (256, 1024, 2048, 5120):
    for count in (32, 64, 128):
        for nm in ((0, 0), (2, 4)):
            W = torch.randn(rows, count, device=dev) * 0.05
            A = torch.randn(count, 2 * count, device=dev)
            U = torch.linalg.cholesky(A @ A.t() / count + 0.1 * torch.eye(count, device=dev), upper=True).contiguous()
            mask1 = (torch.rand(rows, count, device=dev) < 0.5) if nm[0] == 0 else None
            err = torch.empty(rows, count, device=dev)
            mout = torch.zeros(rows, count, dtype=torch.bool, device=dev)
            for _ in range(3):
                SG.sweep_block(W, 0, count, U, mask1, nm[0], nm[1], err, mout)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(20):
                SG.sweep_block(W, 0, count, U, mask1, nm[0], nm[1], err, mout)
            b.record()
            torch.cuda.synchronize()
            print(f"rows {rows:5d} count {count:3d} n:m {nm}: {a.elapsed_time(b) / 20 * 1e3:7.1f} us")
