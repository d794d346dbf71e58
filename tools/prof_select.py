"""Profiling driver: a few launches of one select / stat shape (run under rocprofv3)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vlm-compression_amd"))
import torch
from vlmc import ops
mode = sys.argv[1] if len(sys.argv) > 1 else "row"
out_f = int(sys.argv[2]) if len(sys.argv) > 2 else 5120
in_f = int(sys.argv[3]) if len(sys.argv) > 3 else 2048
dt = torch.float16 if mode == "matrix" else torch.bfloat16
dev = "cuda:0"
W0 = (torch.randn(out_f, in_f, device=dev) * 0.02).to(dt)
s = ops.sqrt_scaler(torch.rand(in_f, device=dev) * 4 + 0.01)
mask = torch.empty(out_f, in_f, dtype=torch.bool, device=dev)
for i in range(10):
    W = W0.clone()
    if mode == "row":
        ops.wanda_select(W, s, "row", k=in_f // 2, mask=mask)
    elif mode == "matrix":
        ops.wanda_select(W, s, "matrix", k=out_f * in_f // 2, mask=mask)
    else:
        ops.wanda_select(W, s, "nm", n=2, m=4, mask=mask)
torch.cuda.synchronize()
