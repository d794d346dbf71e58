"""Sweep the lane vector width of vlmc_act_sqnorm on the bench's activation shapes (GPU only)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "vlm-compression_amd"))
import torch
from vlmc import ops

dev = "cuda:0"
shapes = [("vit 1408", (128, 257, 1408), torch.float16), ("vit 6144", (128, 257, 6144), torch.float16),
          ("enc 2048", (128, 64, 2048), torch.bfloat16), ("enc 5120", (128, 64, 5120), torch.bfloat16),
          ("dec 2048", (128, 16, 2048), torch.bfloat16)]
for name, shp, dt in shapes:
    x = torch.randn(shp, device=dev).to(dt)
    out = torch.empty((shp[0], shp[2]), dtype=torch.float32, device=dev)
    nbytes = x.numel() * 2
    res = []
    for vec in (1, 2, 4, 8):
        os.environ["VLMC_SQNORM_VEC"] = str(vec)
        for _ in range(3):
            ops.act_sqnorm(x, out=out)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            ops.act_sqnorm(x, out=out)
        b.record()
        torch.cuda.synchronize()
        us = a.elapsed_time(b) * 1e3 / 20
        res.append(f"vec{vec}: {us:7.1f} us {nbytes / us / 1e6:5.2f} TB/s")
    print(f"{name:9s}", " | ".join(res))
