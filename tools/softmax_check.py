import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vlm-compression_amd"))
import torch
from vlmc import ops
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
for rows, n, pad in [(37, 40, 160), (1000, 257, 300), (64, 1500, 2000), (5, 1, 64)]:
    x = torch.randn(rows, n, device=dev, generator=g) * 3
    y = ops.softmax_rows(x)
    ref = torch.softmax(x.double(), -1)
    xp = torch.full((rows, pad), torch.finfo(torch.bfloat16).min, device=dev)
    xp[:, :n] = x
    yp = ops.softmax_rows(xp)
    print(rows, n, "max err vs fp64", float((y.double() - ref).abs().max()), "padded rows equal:", torch.equal(yp[:, :n], y), "padding is 0:", bool((yp[:, n:] == 0).all()),
          "torch padded equal:", torch.equal(torch.softmax(xp, -1)[:, :n], torch.softmax(x, -1)))
