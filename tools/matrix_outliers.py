"""Outliers of the fused matrix-wide select over many different sqrt(scaler_row) vectors (a fallback costs ~1 ms): median, max, count."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "vlm-compression_amd"))
import torch
from vlmc import ops
dev = "cuda:0"
shapes = [(4224, 1408), (1408, 1408), (6144, 1408), (1408, 6144)]
torch.manual_seed(1)
W0 = [(torch.randn(o, i, device=dev) * 0.02).half() for o, i in shapes]
W = [w.clone() for w in W0]
masks = [torch.empty(w.shape, dtype=torch.bool, device=dev) for w in W]
parts = [torch.empty(ops.select_partials("matrix", *w.shape), dtype=torch.float64, device=dev) for w in W]
ks = [w.numel() // 2 for w in W]
ts = []
for r in range(600):
    g = torch.Generator(device=dev).manual_seed(r)
    sq = [ops.sqrt_scaler((torch.randn(2048, i, device=dev, generator=g) + 0.1).pow(2).mean(0)) if r % 2 else
          ops.sqrt_scaler(torch.rand(i, device=dev, generator=g) * 4 + 0.01) for o, i in shapes]
    for w, w0 in zip(W, W0): w.copy_(w0)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); ops.wanda_select_batch(W, sq, "matrix", ks=ks, masks=masks, partials=parts); b.record()
    torch.cuda.synchronize()
    ts.append(a.elapsed_time(b) * 1e3)
ts = ts[5:]
import statistics
print("median", statistics.median(ts), "max", max(ts), "over 150 us:", sum(t > 150 for t in ts), "of", len(ts))
