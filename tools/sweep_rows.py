"""Sweep waves-per-row of the per-row select on one shape (GPU only)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "vlm-compression_amd"))
import torch
from vlmc import ops

dev = "cuda:0"
for out_f, in_f in [(2048, 5120), (5120, 2048), (2048, 2048)]:
    copies = 24
    Ws = [(torch.randn(out_f, in_f, device=dev) * 0.02).to(torch.bfloat16) for _ in range(copies)]
    sq = torch.rand(in_f, device=dev) + 0.5
    mask = torch.empty((out_f, in_f), dtype=torch.bool, device=dev)
    parts = torch.empty(out_f, dtype=torch.float64, device=dev)
    res = []
    for nw in (1, 2, 4, 8):
        if in_f // 8 > 64 * nw * 4:
            continue
        os.environ["VLMC_SELECT_NW"] = str(nw)
        plans = [ops.plan_select(w, sq, "row", k=in_f // 2, apply_zero=False, mask=mask, partials=parts) for w in Ws]
        for p in plans[:4]:
            p()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for p in plans:
            p()
        b.record()
        torch.cuda.synchronize()
        us = a.elapsed_time(b) * 1e3 / copies
        res.append(f"nw{nw}: {us:6.1f} us {out_f * in_f * 3 / us / 1e6:5.2f} TB/s(3B/w)")
    print((out_f, in_f), " | ".join(res))
