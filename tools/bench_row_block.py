"""Developer micro-benchmark: per-row select of one Flan-T5-XL block (all linears, one batched call),
one mixed launch vs one launch per row width.   python tools/bench_row_block.py [--reps 30]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vlm-compression_amd"))
import torch
from vlmc import ops

ap = argparse.ArgumentParser(); ap.add_argument("--reps", type=int, default=30)
args = ap.parse_args()
dev = "cuda:0"
enc = [(2048, 2048)] * 4 + [(5120, 2048)] * 2 + [(2048, 5120)]
dec = [(2048, 2048)] * 8 + [(5120, 2048)] * 2 + [(2048, 5120)]
for name, shapes in (("encoder block", enc), ("decoder block", dec)):
    W0 = [(torch.randn(o, i, device=dev) * 0.02).bfloat16() for o, i in shapes]
    sq = [ops.sqrt_scaler(torch.rand(i, device=dev) * 4 + 0.01) for o, i in shapes]
    W = [w.clone() for w in W0]
    masks = [torch.empty(w.shape, dtype=torch.bool, device=dev) for w in W]
    parts = [torch.empty(ops.select_partials("row", *w.shape), dtype=torch.float64, device=dev) for w in W]
    ks = [w.shape[1] // 2 for w in W]
    B = sum(w.numel() * 5 + 4 * w.shape[1] for w in W)
    ref = None
    for mixed in ("0", "1", "0", "1"):
        os.environ["VLMC_SELECT_MIXED"] = mixed
        ts = []
        for r in range(args.reps + 3):
            for w, w0 in zip(W, W0): w.copy_(w0)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); ops.wanda_select_batch(W, sq, "row", ks=ks, masks=masks, partials=parts); b.record()
            torch.cuda.synchronize()
            if r >= 3: ts.append(a.elapsed_time(b) * 1e3)
        ts.sort()
        got = [m.clone() for m in masks] + [w.clone() for w in W]
        if ref is None: ref = got
        same = all(torch.equal(x, y) for x, y in zip(ref, got))
        print(f"{name} mixed={mixed}: med {ts[len(ts)//2]:7.1f} us  min {ts[0]:7.1f} us  {B/ts[len(ts)//2]/1e3:7.1f} GB/s (alg, {B/1e6:.1f} MB)  "
              f"identical to first={same}", flush=True)
