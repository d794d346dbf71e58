"""Developer micro-benchmark: matrix-wide select of one ViT-g block (4 linears, one batched call),
fused kernel vs the four-launch form.   python tools/bench_matrix_block.py [--reps 30]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vlm-compression_amd"))
import torch
from vlmc import ops

ap = argparse.ArgumentParser(); ap.add_argument("--reps", type=int, default=30)
ap.add_argument("--acts", action="store_true", help="sqrt(scaler_row) from synthetic activations N(0.1, 1) like bench.py's kernel pass "
                "(nearly the same for every column) instead of a spread of values")
args = ap.parse_args()
dev = "cuda:0"
shapes = [(4224, 1408), (1408, 1408), (6144, 1408), (1408, 6144)]
W0 = [(torch.randn(o, i, device=dev) * 0.02).half() for o, i in shapes]
if args.acts:
    sq = [ops.sqrt_scaler((torch.randn(128 * 64, i, device=dev) + 0.1).pow(2).mean(0)) for o, i in shapes]
else:
    sq = [ops.sqrt_scaler(torch.rand(i, device=dev) * 4 + 0.01) for o, i in shapes]
W = [w.clone() for w in W0]
masks = [torch.empty(w.shape, dtype=torch.bool, device=dev) for w in W]
parts = [torch.empty(ops.select_partials("matrix", *w.shape), dtype=torch.float64, device=dev) for w in W]
ks = [w.numel() // 2 for w in W]
B = sum(w.numel() * 5 + 4 * w.shape[1] for w in W)
ref = None
for fused in ("0", "1", "0", "1"):
    os.environ["VLMC_MATRIX_FUSED"] = fused
    ts = []
    for r in range(args.reps + 3):
        for w, w0 in zip(W, W0): w.copy_(w0)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); ops.wanda_select_batch(W, sq, "matrix", ks=ks, masks=masks, partials=parts); b.record()
        torch.cuda.synchronize()
        if r >= 3: ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    got = [m.clone() for m in masks] + [w.clone() for w in W]
    if ref is None: ref = got
    same = all(torch.equal(x, y) for x, y in zip(ref, got))
    psum = [float(p.sum()) for p in parts]
    print(f"fused={fused}: med {ts[len(ts)//2]:7.1f} us  min {ts[0]:7.1f} us  {B/ts[len(ts)//2]/1e3:7.1f} GB/s (alg, {B/1e6:.1f} MB)  "
          f"identical to first={same}  sparsity {[round(1-m.float().mean().item(),5) for m in masks]}  parts {psum[0]:.6e}", flush=True)
if os.environ.get("VLMC_LIB"):      # diagnostic build with -DVLMC_FUSED_STAMPS: phase clocks of the last fused launch
    nbytes = ops._lib.load().vlmc_wanda_select_workspace(ops._MODES["matrix"], *shapes[0])
    buf = list(ops._batch_ws._bufs.values())[0].view(torch.int32).cpu().numpy().astype("int64")
    names = ["start", "P1 count", "flush", "barrier A", "P2 bin+cands", "barrier B", "P3 select", "P4 apply"]
    for j, shp in enumerate(shapes):
        base = j * nbytes // 4 + 2048 + 16
        for which, off in (("wg0", 0), ("last", 8)):
            st = buf[base + off: base + off + 8] & 0xFFFFFFFF
            d = [(int(st[i]) - int(st[0])) / 100.0 for i in range(8)]
            print(f"job {j} {shp} {which}: " + "  ".join(f"{n}@{t:.1f}us" for n, t in zip(names, d)), flush=True)
