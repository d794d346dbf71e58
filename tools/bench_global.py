"""Time vlmc_score_select at InstructBLIP-FlanT5-XL scale (588 tensors, 3.70 G weights, fp16 ViT + bf16 T5):
magnitude (weights only), aobd (weights + fp32 gradient statistic), and the torch equivalent of the reference's
get_mask on the GPU for a smaller slice.  python tools/bench_global.py [--blocks-div D]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vlm-compression_amd"))
import torch  # noqa: E402

from vlmc import ops, workload  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--div", type=int, default=1, help="keep every div-th block")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--aobd", type=int, default=1)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    blocks = workload.flan_t5_xl()[::a.div]
    ws, scopes = [], []
    for b in blocks:
        for l in b.linears:
            ws.append((torch.randn(l.out_features, l.in_features, device=dev) * 0.02).to(b.dtype))
            scopes.append(0 if b.tower == "vit" else 1)
    total = sum(w.numel() for w in ws)
    wbytes = sum(w.numel() * w.element_size() for w in ws)
    print(f"{len(ws)} tensors, {total/1e9:.3f} G elements, {wbytes/1e9:.2f} GB of weights")
    keeps = [torch.empty(w.shape, dtype=torch.bool, device=dev) for w in ws]

    def timed(label, fn, algo_bytes):
        fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(a.reps):
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        t = min(ts)
        print(f"{label:34s} {t*1e3:9.2f} ms   {algo_bytes/t/1e9:8.1f} GB/s (algorithmic)")

    for layout, sc, nsc in (("global", [0] * len(ws), 1), ("per-model", scopes, 2), ("layer-wise", list(range(len(ws))), len(ws))):
        sizes = [0] * nsc
        for w, s in zip(ws, sc):
            sizes[s] += w.numel()
        ks = [int(0.5 * n) for n in sizes]
        # 3 histogram reads + 1 apply read + W write + mask write (apply_weights=False keeps the data random between reps)
        timed(f"magnitude / {layout}", lambda: ops.score_select(ws, "weight", scopes=sc, scope_ks=ks, keeps=keeps, apply_weights=False),
              4 * wbytes + total)
    if a.aobd:
        S = [torch.rand(w.shape, device=dev) for w in ws]
        ks = [int(0.5 * total)]
        timed("aobd / global", lambda: ops.score_select(ws, "absw_score", scopes=[0] * len(ws), scope_ks=ks, scores=S, keeps=keeps,
                                                        apply_weights=False), 4 * (wbytes + 4 * total) + total)
        prot = [int(0.2 * w.numel()) for w in ws]
        timed("aobd / global + per-layer cap", lambda: ops.score_select(ws, "absw_score", scopes=[0] * len(ws), scope_ks=ks, scores=S,
                                                                        keeps=keeps, protect_ks=prot, apply_weights=False),
              7 * (wbytes + 4 * total) + total)
        del S
    # what the reference does (:120-130), on the GPU instead of the CPU, for the first ~0.3 G elements
    sub = []
    n = 0
    for w in ws:
        sub.append(w)
        n += w.numel()
        if n > 3e8:
            break

    def torch_get_mask():
        sc = torch.cat([w.float().flatten() for w in sub])
        thr = torch.topk(sc, int(0.5 * sc.numel()), largest=False)[0][-1]
        return [w.float() > thr for w in sub]
    timed(f"torch cat+topk, {n/1e9:.2f} G elements", torch_get_mask, sum(w.numel() * w.element_size() for w in sub))
    ks = [int(0.5 * n)]
    timed(f"score_select, same {n/1e9:.2f} G", lambda: ops.score_select(sub, "weight", scopes=[0] * len(sub), scope_ks=ks,
                                                                         keeps=keeps[:len(sub)], apply_weights=False),
          4 * sum(w.numel() * w.element_size() for w in sub) + n)


if __name__ == "__main__":
    main()
