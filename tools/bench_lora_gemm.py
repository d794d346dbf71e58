"""The fused masked-LoRA GEMMs (csrc/lora_gemm.hip) against the unfused route (W_eff kernel + library GEMMs + G in memory +
vlmc_lora_grad) per layer shape: forward, input gradient, weight gradients; cold weights (a pool of layers per shape, as a
training step meets them).

    python tools/bench_lora_gemm.py [--tokens 1536] [--r 16] [--dtype float16] [--pool 6] [--iters 20]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vlm-compression_amd"))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from vlmc import _lib, sparse_lora as SL  # noqa: E402

SHAPES = {"v7b.qkvo": (4096, 4096), "v7b.gate_up": (11008, 4096), "v7b.down": (4096, 11008), "t5.q": (2048, 2048), "t5.wi": (5120, 2048),
          "t5.wo": (2048, 5120), "vit.qkv": (4224, 1408), "vit.fc1": (6144, 1408), "vit.fc2": (1408, 6144)}


def timed(fn, n, iters):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    for i in range(n):
        fn(i)                                          # once over the pool: warm code, cold weights next time round
    torch.cuda.synchronize()
    ev[0].record()
    for it in range(iters):
        fn(it % n)
    ev[1].record()
    torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / iters * 1e3     # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tokens", type=int, default=1536)
    ap.add_argument("--r", type=int, default=16)
    ap.add_argument("--dtype", default="float16")
    ap.add_argument("--pool", type=int, default=6)
    ap.add_argument("--iters", type=int, default=24)
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    wd = getattr(torch, args.dtype)
    code = {torch.float16: 1, torch.bfloat16: 2}[wd]
    dev = "cuda:0"
    lib = _lib.load()
    M, r = args.tokens, args.r
    print(f"| shape (out x in), M = {M}, r = {r}, {args.dtype} | fused fwd us (TF/s) | W_eff + F.linear us | fused dX us | library dX us | "
          f"fused dA,dB us | dY^T x + vlmc_lora_grad us | fwd+bwd fused / unfused |")
    print("|---|---|---|---|---|---|---|---|")
    for name, (out_f, in_f) in SHAPES.items():
        if args.only and args.only not in name:
            continue
        g = torch.Generator(device=dev).manual_seed(0)
        Ws = [(torch.randn(out_f, in_f, device=dev, generator=g) * 0.02).to(wd) for _ in range(args.pool)]
        Ms = [torch.rand(out_f, in_f, device=dev, generator=g) > 0.5 for _ in range(args.pool)]
        A = torch.randn(r, in_f, device=dev, generator=g) * 0.01
        B = torch.randn(out_f, r, device=dev, generator=g) * 0.01
        x = torch.randn(M, in_f, device=dev, generator=g).to(wd)
        gy = (torch.randn(M, out_f, device=dev, generator=g) * 0.1).to(wd)
        prep = torch.empty(lib.vlmc_sparse_lora_prep_bytes(out_f, in_f), dtype=torch.uint8, device=dev)
        y = torch.empty(M, out_f, dtype=wd, device=dev)
        gx = torch.empty(M, in_f, dtype=wd, device=dev)
        gA, gB = torch.empty_like(A), torch.empty_like(B)
        ws = torch.empty(lib.vlmc_sparse_lora_bwd_weight_workspace(M, out_f, in_f), dtype=torch.uint8, device=dev)
        st = torch.cuda.current_stream().cuda_stream

        def f_fwd(i):
            _lib.check(lib.vlmc_sparse_lora_prep(A.data_ptr(), B.data_ptr(), out_f, in_f, r, code, prep.data_ptr(), st))
            _lib.check(lib.vlmc_sparse_lora_fwd(x.data_ptr(), M, in_f, Ws[i].data_ptr(), code, out_f, in_f, in_f, Ms[i].data_ptr(), prep.data_ptr(), r, 1.0,
                                                1, code, None, y.data_ptr(), out_f, st))

        def u_fwd(i):
            w = SL.effective_weight(Ws[i], A, B, Ms[i], 1.0, SL.FWD_SPARSE, code)
            return F.linear(x, w)

        def f_dx(i):
            _lib.check(lib.vlmc_sparse_lora_bwd_input(gy.data_ptr(), M, out_f, Ws[i].data_ptr(), code, out_f, in_f, in_f, Ms[i].data_ptr(), prep.data_ptr(),
                                                      r, 1.0, 1, code, gx.data_ptr(), in_f, st))

        weffs = [SL.effective_weight(Ws[i], A, B, Ms[i], 1.0, SL.FWD_SPARSE, code) for i in range(args.pool)]

        def u_dx(i):
            return gy @ weffs[i]                       # (the unfused route keeps W_eff from the forward)

        def f_dw(i):
            _lib.check(lib.vlmc_sparse_lora_bwd_weight(gy.data_ptr(), out_f, x.data_ptr(), in_f, M, code, out_f, in_f, Ms[i].data_ptr(), prep.data_ptr(), r,
                                                       1.0, 1, code, gA.data_ptr(), gB.data_ptr(), ws.data_ptr(), ws.numel(), st))

        def u_dw(i):
            G = gy.t() @ x
            return SL.lora_grads(G, A, B, Ms[i], 1.0, True, autocast=code)

        t = {k: timed(f, args.pool, args.iters) for k, f in (("ff", f_fwd), ("uf", u_fwd), ("fx", f_dx), ("ux", u_dx), ("fw", f_dw), ("uw", u_dw))}
        fl = 2.0 * M * out_f * in_f
        print(f"| {name} {out_f} x {in_f} | {t['ff']:.1f} ({fl / t['ff'] / 1e6:.0f}) | {t['uf']:.1f} | {t['fx']:.1f} ({fl / t['fx'] / 1e6:.0f}) | {t['ux']:.1f} | "
              f"{t['fw']:.1f} ({fl / t['fw'] / 1e6:.0f}) | {t['uw']:.1f} | {(t['ff'] + t['fx'] + t['fw']) / (t['uf'] + t['ux'] + t['uw']):.2f} |", flush=True)
        del Ws, Ms, weffs


if __name__ == "__main__":
    main()
