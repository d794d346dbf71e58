"""Micro-benchmarks of the SparseLoRA / SparseGPT / DSnoT kernels at the model shapes of BASELINE.json
configs 3-5 (GPU only).  Prints a markdown table (kept under profiles/).

    python tools/bench_methods.py [--reps 10]
Times are HIP-event medians over fresh (uncached) operands; GB/s use the algorithmic bytes of DESIGN.md §4."""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vlm-compression_amd"))
import torch
import torch.nn as nn
from vlmc import dsnot, ops, sparse_lora, sparsegpt

ap = argparse.ArgumentParser(); ap.add_argument("--reps", type=int, default=10); ap.add_argument("--only", default="")
args = ap.parse_args()
dev = "cuda:0"
rows = []


def timeit(fn, reps=args.reps, setup=None, warm=2):
    ts = []
    for i in range(reps + warm):
        if setup:
            setup()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        if i >= warm:
            ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


def emit(kernel, shape, us, note):
    rows.append(f"| {kernel} | {shape} | {us:.1f} | {note} |")
    print(rows[-1], flush=True)


# ---- SparseLoRA (config 5: Vicuna-7B linears, r = 16, fp16 weights) ------------------------------------
for out_f, in_f in ([(4096, 4096), (11008, 4096), (4096, 11008)] if args.only in ('', 'lora') else []):
    r = 16
    pool = [(torch.randn(out_f, in_f, device=dev) * 0.02).to(torch.float16) for _ in range(6)]
    A = torch.randn(r, in_f, device=dev) * 0.05
    B = torch.randn(out_f, r, device=dev) * 0.05
    M = torch.rand(out_f, in_f, device=dev) > 0.5
    out = torch.empty_like(pool[0])
    it = iter(range(10 ** 9))
    us = timeit(lambda: sparse_lora.effective_weight(pool[next(it) % len(pool)], A, B, M, 1.0, sparse_lora.FWD_SPARSE, 0, out=out))
    nbytes = out_f * in_f * 5 + (out_f + in_f) * r * 4
    emit("`lora_weff_kernel` (forward, sparse)", f"{out_f}x{in_f} r16 fp16", us,
         f"{nbytes / us / 1e3:.0f} GB/s of 5 B/weight (W + mask read, W_eff written); {2 * out_f * in_f * r / us / 1e6:.2f} TFLOP/s f32 MFMA")
    G = [(torch.randn(out_f, in_f, device=dev) * 0.01).to(torch.float16) for _ in range(6)]
    us = timeit(lambda: sparse_lora.lora_grads(G[next(it) % len(G)], A, B, M, 1.0, True, 0))
    nbytes = out_f * in_f * 3
    emit("`lora_grad_fused_kernel` + 2 x `lora_grad_reduce_kernel` (dA and dB)", f"{out_f}x{in_f} r16 fp16", us,
         f"{nbytes / us / 1e3:.0f} GB/s of 3 B/weight (G + mask read once); {4 * out_f * in_f * r / us / 1e6:.2f} TFLOP/s f32 MFMA; "
         "event-timed from an idle stream, i.e. including 3 launch latencies")
    del pool, G

# ---- SparseGPT (config 3: FlanT5-XL shapes, 2:4 and unstructured 50 %) -----------------------------------
for out_f, in_f in ([(2048, 2048), (5120, 2048), (2048, 5120)] if args.only in ('', 'sparsegpt') else []):
    lin = nn.Linear(in_f, out_f, bias=False).to(dev).to(torch.bfloat16)
    g = sparsegpt.SparseGPT(lin)
    xs = [(torch.randn(1, 64, in_f, device=dev) + 0.1).to(torch.bfloat16) for _ in range(128)]
    us_h = timeit(lambda: [g.add_batch(x) for x in xs], reps=3, warm=1)
    emit("Hessian `addmm_` x128 samples (library GEMM)", f"in={in_f}, 64 tokens", us_h,
         f"{2 * 64 * in_f * in_f * 128 / us_h / 1e6:.1f} TFLOP/s fp32")
    H0 = g.H.clone()
    W0 = lin.weight.data.clone()
    for tag, kw in [("2:4", dict(sparsity=0.0, prune_n=2, prune_m=4)), ("50 %", dict(sparsity=0.5))]:
        def setup():
            lin.weight.data.copy_(W0)
        us = timeit(lambda: sparsegpt.fasterprune(lin, H0.clone(), **kw), reps=3, warm=1, setup=setup)
        emit(f"`fasterprune` {tag} (Cholesky chain + {in_f // 128} x [`sparsegpt_sweep_kernel` + trailing GEMM])",
             f"{out_f}x{in_f} bf16", us, "whole linear")
    # the sweep kernel alone
    W = W0.float().clone()
    U = torch.linalg.cholesky(torch.eye(in_f, device=dev) * 2 + 0.01, upper=True).contiguous()
    err = torch.empty((out_f, 128), device=dev)
    us = timeit(lambda: sparsegpt.sweep_block(W, 0, 128, U, None, 2, 4, err))
    emit("`sparsegpt_sweep_kernel` (one 128-column block, 2:4)", f"{out_f} rows", us,
         f"{out_f * 128 * 128 / 2 * 2 / us / 1e6:.2f} TFLOP/s fp32 FMA on the sequential chain")
    del g, xs, H0

# ---- DSnoT (config 4: Vicuna-7B + ViT shapes, wanda init, 100 cycles) -----------------------------------
for out_f, in_f, dt in ([(4096, 4096, torch.float16), (11008, 4096, torch.float16), (4096, 11008, torch.float16),
                         (6144, 1408, torch.float16)] if args.only in ('', 'dsnot') else []):
    xs = [(torch.randn(1, 96, in_f, device=dev) + 0.2).to(dt) for _ in range(16)]
    st = dsnot.DsnotInputStat(in_f, dev)
    for x in xs:
        st.add_call(x)
    st.finalize()
    big = (torch.randn(128, 96, in_f, device=dev) + 0.2).to(dt)
    outm = torch.empty((128, 3, in_f), device=dev)
    from vlmc import _lib
    from vlmc.ops import _dtype_code, _stream
    lib = _lib.load()
    us = timeit(lambda: _lib.check(lib.vlmc_act_moments(big.data_ptr(), _dtype_code(big), 128, 96, in_f, in_f, 96 * in_f,
                                                        outm.data_ptr(), outm.data_ptr() + 4 * in_f, outm.data_ptr() + 8 * in_f,
                                                        _stream())))
    emit("`act_moments_kernel` (128 calls in one launch)", f"[128,96,{in_f}] fp16", us, f"{big.numel() * 2 / us / 1e3:.0f} GB/s")
    del big
    W0 = (torch.randn(out_f, in_f, device=dev) * 0.02).to(dt)
    W = W0.clone()
    for tag, kw in [("unstructured 50 %", dict(ratio=0.5)), ("2:4", dict(ratio=None, prune_n=2, prune_m=4))]:
        ratio = kw.pop("ratio")
        us = timeit(lambda: dsnot.prune_linear(W, st, ratio, max_cycle_time=100, update_threshold=0.1, **kw), reps=5,
                    setup=lambda: W.copy_(W0))
        emit(f"DSnoT `prune_linear` {tag} (select + `dsnot_lists_kernel` + `dsnot_apply_kernel`)", f"{out_f}x{in_f} fp16", us,
             f"whole linear, 100-cycle budget; {out_f * in_f * 5 / us / 1e3:.0f} GB/s of the 5 B/weight a select moves")

print("\n| kernel / step | shape | median us | note |\n|---|---|---|---|")
print("\n".join(rows))
