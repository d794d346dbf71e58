"""vlmc_attn_fwd against the unfused chain on this library's kernels and against vlmc_sdpa_fwd, at the shapes of a prune (us per call)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vlm-compression_amd"))
import torch
from vlmc import ops

dev = "cuda:0"


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(n):
        fn()
    ev[1].record()
    torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) * 1e3 / n


def heads(B, T, H, d, dtype):
    return torch.randn(B, T, H * d, device=dev).to(dtype).view(B, T, H, d).transpose(1, 2)


SHAPES = [("vit-g self (fp16)", torch.float16, 128, 16, 257, 257, 88, 0, False),
          ("t5 enc self 96 (bf16, bias + mask)", torch.bfloat16, 128, 32, 96, 96, 64, 1, True),
          ("t5 enc self 160 (bf16, bias + mask)", torch.bfloat16, 128, 32, 160, 160, 64, 1, True),
          ("t5 dec self 16", torch.bfloat16, 128, 32, 16, 16, 64, 1, True),
          ("t5 dec cross 16 x 160", torch.bfloat16, 128, 32, 16, 160, 64, 1, True),
          ("qformer self 64 (fp16, /8 + mask)", torch.float16, 128, 12, 64, 64, 64, 1, False),
          ("qformer cross 32 x 257", torch.float16, 128, 12, 32, 257, 64, 0, False)]
print("| shape | fused us | unfused chain us | sdpa kernel us (no bias: another op) | score bytes MB |")
print("|---|---|---|---|---|")
for name, dt, B, H, Tq, Tk, d, nadd, f32 in SHAPES:
    q, k, v = heads(B, Tq, H, d, dt), heads(B, Tk, H, d, dt), heads(B, Tk, H, d, dt)
    adds = [torch.randn(B, H, Tq, Tk, device=dev).to(dt)] if nadd else []
    mul = 0.125 if "qformer" in name else None

    def unfused():
        s = ops.attn_matmul(q, k.transpose(-1, -2))
        if mul is not None:
            s = s * mul
        for t in adds:
            s = s + t
        p = ops.softmax_rows(s.float()).type_as(s) if f32 else ops.softmax_rows(s)
        return ops.attn_matmul(p, v)
    tf = timeit(lambda: ops.attn_fused(q, k, v, mul, adds))
    tu = timeit(unfused)
    try:
        ts = timeit(lambda: ops.sdpa(q, k, v))
    except Exception:
        ts = float("nan")
    print(f"| {name} [{B}, {H}, {Tq}, {Tk}, {d}] | {tf:.1f} | {tu:.1f} | {ts:.1f} | {B * H * Tq * Tk * 2 / 1e6:.0f} |", flush=True)
