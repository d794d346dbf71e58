"""Times vlmc_attn_matmul against torch.matmul on the attention shapes of the prune (and aligned neighbours of them):
`python tools/bench_attn.py`.  GB/s = algorithmic bytes (operands once + output once) / time."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vlm-compression_amd"))
import torch  # noqa: E402
from vlmc import ops  # noqa: E402

dev = "cuda:0"


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def eva(B, N, H, d, dtype=torch.float16):
    qkv = (torch.randn(B, N, 3 * H * d, device=dev) * 0.5).to(dtype).reshape(B, N, 3, H, d).permute(2, 0, 3, 1, 4)
    return qkv[0], qkv[1], qkv[2]


def t5(B, T, S, H, d, dtype=torch.bfloat16):
    def shape(t, L):
        return t.view(B, L, H, d).transpose(1, 2)
    return (shape((torch.randn(B, T, H * d, device=dev) * 0.5).to(dtype), T), shape((torch.randn(B, S, H * d, device=dev) * 0.5).to(dtype), S),
            shape((torch.randn(B, S, H * d, device=dev) * 0.5).to(dtype), S))


print("| case | shape | vlmc us | GB/s | torch.matmul us |")
print("|---|---|---|---|---|")
for name, (q, k, v) in (("ViT-g 257 tokens", eva(128, 257, 16, 88)), ("ViT-g 256 tokens (aligned rows)", eva(128, 256, 16, 88)),
                        ("ViT-g 264 tokens (16-B rows)", eva(128, 264, 16, 88)),
                        ("T5 enc 64", t5(128, 64, 64, 32, 64)), ("T5 dec self 16", t5(128, 16, 16, 32, 64)), ("T5 dec cross 16 x 64", t5(128, 16, 64, 32, 64))):
    kt = k.transpose(-2, -1)
    B, H, T, d = q.shape
    S = k.shape[2]
    sc = ops.attn_matmul(q, kt)
    p = torch.softmax(sc.float(), dim=-1).to(sc.dtype)
    us = timeit(lambda: ops.attn_matmul(q, kt))
    lib = timeit(lambda: torch.matmul(q, kt))
    by = 2 * B * H * (T * d + S * d + T * S)
    print(f"| {name} q k^T | {B}x{H}x{T}x{S}x{d} | {us:.1f} | {by / us / 1e3:.0f} | {lib:.1f} |")
    us = timeit(lambda: ops.attn_matmul(p, v))
    lib = timeit(lambda: torch.matmul(p, v))
    by = 2 * B * H * (T * S + S * d + T * d)
    print(f"| {name} attn v | {B}x{H}x{T}x{d}x{S} | {us:.1f} | {by / us / 1e3:.0f} | {lib:.1f} |")
