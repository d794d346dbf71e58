"""vlmc_sdpa_fwd against torch's fused attention at the shapes of the calibration replay (1 x MI355X), interleaved rounds."""
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vlm-compression_amd"))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from vlmc import ops  # noqa: E402

dev = "cuda:0"


def timeit(fn, reps):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


print("| attention | B x H x Tq x Tk x d | vlmc_sdpa_fwd us | TFLOP/s | F.scaled_dot_product_attention us | TFLOP/s |")
print("|---|---|---|---|---|---|")
for name, B, H, Tq, Tk, d, dt in [("ViT-g, 128 samples", 128, 16, 257, 257, 88, torch.float16), ("ViT-g, 16 samples", 16, 16, 257, 257, 88, torch.float16),
                                  ("ViT-g, 1 sample", 1, 16, 257, 257, 88, torch.float16),
                                  ("T5 encoder self, 128", 128, 32, 64, 64, 64, torch.bfloat16), ("T5 decoder self, 128", 128, 32, 16, 16, 64, torch.bfloat16),
                                  ("T5 decoder cross, 128", 128, 32, 16, 64, 64, torch.bfloat16), ("T5 encoder self, 16", 16, 32, 64, 64, 64, torch.bfloat16),
                                  ("Vicuna, 128 x 96 tokens, causal", 128, 32, 96, 96, 128, torch.float16)]:
    # q, k, v as slices of one projection output, like the models make them
    qkv = (torch.randn(B, max(Tq, Tk), 3 * H * d, device=dev) * 0.5).to(dt)
    q, k, v = (t.reshape(B, -1, H, d).transpose(1, 2) for t in qkv.reshape(B, -1, 3, H * d).unbind(2))
    q = q[:, :, :Tq]
    k, v = k[:, :, :Tk], v[:, :, :Tk]
    causal = name.startswith("Vicuna")                     # (the LLaMA tower's self-attention is causal)
    ours, lib = [], []
    for _ in range(5):
        ours.append(timeit(lambda: ops.sdpa(q, k, v, causal=causal), 10))
        lib.append(timeit(lambda: F.scaled_dot_product_attention(q, k, v, is_causal=causal), 10))
    fl = 4.0 * B * H * Tq * Tk * d
    to, tl = statistics.median(ours), statistics.median(lib)
    print(f"| {name} {str(dt)[6:]} | {B} x {H} x {Tq} x {Tk} x {d} | {to * 1e3:.1f} | {fl / to / 1e9:.0f} | {tl * 1e3:.1f} | {fl / tl / 1e9:.0f} |", flush=True)
