"""fp32 batch-invariant GEMM (csrc/gemm_f32.hip) at the Q-Former's shapes: per-group launches of the merged capture and the
launches one padded pass over all 128 samples would make.  TFLOP/s against the 157.3 TFLOP/s fp32 MFMA peak."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vlm-compression_amd"))
import torch  # noqa: E402
from vlmc import ops  # noqa: E402

dev = torch.device("cuda:0")
shapes = [(4096, 768, 768), (4096, 768, 1408), (2048, 768, 768), (1024, 768, 768), (1024, 3072, 768), (700, 768, 768), (700, 3072, 768), (700, 768, 3072), (1800, 768, 1408), (224, 768, 768), (224, 3072, 768),
          (12800, 768, 768), (20480, 768, 768), (8704, 3072, 768), (8704, 768, 3072), (16384, 3072, 768), (4096, 3072, 768), (4096, 768, 3072),
          (32896, 768, 1408), (32896, 1536, 1408), (4096, 2048, 768)]
print("| M | N | K | us | TFLOP/s | of 157.3 |\n|---|---|---|---|---|---|")
for M, N, K in shapes:
    x = torch.randn(M, K, device=dev)
    w = torch.randn(N, K, device=dev) * 0.02
    b = torch.randn(N, device=dev)
    for _ in range(3):
        ops.linear_fwd(x, w, b)
    torch.cuda.synchronize()
    n = 30
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        ops.linear_fwd(x, w, b)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    tf = 2.0 * M * N * K / us / 1e6
    print(f"| {M} | {N} | {K} | {us:.1f} | {tf:.1f} | {tf / 157.3:.2f} |")
