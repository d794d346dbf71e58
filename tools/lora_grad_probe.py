"""Time vlmc_lora_grad (fused / streaming gradient pass of SparseLoRA) with dA and / or dB switched off, per shape."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vlm-compression_amd"))
import torch
from vlmc import sparse_lora, _lib
dev = "cuda:0"


def timeit(fn, reps=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for out_f, in_f in ((4096, 4096), (11008, 4096), (4096, 11008), (16384, 4096)):
    G = torch.randn(out_f, in_f, device=dev).half()
    M = torch.rand(out_f, in_f, device=dev) > 0.5
    A = torch.randn(16, in_f, device=dev) * 0.1
    B = torch.randn(out_f, 16, device=dev) * 0.1
    row = [f"{out_f}x{in_f}"]
    for na, nb in ((True, True), (True, False), (False, True)):
        us = timeit(lambda: sparse_lora.lora_grads(G, A, B, M, 1.0, True, _lib.F16, need_A=na, need_B=nb))
        row.append(f"dA={int(na)} dB={int(nb)}: {us:6.1f} us ({3 * out_f * in_f / us / 1e3:5.0f} GB/s)")
    print(" | ".join(row), flush=True)
