"""torch.sort vs torch.kthvalue vs torch.topk for SparseGPT's per-block threshold (sparsegpt_pruner.py:184)."""
import time, torch
dev = "cuda:0"
for rows in (2048, 5120, 1408, 6144):
    x = torch.rand(rows, 128, device=dev)
    k = int(x.numel() * 0.5)
    for name, fn in (("sort", lambda: torch.sort(x.flatten())[0][k]), ("kthvalue", lambda: torch.kthvalue(x.flatten(), k + 1).values)):
        for _ in range(3):
            v = fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            v = fn()
        torch.cuda.synchronize()
        print(rows, name, f"{(time.perf_counter() - t0) / 50 * 1e6:.1f} us", float(v))
