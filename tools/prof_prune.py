"""Host-side profile of one whole Wanda prune of the synthetic InstructBLIP-FlanT5-XL (cProfile, after two warm-up prunes)."""
import cProfile
import io
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "vlm-compression_amd"))
import torch  # noqa: E402

import bench  # noqa: E402

dev = torch.device("cuda:0")
job = bench.PruneJob(dev)
for _ in range(2):
    job.step()
torch.cuda.synchronize()
t0 = time.perf_counter()
job.step()
torch.cuda.synchronize()
print(f"plain step: {time.perf_counter() - t0:.3f} s")
pr = cProfile.Profile()
pr.enable()
job.step()
torch.cuda.synchronize()
pr.disable()
for key in ("cumulative", "tottime"):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(key).print_stats(45)
    print(s.getvalue()[:9000])
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_callers("named_modules|_named_members|__getattr__|named_parameters", 12)
print(s.getvalue()[:6000])

