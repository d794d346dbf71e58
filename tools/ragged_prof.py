"""The bench headline's prune (reference-op stand-in, ragged text) N times through bench.PruneJob, nothing else around it: the
target of `rocprofv3 --kernel-trace --stats` in tools/collect_r06.sh (kernel time per prune = total / (N + 2))."""
import os, statistics, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from vlmc import forward
from lavis.compression.pruners import calibration as cal

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda:0")
job = bench.PruneJob(dev, reference_ops=True, ragged=True)
ts = []
for rep in range(n + 2):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    job.step()
    torch.cuda.synchronize()
    if rep >= 2:
        ts.append(time.perf_counter() - t0)
env = {k: v for k, v in os.environ.items() if k.startswith("VLMC_")}
print(f"{env}: median {statistics.median(ts) * 1e3:.1f} ms  min {min(ts) * 1e3:.1f}  ({' '.join(f'{x * 1e3:.0f}' for x in ts)})  prunes in this process: {n + 2}")
print("   ", {k: v for k, v in forward.stats.items() if v}, {k: v for k, v in cal.graph_stats.items() if v})
