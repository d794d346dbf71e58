"""Whole Wanda prunes on the reference-op stand-in with ragged calibration text, one configuration per process: median / min of N prunes."""
import os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vlm-compression_amd"))
import torch
from vlmc import synthetic, forward
from lavis.compression.pruners import calibration as cal

dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 7
model = synthetic.InstructBlipT5(reference_ops=True).to(dev).eval()
batches = synthetic.calibration_batches(128, dev, vocab=32128, ragged=True)
ts = []
for rep in range(n + 2):
    dt, model, info = synthetic.time_prune(dev, n_samples=128, model=model, batches=batches)
    if rep >= 2:
        ts.append(dt)
env = {k: v for k, v in os.environ.items() if k.startswith("VLMC_")}
print(f"{env}: median {statistics.median(ts) * 1e3:.1f} ms  min {min(ts) * 1e3:.1f}  ({' '.join(f'{x * 1e3:.0f}' for x in ts)})")
print("   ", {k: v for k, v in forward.stats.items() if k.startswith("attn") or k.startswith("softmax")},
      {k: v for k, v in cal.graph_stats.items() if "tower" in k or "padded" in k or "shared" in k or "merged" in k or "memo" in k})
