"""Which route the Q-Former (a never-pruned tower between the vision tower and the language model) takes in the capture phases."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vlm-compression_amd"))
import torch
from vlmc import synthetic, forward, phases
from lavis.compression.pruners import calibration
dev = torch.device("cuda:0")
model = synthetic.InstructBlipT5().to(dev).eval()
batches = synthetic.calibration_batches(128, dev, vocab=32128, ragged=len(sys.argv) > 1 and sys.argv[1] == "ragged")
for rep in range(3):
    before = dict(calibration.graph_stats)
    dt, model, info = synthetic.time_prune(dev, n_samples=128, model=model, batches=batches)
    d = {k: v - before.get(k, 0) for k, v in calibration.graph_stats.items() if isinstance(v, (int, float)) and v != before.get(k, 0)}
    print(f"prune {rep}: {dt * 1e3:.1f} ms", json.dumps(d))
