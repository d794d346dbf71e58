"""Does the sibling grouping (vlmc/forward.py) disturb the capture engine?  Three prunes, with and without phase timers:
wall-clock, phase sub-totals, the engine's counters."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vlm-compression_amd"))
import torch
from vlmc import synthetic, forward, phases
from lavis.compression.pruners import calibration as cal

dev = torch.device("cuda:0")
model = synthetic.InstructBlipT5().to(dev).eval()
batches = synthetic.calibration_batches(128, dev, vocab=model.t5_model.shared.num_embeddings)
for it in range(5):
    timers = it >= 3
    os.environ["VLMC_PHASE_TIMERS"] = "1" if timers else "0"
    phases.reset()
    g0 = dict(cal.graph_stats)
    f0 = dict(forward.stats)
    dt, _, info = synthetic.time_prune(dev, model=model, batches=batches)
    print(json.dumps({"it": it, "timers": timers, "s": round(dt, 4), "group": os.environ.get("VLMC_LINEAR_GROUP", "1"),
                      "forward": {k: forward.stats[k] - f0[k] for k in f0}, "graph_stats": {k: v - g0.get(k, 0) for k, v in cal.graph_stats.items() if v != g0.get(k, 0)},
                      "phases": {k: round(v, 4) for k, v in phases.times.items()}}), flush=True)
