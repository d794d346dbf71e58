"""Phase sub-totals and engine counters of a prune of the reference-op stand-in (vlmc/synthetic.py reference_ops=True, ragged
calibration text).  `python tools/refops_probe.py [ragged=1] [reference_ops=1]`; under rocprofv3 --kernel-trace --stats for
the per-kernel table."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vlm-compression_amd"))
import torch  # noqa: E402
from vlmc import forward, phases, synthetic  # noqa: E402
from lavis.compression.pruners import calibration as cal  # noqa: E402

ragged = (sys.argv[1] if len(sys.argv) > 1 else "1") == "1"
refops = (sys.argv[2] if len(sys.argv) > 2 else "1") == "1"
dev = torch.device("cuda:0")
model = synthetic.InstructBlipT5(reference_ops=refops).to(dev).eval()
batches = synthetic.calibration_batches(128, dev, vocab=model.t5_model.shared.num_embeddings, ragged=ragged)
for it in range(4):
    timers = it == 3
    os.environ["VLMC_PHASE_TIMERS"] = "1" if timers else "0"
    phases.reset()
    g0, f0 = dict(cal.graph_stats), dict(forward.stats)
    dt, _, info = synthetic.time_prune(dev, model=model, batches=batches)
    print(json.dumps({"it": it, "ragged": ragged, "reference_ops": refops, "timers": timers, "s": round(dt, 4),
                      "forward": {k: forward.stats[k] - f0[k] for k in f0},
                      "graph_stats": {k: v - g0.get(k, 0) for k, v in cal.graph_stats.items() if v != g0.get(k, 0)},
                      "phases": {k: round(v, 4) for k, v in phases.times.items()}}), flush=True)
