"""One process standing for rank 0 of W ranks (VLMC_SIMULATE_WORLD) pruning 6 times, towers synchronised and announced by a
marker kernel count: run under `rocprofv3 --kernel-trace --output-format csv`, then tools/rank_busy.py splits the trace by
tower (the boundaries are printed here as host timestamps relative to the first prune)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vlm-compression_amd"))
W = int(sys.argv[1]) if len(sys.argv) > 1 else 8
if W > 1:
    os.environ["VLMC_SIMULATE_WORLD"] = str(W)
import torch  # noqa: E402
from vlmc import synthetic  # noqa: E402
from lavis.compression.pruners import wanda_pruner as WP  # noqa: E402

dev = torch.device("cuda:0")
model = synthetic.InstructBlipT5().to(dev).eval()
batches = synthetic.calibration_batches(128, dev, vocab=model.t5_model.shared.num_embeddings)
marks = []
real = WP.BLIPT5LayerWandaPruner._tower
mark_buf = torch.zeros(7, device=dev)


def timed(self, cls, **kw):
    from vlmc import phases
    torch.cuda.synchronize()
    mark_buf.cumsum_(0)                       # a kernel that appears nowhere else: the tower's first
    before = dict(phases.times)
    t0 = time.perf_counter()
    out = real(self, cls, **kw)
    torch.cuda.synchronize()
    marks.append((kw["module_to_process"], round((time.perf_counter() - t0) * 1e3, 2)))
    mark_buf.cumsum_(0)                       # ... and its last
    if phases.enabled():
        marks.append({k: round((v - before.get(k, 0.0)) * 1e3, 1) for k, v in phases.times.items() if v - before.get(k, 0.0) > 1e-4})
    return out


WP.BLIPT5LayerWandaPruner._tower = timed
for it in range(int(os.environ.get("RANK_TIMELINE_ITERS", "7"))):
    marks.clear()
    os.environ["VLMC_PHASE_TIMERS"] = "1" if it == 6 else "0"
    dt, _, _ = synthetic.time_prune(dev, model=model, batches=batches)
    print(json.dumps({"it": it, "prune_ms": round(dt * 1e3, 1), "towers": marks}), flush=True)
