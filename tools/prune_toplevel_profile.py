"""cProfile of one warm prune of the bench headline, cumulative time of the pruner-level functions only (what runs outside
capture_block_inputs / walk_blocks): the prelude and the tail of prune().  `python tools/prune_toplevel_profile.py`"""
import cProfile
import os
import pstats
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vlm-compression_amd"))
import torch  # noqa: E402
from vlmc import synthetic  # noqa: E402

dev = torch.device("cuda:0")
model = synthetic.InstructBlipT5(reference_ops=True).to(dev).eval()
batches = synthetic.calibration_batches(128, dev, vocab=32128, ragged=True)
for _ in range(3):
    dt, model, _ = synthetic.time_prune(dev, n_samples=128, model=model, batches=batches)
print(f"warm prune {dt * 1e3:.1f} ms", flush=True)
pr = cProfile.Profile()
pr.enable()
dt, model, _ = synthetic.time_prune(dev, n_samples=128, model=model, batches=batches)
pr.disable()
print(f"profiled prune {dt * 1e3:.1f} ms")
st = pstats.Stats(pr)
rows = []
for (fn, line, name), (cc, nc, tt, ct, callers) in st.stats.items():
    if any(k in fn for k in ("wanda_pruner.py", "layer_single_base_pruner.py", "base_pruner.py", "pruners/utils.py", "synthetic.py")) or \
            name in ("capture_block_inputs", "walk_blocks", "_importance_readback", "quiet_gc", "collect"):
        rows.append((ct, tt, nc, os.path.basename(fn), line, name))
for ct, tt, nc, fn, line, name in sorted(rows, reverse=True)[:45]:
    print(f"{ct * 1e3:9.1f} ms cum {tt * 1e3:8.1f} ms own {nc:7d} calls  {fn}:{line}({name})")
