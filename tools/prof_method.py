"""Host-side profile (cProfile) + phase timers of one whole prune of the synthetic InstructBLIP-FlanT5-XL by any drop-in pruner.
    python tools/prof_method.py dsnot|sparsegpt|wanda[@vicuna]"""
import cProfile
import io
import os
import pstats
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vlm-compression_amd"))
import torch  # noqa: E402

from vlmc import phases, synthetic  # noqa: E402

dev = torch.device("cuda:0")
name, _, family = (sys.argv[1] if len(sys.argv) > 1 else "dsnot").partition("@")
kw = {"t5_model_prefix": "llm_model"} if family == "vicuna" else {}
model = None
for _ in range(2):
    dt, model, info = synthetic.time_prune(dev, f"blipt5_{name}_pruner", model=model, **kw)
print(f"warm prune: {dt:.3f} s")
os.environ["VLMC_PHASE_TIMERS"] = "1"
phases.reset()
dt, model, info = synthetic.time_prune(dev, f"blipt5_{name}_pruner", model=model, **kw)
print(f"with phase timers (a device sync at each phase boundary): {dt:.3f} s", {k: round(v, 4) for k, v in phases.times.items()})
os.environ.pop("VLMC_PHASE_TIMERS")
pr = cProfile.Profile()
pr.enable()
dt, model, info = synthetic.time_prune(dev, f"blipt5_{name}_pruner", model=model, **kw)
pr.disable()
print(f"under cProfile: {dt:.3f} s")
for key in ("cumulative", "tottime"):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(key).print_stats(40)
    print(s.getvalue()[:8000])
