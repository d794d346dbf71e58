"""cProfile of one whole prune by a process standing for rank 0 of W ranks (VLMC_SIMULATE_WORLD): where the HOST time of the
per-rank floor goes.  `python tools/rank_host_profile.py [W=8]`"""
import cProfile
import io
import os
import pstats
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vlm-compression_amd"))
W = int(sys.argv[1]) if len(sys.argv) > 1 else 8
if W > 1:
    os.environ["VLMC_SIMULATE_WORLD"] = str(W)
import torch  # noqa: E402
from vlmc import synthetic  # noqa: E402

dev = torch.device("cuda:0")
model = synthetic.InstructBlipT5().to(dev).eval()
batches = synthetic.calibration_batches(128, dev, vocab=model.t5_model.shared.num_embeddings)
for _ in range(3):
    dt, _, _ = synthetic.time_prune(dev, model=model, batches=batches)
    print("prune ms", round(dt * 1e3, 1), flush=True)
pr = cProfile.Profile()
pr.enable()
dt, _, _ = synthetic.time_prune(dev, model=model, batches=batches)
pr.disable()
print("profiled prune ms", round(dt * 1e3, 1))
for key, n in (("tottime", 45), ("cumulative", 60)):
    buf = io.StringIO()
    pstats.Stats(pr, stream=buf).sort_stats(key).print_stats(n)
    print(buf.getvalue())
buf = io.StringIO()
st = pstats.Stats(pr, stream=buf)
for fn in ("named_modules", "_named_members", "method 'to'", "__setattr__", "find_layers", "named_parameters", "torch.empty", "torch.zeros"):
    st.print_callers(fn)
print(buf.getvalue()[:20000])
