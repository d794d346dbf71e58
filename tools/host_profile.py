"""Where the HOST spends a prune when the GPU has little to do (one rank's share of an 8-GPU run, VLMC_SIMULATE_WORLD=8):
wall-clock, host time to ISSUE the prune (no sync inside), cProfile top list.  `python tools/host_profile.py [world]`"""
import cProfile
import io
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vlm-compression_amd"))
import torch  # noqa: E402
from vlmc import synthetic  # noqa: E402

world = sys.argv[1] if len(sys.argv) > 1 else "8"
os.environ["VLMC_SIMULATE_WORLD"] = world
dev = torch.device("cuda:0")
model = synthetic.InstructBlipT5().to(dev).eval()
batches = synthetic.calibration_batches(128, dev, vocab=model.t5_model.shared.num_embeddings)
for _ in range(3):
    dt, _, _ = synthetic.time_prune(dev, model=model, batches=batches)
print(f"simulated world {world}: warm prune {dt * 1e3:.1f} ms")
ts = []
for _ in range(3):
    dt, _, _ = synthetic.time_prune(dev, model=model, batches=batches)
    ts.append(dt)
print("wall ms:", [round(t * 1e3, 1) for t in ts])
pr = cProfile.Profile()
pr.enable()
dt, _, _ = synthetic.time_prune(dev, model=model, batches=batches)
pr.disable()
print(f"under cProfile: {dt * 1e3:.1f} ms")
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
print(s.getvalue()[:9000])
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(30)
print(s.getvalue()[:7000])
buf = io.StringIO()
st = pstats.Stats(pr, stream=buf)
st.print_callers(r"module.py:\d+\((parameters|named_parameters|named_modules|_named_members)\)")
print(buf.getvalue()[:6000])
