"""Where the HOST spends the prune of the reference-op stand-in with ragged text (15 000 linear launches, ~48 000 eager kernels):
wall-clock, per-call host cost of the vlmc.ops entry points, cProfile top lists.  `python tools/ragged_host_profile.py [ragged=1] [refops=1]`"""
import cProfile
import io
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vlm-compression_amd"))
import torch  # noqa: E402
from vlmc import forward, ops, synthetic  # noqa: E402

ragged = (sys.argv[1] if len(sys.argv) > 1 else "1") == "1"
refops = (sys.argv[2] if len(sys.argv) > 2 else "1") == "1"
dev = torch.device("cuda:0")
model = synthetic.InstructBlipT5(reference_ops=refops).to(dev).eval()
batches = synthetic.calibration_batches(128, dev, vocab=model.t5_model.shared.num_embeddings, ragged=ragged)
for _ in range(3):
    dt, _, _ = synthetic.time_prune(dev, model=model, batches=batches)
ts = []
for _ in range(3):
    dt, _, _ = synthetic.time_prune(dev, model=model, batches=batches)
    ts.append(dt)
print(f"ragged={ragged} reference_ops={refops}: warm prunes ms", [round(t * 1e3, 1) for t in ts], flush=True)

# host microseconds per call of the hot entry points (wrapper + ctypes + launch), measured by wrapping them
acc = {}


def timed(mod, name):
    fn = getattr(mod, name)

    def w(*a, **k):
        t0 = time.perf_counter()
        r = fn(*a, **k)
        e = acc.setdefault(name, [0, 0.0])
        e[0] += 1
        e[1] += time.perf_counter() - t0
        return r
    setattr(mod, name, w)
    return fn


saved = {n: timed(ops, n) for n in ("linear_fwd", "linear_fwd_group", "attn_matmul", "act_sqnorm_batch", "wanda_select_batch",
                                    "wanda_scaler_update_batch")}
f0 = dict(forward.stats)
dt, _, _ = synthetic.time_prune(dev, model=model, batches=batches)
for n, fn in saved.items():
    setattr(ops, n, fn)
print(f"with per-call timers: {dt * 1e3:.1f} ms; forward stats", {k: forward.stats[k] - f0[k] for k in f0})
for n, (c, t) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print(f"  {n:28s} {c:7d} calls  {t * 1e3:8.1f} ms  {t / c * 1e6:6.1f} us/call")

pr = cProfile.Profile()
pr.enable()
dt, _, _ = synthetic.time_prune(dev, model=model, batches=batches)
pr.disable()
print(f"under cProfile: {dt * 1e3:.1f} ms")
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(40)
print(s.getvalue()[:9000])
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(40)
print(s.getvalue()[:9000])
