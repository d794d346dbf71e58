"""Phase breakdown (host timers with syncs) of the Wanda prune on the reference-op stand-in with ragged calibration text."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vlm-compression_amd"))
import torch
from vlmc import synthetic
from lavis.compression.pruners import calibration as cal
import lavis.compression.pruners.wanda_pruner as wp

dev = torch.device("cuda:0")
T = {}


def timed(name, fn):
    def w(*a, **k):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = fn(*a, **k)
        torch.cuda.synchronize()
        T[name] = T.get(name, 0.0) + time.perf_counter() - t0
        return r
    return w


orig_capture, orig_walk = cal.capture_block_inputs, cal.walk_blocks
model = synthetic.InstructBlipT5(reference_ops=True).to(dev).eval()
batches = synthetic.calibration_batches(128, dev, vocab=32128, ragged=True)
for rep in range(3):
    instrument = rep == 2
    if instrument:
        def cap(model_, dl, n, mtp, *a, **k):
            return timed("capture " + mtp, orig_capture)(model_, dl, n, mtp, *a, **k)
        def walk(model_, inps, outs, caches, mtp, *a, **k):
            return timed("walk " + mtp, orig_walk)(model_, inps, outs, caches, mtp, *a, **k)
        cal.capture_block_inputs, cal.walk_blocks = cap, walk
    T.clear()
    dt, model, info = synthetic.time_prune(dev, n_samples=128, model=model, batches=batches)
    print(f"prune {rep}: {dt * 1e3:.1f} ms" + (" (with phase syncs)" if instrument else ""))
for k, v in T.items():
    print(f"   {k:45s} {v * 1e3:7.1f} ms")
