import os, sys, statistics, gc, time
sys.path.insert(0, "/root/repo/vlm-compression_amd")
import torch
from vlmc import synthetic
dev = torch.device("cuda:0")
ref = os.environ.get("REFOPS") == "1"
model = synthetic.InstructBlipT5(reference_ops=ref).to(dev).eval()
batches = synthetic.calibration_batches(128, dev, vocab=32128, ragged=ref)
for mode in ("gc on", "gc off", "gc on"):
    gc.enable() if mode == "gc on" else gc.disable()
    ts = []
    for rep in range(9):
        g0 = [s["collections"] for s in gc.get_stats()]
        dt, model, info = synthetic.time_prune(dev, n_samples=128, model=model, batches=batches)
        g1 = [s["collections"] for s in gc.get_stats()]
        ts.append((dt, [b - a for a, b in zip(g0, g1)]))
    print(mode, "median %.1f" % (statistics.median(t for t, _ in ts[2:]) * 1e3), " ".join(f"{t * 1e3:.0f}{c}" for t, c in ts[2:]), flush=True)
gc.enable()
