"""Host and GPU timelines of the bench headline's prune, by phase, WITHOUT synchronising: when the host has issued a phase
(perf_counter) and when the GPU has finished it (an event recorded at the phase's end, read after the prune).
`python tools/phase_timeline.py [refops=1] [ragged=1]`"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vlm-compression_amd"))
import torch  # noqa: E402
from vlmc import synthetic  # noqa: E402
from lavis.compression.pruners import calibration as cal  # noqa: E402

refops = (sys.argv[1] if len(sys.argv) > 1 else "1") == "1"
ragged = (sys.argv[2] if len(sys.argv) > 2 else "1") == "1"
dev = torch.device("cuda:0")
model = synthetic.InstructBlipT5(reference_ops=refops).to(dev).eval()
batches = synthetic.calibration_batches(128, dev, vocab=32128, ragged=ragged)
for _ in range(3):
    dt, model, _ = synthetic.time_prune(dev, n_samples=128, model=model, batches=batches)
print(f"warm prune {dt * 1e3:.1f} ms", flush=True)
orig_capture, orig_walk = cal.capture_block_inputs, cal.walk_blocks
marks = []


def wrap(name, fn, pos):
    def w(*a, **k):
        t0 = time.perf_counter()
        r = fn(*a, **k)
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        marks.append((name + " " + a[pos], t0, time.perf_counter(), ev))
        return r
    return w


cal.capture_block_inputs = wrap("capture", orig_capture, 3)
cal.walk_blocks = wrap("walk", orig_walk, 4)
for rep in range(3):
    marks.clear()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True)
    e0.record()
    h0 = time.perf_counter()
    dt, model, _ = synthetic.time_prune(dev, n_samples=128, model=model, batches=batches)
    torch.cuda.synchronize()
    print(f"\nprune {rep}: {dt * 1e3:.1f} ms   (time_prune's own clock; restore of the weights precedes the phases)")
    print(f"{'phase':42s} {'host in':>8s} {'host out':>9s} {'host ms':>8s} {'GPU done':>9s} {'GPU ms':>7s} {'GPU behind host':>16s}")
    prev = None
    for name, t0, t1, ev in marks:
        g = e0.elapsed_time(ev)
        gp = g - (prev if prev is not None else 0.0)
        print(f"{name:42s} {(t0 - h0) * 1e3:8.1f} {(t1 - h0) * 1e3:9.1f} {(t1 - t0) * 1e3:8.1f} {g:9.1f} {gp:7.1f} {g - (t1 - h0) * 1e3:16.1f}")
        prev = g
    print("graph_stats", {k: v for k, v in cal.graph_stats.items() if v})
