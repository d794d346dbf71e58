"""GPU-bound time of every phase of the bench headline's prune: the GPU is held by a spin kernel while the host issues the phase,
so the phase's kernels run back to back once it is released; (event after the spin) -> (event at the phase's end).  Phases that
wait for the GPU inside themselves (a capture phase reads its verdicts at the end) show host time = spin + GPU time.
`python tools/phase_gpu_bound.py [spin_ms=250]`"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vlm-compression_amd"))
import torch  # noqa: E402
from vlmc import synthetic  # noqa: E402
from lavis.compression.pruners import calibration as cal  # noqa: E402

spin_ms = float(sys.argv[1]) if len(sys.argv) > 1 else 250.0
dev = torch.device("cuda:0")
model = synthetic.InstructBlipT5(reference_ops=True).to(dev).eval()
batches = synthetic.calibration_batches(128, dev, vocab=32128, ragged=True)
for _ in range(3):
    dt, model, _ = synthetic.time_prune(dev, n_samples=128, model=model, batches=batches)
print(f"warm prune {dt * 1e3:.1f} ms", flush=True)
# cycles per ms of torch.cuda._sleep
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize()
a.record()
torch.cuda._sleep(20_000_000)
b.record()
torch.cuda.synchronize()
per_ms = 20_000_000 / a.elapsed_time(b)
print(f"_sleep: {per_ms:.0f} cycles per ms")
orig_capture, orig_walk = cal.capture_block_inputs, cal.walk_blocks
marks = []


def wrap(name, fn, pos):
    def w(*a_, **k):
        torch.cuda._sleep(int(spin_ms * per_ms))
        e0 = torch.cuda.Event(enable_timing=True)
        e0.record()
        t0 = time.perf_counter()
        r = fn(*a_, **k)
        t1 = time.perf_counter()
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        marks.append((name + " " + a_[pos], t1 - t0, e0, e1))
        return r
    return w


cal.capture_block_inputs = wrap("capture", orig_capture, 3)
cal.walk_blocks = wrap("walk", orig_walk, 4)
for rep in range(2):
    marks.clear()
    dt, model, _ = synthetic.time_prune(dev, n_samples=128, model=model, batches=batches)
    torch.cuda.synchronize()
    print(f"\nprune {rep} (with a {spin_ms:.0f} ms spin in front of every phase)")
    tot = 0.0
    for name, host, e0, e1 in marks:
        g = e0.elapsed_time(e1)
        tot += g
        print(f"{name:42s} host {host * 1e3:7.1f} ms (issue, or spin + GPU when the phase waits)   GPU back to back {g:7.1f} ms")
    print(f"{'sum of the GPU-bound phase times':42s} {tot:7.1f} ms")
