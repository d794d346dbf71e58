"""Which calls make the host wait for the GPU inside the capture phases of the bench headline's prune:
torch.cuda.set_sync_debug_mode("warn") + the Python stack of every warning, counted by call site."""
import collections
import os
import sys
import traceback
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vlm-compression_amd"))
import torch  # noqa: E402
from vlmc import synthetic  # noqa: E402
from lavis.compression.pruners import calibration as cal  # noqa: E402

dev = torch.device("cuda:0")
model = synthetic.InstructBlipT5(reference_ops=True).to(dev).eval()
batches = synthetic.calibration_batches(128, dev, vocab=32128, ragged=True)
for _ in range(2):
    dt, model, _ = synthetic.time_prune(dev, n_samples=128, model=model, batches=batches)
sites = collections.Counter()
phase = ["-"]


def show(message, category, filename, lineno, file=None, line=None):
    if "synchroniz" not in str(message):
        return
    st = [f for f in traceback.extract_stack()[:-1] if "vlm-compression_amd" in f.filename or "tools/" in f.filename or f.filename.endswith("bench.py")]
    key = " <- ".join(f"{os.path.basename(f.filename)}:{f.lineno}({f.name})" for f in reversed(st[-5:]))
    sites[(phase[0], key)] += 1


warnings.showwarning = show
warnings.simplefilter("always")
orig_capture, orig_walk = cal.capture_block_inputs, cal.walk_blocks


def wrap(name, fn, pos):
    def w(*a, **k):
        old, phase[0] = phase[0], name + " " + a[pos]
        try:
            return fn(*a, **k)
        finally:
            phase[0] = old
    return w


cal.capture_block_inputs = wrap("capture", orig_capture, 3)
cal.walk_blocks = wrap("walk", orig_walk, 4)
if len(sys.argv) > 1 and sys.argv[1] == "bench":          # the bench's own step, with its launch probes armed
    sys.path.insert(0, ROOT)
    import bench
    del model
    job = bench.PruneJob(dev, reference_ops=True, ragged=True)
    probe = bench.LaunchProbe(4)
    bench.install_probes(probe)
    for _ in range(2):
        job.step()
    probe.active = True
    torch.cuda.set_sync_debug_mode("warn")
    job.step()
    torch.cuda.set_sync_debug_mode("default")
else:
    torch.cuda.set_sync_debug_mode("warn")
    dt, model, _ = synthetic.time_prune(dev, n_samples=128, model=model, batches=batches)
    torch.cuda.set_sync_debug_mode("default")
for (ph, key), n in sorted(sites.items()):
    print(f"{ph:36s} x{n:4d}  {key}")
