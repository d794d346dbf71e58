"""Where a warm Wanda prune of the synthetic InstructBLIP-FlanT5-XL makes the host wait for the GPU (torch.cuda.set_sync_debug_mode)."""
import collections, os, sys, traceback, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vlm-compression_amd"))
import torch
from vlmc import synthetic
dev = torch.device("cuda:0")
ref = os.environ.get("REFOPS") == "1"
model = synthetic.InstructBlipT5(reference_ops=ref).to(dev).eval()
batches = synthetic.calibration_batches(128, dev, vocab=32128, ragged=ref)
for _ in range(2):
    dt, model, info = synthetic.time_prune(dev, n_samples=128, model=model, batches=batches)
seen = collections.Counter()


def showwarning(message, category, filename, lineno, file=None, line=None):
    if "synchroniz" not in str(message):
        return
    st = [f for f in traceback.extract_stack() if "vlm-compression_amd" in f.filename or "bench" in f.filename][-3:]
    seen[" <- ".join(f"{os.path.basename(f.filename)}:{f.lineno}({f.name})" for f in reversed(st))] += 1


warnings.showwarning = showwarning
warnings.simplefilter("always")
torch.cuda.set_sync_debug_mode("warn")
dt, model, info = synthetic.time_prune(dev, n_samples=128, model=model, batches=batches)
torch.cuda.set_sync_debug_mode("default")
print(f"prune {dt * 1e3:.1f} ms; synchronising calls: {sum(seen.values())}")
for k, v in seen.most_common(30):
    print(f"{v:5d}  {k}")
