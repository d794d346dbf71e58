"""Host profile of a warm SparseGPT prune of the synthetic InstructBLIP-FlanT5-XL (cProfile, cumulative + internal)."""
import cProfile, io, os, pstats, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vlm-compression_amd"))
import torch
from vlmc import synthetic, sparsegpt
dev = torch.device("cuda:0")
model = synthetic.InstructBlipT5().to(dev).eval()
batches = synthetic.calibration_batches(128, dev, vocab=model.t5_model.shared.num_embeddings)
extra = {"prune_n": 2, "prune_m": 4} if (len(sys.argv) > 1 and sys.argv[1] == "2:4") else {}
for _ in range(2):
    dt, _, info = synthetic.time_prune(dev, "blipt5_sparsegpt_pruner", model=model, batches=batches, **extra)
    print(f"prune {dt:.2f} s  routes {sparsegpt.factor_stats}", flush=True)
pr = cProfile.Profile()
pr.enable()
dt, _, _ = synthetic.time_prune(dev, "blipt5_sparsegpt_pruner", model=model, batches=batches, **extra)
pr.disable()
print(f"under cProfile: {dt:.2f} s")
for key, n in (("cumulative", 40), ("tottime", 25)):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(key).print_stats(n)
    print(s.getvalue()[:8000])
