import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vlm-compression_amd"))
import torch, torch.nn as nn
from vlmc import synthetic
from lavis.compression.pruners import calibration
dev = torch.device("cuda:0")
FROZEN = calibration.FROZEN_TOWERS

def run(frozen, per_sample, qformer=True):
    calibration.FROZEN_TOWERS = frozen
    os.environ["VLMC_BATCH_REPLAY"] = "1" if per_sample else "128"
    os.environ["VLMC_TOWER_BATCH"] = "0" if per_sample else "1"
    torch.manual_seed(0)
    model = synthetic.InstructBlipT5(vit_dim=64, vit_hidden=128, vit_heads=4, vit_depth=2, d_model=64, d_ff=128, heads=4, d_kv=16,
                                     enc_depth=2, dec_depth=2, vocab=100, query_tokens=4, qformer=qformer, qformer_dim=64, qformer_heads=4, qformer_hidden=128,
                                     qformer_depth=4, qformer_vocab=50).to(dev).eval()
    batches = synthetic.calibration_batches(12, dev, vit_tokens=9, vit_dim=64, text_len=5, out_len=3, vocab=100)
    synthetic.time_prune(dev, n_samples=12, model=model, batches=batches)
    return {n: m.mask.clone() for n, m in model.named_modules() if isinstance(m, nn.Linear) and hasattr(m, "mask")}

def diff(a, b):
    bad = [k for k in a if not torch.equal(a[k], b[k])]
    return len(bad), bad[:3]

r = {}
for name, args in [("noq/grouped", ((), False, False)), ("noq/per", ((), True, False)), ("eager/grouped", ((), False)), ("eager/per", ((), True)),
                   ("frozen/grouped", (FROZEN, False)), ("frozen/per", (FROZEN, True))]:
    r[name] = run(*args)
print("no Q-Former: grouped vs per-sample", diff(r["noq/grouped"], r["noq/per"]))
print("eager Q-Former: grouped vs per-sample", diff(r["eager/grouped"], r["eager/per"]))
print("frozen: grouped vs per-sample", diff(r["frozen/grouped"], r["frozen/per"]))
print("frozen/grouped vs eager/grouped", diff(r["frozen/grouped"], r["eager/grouped"]))
a = run((), False, False); b = run((), False, False)
print("same config twice (no Q-Former, grouped):", diff(a, b))
a = run((), True, False); b = run((), True, False)
print("same config twice (no Q-Former, per-sample):", diff(a, b))
