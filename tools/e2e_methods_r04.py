"""Whole-prune wall-clock of every drop-in pruner on the final build (warm third run): the README's method row."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vlm-compression_amd"))
import torch
from vlmc import synthetic
dev = torch.device("cuda:0")
for name, family, kw in (("wanda", "", {}), ("dsnot", "", {}), ("sparsegpt", "", {}), ("sparsegpt", "", {"prune_n": 2, "prune_m": 4}),
                         ("wanda", "vicuna", {"t5_model_prefix": "llm_model"}), ("dsnot", "vicuna", {"t5_model_prefix": "llm_model"})):
    model = None
    ts = []
    for it in range(3):
        dt, model, info = synthetic.time_prune(dev, f"blipt5_{name}_pruner", model=model, **kw)
        ts.append(dt)
    if ts:
        print(f"{name + ('@' + family if family else ''):18s} {kw}  runs {[round(t, 3) for t in ts]} s   pruned {info['pruned_fraction']:.4f}", flush=True)
    del model
    torch.cuda.empty_cache()
