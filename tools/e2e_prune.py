"""End-to-end wall-clock of the drop-in pruners on the synthetic InstructBLIP-FlanT5-XL (random weights, true shapes).
    python tools/e2e_prune.py [wanda|dsnot|sparsegpt|mag ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vlm-compression_amd"))
import torch  # noqa: E402

from vlmc import synthetic  # noqa: E402

dev = torch.device("cuda:0")
names = sys.argv[1:] or ["wanda"]          # a name ending in "@vicuna" runs on the InstructBLIP-Vicuna-7B shapes
model = None
last_family = None
for name in names:
    name, _, family = name.partition("@")
    if family != last_family:
        model, last_family = None, family
    for label, env in (("default: grouped replay, invariant GEMM, tower batching", {}),
                       ("second run (warm)", {}),
                       ("library GEMMs in the replay (VLMC_LINEAR_FWD=0)", {"VLMC_LINEAR_FWD": "0"}),
                       ("per-sample loop from HIP graphs (VLMC_BATCH_REPLAY=1, VLMC_TOWER_BATCH=0)",
                        {"VLMC_BATCH_REPLAY": "1", "VLMC_TOWER_BATCH": "0"}),
                       ("the reference's eager per-sample loop (+ VLMC_GRAPH_REPLAY=0)",
                        {"VLMC_BATCH_REPLAY": "1", "VLMC_TOWER_BATCH": "0", "VLMC_GRAPH_REPLAY": "0"})):
        for k in ("VLMC_GRAPH_REPLAY", "VLMC_BATCH_REPLAY", "VLMC_TOWER_BATCH", "VLMC_LINEAR_FWD"):
            os.environ.pop(k, None)
        os.environ.update(env)
        kw = dict(is_global=True) if name in ("mag", "aobd") else {}
        if family == "vicuna":
            kw["t5_model_prefix"] = "llm_model"
        dt, model, info = synthetic.time_prune(dev, f"blipt5_{name}_pruner", model=model, **kw)
        print(f"{name + ('@' + family if family else ''):16s} {label:78s} {dt:8.2f} s   {info['linears'] / dt:8.1f} layers/s   pruned {info['pruned_fraction']:.4f}", flush=True)
