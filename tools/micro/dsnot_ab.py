import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vlm-compression_amd"))
import torch
from vlmc import synthetic
dev = torch.device("cuda:0")
for qf in (True, False):
    model = synthetic.InstructBlipT5(qformer=qf).to(dev).eval()
    ts = []
    for it in range(4):
        dt, model, info = synthetic.time_prune(dev, "blipt5_dsnot_pruner", model=model)
        ts.append(round(dt, 3))
    print("dsnot qformer", qf, ts, flush=True)
    if qf:
        os.environ["VLMC_PHASE_TIMERS"] = "1"
        from vlmc import phases
        dt, model, info = synthetic.time_prune(dev, "blipt5_dsnot_pruner", model=model)
        print("   phases:", {k: round(v, 4) for k, v in getattr(phases, "totals", lambda: {})().items()} if hasattr(phases, "totals") else info.get("phases"))
        os.environ.pop("VLMC_PHASE_TIMERS")
    del model
    torch.cuda.empty_cache()
