"""A/B of environment switches that are read per prune, inside ONE process: whole Wanda prunes of the synthetic
InstructBLIP-FlanT5-XL, the configurations interleaved, median of the rounds.
   python tools/ab_prune_env.py "VLMC_ROW_MEAN=1" "VLMC_ROW_MEAN=0" "VLMC_ATTN_MATMUL=0,VLMC_ROW_MEAN=0" """
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vlm-compression_amd"))
import torch  # noqa: E402
from vlmc import synthetic  # noqa: E402

configs = [dict(kv.split("=") for kv in c.split(",") if kv) for c in sys.argv[1:]] or [{}]
world = os.environ.get("AB_SIMULATE_WORLD")
dev = torch.device("cuda:0")
refops = os.environ.get("AB_REFERENCE_OPS") == "1"            # the reference's attention op sequence + ragged calibration text
model = synthetic.InstructBlipT5(reference_ops=refops).to(dev).eval()
batches = synthetic.calibration_batches(128, dev, vocab=model.t5_model.shared.num_embeddings, ragged=refops)
keys = sorted({k for c in configs for k in c})
times = [[] for _ in configs]
for rnd in range(int(os.environ.get("AB_ROUNDS", "6")) + 1):
    for i, c in enumerate(configs):
        for k in keys:
            os.environ.pop(k, None)
        os.environ.update(c)
        if world:
            os.environ["VLMC_SIMULATE_WORLD"] = world
        dt, _, _ = synthetic.time_prune(dev, model=model, batches=batches)
        if rnd:
            times[i].append(dt)
for c, t in zip(configs, times):
    print(f"{c}: median {statistics.median(t) * 1e3:.1f} ms  min {min(t) * 1e3:.1f}  ({' '.join(f'{x * 1e3:.0f}' for x in t)})", flush=True)
