cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
VLMC_CROSSCHECK=all timeout -k 10 900 python - <<'PY' 2>&1 | grep -v amdgpu | tail -8
import sys, os
sys.path.insert(0, "/root/repo/vlm-compression_amd")
import torch
from vlmc import crosscheck, synthetic
print("routes applied:", sorted(k for k in os.environ if k.startswith("VLMC_"))[:60])
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = synthetic.InstructBlipT5(vit_dim=256, vit_hidden=512, vit_heads=4, vit_depth=3, d_model=256, d_ff=512, heads=4, d_kv=64, enc_depth=3, dec_depth=3, vocab=512, query_tokens=8).to(dev).eval()
synthetic.randomize_(model, 0)
batches = synthetic.calibration_batches(8, dev, vit_tokens=33, vit_dim=256, text_len=9, out_len=5, vocab=512)
for name in ("wanda", "dsnot", "sparsegpt"):
    dt, m, info = synthetic.time_prune(dev, f"blipt5_{name}_pruner", model=None if name != "wanda" else model, batches=batches if name == "wanda" else None, n_samples=8 if name == "wanda" else 128,
                                       **(dict(t5_prune_spec="3-0.5-1.0-1.0", vit_prune_spec="3-0.5-1.0-1.0") if name == "wanda" else {})) if name == "wanda" else (None, None, None)
    if dt is not None:
        print(name, "crosscheck=all prune ok", round(dt, 3), "s pruned", round(info["pruned_fraction"], 4))
PY
