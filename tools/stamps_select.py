"""Diagnostic: per-phase cycle shares of the one-wave-per-row select kernel (VLMC_STAMPS build)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["VLMC_LIB"] = os.path.join(ROOT, "vlm-compression_amd/vlmc/libvlmc_hip_stamps.so")
sys.path.insert(0, os.path.join(ROOT, "vlm-compression_amd"))
import torch
from vlmc import ops
out_f, in_f = int(sys.argv[1]), int(sys.argv[2])
dev = "cuda:0"
W0 = (torch.randn(out_f, in_f, device=dev) * 0.02).to(torch.bfloat16)
s = ops.sqrt_scaler(torch.rand(in_f, device=dev) * 4 + 0.01)
mask = torch.empty(out_f, in_f, dtype=torch.bool, device=dev)
names = ["load issue", "keys(+load wait)", "sample guess", "radix search", "ties", "apply+stores", "-"]
for it in range(3):
    W = W0.clone()
    parts = torch.zeros(out_f, dtype=torch.float64, device=dev)
    ops.wanda_select(W, s, "row", k=in_f // 2, mask=mask, partials=parts)
    torch.cuda.synchronize()
    raw = parts.view(torch.int64)[:9].cpu().tolist()
    waves = raw[8]; tot = sum(raw[:7])
    print(f"waves {waves}  cycles/wave {tot/waves:.0f} (memtime ticks)  " + "  ".join(f"{n}: {100*v/tot:.1f}% ({v/waves:.0f})" for n, v in zip(names, raw[:7])))
