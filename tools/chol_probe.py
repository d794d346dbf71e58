import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "vlm-compression_amd"))
import torch
from vlmc import sparsegpt
torch.manual_seed(0)
X = torch.randn(4096, 128, device="cuda:0")
H = (X.t() @ X) / 4096 + 0.01 * torch.eye(128, device="cuda:0")
for _ in range(3):
    L, info = sparsegpt.blocked_cholesky(H.clone())
torch.cuda.synchronize()
print("info", int(info), "err", float((L @ L.t() - H).abs().max()))
