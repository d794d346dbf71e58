import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vlm-compression_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, torch.nn as nn
import golden_io
from vlmc import sparsegpt as SG
G = golden_io.load("sparsegpt")
for name in sorted({k.split("/")[0] for k in G}):
    W, xs = G[f"{name}/W"], G[f"{name}/xs"].to(G[f"{name}/W"].dtype)
    lin = nn.Linear(W.shape[1], W.shape[0], bias=False); lin.weight.data = W.clone(); lin = lin.to("cuda:0")
    sg = SG.SparseGPT(lin)
    for x in xs: sg.add_batch(x[None].to("cuda:0"), None)
    SG.fasterprune(lin, sg.H, float(G[f"{name}/sparsity"]), int(G[f"{name}/n"]), int(G[f"{name}/m"]))
    got, ref = lin.weight.data.cpu().float(), G[f"{name}/Wn"].float()
    print(name, "rel err %.5f" % float((got - ref).norm() / ref.norm()), "zero agree %.4f" % ((got == 0) == (ref == 0)).float().mean().item(), SG.factor_stats)
