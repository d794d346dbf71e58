import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vlm-compression_amd"))
import torch, torch.nn as nn
from vlmc import synthetic, forward
dev = torch.device("cuda:0")
for dim, heads, hidden, enc in [(64, 4, 128, 64), (768, 12, 3072, 1408)]:
    torch.manual_seed(0)
    layer = synthetic.QFormerLayer(dim, heads, hidden, enc, True).to(dev).half().eval()
    synthetic.randomize_(layer, 0, std=0.05)
    B, T, S, Q = 6, 12, 9, 4
    x = torch.randn(B, T, dim, device=dev).half()
    e = torch.randn(B, S, enc, device=dev).half()
    lins = [m for m in layer.modules() if type(m) is nn.Linear]
    inter = {}
    def hook(name):
        def f(mod, inp, out):
            inter.setdefault(name, []).append((out[0] if isinstance(out, tuple) else out).detach().clone())
        return f
    for n, m in layer.named_modules():
        if n:
            m.register_forward_hook(hook(n))
    with torch.no_grad(), forward.invariant_linears(lins, roots=[layer]):
        before = dict(forward.stats)
        ys = layer(x, None, e, None, query_length=Q)[0]
        stacked = {k: v[-1] for k, v in inter.items()}
        inter.clear()
        singles = []
        for i in range(B):
            singles.append(layer(x[i:i + 1], None, e[i:i + 1], None, query_length=Q)[0])
        print(dim, "stats delta", {k: forward.stats[k] - before.get(k, 0) for k in forward.stats if forward.stats[k] != before.get(k, 0)})
    print(dim, "layer output equal:", torch.equal(ys, torch.cat(singles)))
    for k in stacked:
        per = torch.cat(inter[k]) if inter[k][0].shape[0] == 1 else None
        if per is not None and per.shape == stacked[k].shape and not torch.equal(per, stacked[k]):
            print("   differs at", k, float((per.float() - stacked[k].float()).abs().max()))
