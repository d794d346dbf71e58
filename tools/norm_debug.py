import os, sys
sys.path.insert(0, "/root/repo/vlm-compression_amd")
sys.path.insert(0, "/root/repo/tests")
import torch
from vlmc import forward, ops
from test_rms_norm_gpu import T5StyleNorm, LlamaStyleNorm
dev = "cuda:0"
for cls, dt, n in [(T5StyleNorm, torch.float16, 1000), (T5StyleNorm, torch.float16, 1024), (T5StyleNorm, torch.bfloat16, 1000), (LlamaStyleNorm, torch.float16, 4096)]:
    m = cls(n).to(dt).to(dev)
    with torch.no_grad():
        m.weight.copy_((torch.randn(n) * 0.3 + 1.0).to(dt))
    g = torch.Generator(device=dev).manual_seed(1)
    x = (torch.randn(3, 37, n, generator=g, device=dev) * torch.tensor([0.02, 1.0, 30.0], device=dev)[:, None, None]).to(dt)
    with torch.no_grad(), forward.invariant_matmuls():
        want = m(x)
        for mode in (0, 1, 2):
            got = ops.rms_norm(x, m.weight.detach(), m.variance_epsilon, mode)
            d = (got != want)
            print(cls.__name__, dt, n, "mode", mode, "mismatches", int(d.sum()), "of", d.numel(), "rows with mismatch", int(d.any(-1).sum()),
                  "max abs diff", float((got.float() - want.float()).abs().max()))
        v = x.to(torch.float32).pow(2).mean(-1, keepdim=True)
        r = torch.rsqrt(v + m.variance_epsilon)
        for name, cand in (("1/sqrt", 1.0 / torch.sqrt(v + m.variance_epsilon)),):
            print("   torch.rsqrt vs", name, int((cand != r).sum()), "of", r.numel())
print("---- emulation")
for cls, dt, n in [(T5StyleNorm, torch.float16, 1000)]:
    m = cls(n).to(dt).to(dev)
    with torch.no_grad():
        m.weight.copy_((torch.randn(n) * 0.3 + 1.0).to(dt))
    g = torch.Generator(device=dev).manual_seed(1)
    x = (torch.randn(3, 37, n, generator=g, device=dev) * torch.tensor([0.02, 1.0, 30.0], device=dev)[:, None, None]).to(dt)
    with torch.no_grad(), forward.invariant_matmuls():
        want = m(x)
        v = x.to(torch.float32).pow(2).mean(-1, keepdim=True)
        r = torch.rsqrt(v + m.variance_epsilon)
        h = (x.float() * r).to(dt)
        y = (m.weight.float() * h.float()).to(dt)
        print("emulated (fp32 mul, round, fp32 mul, round) vs module:", int((y != want).sum()))
        h2 = (x * r).to(dt)
        y2 = m.weight * h2
        print("literal ops vs module:", int((y2 != want).sum()), " h vs h2:", int((h != h2).sum()))
        got = ops.rms_norm(x, m.weight.detach(), m.variance_epsilon, 2)
        bad = (got != want).nonzero()
        for idx in bad[:6]:
            i, j, k = idx.tolist()
            print("  at", (i, j, k), "x", float(x[i, j, k]), "r", float(r[i, j, 0]), "x*r", float(x[i, j, k].float() * r[i, j, 0]), "h", float(h[i, j, k]), "w", float(m.weight[k]),
                  "want", float(want[i, j, k]), "got", float(got[i, j, k]), "w*h fp32", float(m.weight[k].float() * h[i, j, k].float()))
