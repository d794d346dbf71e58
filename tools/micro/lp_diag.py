import sys; sys.path.insert(0, "/root/repo/vlm-compression_amd")
import torch, torch.nn.functional as F
from vlmc import ops
DEV="cuda:0"
for (M,N,K) in [(300,1408,6144),(771,6144,1408),(64,2048,2048)]:
    g = torch.Generator(device=DEV).manual_seed(M + N + K)
    x = (torch.randn(M, K, generator=g, device=DEV) * 0.5).half()
    w = (torch.randn(N, K, generator=g, device=DEV) * 0.05).half()
    b = torch.randn(N, generator=g, device=DEV).half()
    y = ops.linear_fwd(x, w, b)
    got = ops.linear_fwd_post(x, w, b, act=1); want = F.gelu(y)
    d = (got.view(torch.int16) != want.view(torch.int16))
    print(M,N,K, int(d.sum()), "differ")
    if d.any():
        idx = d.nonzero()[:8]
        for r,c in idx.tolist():
            print("  ", r, c, float(y[r,c]), float(got[r,c]), float(want[r,c]))
        print("  rows:", sorted(set(idx[:,0].tolist()))[:10], "cols:", sorted(set(d.nonzero()[:,1].tolist()))[:20])
