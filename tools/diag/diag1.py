import sys; sys.path.insert(0,'vlm-compression_amd'); sys.path.insert(0,'.')
import numpy as np, torch
from vlmc import ops
rng = np.random.default_rng(1)
x = (rng.standard_normal((1, 1, 1 << 16)) * 10.0 ** rng.integers(-22, 18, (1, 1, 1 << 16))).astype(np.float32)
got = ops.act_sqnorm(torch.from_numpy(x).to('cuda')).cpu().numpy()[0]
sq=(x[0,0]*x[0,0]).astype(np.float32)
r = np.sqrt(sq, dtype=np.float32); want=r*r
bad = np.nonzero(got.view(np.uint32)!=want.view(np.uint32))[0]
print(len(bad))
for i in bad[:20]:
    print(x[0,0,i], sq[i], r[i], want[i], got[i], hex(got.view(np.uint32)[i]), hex(want.view(np.uint32)[i]))
