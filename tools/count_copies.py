"""Who issues device-to-device copies during one whole Wanda prune (clone / copy_ / contiguous / to by caller line)."""
import collections
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "vlm-compression_amd"))
import torch  # noqa: E402

import bench  # noqa: E402

dev = torch.device("cuda:0")
job = bench.PruneJob(dev)
for _ in range(2):
    job.step()
torch.cuda.synchronize()
stats = collections.defaultdict(lambda: [0, 0])


def caller():
    for f in reversed(traceback.extract_stack()[:-2]):
        if "count_copies" not in f.filename and "/torch/" not in f.filename:
            return f"{os.path.relpath(f.filename, ROOT)}:{f.lineno}"
    return "?"


def wrap(cls, name, nbytes):
    real = getattr(cls, name)

    def w(self, *a, **k):
        out = real(self, *a, **k)
        try:
            if isinstance(self, torch.Tensor) and self.is_cuda:
                n = nbytes(self, out, a)
                if n:
                    ent = stats[(name, caller())]
                    ent[0] += 1
                    ent[1] += n
        except Exception:
            pass
        return out
    setattr(cls, name, w)


wrap(torch.Tensor, "clone", lambda s, o, a: s.numel() * s.element_size())
wrap(torch.Tensor, "copy_", lambda s, o, a: s.numel() * s.element_size())
wrap(torch.Tensor, "contiguous", lambda s, o, a: 0 if o.data_ptr() == s.data_ptr() else s.numel() * s.element_size())
wrap(torch.Tensor, "to", lambda s, o, a: 0 if o.data_ptr() == s.data_ptr() else o.numel() * o.element_size())
wrap(torch.Tensor, "float", lambda s, o, a: 0 if o.data_ptr() == s.data_ptr() else o.numel() * o.element_size())
wrap(torch.Tensor, "reshape", lambda s, o, a: 0 if o.data_ptr() == s.data_ptr() or o.numel() == 0 else o.numel() * o.element_size())
job.step()
torch.cuda.synchronize()
rows = sorted(stats.items(), key=lambda kv: -kv[1][1])
print(f"{'op':12s} {'calls':>7s} {'MB':>10s}  caller")
for (name, where), (n, b) in rows[:40]:
    print(f"{name:12s} {n:7d} {b / 1e6:10.1f}  {where}")
