"""Who clones / copies tensors during the headline's prune: calls of Tensor.clone / copy_ / contiguous (when it copies) / torch.cat by
calling line, one prune after two warm ones.  `python tools/count_copies.py`"""
import collections, os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench

dev = torch.device("cuda:0")
job = bench.PruneJob(dev, reference_ops=True, ragged=True)
job.step(); job.step()
counts = collections.Counter()
byt = collections.Counter()
T = torch.Tensor
real = {n: getattr(T, n) for n in ("clone", "copy_", "contiguous", "to")}
real_cat = torch.cat


def where():
    for f in reversed(traceback.extract_stack(limit=8)[:-2]):
        if "count_copies" not in f.filename:
            return f"{os.path.basename(f.filename)}:{f.lineno}"
    return "?"


def wrap(name):
    fn = real[name]

    def w(self, *a, **k):
        r = fn(self, *a, **k)
        if name in ("clone", "copy_") or (isinstance(r, T) and r is not self and r.data_ptr() != self.data_ptr()):
            key = (name, where())
            counts[key] += 1
            byt[key] += r.numel() * r.element_size() if isinstance(r, T) else 0
        return r
    return w


for n in real:
    setattr(T, n, wrap(n))


def cat(ts, *a, **k):
    r = real_cat(ts, *a, **k)
    key = ("cat", where())
    counts[key] += 1
    byt[key] += r.numel() * r.element_size()
    return r


torch.cat = cat
job.step()
torch.cuda.synchronize()
for n, fn in real.items():
    setattr(T, n, fn)
torch.cat = real_cat
tot = sum(counts.values())
print(f"{tot} copying calls in one prune")
for key, c in counts.most_common(40):
    print(f"{c:6d}  {byt[key] / 1e6:10.1f} MB  {key[0]:10s} {key[1]}")
