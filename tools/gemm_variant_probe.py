"""Which launch-time variant of the GEMM engine disagrees with an fp64 product on a given shape (child process per variant)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import json, os, sys, torch
sys.path.insert(0, os.path.join(%r, 'vlm-compression_amd'))
from vlmc import ops
out = []
g = torch.Generator(device='cuda:0').manual_seed(5)
for dt, M, N, K in json.loads(sys.argv[1]):
    dt = getattr(torch, dt)
    x = (torch.randn(M, K, generator=g, device='cuda:0') * 0.5 + 0.1).to(dt)
    w = (torch.randn(N, K, generator=g, device='cuda:0') * 0.05).to(dt)
    b = (torch.randn(N, generator=g, device='cuda:0') * 0.1).to(dt)
    y = ops.linear_fwd(x, w, b).double()
    ref = x.double() @ w.double().t() + b.double()
    bad = ((y - ref).abs() > 0.05 + 0.02 * ref.abs())
    rows = bad.any(1).nonzero().flatten().tolist()
    cols = bad.any(0).nonzero().flatten().tolist()
    out.append([int(bad.sum()), rows[:3] + rows[-3:], cols[:3] + cols[-3:]])
print(json.dumps(out))
""" % ROOT

SHAPES = [("float16", 40, 1000, 1408), ("bfloat16", 40, 1000, 1408), ("float16", 40, 1024, 1408), ("float16", 64, 1000, 1408),
          ("float16", 40, 1000, 1440), ("float16", 300, 1000, 1408)]
KEYS = ("RING", "BIG_TILES", "PINGPONG", "PERSIST", "EDGE", "WIDE", "SMALL_TILES")
CONFIGS = [("1", "200", "1", "1", "1", "1", "1"), ("0", "200", "1", "1", "1", "1", "1"), ("1", "1", "1", "1", "1", "1", "1"),
           ("1", "1", "1", "0", "1", "1", "1"), ("1", "1", "0", "1", "1", "1", "1"), ("0", "0", "1", "1", "1", "1", "1"),
           ("1", "0", "1", "1", "1", "1", "1"), ("1", "1", "1", "1", "0", "1", "1"), ("1", "1", "1", "1", "1", "0", "1"),
           ("1", "0", "1", "1", "1", "0", "1"), ("1", "200", "1", "1", "1", "1", "0"), ("0", "200", "1", "1", "1", "1", "0"),
           ("1", "200", "0", "0", "1", "1", "0")]
if len(sys.argv) > 1 and sys.argv[1] == "shapes":
    KEYS = ("SHAPE", "RING", "WIDE", "WIDE_SLOTS")
    CONFIGS = [(sh, r, w, n) for sh in ("64", "p32", "32", "128") for r, w, n in (("1", "1", "0"), ("1", "1", "2"), ("1", "0", "0"), ("0", "1", "0"))]
    SHAPES += [("bfloat16", 16, 2048, 2048), ("bfloat16", 700, 1000, 2048), ("float16", 257, 1408, 352), ("bfloat16", 2100, 520, 96)]
for cfg in CONFIGS:
    env = dict(os.environ, **{f"VLMC_GEMM_{k}": v for k, v in zip(KEYS, cfg)})
    r = subprocess.run([sys.executable, "-c", CHILD, json.dumps(SHAPES)], env=env, capture_output=True, text=True, timeout=300)
    print(dict(zip(KEYS, cfg)), r.stdout.strip().splitlines()[-1] if r.returncode == 0 else r.stderr[-400:], flush=True)
