"""A/B of the GEMM engine's launch-time switches on one box: every configuration is a child process (the switches are read
once per process), same shapes, interleaved rounds inside each child, median.  `python tools/gemm_ab.py [shape-set]`."""
import itertools
import json
import os
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPES = {
    "vit": [("vit.qkv", "float16", 128 * 257, 4224, 1408), ("vit.proj", "float16", 128 * 257, 1408, 1408),
            ("vit.fc1", "float16", 128 * 257, 6144, 1408), ("vit.fc2", "float16", 128 * 257, 1408, 6144)],
    "t5": [("t5enc.q", "bfloat16", 8192, 2048, 2048), ("t5enc.wi", "bfloat16", 8192, 5120, 2048), ("t5enc.wo", "bfloat16", 8192, 2048, 5120),
           ("t5dec.wi", "bfloat16", 2048, 5120, 2048)],
    "dec": [("t5dec.q", "bfloat16", 2048, 2048, 2048), ("t5dec.wi", "bfloat16", 2048, 5120, 2048), ("t5dec.wo", "bfloat16", 2048, 2048, 5120),
            ("t5dec.qkv-as-one", "bfloat16", 2048, 6144, 2048), ("N=8 t5dec.q", "bfloat16", 256, 2048, 2048),
            ("N=8 t5enc.q", "bfloat16", 1024, 2048, 2048), ("N=8 t5enc.wo", "bfloat16", 1024, 2048, 5120)],
    # few rows of X: one ragged sample / a small group, one rank's share of the decoder at N = 8
    "small": [("1 sample t5enc.q", "bfloat16", 64, 2048, 2048), ("1 sample t5enc.wi", "bfloat16", 64, 5120, 2048),
              ("1 sample t5enc.wo", "bfloat16", 64, 2048, 5120), ("1 sample t5dec.q", "bfloat16", 16, 2048, 2048),
              ("4 samples t5enc.q", "bfloat16", 200, 2048, 2048), ("4 samples t5enc.wo", "bfloat16", 200, 2048, 5120),
              ("N=8 t5dec.q", "bfloat16", 256, 2048, 2048), ("N=8 t5dec.wi", "bfloat16", 256, 5120, 2048),
              ("N=8 t5dec.wo", "bfloat16", 256, 2048, 5120), ("N=8 t5enc.q", "bfloat16", 1024, 2048, 2048),
              ("1 sample vit.proj", "float16", 257, 1408, 1408), ("1 sample vit.fc2", "float16", 257, 1408, 6144)],
    "rank": [("N=8 vit.fc1", "float16", 16 * 257, 6144, 1408), ("N=8 vit.fc2", "float16", 16 * 257, 1408, 6144),
             ("N=8 vit.qkv", "float16", 16 * 257, 4224, 1408), ("N=8 t5enc.wi", "bfloat16", 1024, 5120, 2048),
             ("N=4 vit.fc2", "float16", 32 * 257, 1408, 6144), ("N=2 vit.fc2", "float16", 64 * 257, 1408, 6144)],
}

CHILD = r"""
import json, os, statistics, sys, torch
sys.path.insert(0, os.path.join(%r, 'vlm-compression_amd'))
from vlmc import ops
shapes = json.loads(sys.argv[1])
dev = 'cuda:0'
def timeit(fn, reps):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
out = {}
data = []
cold = os.environ.get('GEMM_AB_COLD', '0') == '1'      # every launch reads another copy of W: none of it is in L2 / Infinity Cache
for name, dt, M, N, K in shapes:
    dt = getattr(torch, dt)
    w = (torch.randn(N, K, device=dev) * 0.05).to(dt)
    n = max(2, min(128, -(-(768 << 20) // (N * K * 2)))) if cold else 1
    data.append(((torch.randn(M, K, device=dev) * 0.5).to(dt), [w] + [w.clone() for _ in range(n - 1)]))
ts = [[] for _ in shapes]
def timeit_pool(x, ws, reps):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for r in range(reps): ops.linear_fwd(x, ws[r %% len(ws)])
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
for r in range(5):
    for i, (x, ws) in enumerate(data):
        ts[i].append(timeit_pool(x, ws, max(10, len(ws))))
print(json.dumps([statistics.median(t) * 1e3 for t in ts]))
""" % ROOT


def main():
    sets = sys.argv[1:] or ["vit", "t5"]
    shapes = [s for k in sets for s in SHAPES[k]]
    configs = [{"VLMC_GEMM_WIDE": "0"}, {"VLMC_GEMM_WIDE": "1"}]
    extra = os.environ.get("GEMM_AB_CONFIGS")
    if extra:
        configs = [dict(kv.split("=") for kv in c.split(",")) for c in extra.split(";")]
    print("| " + " | ".join(["config"] + [f"{n} {M}x{N}x{K}" for n, _, M, N, K in shapes]) + " |")
    print("|" + "---|" * (len(shapes) + 1))
    for rep in range(2):
        for cfg in configs:
            env = dict(os.environ, **cfg)
            r = subprocess.run([sys.executable, "-c", CHILD, json.dumps(shapes)], env=env, capture_output=True, text=True, timeout=900)
            if r.returncode != 0:
                print(cfg, "FAILED", r.stderr[-500:])
                continue
            us = json.loads(r.stdout.strip().splitlines()[-1])
            cells = [f"{u:.1f} us ({2.0 * M * N * K / u / 1e6:.0f})" for u, (_, _, M, N, K) in zip(us, shapes)]
            print("| " + " | ".join([" ".join(f"{k[10:]}={v}" for k, v in cfg.items())] + cells) + " |", flush=True)


if __name__ == "__main__":
    main()
