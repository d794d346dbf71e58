"""How well do batch-1 block forwards overlap across HIP streams on this GPU?  One Flan-T5-XL encoder block (bf16, 64 tokens),
captured per stream, replayed round-robin; also a chain of 24 different blocks per replay (the tower graph of the capture)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vlm-compression_amd"))
import torch  # noqa: E402

from vlmc import synthetic as S  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
blocks = [S.T5Block(2048, 5120, 32, 64, False).to(dev).bfloat16().eval() for _ in range(24)]


def capture(fn):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = fn()
    return g, out


def run(nstreams, chain, reps=256):
    streams = [torch.cuda.Stream(device=dev) for _ in range(nstreams)]
    graphs = []
    for s in streams:
        x = torch.randn(1, 64, 2048, device=dev).bfloat16()

        def fn():
            h = x
            with torch.no_grad():
                for b in blocks[:chain]:
                    h = b(h)[0]
            return h
        fn()
        graphs.append(capture(fn))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for j in range(reps):
        with torch.cuda.stream(streams[j % nstreams]):
            graphs[j % nstreams][0].replay()
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return dt / reps * 1e6, t_issue / reps * 1e6


for chain in (1, 24):
    for ns in (1, 2, 4, 8):
        us, issue = run(ns, chain, 512 if chain == 1 else 128)
        print(f"chain of {chain:2d} blocks, {ns} streams: {us:8.1f} us per replay (host issue {issue:6.1f} us)", flush=True)
