"""Does a HIP graph with S parallel branches (S independent batch-1 block forwards captured on forked streams) run them
concurrently?  Time per sample-forward for S = 1, 2, 4, 8 (synthetic T5 / ViT blocks at model size)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vlm-compression_amd"))
import torch
from vlmc import synthetic
dev = torch.device("cuda:0")
for kind in ("t5", "vit"):
    if kind == "t5":
        blk = synthetic.T5Block(2048, 5120, 32, 64, False).to(torch.bfloat16).to(dev).eval()
        mk = lambda: (torch.randn(1, 64, 2048, device=dev) * 0.5).to(torch.bfloat16)
        call = lambda x: blk(x)[0]
    else:
        blk = synthetic.ViTBlock(1408, 6144, 16).to(torch.float16).to(dev).eval()
        mk = lambda: (torch.randn(1, 257, 1408, device=dev) * 0.5).half()
        call = lambda x: blk(x, None)
    for S in (1, 2, 4, 8):
        xs = [mk() for _ in range(S)]
        streams = [torch.cuda.Stream(device=dev) for _ in range(S)]
        with torch.no_grad():
            for x in xs:
                call(x)
            torch.cuda.synchronize()
            tc = time.perf_counter()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                main = torch.cuda.current_stream()
                ys = []
                for x, st in zip(xs, streams):
                    st.wait_stream(main)
                    with torch.cuda.stream(st):
                        ys.append(call(x))
                for st in streams:
                    main.wait_stream(st)
        torch.cuda.synchronize()
        cap_ms = (time.perf_counter() - tc) * 1e3
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 64
        for _ in range(n):
            g.replay()
        enq = (time.perf_counter() - t0) / n          # host time to enqueue (the GPU may still be running)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        print(f"{kind} block, {S} parallel branch(es): {dt * 1e6:8.1f} us per replay, {dt * 1e6 / S:7.1f} us per sample, capture {cap_ms:.1f} ms, host enqueue {enq * 1e6:.0f} us per replay", flush=True)

# ---- S single-branch graphs, each replayed on its own stream (no multi-branch graph) -------------------------------------
print("separate single-branch graphs on separate streams:")
for kind in ("t5", "vit"):
    if kind == "t5":
        blk = synthetic.T5Block(2048, 5120, 32, 64, False).to(torch.bfloat16).to(dev).eval()
        mk = lambda: (torch.randn(1, 64, 2048, device=dev) * 0.5).to(torch.bfloat16)
        call = lambda x: blk(x)[0]
    else:
        blk = synthetic.ViTBlock(1408, 6144, 16).to(torch.float16).to(dev).eval()
        mk = lambda: (torch.randn(1, 257, 1408, device=dev) * 0.5).half()
        call = lambda x: blk(x, None)
    for S in (1, 2, 4):
        xs = [mk() for _ in range(S)]
        streams = [torch.cuda.Stream(device=dev) for _ in range(S)]
        graphs = []
        with torch.no_grad():
            for x in xs:
                call(x)
            torch.cuda.synchronize()
            for x in xs:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, capture_error_mode="thread_local"):
                    y = call(x)
                graphs.append((g, y))
        torch.cuda.synchronize()
        n = 64
        t0 = time.perf_counter()
        for _ in range(n):
            for (g, _), st in zip(graphs, streams):
                with torch.cuda.stream(st):
                    g.replay()
        enq = (time.perf_counter() - t0) / n
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        print(f"{kind} block, {S} graph(s) on {S} stream(s): {dt * 1e6 / S:7.1f} us per sample, host enqueue {enq * 1e6 / S:.0f} us per replay", flush=True)
