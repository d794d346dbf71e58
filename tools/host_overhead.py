"""How long does the host need to ISSUE one bench step (no sync) vs. the GPU to run it?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "vlm-compression_amd"))
import torch, bench
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
blocks, acts, sets, n_local = bench.build_workload(dev, 0, 1, 2)
state = bench.alloc_state(blocks, n_local, 1, dev)
plans = [bench.build_plans(blocks, acts, w, n_local, 1, dev, state) for w in sets]
for i in range(2): bench.run_step(plans[i % 2], state, 1)
torch.cuda.synchronize()
for with_events in (False, True):
    ev = {"stat": [], "rows": []} if with_events else None
    t0 = time.perf_counter(); bench.run_step(plans[0], state, 1, ev); t1 = time.perf_counter()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"events={with_events}: issue {1e3*(t1-t0):.2f} ms, until done {1e3*(t2-t0):.2f} ms")
