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
from vlmc import _lib
hipev, set_events = bench.HipEvents(), _lib.load().vlmc_set_launch_events
for stride in (0, 4, 1):                      # 0 = no events; N = HIP events carried by every N-th block's launches
    ev = {"stat": [], "rows": []} if stride else None
    t0 = time.perf_counter(); bench.run_step(plans[0], state, 1, ev, hipev, set_events, 0, max(1, stride)); t1 = time.perf_counter()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"event stride={stride}: issue {1e3*(t1-t0):.2f} ms, until done {1e3*(t2-t0):.2f} ms")
