"""Phase sub-totals of a whole Wanda prune for several settings of the capture streams / replay switches (1x MI355X)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "vlm-compression_amd"))
import torch  # noqa: E402

import bench  # noqa: E402
from vlmc import phases  # noqa: E402

job = bench.PruneJob(torch.device("cuda:0"))
settings = [dict(VLMC_CAPTURE_STREAMS=s) for s in sys.argv[1:]] or [dict(VLMC_CAPTURE_STREAMS="1"), dict(VLMC_CAPTURE_STREAMS="2"),
                                                                    dict(VLMC_CAPTURE_STREAMS="4"), dict(VLMC_CAPTURE_STREAMS="8"),
                                                                    dict(VLMC_CAPTURE_STREAMS="4", VLMC_LINEAR_FWD="0")]
for env in settings:
    os.environ.update(env)
    for _ in range(2):
        job.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        job.step()
    torch.cuda.synchronize()
    plain = (time.perf_counter() - t0) / 3
    os.environ["VLMC_PHASE_TIMERS"] = "1"
    phases.reset()
    job.step()
    os.environ["VLMC_PHASE_TIMERS"] = "0"
    print(env, f"plain {plain:.3f} s | phases:", {k: round(v, 3) for k, v in phases.times.items()}, flush=True)
    for k in env:
        os.environ.pop(k)
