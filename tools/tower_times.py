"""Seconds per TOWER of the Wanda prune (synchronised at the tower boundaries) for one process standing for rank 0 of W ranks
(VLMC_SIMULATE_WORLD): what sharding each tower's calibration samples buys on one rank.  `python tools/tower_times.py 1 2 4 8`"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vlm-compression_amd"))
import torch  # noqa: E402
from vlmc import synthetic  # noqa: E402
from lavis.compression.pruners import wanda_pruner as WP  # noqa: E402

dev = torch.device("cuda:0")
model = synthetic.InstructBlipT5().to(dev).eval()
batches = synthetic.calibration_batches(128, dev, vocab=model.t5_model.shared.num_embeddings)
times = {}
real = WP.BLIPT5LayerWandaPruner._tower


PROFILE = {"on": False, "pr": None}


def timed(self, cls, **kw):
    from vlmc import phases
    torch.cuda.synchronize()
    before = dict(phases.times)
    prof = PROFILE["on"] and kw["module_to_process"].endswith("decoder.block")
    if prof:
        import cProfile
        PROFILE["pr"] = cProfile.Profile()
        PROFILE["pr"].enable()
    t0 = time.perf_counter()
    out = real(self, cls, **kw)
    torch.cuda.synchronize()
    if prof:
        PROFILE["pr"].disable()
    times.setdefault(kw["module_to_process"], []).append(time.perf_counter() - t0)
    if phases.enabled():
        print("      phases of", kw["module_to_process"], {k: round((v - before.get(k, 0.0)) * 1e3, 1) for k, v in phases.times.items()
                                                          if v - before.get(k, 0.0) > 1e-4})
    return out


WP.BLIPT5LayerWandaPruner._tower = timed
for w in [int(a) for a in sys.argv[1:]] or [1, 8]:
    if w > 1:
        os.environ["VLMC_SIMULATE_WORLD"] = str(w)
    else:
        os.environ.pop("VLMC_SIMULATE_WORLD", None)
    times.clear()
    tot = []
    for _ in range(5):
        dt, _, _ = synthetic.time_prune(dev, model=model, batches=batches)
        tot.append(dt)
    print(f"world {w}: prune ms {[round(t * 1e3, 1) for t in tot]}", flush=True)
    for k, v in times.items():
        print(f"   {k:32s} ms {[round(t * 1e3, 1) for t in v]}")
    from vlmc import phases
    os.environ["VLMC_PHASE_TIMERS"] = "1"
    phases.reset()
    synthetic.time_prune(dev, model=model, batches=batches)
    os.environ["VLMC_PHASE_TIMERS"] = "0"
    if w == 8:
        import io
        import pstats
        PROFILE["on"] = True
        synthetic.time_prune(dev, model=model, batches=batches)
        PROFILE["on"] = False
        for key in ("tottime", "cumulative"):
            buf = io.StringIO()
            pstats.Stats(PROFILE["pr"], stream=buf).sort_stats(key).print_stats(28)
            print(buf.getvalue()[:6000])
