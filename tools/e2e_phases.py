"""Phase breakdown of one end-to-end Wanda prune of the synthetic InstructBLIP-FlanT5-XL (host timers with syncs)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vlm-compression_amd"))
import torch
from vlmc import synthetic
from lavis.compression.pruners import calibration as cal

dev = torch.device("cuda:0")
T = {}


def timed(name, fn):
    def w(*a, **k):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = fn(*a, **k)
        torch.cuda.synchronize()
        T[name] = T.get(name, 0.0) + time.perf_counter() - t0
        return r
    return w


cal.capture_block_inputs = timed("capture (model forward up to the tower)", cal.capture_block_inputs)
orig_walk = cal.walk_blocks


def walk(model, inps, outs, caches, mtp, n, autocast, prune_block, tuple_output, **kw):
    def pb(i, layer, subset, run_pass, state):
        def rp(before_sample=None, **kw):
            return timed("replay pass with hooks" if before_sample is not None else "replay pass plain", run_pass)(before_sample, **kw)
        return timed("prune_block total (incl. hooked pass)", prune_block)(i, layer, subset, rp, state)
    return orig_walk(model, inps, outs, caches, mtp, n, autocast, pb, tuple_output, **kw)


cal.walk_blocks = timed("walk_blocks total", walk)
import lavis.compression.pruners.wanda_pruner as wp
import lavis.compression.pruners.sparsegpt_pruner as sp
import lavis.compression.pruners.dsnot_pruner as dp
wp.cal = sp.cal = dp.cal = cal
NAME = sys.argv[1] if len(sys.argv) > 1 else "wanda"
if NAME == "sparsegpt":
    from vlmc import sparsegpt as SGm
    _fp = SGm.fasterprune
    def fp_shape(layer, *a, **k):
        return timed(f"fasterprune {tuple(layer.weight.shape)} cached={'U' in (k.get('factor_cache') or {})}", _fp)(layer, *a, **k)
    SGm.fasterprune = timed("fasterprune (all linears)", fp_shape)
    SGm.SparseGPT.add_batch = timed("SparseGPT.add_batch (all calls)", SGm.SparseGPT.add_batch)
for mode, env in (("graph", {}),):
    os.environ.pop("VLMC_GRAPH_REPLAY", None)
    os.environ.update(env)
    T.clear()
    dt, model, info = synthetic.time_prune(dev, f"blipt5_{NAME}_pruner")
    print(mode, f"total {dt:.2f} s")
    if NAME == "sparsegpt":
        print("   factor routes", SGm.factor_stats)
    for k, v in T.items():
        print(f"   {k:45s} {v:7.2f} s")
    del model
