"""Who holds which calibration samples.

Under an initialised `torch.distributed` group every rank captures and replays a contiguous
share of the calibration samples and the pruning methods exchange statistics; with
`VLMC_SHARD_CALIB=0` every rank is a replica that sees ALL samples (the reference's behaviour,
runner_base.py:864-870 aside) and nothing is exchanged.  Capture, replay and every statistics
exchange ask this one function, so they cannot disagree about the mode."""
from __future__ import annotations

import os


def simulated_world():
    """`VLMC_SIMULATE_WORLD=W` (bench.py --calib-local, W = 128 / L): ONE process stands for rank 0 of W -- it captures
    and replays its share of the calibration samples and runs every kernel a rank runs; where the ranks would exchange
    statistics it fills in W copies of its own rows (same bytes through the same recurrence).  What it measures is a
    rank's floor: everything but the collectives and the other ranks' arrival skew.  0 = off."""
    try:
        w = int(os.environ.get("VLMC_SIMULATE_WORLD", "0"))
    except ValueError:
        return 0
    if w > 1 and not _warned:
        import warnings
        _warned.append(w)
        warnings.warn(f"VLMC_SIMULATE_WORLD={w}: this process REHEARSES rank 0 of {w} -- statistics of the other ranks are filled in "
                      "with copies of this rank's rows, so the masks are NOT those of a real prune (a measurement aid of bench.py "
                      "--calib-local; unset the variable for real runs)", RuntimeWarning, stacklevel=2)
    return w if w > 1 else 0


_warned = []


def require_real_exchange(what):
    """Paths whose exchange cannot be rehearsed by repeating this rank's rows (SparseGPT's Hessian all-reduce and weight
    broadcast) refuse to run under VLMC_SIMULATE_WORLD instead of crashing inside torch.distributed without a process group."""
    if simulated_world():
        raise RuntimeError(f"{what}: VLMC_SIMULATE_WORLD rehearses the Wanda / DSnoT statistics exchange only; run it under a real "
                           "torch.distributed launch (or unset VLMC_SIMULATE_WORLD)")


def calibration_shard():
    """(rank, world) for sample sharding, or (0, 1) when running as replicas."""
    import torch.distributed as dist
    if os.environ.get("VLMC_SHARD_CALIB", "1") == "0":
        return 0, 1
    if simulated_world():
        return 0, simulated_world()
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        return dist.get_rank(), dist.get_world_size()
    return 0, 1
