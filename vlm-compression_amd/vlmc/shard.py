"""Who holds which calibration samples.

Under an initialised `torch.distributed` group every rank captures and replays a contiguous
share of the calibration samples and the pruning methods exchange statistics; with
`VLMC_SHARD_CALIB=0` every rank is a replica that sees ALL samples (the reference's behaviour,
runner_base.py:864-870 aside) and nothing is exchanged.  Capture, replay and every statistics
exchange ask this one function, so they cannot disagree about the mode."""
from __future__ import annotations

import os


def calibration_shard():
    """(rank, world) for sample sharding, or (0, 1) when running as replicas."""
    import torch.distributed as dist
    if os.environ.get("VLMC_SHARD_CALIB", "1") == "0":
        return 0, 1
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        return dist.get_rank(), dist.get_world_size()
    return 0, 1
