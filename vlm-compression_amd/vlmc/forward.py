"""The dense calibration forward of a block's linears on the batch-invariant MFMA kernel (`vlmc_linear_fwd`).

The reference replays every transformer block once per calibration sample (`layer(inps[j], **caches[j])`,
wanda_pruner.py:308-311, :343-346).  The replay engine forwards groups of samples instead
(`lavis/compression/pruners/calibration.py: walk_blocks`); for the statistics -- and therefore the masks -- not to depend
on how the samples were grouped (group size, ragged shapes, sharding over GPUs), the GEMMs of the block must give a row
the same bits whatever else is in the launch.  A GEMM library picks its kernel by problem size; `vlmc_linear_fwd` does
not (csrc/gemm_nt.hip).  While `invariant_linears(modules)` is active, the forward of those `nn.Linear` modules (and the
dense branch of the SparseLoRA `Linear`) runs on it whenever it can: 16-bit weights and activations of one dtype (an
active autocast to the weights' dtype casts the input like autocast would), no gradients.  Everything else -- fp32
models, odd widths -- stays with `F.linear`.  `VLMC_LINEAR_FWD=0` switches the kernel off."""
from __future__ import annotations

import contextlib
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops

_active = 0
stats = {"kernel": 0, "library": 0}


def enabled():
    return os.environ.get("VLMC_LINEAR_FWD", "1") != "0"


def linear(x, weight, bias=None):
    """`F.linear` for a calibration forward: the invariant kernel when the replay engine asked for it and the call fits."""
    if _active and not torch.is_grad_enabled() and weight.is_cuda:
        xin = x
        if torch.is_autocast_enabled() and x.is_floating_point() and x.dtype != weight.dtype and \
                torch.get_autocast_gpu_dtype() == weight.dtype:
            xin = x.to(weight.dtype)                            # what autocast does to the input of a linear
        b = bias
        if b is not None and b.dtype != weight.dtype and torch.is_autocast_enabled() and torch.get_autocast_gpu_dtype() == weight.dtype:
            b = b.to(weight.dtype)
        if ops.linear_fwd_supported(xin, weight, b) and (not torch.is_autocast_enabled() or torch.get_autocast_gpu_dtype() == weight.dtype):
            stats["kernel"] += 1
            return ops.linear_fwd(xin, weight, b, _checked=True)     # (linear_fwd_supported has just said yes)
    stats["library"] += 1
    return F.linear(x, weight, bias)


@contextlib.contextmanager
def invariant_linears(modules):
    """Route the forward of the given `nn.Linear` modules (exact type) through `linear` for the duration."""
    global _active
    if not enabled():
        yield
        return
    patched = []
    for m in modules:
        if type(m) is nn.Linear and "forward" not in m.__dict__:
            m.forward = (lambda mod: (lambda x: linear(x, mod.weight, mod.bias)))(m)
            patched.append(m)
    _active += 1
    try:
        yield
    finally:
        _active -= 1
        for m in patched:
            m.__dict__.pop("forward", None)
