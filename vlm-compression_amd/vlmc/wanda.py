"""Wanda pruning engine pieces shared by the drop-in pruner (`lavis.compression`) and
bench.py: per-input activation statistics, the multi-GPU statistics exchange, and the
per-linear fused select.  Everything numeric runs in the HIP kernels behind `vlmc.ops`;
this module only sequences launches and owns the small state tensors.

Multi-GPU: calibration samples are sharded over ranks in contiguous
ranges (rank r owns samples [r*n, (r+1)*n)).  Each rank reduces its own samples to
per-sample squared norms ([n, in] fp32); ONE all-gather per transformer block (RCCL
over xGMI, or gloo on CPU in the tests) concatenates them in sample order, and every
rank then runs the reference's running-mean recurrence over all samples in order --
so scaler_row, and therefore every mask, is bit-identical for any number of GPUs.
The select itself (<= 125 MB of traffic per linear) is replicated, because every rank
needs the pruned weights for the next block's forward.
"""
from __future__ import annotations

import torch

from . import ops


class InputStat:
    """Statistics of ONE distinct linear input over the calibration set
    (the state of `WrappedGPT`, wanda_pruner.py:51-81, for every linear sharing it)."""

    def __init__(self, in_features: int, device, capacity: int = 0):
        self.in_features = in_features
        self.device = device
        self.rows = []                 # per-call squared norms, [calls_i, in] tensors in call order
        self.batches = []              # samples per call (the reference's `tmp`)
        self.scaler_row = None
        self.sqrt_row = None
        self.nsamples = 0
        self._buf = torch.empty((capacity, in_features), dtype=torch.float32, device=device) if capacity else None
        self._used = 0

    # -- accumulation ---------------------------------------------------------------
    def add_call(self, x: torch.Tensor):
        """One forward-hook call with input x [b, T, in] (or [T, in])."""
        b = x.shape[0] if x.dim() == 3 else 1
        x = x.reshape(1, -1, x.shape[-1])
        out = None
        if self._buf is not None and self._used < self._buf.shape[0]:
            out = self._buf[self._used:self._used + 1]
            self._used += 1
        self.rows.append(ops.act_sqnorm(x, out=out))
        self.batches.append(b)

    def add_samples(self, x: torch.Tensor, out: torch.Tensor | None = None):
        """Many batch-1 calls at once: x [samples, T, in] -> one launch."""
        self.rows.append(ops.act_sqnorm(x, out=out))
        self.batches.extend([1] * x.shape[0])

    def local_normsq(self) -> torch.Tensor:
        if len(self.rows) == 1:
            return self.rows[0]
        if self._buf is not None and self._used == len(self.rows):
            return self._buf[:self._used]
        return torch.cat(self.rows, dim=0)

    # -- finalisation ---------------------------------------------------------------
    def finalize(self, normsq: torch.Tensor | None = None, batches=None):
        """Run the running-mean recurrence over `normsq` (default: the local calls) in
        order and produce scaler_row and sqrt(scaler_row)."""
        finalize_stats([self], [self.local_normsq() if normsq is None else normsq], [batches or self.batches])
        return self


def finalize_stats(stats, normsqs, batches_list):
    """Running-mean recurrence for several statistics.  Statistics that saw the same call pattern (the
    inputs of one transformer block) share ONE launch per run of equal batch sizes."""
    # one zero-filled and one uninitialised buffer for all statistics of the call (a block's 4-7 distinct inputs), handed out as
    # views: two allocations and one fill instead of two per statistic -- on one rank's share of the calibration set the towers are
    # bound by the number of dispatches (profiles/r04_scaling_floor.md).  Offsets are multiples of 64 floats: 256-byte aligned views.
    by_dev = {}
    for st in stats:
        by_dev.setdefault(st.device, []).append(st)
    for dev, members_ in by_dev.items():
        sizes = [(st.in_features + 63) // 64 * 64 for st in members_]
        zeros = torch.zeros(sum(sizes), dtype=torch.float32, device=dev)
        empty = torch.empty(sum(sizes), dtype=torch.float32, device=dev)
        off = 0
        for st, sz in zip(members_, sizes):
            st.scaler_row = zeros[off:off + st.in_features]
            st.sqrt_row = empty[off:off + st.in_features]
            off += sz
    groups = {}
    for st, nsq, bt in zip(stats, normsqs, batches_list):
        groups.setdefault(tuple(bt), []).append((st, nsq))
    for bt, members in groups.items():
        n, start = 0, 0
        while start < len(bt):                 # consecutive calls with the same batch size: one launch
            end = start
            while end < len(bt) and bt[end] == bt[start]:
                end += 1
            last = end == len(bt)
            n = ops.wanda_scaler_update_batch([st.scaler_row for st, _ in members], n,
                                              [nsq[start:end] for _, nsq in members], bt[start],
                                              [st.sqrt_row if last else None for st, _ in members])
            start = end
        if not bt:
            for st, _ in members:
                ops.wanda_scaler_update(st.scaler_row, 0, None, 1, sqrt_out=st.sqrt_row)
        for st, _ in members:
            st.nsamples = n
    return stats


def gather_stats(stats, group=None):
    """Multi-GPU exchange for one transformer block: all-gather the per-sample squared
    norms of every distinct input in ONE collective, then finalise each statistic over
    the full, ordered sample set.  With no process group this is just the finalisation."""
    import torch.distributed as dist
    from .shard import calibration_shard
    # replicas (VLMC_SHARD_CALIB=0) have every sample already: no exchange, or the rows would be counted world times
    world = dist.get_world_size(group) if group is not None else calibration_shard()[1]
    if world == 1:
        return finalize_stats(stats, [st.local_normsq() for st in stats], [st.batches for st in stats])
    local = [st.local_normsq() for st in stats]
    n_local = local[0].shape[0]
    assert all(t.shape[0] == n_local for t in local), "every input must see the same number of calls"
    flat = torch.cat(local, dim=1).contiguous()                        # [n_local, sum(in)]
    from .shard import simulated_world
    if group is None and simulated_world():
        gathered = flat.repeat(world, 1)                               # one rank rehearsing W: its own rows W times
    else:
        gathered = torch.empty((world * n_local, flat.shape[1]), dtype=flat.dtype, device=flat.device)
        from . import phases
        with phases.phase("exchange"):                                 # (a timer only under VLMC_PHASE_TIMERS=1: bench.py's sub-totals)
            dist.all_gather_into_tensor(gathered, flat, group=group)   # rank-major == sample order
    parts, off = [], 0
    for st in stats:
        parts.append(gathered[:, off:off + st.in_features])            # column slices: the kernel takes a row stride
        off += st.in_features
    return finalize_stats(stats, parts, [st.batches * world for st in stats])


def prune_linear(weight: torch.Tensor, stat: InputStat, mode: str, *, ratio=None, n=0, m=0, apply_zero=True,
                 mask: torch.Tensor | None = None, partials: torch.Tensor | None = None):
    """Loop body of wanda_pruner.py:316-341 / :664-687 for one linear."""
    if mode == "nm":
        return ops.wanda_select(weight, stat.sqrt_row, "nm", n=n, m=m, apply_zero=apply_zero, mask=mask, partials=partials)
    if mode == "row":
        k = int(weight.shape[1] * ratio)                               # :336
    elif mode == "matrix":
        k = int(weight.numel() * ratio)                                # :682
    else:
        raise ValueError(mode)
    return ops.wanda_select(weight, stat.sqrt_row, mode, k=k, apply_zero=apply_zero, mask=mask, partials=partials)


def prune_block(weights, stats, mode: str, *, ratios=None, n=0, m=0, apply_zero=True, partials=None):
    """`for name in subset:` of wanda_pruner.py:316-341 / :664-687 for ALL linears of a block with batched
    launches.  Linears of different dtypes go in separate batches.  Returns the list of keep masks."""
    masks = [None] * len(weights)
    by_dtype = {}
    for i, w in enumerate(weights):
        by_dtype.setdefault(w.dtype, []).append(i)
    for idx in by_dtype.values():
        ws = [weights[i] for i in idx]
        if mode == "nm":
            ks = None
        elif mode == "row":
            ks = [int(w.shape[1] * ratios[i]) for w, i in zip(ws, idx)]                     # :336
        elif mode == "matrix":
            ks = [int(w.numel() * ratios[i]) for w, i in zip(ws, idx)]                      # :682
        else:
            raise ValueError(mode)
        out, _ = ops.wanda_select_batch(ws, [stats[i].sqrt_row for i in idx], mode, ks=ks, n=n, m=m, apply_zero=apply_zero,
                                        partials=None if partials is None else [partials[i] for i in idx])
        for i, mk in zip(idx, out):
            masks[i] = mk
    return masks
