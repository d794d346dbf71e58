"""DSnoT on the GPU: per-input statistics and the training-free prune/regrow refinement.

Mirrors the DSnoT `WrappedGPT` state (dsnot_pruner.py:53-105) and the per-linear body of
`_prune` (:378-751).  Per linear:
    initial mask   vlmc_wanda_select (row rule with k = round(in*ratio), or n:m; magnitude init =
                   the same kernel with a unit scale) -- no weights touched
    refinement     vlmc_dsnot_refine: every row's whole cycle loop in ONE launch (the reference
                   issues ~20 small kernels per cycle for up to 100 cycles after three full-row sorts)
    replay/apply   vlmc_dsnot_apply with C = min(max_cycle, max over rows of the stop cycle)
"""
from __future__ import annotations

import torch

from . import _lib, ops
from .ops import _dtype_code, _need_gpu, _stream


def act_moments(x: torch.Tensor) -> torch.Tensor:
    """[3, in] fp32 for ONE hook call x [1, tokens, in]: (||x||_2)^2, sum and biased variance over the tokens,
    per input channel, in one pass (dsnot_pruner.py:88-100)."""
    if x.stride(-1) != 1:
        x = x.contiguous()
    _need_gpu(x)
    out = torch.empty((3, x.shape[-1]), dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().vlmc_act_moments(x.data_ptr(), _dtype_code(x), 1, x.shape[1], x.shape[2], x.stride(1), 0,
                                            out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(), _stream()))
    return out


def stats_update(in_features, device, normsq, sums, vars_, tokens, batch):
    """Fold the per-call moments ([n_calls, in] each, sample order) into the running statistics from a zero
    state.  Returns (scaler_row, sum_row, var_row, sqrt(scaler_row)), all [in] fp32."""
    scaler = torch.zeros(in_features, dtype=torch.float32, device=device)
    _need_gpu(scaler)
    sum_row, var_row, sqrt_row = torch.zeros_like(scaler), torch.zeros_like(scaler), torch.empty_like(scaler)
    tok = torch.tensor(list(tokens), dtype=torch.int64, device=device)
    normsq, sums, vars_ = normsq.contiguous(), sums.contiguous(), vars_.contiguous()
    _lib.check(_lib.load().vlmc_dsnot_stats_update(
        scaler.data_ptr(), sum_row.data_ptr(), var_row.data_ptr(), in_features, 0, 0, normsq.data_ptr(), sums.data_ptr(),
        vars_.data_ptr(), tok.data_ptr(), len(tokens), batch, sqrt_row.data_ptr(), _stream()))
    # the inputs must outlive the launch: callers keep them referenced through DsnotInputStat._keep
    return scaler, sum_row, var_row, sqrt_row, (normsq, sums, vars_, tok)


def act_moments_calls(x: torch.Tensor, calls: int) -> torch.Tensor:
    """[3, calls, in] fp32 for `calls` hook calls stacked along dim 0 of x [calls * b, tokens, in] (a grouped replay
    forward, calibration.stacked_samples): the same per-call moments as `act_moments`, ONE launch for all of them
    (a launch per sample and distinct linear input was 54 000 launches -- 0.7 s of host time -- per FlanT5-XL prune)."""
    x = x.reshape(calls, -1, x.shape[-1])
    if x.stride(-1) != 1:
        x = x.contiguous()
    _need_gpu(x)
    out = torch.empty((3, calls, x.shape[-1]), dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().vlmc_act_moments(x.data_ptr(), _dtype_code(x), calls, x.shape[1], x.shape[2], x.stride(1), x.stride(0),
                                            out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(), _stream()))
    return out


class DsnotInputStat:
    """Statistics of ONE distinct linear input (shared by the linears that receive it)."""

    def __init__(self, in_features: int, device):
        self.in_features, self.device = in_features, device
        # per launch: (moments [3, n, in], tokens per call [n], batch per call [n], calibration sample of each call [n] or None)
        self.blocks = []
        self.scaler_row = self.sum_row = self.var_row = self.sqrt_row = None
        self.nsamples = self.ntokens = 0

    def add_call(self, x: torch.Tensor, sample=None):
        b = x.shape[0] if x.dim() == 3 else 1
        x = x.reshape(1, -1, x.shape[-1])
        self.blocks.append((act_moments(x).unsqueeze(1), [x.shape[1]], [b], None if sample is None else [sample]))

    def add_calls(self, x: torch.Tensor, calls: int, samples):
        """x [calls * b, tokens, in]: `calls` hook calls of the reference in one tensor, for calibration samples `samples`."""
        if calls == 1:
            return self.add_call(x, samples[0])
        b = x.shape[0] // calls
        self.blocks.append((act_moments_calls(x, calls), [b * x.shape[1]] * calls, [b] * calls, list(samples)))

    def extend(self, other: "DsnotInputStat"):
        """Append the call records of `other` (used when hook-level records are merged per linear input)."""
        self.blocks += other.blocks

    def ordered(self):
        """(moments [3, N, in], tokens [N], batches [N]) of all calls, in calibration-sample order (stable: calls without a
        sample index keep their arrival order)."""
        if not self.blocks:
            return torch.zeros((3, 0, self.in_features), dtype=torch.float32, device=self.device), [], []
        mom = self.blocks[0][0] if len(self.blocks) == 1 else torch.cat([b[0] for b in self.blocks], dim=1)
        tokens = [t for b in self.blocks for t in b[1]]
        batches = [t for b in self.blocks for t in b[2]]
        if all(b[3] is not None for b in self.blocks):
            sample = [t for b in self.blocks for t in b[3]]
            order = sorted(range(len(sample)), key=lambda i: sample[i])
            if order != list(range(len(sample))):
                mom = mom.index_select(1, torch.tensor(order, dtype=torch.int64, device=mom.device))
                tokens, batches = [tokens[i] for i in order], [batches[i] for i in order]
        return mom, tokens, batches

    def finalize(self, gathered=None):
        """gathered = (moments [3, n_calls, in], tokens list, batches list) replaces the local records."""
        mom, tokens, batches = self.ordered() if gathered is None else gathered
        assert len(set(batches)) <= 1, "DSnoT statistics expect a constant calibration batch size"
        b = batches[0] if batches else 1
        self.scaler_row, self.sum_row, self.var_row, self.sqrt_row, self._keep = stats_update(
            self.in_features, self.device, mom[0], mom[1], mom[2], tokens, b)
        self.nsamples = len(tokens) * b
        self.ntokens = int(sum(tokens))
        return self


def gather_stats(stats):
    """Multi-GPU: all-gather the per-call moments (rank-major = sample order), then finalise."""
    import torch.distributed as dist
    from .shard import calibration_shard, simulated_world
    world = calibration_shard()[1]          # 1 for replicas (VLMC_SHARD_CALIB=0): they already hold every sample
    if world == 1:
        for st in stats:
            st.finalize()
        return stats
    for st in stats:
        mom, tokens, batches = st.ordered()
        local = mom.permute(1, 0, 2).contiguous()                         # [calls, 3, in]
        tok = torch.tensor(tokens, dtype=torch.int64, device=local.device)
        if simulated_world():                                             # one rank rehearsing W (vlmc/shard.py)
            allm, allt = local.repeat(world, 1, 1), tok.repeat(world)
        else:
            allm = torch.empty((world * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
            dist.all_gather_into_tensor(allm, local)
            allt = torch.empty(world * tok.numel(), dtype=torch.int64, device=local.device)
            dist.all_gather_into_tensor(allt, tok)
        st.finalize((allm.permute(1, 0, 2), allt.cpu().tolist(), batches * world))
    return stats


@torch.no_grad()
def reorder_indices(input_tensor: torch.Tensor) -> torch.Tensor:
    """`return_reorder_indice(input_tensor)` (dsnot_pruner.py:1881-1925) on the GPU: int64 [rows, cols]; per row the indices of the
    negative entries in order, then one 0 per entry that is neither negative nor positive, then the indices of the positive
    entries reversed (include/vlmc.h: vlmc_reorder_indices)."""
    _need_gpu(input_tensor)
    if input_tensor.dim() != 2:
        raise ValueError("return_reorder_indice expects a 2-D tensor (the reference indexes input_tensor.shape[1], :1906)")
    x = input_tensor.detach()
    if x.dtype not in ops._DT:
        x = x.to(torch.float32)                                   # (integer inputs: exact in fp32 up to 2^24; only the sign is read)
    if x.shape[1] and x.stride(1) != 1:
        x = x.contiguous()
    rows, cols = x.shape
    out = torch.empty((rows, cols), dtype=torch.int64, device=x.device)
    if rows and cols:
        _lib.check(_lib.load().vlmc_reorder_indices(x.data_ptr(), _dtype_code(x), rows, cols, x.stride(0) if rows > 1 else cols,
                                                    out.data_ptr(), cols, _stream()))
    return out


def prune_linear(weight: torch.Tensor, stat: DsnotInputStat, ratio, *, prune_n=0, prune_m=0, initial_method="wanda",
                 without_DSnoT=False, max_cycle_time=100, update_threshold=0.1, pow_of_var_regrowing=1.0,
                 without_same_sign=True, apply_zero=True):
    """Per-linear body of the DSnoT `_prune` loops.  Returns the keep mask (torch.bool, True = keep),
    or None when ratio == 0 in the unstructured branch (the reference skips the linear, :560-561)."""
    _need_gpu(weight)
    if initial_method not in ("wanda", "magnitude"):
        raise ValueError("initial_method must be 'wanda' or 'magnitude' (the reference's 'sparsegpt' branch cannot run: "
                         "its Hessian is never allocated, dsnot_pruner.py:340,386)")
    out_f, in_f = weight.shape
    init_scale = stat.sqrt_row if initial_method == "wanda" else torch.ones_like(stat.sqrt_row)
    max_cycle = int(max_cycle_time)
    if prune_n != 0:
        keep, _ = ops.wanda_select(weight, init_scale, "nm", n=prune_n, m=prune_m, apply_zero=False)
    else:
        if ratio == 0.:
            return None
        keep, _ = ops.wanda_select(weight, init_scale, "row", k=round(in_f * ratio), apply_zero=False)     # :562 round()
    lib = _lib.load()
    ncyc = torch.zeros(1, dtype=torch.int32, device=weight.device)
    events = torch.empty((out_f, max(max_cycle, 1)), dtype=torch.int32, device=weight.device)
    if not (prune_n == 0 and without_DSnoT):
        if max_cycle >= in_f:
            raise RuntimeError(f"DSnoT needs in_features ({in_f}) > max_cycle_time ({max_cycle}); the reference's pointers "
                               "run out of range in that case (SURVEY.md Appendix B.1)")
        stop = torch.empty(out_f, dtype=torch.int32, device=weight.device)
        _lib.check(lib.vlmc_dsnot_refine(
            weight.data_ptr(), _dtype_code(weight), out_f, in_f, weight.stride(0), keep.data_ptr(), stat.sqrt_row.data_ptr(),
            stat.sum_row.data_ptr(), stat.var_row.data_ptr(), int(initial_method == "wanda"), int(prune_n), int(prune_m),
            max_cycle, float(update_threshold), float(pow_of_var_regrowing or 0.0), int(bool(without_same_sign)),
            events.data_ptr(), stop.data_ptr(), _stream()))
        ncyc = stop.max().clamp(max=max_cycle).to(torch.int32).reshape(1)
    _lib.check(lib.vlmc_dsnot_apply(weight.data_ptr(), _dtype_code(weight), out_f, in_f, weight.stride(0), keep.data_ptr(),
                                    events.data_ptr(), ncyc.data_ptr(), max(max_cycle, 1), int(prune_n != 0),
                                    int(bool(apply_zero)), _stream()))
    return keep
