"""Tensor-level wrappers over the C ABI (include/vlmc.h).

Every function takes CUDA(HIP) tensors, passes raw device pointers + the current
torch stream, and never synchronises.  Mirrors, op for op, what the reference's
pruner loop does with PyTorch ops (file:line under /root/reference cited per op).
"""
from __future__ import annotations

import torch

from . import _lib

_DT = {torch.float32: _lib.F32, torch.float16: _lib.F16, torch.bfloat16: _lib.BF16}


def _dtype_code(t: torch.Tensor) -> int:
    try:
        return _DT[t.dtype]
    except KeyError:
        raise TypeError(f"vlmc: unsupported dtype {t.dtype}") from None


def _need_gpu(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("vlmc ops run on the GPU only (no CPU fallback); got a tensor on " + str(t.device))


def _stream():
    return torch.cuda.current_stream().cuda_stream


class Workspace:
    """Grow-only device scratch buffer, one per (device, purpose)."""

    def __init__(self):
        self.buf = None

    def get(self, nbytes: int, device) -> torch.Tensor:
        if self.buf is None or self.buf.numel() < nbytes or self.buf.device != device:
            self.buf = torch.empty(max(nbytes, 1 << 16), dtype=torch.uint8, device=device)
        return self.buf


_select_ws = Workspace()
_MODES = {"row": _lib.SEL_ROW, "matrix": _lib.SEL_MATRIX, "nm": _lib.SEL_NM}


def act_sqnorm(x: torch.Tensor, out: torch.Tensor | None = None) -> torch.Tensor:
    """normsq[c, ch] = (||x[c, :, ch]||_2)**2 for hook inputs x [calls, tokens, in]
    (or [tokens, in] = one call).  wanda_pruner.py:73-81 without the running mean."""
    _need_gpu(x, out)
    if x.dim() == 2:
        x = x.unsqueeze(0)
    if x.dim() != 3:
        raise ValueError("act_sqnorm expects [calls, tokens, in]")
    if x.stride(-1) != 1 or (x.shape[1] > 1 and x.stride(1) < x.shape[2]):
        x = x.contiguous()
    calls, tokens, in_f = x.shape
    if out is None:
        out = torch.empty((calls, in_f), dtype=torch.float32, device=x.device)
    assert out.shape == (calls, in_f) and out.dtype == torch.float32 and out.is_contiguous()
    row_stride = x.stride(1) if tokens > 1 else in_f
    call_stride = x.stride(0) if calls > 1 else tokens * row_stride
    _lib.check(_lib.load().vlmc_act_sqnorm(x.data_ptr(), _dtype_code(x), calls, tokens, in_f, row_stride, call_stride,
                                           out.data_ptr(), _stream()))
    return out


def wanda_scaler_update(scaler_row: torch.Tensor, nsamples_before: int, normsq: torch.Tensor | None, batch: int = 1,
                        sqrt_out: torch.Tensor | None = None) -> int:
    """Apply the running-mean recurrence of wanda_pruner.py:77-81 in place for every
    row of `normsq` ([calls, in], in call order); optionally also write
    sqrt(scaler_row) (the score factor of wanda_pruner.py:318) into `sqrt_out`.
    Returns the new sample count."""
    _need_gpu(scaler_row, normsq, sqrt_out)
    assert scaler_row.dtype == torch.float32 and scaler_row.is_contiguous()
    calls = 0
    if normsq is not None:
        assert normsq.dtype == torch.float32 and normsq.is_contiguous() and normsq.shape[-1] == scaler_row.numel()
        calls = normsq.shape[0] if normsq.dim() == 2 else 1
    if sqrt_out is not None:
        assert sqrt_out.dtype == torch.float32 and sqrt_out.is_contiguous() and sqrt_out.numel() == scaler_row.numel()
    _lib.check(_lib.load().vlmc_wanda_scaler_update(
        scaler_row.data_ptr(), scaler_row.numel(), nsamples_before, normsq.data_ptr() if calls else None, calls, batch,
        sqrt_out.data_ptr() if sqrt_out is not None else None, _stream()))
    return nsamples_before + calls * batch


def sqrt_scaler(scaler_row: torch.Tensor) -> torch.Tensor:
    """IEEE sqrt(scaler_row) on the device (torch.sqrt(scaler_row), wanda_pruner.py:318)."""
    out = torch.empty_like(scaler_row)
    wanda_scaler_update(scaler_row, 0, None, 1, sqrt_out=out)
    return out


def wanda_select(weight: torch.Tensor, sqrt_scaler_row: torch.Tensor, mode: str, *, k: int = 0, n: int = 0, m: int = 0,
                 apply_zero: bool = True, mask: torch.Tensor | None = None, partials: torch.Tensor | None = None):
    """Fused score + select + apply for one linear (wanda_pruner.py:318-341 / :666-687).
    `sqrt_scaler_row` = sqrt(scaler_row) as produced by wanda_scaler_update(sqrt_out=...).

    mode "row": prune the k lowest-score columns of every row (stable);
    mode "matrix": prune score < sort(score.flatten())[k];  mode "nm": n of every m.
    Writes `mask` (torch.bool [out,in], True = keep), zeroes pruned weights in place
    when apply_zero, writes partial sums of the scores into `partials` (float64
    [select_partials(...)]; `partials.sum() / weight.numel()` is the importance_score).
    Returns (mask, partials).
    """
    _need_gpu(weight, sqrt_scaler_row, mask, partials)
    if weight.dim() != 2 or weight.stride(1) != 1:
        raise ValueError("wanda_select expects a row-major 2-D weight")
    out_f, in_f = weight.shape
    assert (sqrt_scaler_row.dtype == torch.float32 and sqrt_scaler_row.is_contiguous()
            and sqrt_scaler_row.numel() == in_f)
    if mask is None:
        mask = torch.empty((out_f, in_f), dtype=torch.bool, device=weight.device)
    assert mask.dtype == torch.bool and mask.is_contiguous() and mask.shape == weight.shape
    code = _MODES[mode]
    lib = _lib.load()
    nparts = lib.vlmc_wanda_select_partials(code, out_f, in_f)
    if partials is None:
        partials = torch.empty(nparts, dtype=torch.float64, device=weight.device)
    assert partials.dtype == torch.float64 and partials.is_contiguous() and partials.numel() >= nparts
    nbytes = lib.vlmc_wanda_select_workspace(code, out_f, in_f)
    ws = _select_ws.get(nbytes, weight.device) if nbytes else None
    _lib.check(lib.vlmc_wanda_select(weight.data_ptr(), _dtype_code(weight), out_f, in_f, weight.stride(0),
                                     sqrt_scaler_row.data_ptr(), code, int(k), int(n), int(m), int(bool(apply_zero)),
                                     mask.data_ptr(), partials.data_ptr(), ws.data_ptr() if nbytes else None,
                                     ws.numel() if nbytes else 0, _stream()))
    return mask, partials[:nparts]


def select_partials(mode: str, out_f: int, in_f: int) -> int:
    """Number of float64 partial sums `wanda_select` writes for this shape."""
    return int(_lib.load().vlmc_wanda_select_partials(_MODES[mode], out_f, in_f))


# ---------------------------------------------------------------------------------------
# Launch plans: pre-validated, pre-bound C-ABI calls for hot loops that issue thousands of
# launches per step (bench.py, the per-block pruner loop).  A plan is a zero-argument
# callable; tensors referenced by a plan must stay alive and must not be reallocated.
# ---------------------------------------------------------------------------------------
def _bind(fn, args):
    check = _lib.check

    def run():
        rc = fn(*args)
        if rc:
            check(rc)
    return run


def plan_act_sqnorm(x: torch.Tensor, out: torch.Tensor):
    _need_gpu(x, out)
    assert x.dim() == 3 and x.is_contiguous() and out.is_contiguous() and out.shape == (x.shape[0], x.shape[2])
    calls, tokens, in_f = x.shape
    return _bind(_lib.load().vlmc_act_sqnorm, (x.data_ptr(), _dtype_code(x), calls, tokens, in_f, in_f, tokens * in_f,
                                               out.data_ptr(), _stream()))


def plan_scaler_update(scaler_row: torch.Tensor, nsamples_before: int, normsq: torch.Tensor, batch: int,
                       sqrt_out: torch.Tensor):
    _need_gpu(scaler_row, normsq, sqrt_out)
    assert normsq.is_contiguous() and normsq.dtype == torch.float32
    return _bind(_lib.load().vlmc_wanda_scaler_update, (scaler_row.data_ptr(), scaler_row.numel(), nsamples_before,
                                                        normsq.data_ptr(), normsq.shape[0], batch, sqrt_out.data_ptr(),
                                                        _stream()))


def plan_select(weight: torch.Tensor, sqrt_scaler_row: torch.Tensor, mode: str, *, k=0, n=0, m=0, apply_zero=True,
                mask: torch.Tensor, partials: torch.Tensor):
    _need_gpu(weight, sqrt_scaler_row, mask, partials)
    out_f, in_f = weight.shape
    code = _MODES[mode]
    lib = _lib.load()
    assert partials.numel() >= lib.vlmc_wanda_select_partials(code, out_f, in_f)
    nbytes = lib.vlmc_wanda_select_workspace(code, out_f, in_f)
    ws = _select_ws.get(nbytes, weight.device) if nbytes else None
    return _bind(lib.vlmc_wanda_select, (weight.data_ptr(), _dtype_code(weight), out_f, in_f, weight.stride(0),
                                         sqrt_scaler_row.data_ptr(), code, int(k), int(n), int(m), int(bool(apply_zero)),
                                         mask.data_ptr(), partials.data_ptr(), ws.data_ptr() if nbytes else None,
                                         ws.numel() if nbytes else 0, _stream()))
