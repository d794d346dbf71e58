"""Tensor-level wrappers over the C ABI (include/vlmc.h).

Every function takes CUDA(HIP) tensors, passes raw device pointers + the current
torch stream, and never synchronises.  Mirrors, op for op, what the reference's
pruner loop does with PyTorch ops (file:line under /root/reference cited per op).

There is ONE backend: these kernels.  The `VLMC_*` switches that send an op back to a torch / library call
(`VLMC_LINEAR_FWD=0`, `VLMC_SDPA=0`, `VLMC_ATTN_MATMUL=0`, `VLMC_SGPT_PERSISTENT=0`, ... -- `vlmc/crosscheck.py` lists them by
name) exist so that the tests can hold every kernel against the route it replaced, on the GPU, inside whole prunes; they are
test and measurement aids, not a supported second path: nothing in the product selects them, a missing library raises
(`_lib.load`), CPU tensors raise (`_need_gpu`).
"""
from __future__ import annotations

import ctypes
import threading

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


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream():
    """The current stream's handle (every launch asks: the raw getter is ~0.3 us, a `torch.cuda.Stream` object ~3 us)."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


class Workspace:
    """Grow-only device scratch buffers for one purpose, one per (device, stream): calls on different streams
    or devices never share scratch memory, calls on one stream are ordered by the stream itself."""

    def __init__(self, zeroed: bool = False):
        self._bufs = {}
        self._lock = threading.Lock()
        self._alloc = torch.zeros if zeroed else torch.empty     # SEL_MATRIX workspaces: zero-filled at first use (vlmc.h)

    def get(self, nbytes: int, device) -> torch.Tensor:
        device = torch.device(device)
        key = (device.index if device.index is not None else torch.cuda.current_device(), torch.cuda.current_stream(device).cuda_stream)
        with self._lock:
            buf = self._bufs.get(key)
            if buf is None or buf.numel() < nbytes:
                buf = self._bufs[key] = self._alloc(max(nbytes, 1 << 16), dtype=torch.uint8, device=device)
            return buf


_select_ws = Workspace(zeroed=True)
_MODES = {"row": _lib.SEL_ROW, "matrix": _lib.SEL_MATRIX, "nm": _lib.SEL_NM}


_hessian_ws = Workspace()


_FAST_ENTRY_POINTS = ("linear_fwd", "linear_fwd_group", "attn_matmul", "row_mean", "rms_norm", "sdpa")


def _load_fast():
    """The compiled host path (csrc/fastpath/fast_bind.cpp -> vlmc/_fast*.so) for linear_fwd / linear_fwd_group / attn_matmul:
    the same C ABI, ~2 us of host time per call instead of 8-22 through ctypes.  None when it has not been built, when
    `VLMC_FAST=0`, or when `VLMC_LIB` points at another build of the library (the module links the in-tree one)."""
    import os
    if os.environ.get("VLMC_FAST", "1") == "0" or os.environ.get("VLMC_LIB"):
        return None
    try:
        from . import _fast
    except ImportError:
        return None
    if _fast.abi_version() != _lib.header_abi_version():
        raise ImportError("vlmc/_fast was built against another ABI version of libvlmc_hip.so; rebuild (make -C vlm-compression_amd/csrc/fastpath)")
    _lib.load()
    missing = [n for n in _FAST_ENTRY_POINTS if not hasattr(_fast, n)]
    if missing:                                                          # a stale build at the same ABI: one route for everything, not a mix
        import warnings
        warnings.warn(f"vlmc/_fast lacks {missing}: rebuild it (make -C vlm-compression_amd/csrc/fastpath); using the ctypes route")
        return None
    return _fast


_fast = _load_fast()


def linear_f32_supported(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor | None = None) -> bool:
    """Whether `linear_fwd` takes this call on its fp32 kernel: fp32 activations, weights (and bias) on the GPU, weight rows contiguous."""
    f32 = torch.float32
    return (x.is_cuda and weight.is_cuda and x.dtype is f32 and weight.dtype is f32 and weight.dim() == 2 and x.dim() >= 1
            and x.shape[-1] == weight.shape[1] and weight.stride(1) == 1 and weight.shape[1] > 0
            and (bias is None or (bias.is_cuda and bias.dtype is f32 and bias.is_contiguous())))


def _linear_fwd_f32(x, weight, bias):
    N, K = weight.shape
    x2 = x.reshape(-1, K)
    if x2.stride(1) != 1 or (x2.shape[0] > 1 and x2.stride(0) < K):
        x2 = x2.contiguous()
    M = x2.shape[0]
    y = torch.empty((M, N), dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().vlmc_linear_fwd(x2.data_ptr(), weight.data_ptr(), bias.data_ptr() if bias is not None else None, _lib.F32, M, N, K,
                                           x2.stride(0) if M > 1 else K, weight.stride(0), y.data_ptr(), N, _stream()))
    return y.reshape(*x.shape[:-1], N)


def linear_fwd_supported(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor | None = None) -> bool:
    """Whether `linear_fwd` takes this call on its 16-bit kernels: 16-bit activations and weights of one dtype on the GPU, K a multiple of 8."""
    return (x.is_cuda and weight.is_cuda and x.dtype == weight.dtype and x.dtype in (torch.float16, torch.bfloat16)
            and weight.dim() == 2 and x.shape[-1] == weight.shape[1] and weight.shape[1] % 8 == 0 and weight.stride(1) == 1
            and weight.stride(0) % 8 == 0 and weight.data_ptr() % 16 == 0
            and (bias is None or (bias.is_cuda and bias.dtype == weight.dtype and bias.is_contiguous())))


def linear_fwd(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor | None = None, _checked: bool = False,
               _try: bool = False) -> torch.Tensor:
    """y = x @ weight.T + bias on the batch-invariant MFMA kernel (`F.linear` inside the calibration replay's block
    forwards, wanda_pruner.py:308-311,343-346): a row of y depends on its row of x only, whatever else is in the call.
    `_try`: return None instead of raising when the call is not one the kernel takes.  fp32 tensors (the reference's Q-Former): the
    fp32 matrix-core kernel, the same invariance."""
    if weight.dtype is torch.float32 and x.dtype is torch.float32:
        y = None
        if _fast is not None and x.is_cuda:
            y = _fast.linear_fwd(x, weight, bias, _stream())
        elif linear_f32_supported(x, weight, bias):
            y = _linear_fwd_f32(x, weight, bias)
        if y is None and not _try:
            _need_gpu(x, weight, bias)
            raise TypeError("vlmc.linear_fwd: fp32 x [.., K], weight [N, K] with contiguous rows and a contiguous fp32 bias expected")
        return y
    if _fast is not None:
        if not x.is_cuda:
            _need_gpu(x, weight, bias)
        y = _fast.linear_fwd(x, weight, bias, _stream())
        if y is None and not _try:
            _need_gpu(x, weight, bias)
            raise TypeError("vlmc.linear_fwd: fp16/bf16 tensors of one dtype with in_features % 8 == 0 expected")
        return y
    if _try and not (x.is_cuda and linear_fwd_supported(x, weight, bias)):
        return None
    _need_gpu(x, weight, bias)
    if not _checked and not _try and not linear_fwd_supported(x, weight, bias):
        raise TypeError("vlmc.linear_fwd: fp16/bf16 tensors of one dtype with in_features % 8 == 0 expected")
    N, K = weight.shape
    x2 = x.reshape(-1, K)
    if x2.stride(1) != 1 or x2.stride(0) % 8 != 0 or x2.stride(0) < K or x2.data_ptr() % 16 != 0:
        x2 = x2.contiguous()                                 # (an expanded input has stride(0) == 0 < K)
    M = x2.shape[0]
    y = torch.empty((M, N), dtype=x.dtype, device=x.device)
    _lib.check(_lib.load().vlmc_linear_fwd(x2.data_ptr(), weight.data_ptr(), bias.data_ptr() if bias is not None else None,
                                           _dtype_code(x), M, N, K, x2.stride(0), weight.stride(0), y.data_ptr(), N, _stream()))
    return y.reshape(*x.shape[:-1], N)


LINEAR_GROUP_MAX = 4


_group_jobs = {}           # (W pointer, N, ldw, bias pointer) per member -> the launch's job table, everything but Y filled in


def linear_fwd_group(x: torch.Tensor, weights, biases=None, _checked: bool = False) -> list:
    """[x @ w.T + b for w, b in zip(weights, biases)] in ONE launch of the batch-invariant kernel: up to 4 linears fed the
    same activations (q / k / v, wi_0 / wi_1, cross-attention k / v).  Every output has the bits `linear_fwd` gives it;
    the launch shares the chip between the products (include/vlmc.h: vlmc_linear_fwd_group)."""
    n = len(weights)
    if biases is None:
        biases = [None] * n
    if _fast is not None:
        if not x.is_cuda:
            _need_gpu(x, *weights)
        outs = _fast.linear_fwd_group(x, list(weights), list(biases), _stream())
        if outs is None:
            if not 1 <= n <= LINEAR_GROUP_MAX or len(biases) != n:
                raise ValueError(f"linear_fwd_group takes 1..{LINEAR_GROUP_MAX} weights and as many biases")
            _need_gpu(x, *weights, *[b for b in biases if b is not None])
            raise TypeError("vlmc.linear_fwd_group: fp16/bf16 tensors of one dtype with in_features % 8 == 0 expected")
        return outs
    if not _checked:
        if not 1 <= n <= LINEAR_GROUP_MAX or len(biases) != n:
            raise ValueError(f"linear_fwd_group takes 1..{LINEAR_GROUP_MAX} weights and as many biases")
        _need_gpu(x, *weights, *[b for b in biases if b is not None])
        for w, b in zip(weights, biases):
            if not linear_fwd_supported(x, w, b):
                raise TypeError("vlmc.linear_fwd_group: fp16/bf16 tensors of one dtype with in_features % 8 == 0 expected")
    K = weights[0].shape[1]
    x2 = x.reshape(-1, K)
    if x2.stride(1) != 1 or x2.stride(0) % 8 != 0 or x2.stride(0) < K or x2.data_ptr() % 16 != 0:
        x2 = x2.contiguous()
    M = x2.shape[0]
    # the table holds nothing but what its key says (a recycled address of an equal shape describes the same job)
    key = tuple((w.data_ptr(), w.shape[0], w.stride(0), b.data_ptr() if b is not None else 0) for w, b in zip(weights, biases))
    jobs = _group_jobs.get(key)
    if jobs is None:
        if len(_group_jobs) > 4096:
            _group_jobs.clear()
        jobs = _group_jobs[key] = (_lib.LinearJob * n)()
        for g, (wp, N, ldw, bp) in enumerate(key):
            jobs[g] = _lib.LinearJob(wp, bp or None, None, N, ldw, N)
    outs = []
    lead = x.shape[:-1]
    for g in range(n):
        y = torch.empty((M, key[g][1]), dtype=x.dtype, device=x.device)
        jobs[g].Y = y.data_ptr()
        outs.append(y)
    _lib.check(_lib.load().vlmc_linear_fwd_group(x2.data_ptr(), jobs, n, _DT[x.dtype], M, K, x2.stride(0), _stream()))
    return [y.reshape(*lead, y.shape[1]) for y in outs]


def linear_fwd_rows(x: torch.Tensor, weights, biases, rowmap: torch.Tensor, n_real: int) -> list:
    """[x @ w.T + b for w, b in zip(weights, biases)] over the rows `rowmap[:n_real]` of the flattened x only; rows
    `rowmap[n_real:]` of every output are zeros (include/vlmc.h: vlmc_linear_fwd_rows).  x [.., K] flattens to M rows; `rowmap`:
    int32 device tensor [M], a permutation of 0 .. M - 1 with the rows that are real first (a padded group of ragged calibration
    samples: calibration.plan_padded builds it).  A computed row has the bits `linear_fwd` gives it."""
    n = len(weights)
    if biases is None:
        biases = [None] * n
    if _fast is not None and x.is_cuda and hasattr(_fast, "linear_fwd_rows"):
        outs = _fast.linear_fwd_rows(x, list(weights), list(biases), rowmap, int(n_real), _stream())
        if outs is not None:
            return outs                                               # (None: not a call the kernel takes -- the checks below say why)
    if not 1 <= n <= LINEAR_GROUP_MAX or len(biases) != n:
        raise ValueError(f"linear_fwd_rows takes 1..{LINEAR_GROUP_MAX} weights and as many biases")
    _need_gpu(x, rowmap, *weights, *[b for b in biases if b is not None])
    f32 = x.dtype is torch.float32
    for w, b in zip(weights, biases):
        if not (linear_f32_supported(x, w, b) if f32 else linear_fwd_supported(x, w, b)):
            raise TypeError("vlmc.linear_fwd_rows: fp16/bf16 tensors of one dtype with in_features % 8 == 0, or fp32 tensors, expected")
    K = weights[0].shape[1]
    x2 = x.reshape(-1, K)
    if x2.stride(1) != 1 or x2.stride(0) < K or (not f32 and (x2.stride(0) % 8 != 0 or x2.data_ptr() % 16 != 0)):
        x2 = x2.contiguous()
    M = x2.shape[0]
    if rowmap.dtype != torch.int32 or rowmap.dim() != 1 or rowmap.shape[0] != M or not rowmap.is_contiguous() or not 1 <= n_real <= M:
        raise ValueError("vlmc.linear_fwd_rows: rowmap must be a contiguous int32 [rows of x] tensor and 1 <= n_real <= rows")
    jobs = (_lib.LinearJob * n)()
    outs = []
    for g, (w, b) in enumerate(zip(weights, biases)):
        y = torch.empty((M, w.shape[0]), dtype=x.dtype, device=x.device)
        jobs[g] = _lib.LinearJob(w.data_ptr(), b.data_ptr() if b is not None else None, y.data_ptr(), w.shape[0], w.stride(0), w.shape[0])
        outs.append(y)
    _lib.check(_lib.load().vlmc_linear_fwd_rows(x2.data_ptr(), jobs, n, _DT[x.dtype], M, K, x2.stride(0), rowmap.data_ptr(), int(n_real),
                                                _stream()))
    lead = x.shape[:-1]
    return [y.reshape(*lead, y.shape[1]) for y in outs]


def linear_fwd_gather(x: torch.Tensor, weight: torch.Tensor, bias, x_rows: torch.Tensor, y_rows: torch.Tensor, n_real: int, out_rows: int,
                      pitch: int) -> torch.Tensor:
    """fp32: y[y_rows[i]] = x_base[x_rows[i]] @ weight.T + bias for i < n_real, y[y_rows[i]] = 0 for the other entries of y_rows;
    y is a fresh [out_rows, N] tensor every row of which y_rows names once.  `x` is any fp32 CUDA tensor whose element (row r, k) lies
    at x.data_ptr() + (r * pitch + k) * 4 for the rows x_rows names -- a token slice of a padded stack, read in place
    (include/vlmc.h: vlmc_linear_fwd_gather).  A computed row has the bits `linear_fwd` gives it."""
    _need_gpu(x, weight, x_rows, y_rows, *([bias] if bias is not None else []))
    N, K = weight.shape
    if x.dtype is not torch.float32 or weight.dtype is not torch.float32 or weight.stride(1) != 1 or weight.stride(0) < K or pitch < K or \
            (bias is not None and (bias.dtype is not torch.float32 or not bias.is_contiguous() or bias.shape != (N,))):
        raise TypeError("vlmc.linear_fwd_gather: fp32 x, weight [N, K] with contiguous rows and a contiguous fp32 bias expected")
    for t, n_ in ((x_rows, n_real), (y_rows, out_rows)):
        if t.dtype != torch.int32 or t.dim() != 1 or not t.is_contiguous() or t.shape[0] != n_:
            raise ValueError("vlmc.linear_fwd_gather: x_rows int32 [n_real], y_rows int32 [out_rows], contiguous")
    if not 0 <= n_real <= out_rows or out_rows < 1:
        raise ValueError("vlmc.linear_fwd_gather: 0 <= n_real <= out_rows")
    y = torch.empty((out_rows, N), dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().vlmc_linear_fwd_gather(x.data_ptr(), weight.data_ptr(), bias.data_ptr() if bias is not None else None, _lib.F32, N, K,
                                                  pitch, weight.stride(0), y.data_ptr(), N, x_rows.data_ptr(), y_rows.data_ptr(), int(n_real),
                                                  int(out_rows - n_real), _stream()))
    return y


_16BIT = (torch.float16, torch.bfloat16)


def attn_matmul_plan(a: torch.Tensor, b: torch.Tensor, _cuda_only: bool = True):
    """The arguments `vlmc_attn_matmul` takes for `torch.matmul(a, b)`, or None if the call is not one it computes: 3-D or
    4-D CUDA tensors of one 16-bit dtype, equal (or broadcast) batch dimensions, `a` contiguous along its last dimension,
    `b` along its last (attn @ v) or its second to last (q @ k.transpose(-2, -1)) -- read in place through their strides."""
    nd = a.dim()
    if nd != b.dim() or nd < 3 or nd > 4 or a.dtype != b.dtype or (a.dtype not in _16BIT and a.dtype is not torch.float32) or \
            (_cuda_only and not (a.is_cuda and b.is_cuda)):
        return None
    ash, bsh = a.shape, b.shape
    M, K, N = ash[-2], ash[-1], bsh[-1]
    if K != bsh[-2] or M == 0 or N == 0 or K == 0:
        return None
    sa, sb = a.stride(), b.stride()
    if sa[-1] != 1 and K != 1:
        return None
    if K == 1 or sb[-2] == 1:
        sbk, sbn = 1, sb[-1]
    elif sb[-1] == 1 or N == 1:
        sbk, sbn = sb[-2], 1
    else:
        return None
    if sa[-2] < 0 or sbk < 0 or sbn < 0:
        return None
    batch, sab, sbb = [], [], []
    for i in range(nd - 2):
        x, y = ash[i], bsh[i]
        if x != y and x != 1 and y != 1:
            return None
        n = x if x != 1 else y
        if n == 0:
            return None
        batch.append(n)
        sab.append(sa[i] if x != 1 else 0)
        sbb.append(sb[i] if y != 1 else 0)
    if nd == 3:
        batch, sab, sbb = [1] + batch, [0] + sab, [0] + sbb
    return (batch, M, N, K, sab[0], sab[1], sa[-2], sbb[0], sbb[1], sbk, sbn)


def attn_matmul(a: torch.Tensor, b: torch.Tensor, _plan=None, _try: bool = False) -> torch.Tensor:
    """`torch.matmul(a, b)` for the batched products of attention (q @ k^T, attn @ v: eva_vit.py:147,164;
    modeling_t5.py:590,638) on the batch-invariant MFMA kernel: an output element has the same bits whatever the batch
    count, M or N (include/vlmc.h: vlmc_attn_matmul).  `_try`: None instead of an error for a call the kernel does not take.
    fp32 operands (the reference's Q-Former): the fp32 matrix-core kernel, at most 65535 matrices per call."""
    if a.dtype is torch.float32 and _fast is None:
        plan = _plan if _plan is not None else attn_matmul_plan(a, b)
        if plan is not None and plan[0][0] * plan[0][1] > 65535:
            plan = None
        if plan is None:
            if _try:
                return None
            _need_gpu(a, b)
            raise TypeError("vlmc.attn_matmul: 3-D / 4-D tensors of one dtype expected, a contiguous along k, b along k or n")
        batch, M, N, K, sa0, sa1, sam, sb0, sb1, sbk, sbn = plan
        out = torch.empty((*(batch if a.dim() == 4 else batch[1:]), M, N), dtype=a.dtype, device=a.device)
        _lib.check(_lib.load().vlmc_attn_matmul(a.data_ptr(), b.data_ptr(), out.data_ptr(), _lib.F32, batch[0], batch[1], M, N, K, sa0, sa1, sam,
                                                sb0, sb1, sbk, sbn, batch[1] * M * N, M * N, N, _stream()))
        return out
    if _fast is not None:
        if not a.is_cuda:
            if _try:
                return None
            _need_gpu(a, b)
        out = _fast.attn_matmul(a, b, _stream())
        if out is None and not _try:
            _need_gpu(a, b)
            raise TypeError("vlmc.attn_matmul: 3-D / 4-D fp16 / bf16 tensors of one dtype expected, a contiguous along k, b along k or n")
        return out
    plan = _plan if _plan is not None else attn_matmul_plan(a, b)
    if plan is None:
        if _try:
            return None
        _need_gpu(a, b)
        raise TypeError("vlmc.attn_matmul: 3-D / 4-D fp16 / bf16 tensors of one dtype expected, a contiguous along k, b along k or n")
    batch, M, N, K, sa0, sa1, sam, sb0, sb1, sbk, sbn = plan
    nb0, nb1 = batch
    out = torch.empty((*(batch if a.dim() == 4 else batch[1:]), M, N), dtype=a.dtype, device=a.device)
    _lib.check(_lib.load().vlmc_attn_matmul(a.data_ptr(), b.data_ptr(), out.data_ptr(), _DT[a.dtype], nb0, nb1, M, N, K, sa0, sa1, sam,
                                            sb0, sb1, sbk, sbn, nb1 * M * N, M * N, N, _stream()))
    return out


def rms_norm(x: torch.Tensor, weight: torch.Tensor, eps: float, rsqrt_mode: int = 0) -> torch.Tensor:
    """`weight * (x * rsqrt(x.float().pow(2).mean(-1, keepdim=True) + eps)).to(x.dtype)` -- T5LayerNorm / LlamaRMSNorm, op for op --
    in one launch (include/vlmc.h: vlmc_rms_norm); the mean is `row_mean`'s.  16-bit CUDA x [.., n] and weight [n] of one dtype."""
    _need_gpu(x, weight)
    if _fast is not None and hasattr(_fast, "rms_norm"):
        out = _fast.rms_norm(x, weight, float(eps), int(rsqrt_mode), _stream())
        if out is None:
            raise TypeError("vlmc.rms_norm: fp16 / bf16 x [.., n] and a contiguous weight [n] of the same dtype expected")
        return out
    n = x.shape[-1]
    if x.dtype not in _16BIT or weight.dtype != x.dtype or weight.shape != (n,) or not weight.is_contiguous() or n == 0:
        raise TypeError("vlmc.rms_norm: fp16 / bf16 x [.., n] and a contiguous weight [n] of the same dtype expected")
    x2 = x.reshape(-1, n)
    if x2.stride(1) != 1 or (x2.shape[0] > 1 and x2.stride(0) < n):
        x2 = x2.contiguous()
    rows = x2.shape[0]
    out = torch.empty((rows, n), dtype=x.dtype, device=x.device)
    _lib.check(_lib.load().vlmc_rms_norm(x2.data_ptr(), _DT[x.dtype], rows, n, x2.stride(0) if rows > 1 else n, weight.data_ptr(), float(eps),
                                         int(rsqrt_mode), out.data_ptr(), n, _stream()))
    return out.reshape(x.shape)


_sdpa_max_keys = {}


def sdpa_plan(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor):
    """(B, H, Tq, Tk, d) if `vlmc_sdpa_fwd` computes `F.scaled_dot_product_attention(q, k, v)` for these operands, else None:
    4-D CUDA tensors [B, H, T, d] of one 16-bit dtype, unit stride along d, d a multiple of 8 up to 128, the head's keys
    within what its K and V may take of LDS."""
    if q.dim() != 4 or k.dim() != 4 or v.dim() != 4 or q.dtype not in _16BIT or k.dtype != q.dtype or v.dtype != q.dtype or \
            not (q.is_cuda and k.is_cuda and v.is_cuda):
        return None
    B, H, Tq, d = q.shape
    Tk = k.shape[2]
    if k.shape != (B, H, Tk, d) or v.shape != (B, H, Tk, d) or d % 8 or d > 128 or min(B, H, Tq, Tk, d) <= 0:
        return None
    if q.stride(3) != 1 or k.stride(3) != 1 or v.stride(3) != 1 or min(q.stride(2), k.stride(2), v.stride(2)) < 0:
        return None
    mk = _sdpa_max_keys.get(d)
    if mk is None:
        mk = _sdpa_max_keys[d] = int(_lib.load().vlmc_sdpa_max_keys(d))
    if Tk > mk:
        return None
    return B, H, Tq, Tk, d


def sdpa(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, scale=None, _try: bool = False, causal: bool = False):
    """`F.scaled_dot_product_attention(q, k, v[, is_causal=True])` (no mask, no dropout) on the fused, batch-invariant MFMA kernel
    (include/vlmc.h: vlmc_sdpa_fwd).  The result is a [B, H, Tq, d] view of a [B, Tq, H, d] buffer: the `transpose(1, 2)
    .reshape(B, Tq, H * d)` that follows in every model file is then free.  LAYOUT CONTRACT: unlike torch's math path the
    result is NOT contiguous in [B, H, Tq, d] order -- code that goes on with `.view(B * H, Tq, d)` needs `.contiguous()` first
    (`.reshape` and `.transpose(1, 2).reshape(...)` work as they are).  `_try`: None for a call the kernel does not take, also
    when only the C entry point refuses it."""
    if _fast is not None and q.is_cuda and hasattr(_fast, "sdpa"):
        out = _fast.sdpa(q, k, v, float(q.shape[-1] ** -0.5 if scale is None else scale), bool(causal), _stream())
        if out is None and not _try:
            raise TypeError("vlmc.sdpa: [B, H, T, d] fp16 / bf16 CUDA tensors of one dtype expected, d a multiple of 8 up to 128, "
                            "unit stride along d, at most vlmc_sdpa_max_keys(d) keys, a positive scale")
        return out
    plan = sdpa_plan(q, k, v) if (scale is None or 0.0 < float(scale) < float("inf")) else None
    if plan is None:
        if _try:
            return None
        _need_gpu(q, k, v)
        raise TypeError("vlmc.sdpa: [B, H, T, d] fp16 / bf16 CUDA tensors of one dtype expected, d a multiple of 8 up to 128, "
                        "unit stride along d, at most vlmc_sdpa_max_keys(d) keys")
    B, H, Tq, Tk, d = plan
    out = torch.empty((B, Tq, H, d), dtype=q.dtype, device=q.device)
    sq, sk, sv = q.stride(), k.stride(), v.stride()
    rc = _lib.load().vlmc_sdpa_fwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), _DT[q.dtype], B, H, Tq, Tk, d,
                                   sq[0], sq[1], sq[2], sk[0], sk[1], sk[2], sv[0], sv[1], sv[2], Tq * H * d, d, H * d,
                                   float(d ** -0.5 if scale is None else scale), int(bool(causal)), _stream())
    if rc:
        if _try and rc == _lib.VLMC_EINVAL:                               # the entry point refuses the call: the caller's own op runs
            return None
        _lib.check(rc)
    return out.transpose(1, 2)


def row_mean(x: torch.Tensor, keepdim: bool = False) -> torch.Tensor:
    """`x.mean(-1, keepdim=keepdim)` for fp32 CUDA tensors with a fixed, batch-invariant summation order (the norms of a
    replayed block: include/vlmc.h: vlmc_row_mean)."""
    _need_gpu(x)
    if _fast is not None:
        out = _fast.row_mean(x, bool(keepdim), _stream())
        if out is None:
            raise TypeError("vlmc.row_mean: a non-empty fp32 tensor expected")
        return out
    if x.dtype != torch.float32 or x.dim() < 1 or x.shape[-1] == 0:
        raise TypeError("vlmc.row_mean: a non-empty fp32 tensor expected")
    n = x.shape[-1]
    x2 = x.reshape(-1, n)
    if x2.stride(1) != 1 or (x2.shape[0] > 1 and x2.stride(0) < n):
        x2 = x2.contiguous()
    rows = x2.shape[0]
    out = torch.empty(x.shape[:-1] + ((1,) if keepdim else ()), dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().vlmc_row_mean(x2.data_ptr(), rows, n, x2.stride(0) if rows > 1 else n, out.data_ptr(), _stream()))
    return out


def gelu(x: torch.Tensor, approximate: str = "none") -> torch.Tensor:
    """`F.gelu(x, approximate=...)` for 16-bit CUDA tensors with one instruction sequence for every element (include/vlmc.h:
    vlmc_gelu): torch's body arithmetic also where torch itself switches to another (its kernels' last partial block)."""
    _need_gpu(x)
    if (x.dtype not in _16BIT and x.dtype != torch.float32) or approximate not in ("none", "tanh"):
        raise TypeError("vlmc.gelu: an fp16 / bf16 / fp32 tensor and approximate 'none' or 'tanh' expected")
    xc = x if x.is_contiguous() else x.contiguous()
    y = torch.empty_like(xc)
    _lib.check(_lib.load().vlmc_gelu(xc.data_ptr(), y.data_ptr(), xc.numel(), _DT[x.dtype], 1 if approximate == "tanh" else 0, _stream()))
    return y


def softmax_rows(x: torch.Tensor, out_dtype=None) -> torch.Tensor:
    """`torch.softmax(x, -1[, dtype=out_dtype])` on the padding-invariant kernel (include/vlmc.h: vlmc_softmax_rows): fp32 -> fp32,
    fp16 / bf16 -> the same dtype (fp32 arithmetic, one rounding) or fp32."""
    _need_gpu(x)
    out_dtype = x.dtype if out_dtype is None else out_dtype
    if x.dim() < 1 or x.shape[-1] == 0 or not ((x.dtype == torch.float32 and out_dtype == torch.float32) or
                                               (x.dtype in _16BIT and out_dtype in (x.dtype, torch.float32))):
        raise TypeError("vlmc.softmax_rows: a non-empty fp32 tensor, or fp16 / bf16 with the same or fp32 output dtype, expected")
    n = x.shape[-1]
    x2 = x.reshape(-1, n)
    if x2.stride(1) != 1 or (x2.shape[0] > 1 and x2.stride(0) < n):
        x2 = x2.contiguous()
    rows = x2.shape[0]
    out = torch.empty(x.shape, dtype=out_dtype, device=x.device)
    _lib.check(_lib.load().vlmc_softmax_rows(x2.data_ptr(), _DT[x.dtype], rows, n, x2.stride(0) if rows > 1 else n, out.data_ptr(),
                                             _DT[out_dtype], n, _stream()))
    return out


_attn_max_keys = {}
_I64x3 = ctypes.c_int64 * 3
_I64x4 = ctypes.c_int64 * 4


def attn_fused_plan(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, adds=(), _cuda_only: bool = True):
    """(B, H, Tq, Tk, d) if `vlmc_attn_fwd` computes the reference-op attention chain for these operands, else None: q [B, H, Tq, d],
    k, v [B, H, Tk, d] CUDA tensors of one 16-bit dtype with unit stride along d (k is the operand BEFORE its transpose), d a multiple
    of 8 up to 128, Tk within what a head's K and V may take of LDS; at most two addends of the same dtype, broadcastable to
    [B, H, Tq, Tk] (any strides: T5's position bias is a permuted table lookup)."""
    if q.dim() != 4 or k.dim() != 4 or v.dim() != 4 or q.dtype not in _16BIT or k.dtype != q.dtype or v.dtype != q.dtype or \
            (_cuda_only and not (q.is_cuda and k.is_cuda and v.is_cuda)):
        return None
    B, H, Tq, d = q.shape
    Tk = k.shape[2]
    if k.shape != (B, H, Tk, d) or v.shape != (B, H, Tk, d) or d % 8 or d > 128 or min(B, H, Tq, Tk, d) <= 0:
        return None
    for t in (q, k, v):
        st = t.stride()
        if (st[3] != 1 and d != 1) or min(st[0], st[1], st[2]) < 0:
            return None
    mk = _attn_max_keys.get(d)
    if mk is None:
        mk = _attn_max_keys[d] = int(_lib.load().vlmc_attn_max_keys(d))
    if Tk > mk or len(adds) > 2:
        return None
    for t in adds:
        if t.dtype != q.dtype or t.device != q.device or t.dim() > 4 or t.dim() == 0 or t.requires_grad:
            return None
        sh = (1,) * (4 - t.dim()) + tuple(t.shape)
        if sh[3] != Tk or sh[0] not in (1, B) or sh[1] not in (1, H) or sh[2] not in (1, Tq) or (t.stride(-1) < 1 and Tk != 1) or min(t.stride()) < 0:
            return None
    return B, H, Tq, Tk, d


def _add_strides(t, B, H, Tq):
    sh = (1,) * (4 - t.dim()) + tuple(t.shape)
    st = (0,) * (4 - t.dim()) + tuple(t.stride())
    return _I64x4(*(0 if n == 1 else s for n, s in zip(sh[:3], st[:3])), max(1, st[3]))


def attn_fused(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, mul=None, adds=(), _plan=None, q_len=None, k_len=None) -> torch.Tensor:
    """The reference-op attention chain in one launch (include/vlmc.h: vlmc_attn_fwd):
        s = q @ k^T;  [s = s * mul];  [s = s + adds[0] [+ adds[1]]];  p = softmax in fp32, rounded;  out = p @ v
    every intermediate rounded to the dtype where the tensor op would round it -- the bits of the unfused sequence on
    `attn_matmul`, torch's elementwise ops and `softmax_rows`.  Returns [B, H, Tq, d] as a VIEW of a contiguous [B, Tq, H, d]
    tensor (the model's `.transpose(1, 2).reshape(B, Tq, H * d)` is free).  `mul`: a Python float already rounded to fp32.
    `q_len` / `k_len` (int32 device tensors [B], a padded group of ragged samples): output rows behind q_len[b] are zeros; keys
    behind k_len[b] -- which an addend must mask -- are skipped (include/vlmc.h: vlmc_attn_fwd_lens)."""
    _need_gpu(q, k, v, q_len, k_len)
    plan = _plan if _plan is not None else attn_fused_plan(q, k, v, adds)
    if plan is None:
        raise TypeError("vlmc.attn_fused: q [B, H, Tq, d], k, v [B, H, Tk, d] of one 16-bit dtype with unit stride along d, at most "
                        "vlmc_attn_max_keys(d) keys, at most two broadcastable 16-bit addends expected")
    B, H, Tq, Tk, d = plan
    out = torch.empty((B, Tq, H, d), dtype=q.dtype, device=q.device)
    a0 = adds[0] if len(adds) > 0 else None
    a1 = adds[1] if len(adds) > 1 else None
    for ln in (q_len, k_len):
        if ln is not None and (ln.dtype != torch.int32 or ln.shape != (B,) or not ln.is_contiguous()):
            raise TypeError("vlmc.attn_fused: q_len / k_len must be contiguous int32 [batch] tensors")
    if k_len is not None and a0 is None:
        raise ValueError("vlmc.attn_fused: k_len needs an addend that masks the keys behind it")
    _lib.check(_lib.load().vlmc_attn_fwd_lens(
        q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), _DT[q.dtype], B, H, Tq, Tk, d, _I64x3(*q.stride()[:3]),
        _I64x3(*k.stride()[:3]), _I64x3(*v.stride()[:3]), 0 if mul is None else 1, 0.0 if mul is None else float(mul),
        None if a0 is None else a0.data_ptr(), None if a0 is None else _add_strides(a0, B, H, Tq),
        None if a1 is None else a1.data_ptr(), None if a1 is None else _add_strides(a1, B, H, Tq),
        None if q_len is None else q_len.data_ptr(), None if k_len is None else k_len.data_ptr(), _stream()))
    return out.permute(0, 2, 1, 3)


def pack_24(weight: torch.Tensor, mask: torch.Tensor):
    """(values [out, in / 2], meta [out, in / 8] uint8) of a 16-bit weight whose `mask` (bool, 1 = keep) keeps exactly two of every
    four consecutive input columns (include/vlmc.h: vlmc_pack_24).  Raises if a group keeps another number."""
    _need_gpu(weight, mask)
    out_f, in_f = weight.shape
    if weight.dtype not in _16BIT or mask.shape != weight.shape or mask.dtype not in (torch.bool, torch.uint8) or in_f % 8 or \
            weight.stride(1) != 1 or mask.stride(1) != 1:
        raise TypeError("vlmc.pack_24: a 16-bit weight [out, in] (in % 8 == 0) and a bool mask of the same shape, rows contiguous, expected")
    values = torch.empty((out_f, in_f // 2), dtype=weight.dtype, device=weight.device)
    meta = torch.empty((out_f, in_f // 8), dtype=torch.uint8, device=weight.device)
    bad = torch.zeros(1, dtype=torch.int32, device=weight.device)
    _lib.check(_lib.load().vlmc_pack_24(weight.data_ptr(), _DT[weight.dtype], out_f, in_f, weight.stride(0), mask.data_ptr(), mask.stride(0),
                                        values.data_ptr(), meta.data_ptr(), bad.data_ptr(), _stream()))
    n_bad = int(bad.item())
    if n_bad:
        raise ValueError(f"vlmc.pack_24: {n_bad} groups of four columns do not keep exactly two weights: not a 2:4 mask")
    return values, meta


def unpack_24(values: torch.Tensor, meta: torch.Tensor, want_mask: bool = True):
    """The dense [out, in] weight (zeros where nothing was kept) and the bool mask of `pack_24`'s output (vlmc_unpack_24)."""
    _need_gpu(values, meta)
    out_f, half = values.shape
    in_f = 2 * half
    if values.dtype not in _16BIT or meta.dtype != torch.uint8 or in_f % 8 or meta.shape != (out_f, in_f // 8) or \
            not values.is_contiguous() or not meta.is_contiguous():
        raise TypeError("vlmc.unpack_24: contiguous values [out, in / 2] (16-bit) and meta [out, in / 8] (uint8) expected")
    w = torch.empty((out_f, in_f), dtype=values.dtype, device=values.device)
    mask = torch.empty((out_f, in_f), dtype=torch.bool, device=values.device) if want_mask else None
    bad = torch.zeros(1, dtype=torch.int32, device=values.device)
    _lib.check(_lib.load().vlmc_unpack_24(values.data_ptr(), meta.data_ptr(), _DT[values.dtype], out_f, in_f, w.data_ptr(), in_f,
                                          None if mask is None else mask.data_ptr(), in_f, bad.data_ptr(), _stream()))
    if int(bad.item()):
        raise ValueError("vlmc.unpack_24: the metadata holds codes that pack_24 never writes")
    return w, mask


def hessian_accum(H: torch.Tensor, x: torch.Tensor, alpha: float, beta: float) -> torch.Tensor:
    """H = alpha * H + beta * x^T x on the tiles on and below the diagonal (SparseGPT.add_batch, sparsegpt_pruner.py:76-79,
    with alpha = n/(n+b) and beta = 2/(n+b)); x [rows, in] fp16 / bf16 / fp32.  `symmetrize_lower(H)` completes H."""
    _need_gpu(H, x)
    assert H.dtype == torch.float32 and H.dim() == 2 and H.shape[0] == H.shape[1] == x.shape[-1] and H.stride(1) == 1
    x2 = x.reshape(-1, x.shape[-1])
    if x2.stride(1) != 1:
        x2 = x2.contiguous()
    lib = _lib.load()
    code = _dtype_code(x2)
    need = lib.vlmc_hessian_workspace(code, x2.shape[0], x2.shape[1])
    ws = _hessian_ws.get(need, x.device)
    _lib.check(lib.vlmc_hessian_accum(x2.data_ptr(), code, x2.shape[0], x2.shape[1], x2.stride(0), H.data_ptr(), H.stride(0),
                                      float(alpha), float(beta), ws.data_ptr(), ws.numel(), _stream()))
    return H


def symmetrize_lower(H: torch.Tensor) -> torch.Tensor:
    """Copy the lower triangle of H onto the upper one (once, before a Hessian from `hessian_accum` is read in full)."""
    _need_gpu(H)
    assert H.dtype == torch.float32 and H.dim() == 2 and H.shape[0] == H.shape[1] and H.stride(1) == 1
    _lib.check(_lib.load().vlmc_symmetrize_lower(H.data_ptr(), H.shape[0], H.stride(0), _stream()))
    return H


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
# Batched entry points: all hook inputs / statistics / linears of one transformer block per launch
# ---------------------------------------------------------------------------------------
def _stat_jobs(xs, outs, call_tokens=None):
    jobs = (_lib.StatJob * len(xs))()
    calls = xs[0].shape[0]
    for j, (x, out) in enumerate(zip(xs, outs)):
        ct = call_tokens[j] if call_tokens is not None else None
        if ct is not None:
            assert ct.dtype == torch.int32 and ct.is_cuda and ct.is_contiguous() and ct.numel() == calls
        _need_gpu(x, out)
        assert x.dim() == 3 and x.stride(2) == 1 and x.shape[0] == calls and x.dtype == xs[0].dtype
        assert out.dtype == torch.float32 and out.shape == (calls, x.shape[2]) and out.stride(1) == 1
        tokens, in_f = x.shape[1], x.shape[2]
        row_stride = x.stride(1) if tokens > 1 else in_f
        call_stride = x.stride(0) if calls > 1 else tokens * row_stride
        assert row_stride >= in_f
        jobs[j] = _lib.StatJob(x.data_ptr(), out.data_ptr(), in_f, tokens, row_stride, call_stride,
                               out.stride(0) if calls > 1 else in_f, ct.data_ptr() if ct is not None else None)
    return jobs, calls


def act_sqnorm_batch(xs, outs=None, call_tokens=None):
    """`act_sqnorm` for several hook inputs ([calls, tokens, in_j], same dtype and calls) in one launch.
    `outs[j]` may be a column slice of a wider fp32 buffer (row stride = the buffer's width).  `call_tokens[j]` (int32 device
    tensor [calls] or None): only the first call_tokens[j][c] token rows of call c of input j count -- a padded group of ragged
    calibration samples."""
    xs = [x if x.stride(-1) == 1 else x.contiguous() for x in xs]
    if outs is None:
        outs = [torch.empty((x.shape[0], x.shape[2]), dtype=torch.float32, device=x.device) for x in xs]
    jobs, calls = _stat_jobs(xs, outs, call_tokens)
    _lib.check(_lib.load().vlmc_act_sqnorm_batch(jobs, len(xs), _dtype_code(xs[0]), calls, _stream()))
    return outs


def _update_jobs(scalers, normsqs, sqrt_outs):
    jobs = (_lib.UpdateJob * len(scalers))()
    calls = normsqs[0].shape[0]
    for j, (s, nsq, sq) in enumerate(zip(scalers, normsqs, sqrt_outs)):
        _need_gpu(s, nsq, sq)
        assert s.dtype == torch.float32 and s.is_contiguous() and nsq.dtype == torch.float32
        assert nsq.shape == (calls, s.numel()) and nsq.stride(1) == 1
        assert sq is None or (sq.dtype == torch.float32 and sq.is_contiguous() and sq.numel() == s.numel())
        jobs[j] = _lib.UpdateJob(s.data_ptr(), nsq.data_ptr(), sq.data_ptr() if sq is not None else None, s.numel(),
                                 nsq.stride(0) if calls > 1 else s.numel())
    return jobs, calls


def wanda_scaler_update_batch(scalers, nsamples_before: int, normsqs, batch: int = 1, sqrt_outs=None) -> int:
    """`wanda_scaler_update` for several statistics that saw the same calls, in one launch."""
    sqrt_outs = [None] * len(scalers) if sqrt_outs is None else sqrt_outs
    jobs, calls = _update_jobs(scalers, normsqs, sqrt_outs)
    _lib.check(_lib.load().vlmc_wanda_scaler_update_batch(jobs, len(scalers), nsamples_before, calls, batch, _stream()))
    return nsamples_before + calls * batch


_batch_ws = Workspace(zeroed=True)


def _select_jobs(weights, sqrt_rows, code, ks, masks, partials, ws_holder):
    lib = _lib.load()
    n = len(weights)
    jobs = (_lib.SelectJob * n)()
    nbytes = lib.vlmc_wanda_select_workspace(code, *weights[0].shape)
    ws = ws_holder.get(nbytes * n, weights[0].device) if nbytes else None
    for j, (w, sq, k, mk, pt) in enumerate(zip(weights, sqrt_rows, ks, masks, partials)):
        _need_gpu(w, sq, mk, pt)
        if w.dim() != 2 or w.stride(1) != 1:
            raise ValueError("wanda_select expects a row-major 2-D weight")
        assert w.dtype == weights[0].dtype
        out_f, in_f = w.shape
        assert sq.dtype == torch.float32 and sq.is_contiguous() and sq.numel() == in_f
        assert mk.dtype == torch.bool and mk.is_contiguous() and mk.shape == w.shape
        assert pt is None or (pt.dtype == torch.float64 and pt.is_contiguous()
                              and pt.numel() >= lib.vlmc_wanda_select_partials(code, out_f, in_f))
        jobs[j] = _lib.SelectJob(w.data_ptr(), out_f, in_f, w.stride(0), sq.data_ptr(), int(k), mk.data_ptr(),
                                 pt.data_ptr() if pt is not None else None,
                                 ws.data_ptr() + j * nbytes if nbytes else None, nbytes)
    return jobs


def wanda_select_batch(weights, sqrt_rows, mode: str, *, ks=None, n: int = 0, m: int = 0, apply_zero: bool = True,
                       masks=None, partials=None):
    """`wanda_select` for all linears of a block (same dtype, mode, n:m) with as few launches as the shapes
    allow.  Returns (masks, partials) as lists."""
    code = _MODES[mode]
    lib = _lib.load()
    ks = [0] * len(weights) if ks is None else ks
    if masks is None:
        masks = [torch.empty(tuple(w.shape), dtype=torch.bool, device=w.device) for w in weights]
    if partials is None:
        partials = [torch.empty(lib.vlmc_wanda_select_partials(code, *w.shape), dtype=torch.float64, device=w.device)
                    for w in weights]
    jobs = _select_jobs(weights, sqrt_rows, code, ks, masks, partials, _batch_ws)
    _lib.check(lib.vlmc_wanda_select_batch(jobs, len(weights), _dtype_code(weights[0]), code, int(n), int(m),
                                           int(bool(apply_zero)), _stream()))
    return masks, [p[:lib.vlmc_wanda_select_partials(code, *w.shape)] if p is not None else None
                   for p, w in zip(partials, weights)]


# ---------------------------------------------------------------------------------------
# Launch plans: pre-validated, pre-bound C-ABI calls for hot loops that issue thousands of
# launches per step (bench.py, the per-block pruner loop).  A plan is a zero-argument
# callable; tensors referenced by a plan must stay alive and must not be reallocated.
# ---------------------------------------------------------------------------------------
_c_int64 = ctypes.c_int64
_SCORE_MODES = {"weight": _lib.SCORE_W, "score": _lib.SCORE_S, "absw_score": _lib.SCORE_ABSW_S}
_score_ws = Workspace()


def score_select(weights, mode: str, *, scopes, scope_ks, scores=None, prev_keeps=None, protect_ks=None, apply_weights: bool = True,
                 keeps=None):
    """One threshold per SCOPE over many tensors (global_pruner.py:107-148, :166-169, :188-190).

    `weights[i]`: contiguous parameter tensor (any of fp32/fp16/bf16, may differ per tensor; None allowed in
    mode "score"); `scores[i]`: fp32 tensor of the same numel (modes "score", "absw_score"); `scopes[i]`: scope id;
    `scope_ks[s]`: rank of the threshold (the k-th smallest score of the scope, 1-based); keep = score > threshold;
    `protect_ks[i]`: the protect_k largest scores of tensor i count as FLT_MAX; `prev_keeps[i]`: bool mask multiplied
    into the scores.  Multiplies the weights by the masks in place when `apply_weights`.  Returns the keep masks."""
    n = len(scopes)
    weights = list(weights) if weights is not None else [None] * n
    scores = list(scores) if scores is not None else [None] * n
    prev_keeps = list(prev_keeps) if prev_keeps is not None else [None] * n
    protect_ks = list(protect_ks) if protect_ks is not None else [0] * n
    code = _SCORE_MODES[mode]
    first = next(t for t in list(weights) + list(scores) if t is not None)
    if keeps is None:
        keeps = [torch.empty((weights[i] if weights[i] is not None else scores[i]).shape, dtype=torch.bool, device=first.device)
                 for i in range(n)]
    jobs = (_lib.ScoreJob * n)()
    for i in range(n):
        w, s, pk, kp = weights[i], scores[i], prev_keeps[i], keeps[i]
        _need_gpu(w, s, pk, kp)
        numel = kp.numel()
        for t, dt in ((w, None), (s, torch.float32), (pk, torch.bool), (kp, torch.bool)):
            if t is None:
                continue
            if not t.is_contiguous() or t.numel() != numel or (dt is not None and t.dtype != dt):
                raise ValueError("score_select: operands of a job must be contiguous, of the same numel, scores fp32, masks bool")
        jobs[i] = _lib.ScoreJob(w.data_ptr() if w is not None else None, s.data_ptr() if s is not None else None,
                                pk.data_ptr() if pk is not None else None, kp.data_ptr(), numel, int(protect_ks[i]), int(scopes[i]),
                                _dtype_code(w) if w is not None else _lib.F32)
    ks = (_c_int64 * len(scope_ks))(*[int(k) for k in scope_ks])
    lib = _lib.load()
    nbytes = lib.vlmc_score_select_workspace(n, len(scope_ks))
    ws = _score_ws.get(nbytes, first.device)
    _lib.check(lib.vlmc_score_select(jobs, n, ks, len(scope_ks), code, int(bool(apply_weights)), ws.data_ptr(), ws.numel(), _stream()))
    return keeps


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


def plan_act_sqnorm_batch(xs, outs):
    jobs, calls = _stat_jobs(xs, outs)
    return _bind(_lib.load().vlmc_act_sqnorm_batch, (jobs, len(xs), _dtype_code(xs[0]), calls, _stream()))


def plan_scaler_update_batch(scalers, nsamples_before: int, normsqs, batch: int, sqrt_outs):
    jobs, calls = _update_jobs(scalers, normsqs, sqrt_outs)
    return _bind(_lib.load().vlmc_wanda_scaler_update_batch, (jobs, len(scalers), nsamples_before, calls, batch, _stream()))


def plan_select_batch(weights, sqrt_rows, mode: str, *, ks=None, n=0, m=0, apply_zero=True, masks, partials):
    """Pre-bound batched select.  The plan owns its SEL_MATRIX workspace (plans may be replayed in any order)."""
    code = _MODES[mode]
    ks = [0] * len(weights) if ks is None else ks
    holder = Workspace(zeroed=True)
    jobs = _select_jobs(weights, sqrt_rows, code, ks, masks, partials, holder)
    run = _bind(_lib.load().vlmc_wanda_select_batch, (jobs, len(weights), _dtype_code(weights[0]), code, int(n), int(m),
                                                     int(bool(apply_zero)), _stream()))
    run.workspace = holder
    return run
