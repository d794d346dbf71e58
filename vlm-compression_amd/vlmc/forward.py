"""The dense calibration forward of a block's linears on the batch-invariant MFMA kernel (`vlmc_linear_fwd`).

The reference replays every transformer block once per calibration sample (`layer(inps[j], **caches[j])`,
wanda_pruner.py:308-311, :343-346).  The replay engine forwards groups of samples instead
(`lavis/compression/pruners/calibration.py: walk_blocks`); for the statistics -- and therefore the masks -- not to depend
on how the samples were grouped (group size, ragged shapes, sharding over GPUs), the GEMMs of the block must give a row
the same bits whatever else is in the launch.  A GEMM library picks its kernel by problem size; `vlmc_linear_fwd` does
not (csrc/gemm_nt.hip).  While `invariant_linears(modules)` is active, the forward of those `nn.Linear` modules (and the
dense branch of the SparseLoRA `Linear`) runs on it whenever it can: 16-bit weights and activations of one dtype (an
active autocast to the weights' dtype casts the input like autocast would), no gradients.  Everything else -- fp32
models, odd widths -- stays with `F.linear`.  `VLMC_LINEAR_FWD=0` switches the kernel off.

Sibling linears share a launch (`VLMC_LINEAR_GROUP=0`: never).  The model calls `self.q(h)`, `self.k(h)`, `self.v(h)`
one after the other on the SAME tensor (modeling_t5.py:546-572, modeling_llama.py:204-206; wi_0 / wi_1 of the gated FFN,
modeling_t5.py:337-341; k / v of a cross-attention): three launches that each fill a fraction of the chip (a decoder
projection of 128 x 16 tokens is 64 tiles of 256 x 256 on 256 CUs).  Which linears are siblings is LEARNED from the
calls -- consecutive calls of patched linears whose input is the very same tensor (same storage, offset, shape, strides,
version counter, kept alive in between) -- never assumed from names.  Once a group is known, the call of its first member
computes every member's product in one `vlmc_linear_fwd_group` launch and keeps the others' outputs for their calls, which
check that they are handed the tensor the products were computed from (anything else: the stash is dropped and the linear
computes for itself).  The products are the bits the single launches give (tests/test_gemm_gpu.py); module hooks fire per
module as before, only `forward` is replaced."""
from __future__ import annotations

import contextlib
import os
import threading
import weakref

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops

_active = 0
stats = {"kernel": 0, "library": 0, "grouped_launches": 0, "served_from_group": 0, "stash_dropped": 0, "attn_kernel": 0,
         "attn_library": 0, "mean_kernel": 0, "sdpa_kernel": 0, "sdpa_library": 0, "norm_kernel": 0, "softmax_kernel": 0,
         "attn_fused": 0, "attn_chain_unfused": 0, "attn_fused_checks": 0, "gelu_kernel": 0, "kernel_rows": 0, "attn_fused_lens": 0,
         "kernel_slices": 0}

# first member of a learned sibling group -> tuple of weak references to all members, in call order
_SIBLINGS = weakref.WeakKeyDictionary()


def enabled():
    return os.environ.get("VLMC_LINEAR_FWD", "1") != "0"


def grouping_enabled():
    return os.environ.get("VLMC_LINEAR_GROUP", "1") != "0"


def f32_enabled():
    """fp32 linears / attention products / GELU of a replayed forward on the batch-invariant fp32 kernels (the reference's fp32
    Q-Former); `VLMC_LINEAR_F32=0`: fp32 modules stay with the library, as before round 6 (a finished fp32 tower is then not stacked)."""
    return os.environ.get("VLMC_LINEAR_F32", "1") != "0"


def _prepare(x, weight, bias):
    """What the kernel would be fed for `F.linear(x, weight, bias)`, or None if the call stays with the library."""
    if not (_active and not torch.is_grad_enabled() and weight.is_cuda):
        return None
    xin = x
    autocast = torch.is_autocast_enabled()
    if autocast and x.is_floating_point() and x.dtype != weight.dtype and torch.get_autocast_gpu_dtype() == weight.dtype:
        xin = x.to(weight.dtype)                                # what autocast does to the input of a linear
    b = bias
    if b is not None and b.dtype != weight.dtype and autocast and torch.get_autocast_gpu_dtype() == weight.dtype:
        b = b.to(weight.dtype)
    if ops.linear_fwd_supported(xin, weight, b) and (not autocast or torch.get_autocast_gpu_dtype() == weight.dtype):
        return xin, b
    return None


# ---- a PADDED group of ragged calibration samples: the linears skip the padding rows ------------------------------------------
# `calibration.walk_blocks` forwards ragged samples of a block as ONE [samples, longest, d] batch (zero rows behind a sample's own,
# masks at the dtype's minimum).  On the bench's ragged set (prompts of 8..128 tokens) more than half of those rows are padding, and
# the linears are where a block's time goes.  While such a forward runs, `_ROW_MAPS` maps (samples, padded tokens) to the row map of
# that layout (real rows first; built by calibration.plan_padded): a linear whose input has that leading shape runs on
# `vlmc_linear_fwd_rows` -- padding rows are neither loaded nor multiplied, and come out as zeros, so they stay zero through the
# whole block (every op of a block is row-wise but attention, which masks the padded keys).  A sample's own rows keep the bits of
# its own forward (tests/test_gemm_gpu.py).  `VLMC_ROW_MAP=0`: every row is computed, as before round 6.
_ROW_MAPS = None


def row_map_enabled():
    return os.environ.get("VLMC_ROW_MAP", "1") != "0"


_PAD_LENGTHS = None           # {padded tokens: int32 device tensor [samples]}: each sample's own token rows (the fused attention skips the rest)
# fp32 blocks (the reference's Q-Former) that SLICE the padded stack along its tokens -- a BERT layer of the Q-Former sends
# `attention_output[:, :query_length]` and `attention_output[:, query_length:]` through feed-forward halves of their own
# (Qformer.py:434-466): `_PAD_HOST` {(samples, padded tokens): each sample's own token rows, on the host} lets a linear fed such a
# view (recognised by what it is a view OF: `_slice_rows`) read the slice's real rows in place and write a compact
# [samples, slice tokens, N] output, whose rows carry their map as an attribute of the tensor (`_vlmc_rows`) -- through `F.gelu`
# too -- for the next linear: nothing is inferred from the shape of a derived tensor.
_PAD_HOST = None
_SLICES = None                # (samples, padded tokens, first token, slice tokens) -> (x_rows, y_rows, n_real, out_rows) of this context


def int32_on(values, device):
    """A small host array (token counts, a row map) as an int32 tensor on `device` WITHOUT draining the GPU: `torch.tensor(.., device=)`
    copies from pageable memory and synchronises the stream -- in the middle of a capture phase the host then waits for the whole walk
    before it and issues the rest of the phase against an idle GPU (tools/find_syncs.py, tools/phase_timeline.py).  Pinned staging
    buffer, asynchronous copy; the caching host allocator keeps the buffer until the copy has run."""
    t = torch.as_tensor(values, dtype=torch.int32)
    device = torch.device(device)
    if device.type != "cuda":
        return t.to(device)
    return t.pin_memory().to(device, non_blocking=True)


@contextlib.contextmanager
def padded_rows(maps, lengths=None, host=None):
    """`maps`: {(samples, padded tokens): (int32 device tensor [samples * padded tokens], number of real rows)} or None;
    `lengths`: {padded tokens: int32 device tensor [samples]} or None -- what the fused attention takes as q_len / k_len;
    `host`: {(samples, padded tokens): sequence of each sample's own token rows} or None -- for token slices of the stack (fp32)."""
    global _ROW_MAPS, _PAD_LENGTHS, _PAD_HOST, _SLICES
    prev = _ROW_MAPS, _PAD_LENGTHS, _PAD_HOST, _SLICES
    on = row_map_enabled()
    _ROW_MAPS = maps if (maps and on) else None
    _PAD_LENGTHS = lengths if (lengths and on) else None
    _PAD_HOST = host if (host and on and os.environ.get("VLMC_ROW_SLICES", "1") != "0") else None
    _SLICES = {}
    try:
        yield
    finally:
        _ROW_MAPS, _PAD_LENGTHS, _PAD_HOST, _SLICES = prev


def _slice_rows(x, K):
    """(x_rows, y_rows, n_real, out_rows) for a [samples, L, K] view that IS rows [a, a + L) of every sample of a contiguous
    [samples, P, K] tensor whose samples own their first T_j token rows (`_PAD_HOST[(samples, P)]`): sample j's rows a .. min(T_j, a + L)
    of the base -- relative to the view's first element, pitch K -- and the compact output's rows; else None."""
    base = x._base
    if base is None or x.dim() != 3 or not base.is_contiguous() or x.stride(2) != 1 or x.stride(1) != K or x.stride(0) % K:
        return None
    n, L, _ = x.shape
    P = x.stride(0) // K                                   # token rows per sample of what x is a view of: n * P rows of K in all
    T = _PAD_HOST.get((n, P))
    if T is None or L >= P or base.numel() != n * P * K:
        return None
    off = x.storage_offset() - base.storage_offset()
    if off < 0 or off % K or off // K + L > P:
        return None
    a = off // K
    ent = _SLICES.get((n, P, a, L))
    if ent is None:
        import numpy as np
        ln = np.clip(np.asarray(T, dtype=np.int64) - a, 0, L)
        tok = np.arange(L, dtype=np.int64)[None, :]
        real = tok < ln[:, None]
        xi = np.arange(n, dtype=np.int64)[:, None] * P + tok
        yi = np.arange(n, dtype=np.int64)[:, None] * L + tok
        ent = _SLICES[(n, P, a, L)] = (int32_on(xi[real].astype(np.int32), x.device),
                                       int32_on(np.concatenate([yi[real], yi[~real]]).astype(np.int32), x.device), int(real.sum()), n * L)
    return ent


def _pad_lengths(q, k, adds):
    """(q_len, k_len) for `ops.attn_fused` inside a padded group: the samples' own query / key counts where the operands' token
    counts are a padded length of the group (k_len only with an addend: the mask that hides the padding keys)."""
    ql = _PAD_LENGTHS.get(q.shape[2])
    kl = _PAD_LENGTHS.get(k.shape[2]) if adds else None
    if ql is not None and ql.shape[0] != q.shape[0]:
        ql = None
    if kl is not None and kl.shape[0] != k.shape[0]:
        kl = None
    return ql, kl


def _rows_for(x):
    return _ROW_MAPS.get((x.shape[0], x.shape[1])) if x.dim() == 3 else None


def _linear_rows_f32(x, weight, bias):
    """An fp32 linear inside a padded group, over the real rows only: the input carries its row map (the output of a slice's linear, or
    its GELU), has a padded leading shape (`_ROW_MAPS`), or is a token slice of a padded stack (`_slice_rows`); else None."""
    if torch.is_autocast_enabled() or not f32_enabled() or type(x) is not torch.Tensor or not ops.linear_f32_supported(x, weight, bias):
        return None
    tag = getattr(x, "_vlmc_rows", None)
    if tag is not None and x.dim() == 3 and x.is_contiguous() and tag[0].shape[0] == x.shape[0] * x.shape[1]:
        rm = tag
    else:
        rm = _rows_for(x) if (tag is None and x.is_contiguous()) else None
    if rm is not None:
        y = ops.linear_fwd_rows(x, [weight], [bias], rm[0], rm[1])[0]
        if tag is not None:
            y._vlmc_rows = tag
        stats["kernel"] += 1
        stats["kernel_rows"] += 1
        return y
    if _PAD_HOST is None or tag is not None:
        return None
    sl = _slice_rows(x, weight.shape[1])
    if sl is None:
        return None
    x_rows, y_rows, n_real, out_rows = sl
    y = ops.linear_fwd_gather(x, weight, bias, x_rows, y_rows, n_real, out_rows, weight.shape[1]).view(x.shape[0], x.shape[1], weight.shape[0])
    if n_real < out_rows:
        y._vlmc_rows = (y_rows, n_real)              # (a compact [samples, L, N] stack: its row map IS y_rows -- real rows first)
    stats["kernel"] += 1
    stats["kernel_rows"] += 1
    stats["kernel_slices"] += 1
    return y


def linear(x, weight, bias=None):
    """`F.linear` for a calibration forward: the invariant kernel when the replay engine asked for it and the call fits."""
    if _active and not torch.is_grad_enabled() and weight.is_cuda:
        if _ROW_MAPS is not None:
            if weight.dtype is torch.float32:
                y = _linear_rows_f32(x, weight, bias)
                if y is not None:
                    return y
            else:
                rm = _rows_for(x)
                if rm is not None:
                    p = _prepare(x, weight, bias)
                    if p is not None:
                        stats["kernel"] += 1
                        stats["kernel_rows"] += 1
                        return ops.linear_fwd_rows(p[0], [weight], [p[1]], rm[0], rm[1])[0]
        if not torch.is_autocast_enabled():
            y = ops.linear_fwd(x, weight, bias, _try=True) if (weight.dtype is not torch.float32 or f32_enabled()) else None   # (None: not one it takes)
            if y is not None:
                stats["kernel"] += 1
                return y
        else:
            p = _prepare(x, weight, bias)                            # what autocast would do to x and the bias
            if p is not None:
                stats["kernel"] += 1
                return ops.linear_fwd(p[0], weight, p[1], _checked=True)
    stats["library"] += 1
    return F.linear(x, weight, bias)


# first member of a group -> consecutive launches whose products were wasted (a sibling was not called on the tensor, or the
# group could not be formed): after three the group is forgotten -- a later call pattern in which the first member is
# called alone would otherwise keep computing its siblings' products for nothing; it is learned again if the
# calls show it again
_TROUBLE = weakref.WeakKeyDictionary()


def _trouble(first):
    n = _TROUBLE.get(first, 0) + 1
    if n >= 3:
        _SIBLINGS.pop(first, None)
        _TROUBLE.pop(first, None)
    else:
        _TROUBLE[first] = n


def _input_key(x):
    return (x.data_ptr(), x._version, tuple(x.shape), tuple(x.stride()), x.dtype)


def register_siblings(modules):
    """Declare `modules` (in call order) linears that are fed one tensor -- what `_Tracker` learns by itself from a
    forward; `walk_blocks` carries the groups it learned on a tower's first block over to the next blocks by name."""
    modules = tuple(modules)
    if len(modules) >= 2 and len(modules) <= ops.LINEAR_GROUP_MAX:
        _SIBLINGS[modules[0]] = tuple(weakref.ref(m) for m in modules)


def sibling_groups(modules):
    """The learned groups whose first member is one of `modules`: list of tuples of modules."""
    out = []
    for m in modules:
        refs = _SIBLINGS.get(m)
        if refs:
            group = tuple(r() for r in refs)
            if all(g is not None for g in group):
                out.append(group)
    return out


class _Tracker:
    """Per `invariant_linears` context: learns sibling groups from the calls, serves them."""

    def __init__(self, patched):
        self.patched = {id(m) for m in patched}
        self.run, self.run_x, self.run_key = [], None, None      # consecutive calls on one tensor (the tensor kept alive)
        self.stash = {}                                          # id(module) -> (x, key, y)
        self.grouping = grouping_enabled()

    def _close_run(self):
        if len(self.run) >= 2 and len({id(m) for m in self.run}) == len(self.run) and self.run[0] not in _SIBLINGS:
            register_siblings(self.run[:ops.LINEAR_GROUP_MAX])
        self.run, self.run_x, self.run_key = [], None, None

    def call(self, mod, x):
        if type(x) is LazyScores:
            x = x._realize()
        if not isinstance(x, torch.Tensor):
            return linear(x, mod.weight, mod.bias)
        if not self.grouping:
            return self.single(mod, x)
        key = _input_key(x)
        # ---- served from a sibling's launch? --------------------------------------------------------------------------
        kept = self.stash.pop(id(mod), None)
        if kept is not None:
            if kept[1] == key and kept[0].untyped_storage().data_ptr() == x.untyped_storage().data_ptr():
                stats["served_from_group"] += 1
                _TROUBLE.pop(kept[3], None)
                return kept[2]
            stats["stash_dropped"] += 1
            _trouble(kept[3])
        # ---- learn: consecutive calls on the very same tensor -------------------------------------------------------------
        if self.run and key == self.run_key:
            self.run.append(mod)
        else:
            self._close_run()
            self.run, self.run_x, self.run_key = [mod], x, key
        # ---- first member of a known group: one launch for all ------------------------------------------------------------
        refs = _SIBLINGS.get(mod)
        if refs:
            group = [r() for r in refs]
            if all(g is not None and id(g) in self.patched for g in group):
                # the activation is cast (autocast) and checked ONCE, against the first member; the others must agree with it
                # in dtype, width and bias handling -- `_prepare` on each would cast x again per member
                p0 = _prepare(x, mod.weight, mod.bias)
                if p0 is not None:
                    xin, w0 = p0[0], mod.weight
                    biases = [p0[1]]
                    ok = True
                    for g in group[1:]:
                        w, b = g.weight, g.bias
                        if w.dtype != w0.dtype or w.shape[1] != w0.shape[1] or not w.is_cuda:
                            ok = False
                            break
                        if b is not None and b.dtype != w.dtype:
                            if not (torch.is_autocast_enabled() and torch.get_autocast_gpu_dtype() == w.dtype):
                                ok = False
                                break
                            b = b.to(w.dtype)
                        if not ops.linear_fwd_supported(xin, w, b):
                            ok = False
                            break
                        biases.append(b)
                    if ok:
                        rm = _rows_for(xin) if _ROW_MAPS is not None else None
                        if rm is not None:                       # a padded group of ragged samples: the padding rows are skipped
                            outs = ops.linear_fwd_rows(xin, [g.weight for g in group], biases, rm[0], rm[1])
                            stats["kernel_rows"] += len(group)
                        else:
                            outs = ops.linear_fwd_group(xin, [g.weight for g in group], biases, _checked=True)
                        stats["kernel"] += len(group)
                        stats["grouped_launches"] += 1
                        for g, y in zip(group[1:], outs[1:]):
                            self.stash[id(g)] = (x, key, y, mod)
                        return outs[0]
            _trouble(mod)                                        # (the group could not be formed for this call)
        return self.single(mod, x)

    def single(self, mod, x):
        """A linear on its own."""
        return linear(x, mod.weight, mod.bias)

    def close(self):
        self._close_run()
        if self.stash:
            stats["stash_dropped"] += len(self.stash)
            for first in {id(k[3]): k[3] for k in self.stash.values()}.values():
                _trouble(first)
        self.stash.clear()


# ---- the batched matmuls of attention (q @ k^T, attn @ v) on the batch-invariant kernel -----------------------------------
# The reference's model files write attention as batched `torch.matmul`s / `@` (eva_vit.py:147,164; modeling_t5.py:590,638;
# modeling_llama.py likewise).  The GEMM library picks its batched kernel by batch count, so a grouped block forward gives a
# sample other last bits than its own forward would -- and near-tie mask bits follow.  While a block is replayed, `torch.matmul`,
# `torch.bmm`, `Tensor.__matmul__`, `Tensor.matmul` and `Tensor.bmm` are plain attribute patches (no `TorchFunctionMode`: that
# would tax every op of a host-bound loop) that send 3-D / 4-D 16-bit products without gradients to `vlmc_attn_matmul`
# (csrc/attn_matmul.hip) and everything else to the original.  `VLMC_ATTN_MATMUL=0`: off.
# The attributes are process-wide, the ROUTING is not: every patched function first asks whether the calling thread is the
# one that installed the patches (`_mm_owner`) and hands any other thread -- a data-loader worker, a DDP hook -- straight to
# the original.  A second thread that enters `invariant_matmuls` while another one holds the patches
# runs unpatched (its products go to the library: the replay of ONE prune is what the kernels are promised for).
_mm_depth = 0
_mm_saved = {}
_mm_owner = None
_ident = threading.get_ident


def attn_matmul_enabled():
    return os.environ.get("VLMC_ATTN_MATMUL", "1") != "0"


# ---- the reference-op attention chain in ONE launch ---------------------------------------------------------------------------
# The reference's model files write attention as separate tensor ops (eva_vit.py:145-164, modeling_t5.py:588-640,
# Qformer.py:205-246):  scores = q @ k^T;  [scores / sqrt(d)];  [scores + bias (+ mask)];  softmax (in fp32);  probs @ v -- with
# the patches above that is 5-7 launches and ~8 passes over the [B, H, Tq, Tk] scores per attention.  `vlmc_attn_fwd`
# (csrc/attn_fused_kernel.hpp) computes the same chain, rounding where each op rounds, in one launch.  To splice it under model
# code that nobody here may rewrite, the first product is answered LAZILY: the patched `matmul` hands out a `LazyScores` -- a
# tensor subclass without storage that only remembers (q, k) -- and every op the model then applies to it arrives in
# `__torch_function__`:
#   * the ops of the chain (`+=` / `+` a 16-bit tensor, `/` or `*` a Python number, `.float()`, softmax over the last dim,
#     `.type_as` / `.to(dtype)`, dropout in eval mode) are RECORDED, twice: literally (function + arguments, for a replay) and as
#     the fused kernel's arguments;
#   * the second product `matmul(lazy, v)` runs the fused kernel if the recorded chain is one it computes;
#   * ANY other op -- or a chain the kernel does not take -- first REALIZES the tensor: the literal record is replayed on the
#     unfused kernels (`vlmc_attn_matmul`, torch's elementwise ops, `vlmc_softmax_rows`), and the op runs on the result.
# So the model's code sees exactly the values it would have seen; the only thing that changes is when they are computed.  The
# first time a chain SIGNATURE (dtype, head_dim, multiplier?, addend broadcast pattern, fp32 detour?) reaches the fused kernel,
# its output is compared bit for bit with the unfused sequence on the same operands (one extra pass, once per process and
# signature); a mismatch switches the signature off for good.  `VLMC_ATTN_FUSED=0`: never lazy.
_SELF = object()
_FUSED_OK = {}                # chain signature -> True / False
_LAZY_LIVE = weakref.WeakSet()    # lazy tensors somebody still holds: realized when the patches go (their record replays patched functions)


def attn_fused_enabled():
    return os.environ.get("VLMC_ATTN_FUSED", "1") != "0"


_LAZY_META_PROPS = {"shape", "dtype", "device", "ndim", "is_cuda", "requires_grad", "layout", "is_sparse", "is_quantized", "is_meta",
                    "is_leaf", "grad_fn", "is_cpu", "is_nested", "names"}
_LAZY_META_FUNCS = set()


class LazyScores(torch.Tensor):
    """`q @ k^T` of an attention, not computed yet (see above).  `_state`: s16 scores in the dtype, s32 after `.float()`,
    p32 / p16 probabilities in fp32 / the dtype."""

    @staticmethod
    def __new__(cls, q, kt, shape, dtype, state="s16", ops_=(), mul=None, adds=(), base=None):
        r = torch.Tensor._make_wrapper_subclass(cls, shape, dtype=dtype, device=q.device)
        r._q, r._kt, r._state, r._ops, r._mul, r._adds, r._real = q, kt, state, list(ops_), mul, list(adds), None
        r._dt16 = q.dtype if base is None else base
        r._consumed = False
        _LAZY_LIVE.add(r)
        return r

    def _child(self, state, dtype=None):
        return LazyScores(self._q, self._kt, self.shape, self.dtype if dtype is None else dtype, state, self._ops, self._mul, self._adds,
                          self._dt16)

    def _realize(self):
        """The tensor the model's ops would have produced so far, on the unfused kernels."""
        if self._real is None:
            t = ops.attn_matmul(self._q, self._kt)
            for func, args, kwargs in self._ops:
                t = func(*[t if a_ is _SELF else a_ for a_ in args], **{k_: (t if v_ is _SELF else v_) for k_, v_ in kwargs.items()})
            self._real = t
            stats["attn_kernel"] += 1
            stats["attn_chain_unfused"] += 1
        return self._real

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        h = _LAZY_HANDLERS.get(func)
        if h is not None:
            r = h(func, args, kwargs)
            if r is not NotImplemented:
                return r
        elif func in _LAZY_META_FUNCS or (getattr(func, "__name__", None) == "__get__" and
                                         getattr(getattr(func, "__self__", None), "__name__", None) in _LAZY_META_PROPS):
            with torch._C.DisableTorchFunctionSubclass():
                return func(*args, **kwargs)
        return func(*_real_args(args), **{k_: _real_arg(v_) for k_, v_ in kwargs.items()})


def _lazy_dispatch(cls, func, types, args=(), kwargs=None):
    """Below `__torch_function__` (an op that reached the dispatcher with a lazy operand all the same): its value."""
    return func(*_real_args(args), **{k_: _real_arg(v_) for k_, v_ in (kwargs or {}).items()})


LazyScores.__torch_dispatch__ = classmethod(_lazy_dispatch)


def _realize_escaped():
    """Lazy products somebody still holds when the route goes down (they escaped the code that was to consume them): their
    values, now, on the kernels their records name.  (One whose chain reached a fused kernel is done.)"""
    for z in list(_LAZY_LIVE):
        if z._real is None and not z._consumed:
            z._realize()


def _real_arg(v):
    if type(v) is LazyScores:
        return v._realize()
    if type(v) in (tuple, list):
        return type(v)(_real_arg(e) for e in v)
    return v


def _real_args(args):
    return [_real_arg(v) for v in args]


def _is_number(v):
    return type(v) in (int, float)


def _fp32(v):
    import numpy as np
    return float(np.float32(v))


def _lz_add(func, args, kw, inplace=False):
    if kw or len(args) != 2:
        return NotImplemented
    a, b = args
    if type(b) is LazyScores:
        if inplace:
            return NotImplemented
        a, b = b, a
        pos = (b, _SELF)
    else:
        pos = (_SELF, b)
    if type(a) is not LazyScores or a._real is not None or a._state != "s16" or len(a._adds) >= 2 or type(b) is not torch.Tensor or \
            b.dtype != a._dt16 or b.device != a._q.device or b.requires_grad or b.dim() > 4 or b.dim() == 0:
        return NotImplemented
    sh = (1,) * (4 - b.dim()) + tuple(b.shape)
    if any(n not in (1, m) for n, m in zip(sh[:3], a.shape[:3])) or sh[3] != a.shape[3] or min(b.stride()) < 0 or (b.stride(-1) < 1 and sh[3] != 1):
        return NotImplemented
    r = a if inplace else a._child("s16")
    r._ops.append((func, pos, {}))
    r._adds.append(b)
    return r


def _lz_iadd(func, args, kw):
    return _lz_add(func, args, kw, True)


def _lz_scale(func, args, kw, divide):
    if kw or len(args) != 2:
        return NotImplemented
    a, c = args
    if type(a) is not LazyScores or a._real is not None or a._state != "s16" or a._adds or a._mul is not None or not _is_number(c) or \
            (divide and c == 0):
        return NotImplemented
    import numpy as np
    # torch divides a 16-bit CUDA tensor by a CPU scalar as a multiplication with the fp32 reciprocal (BinaryDivTrueKernel.cu)
    m = float(np.float32(1.0) / np.float32(c)) if divide else _fp32(c)
    if not (m == m and abs(m) != float("inf")):
        return NotImplemented
    r = a._child("s16")
    r._ops.append((func, (_SELF, c), {}))
    r._mul = m
    return r


def _lz_div(func, args, kw):
    return _lz_scale(func, args, kw, True)


def _lz_mul(func, args, kw):
    if len(args) == 2 and type(args[0]) is not LazyScores:
        return NotImplemented                                          # (number * lazy arrives as __rmul__(lazy, number))
    return _lz_scale(func, args, kw, False)


def _lz_cast(a, dtype, func, args, kw):
    if type(a) is not LazyScores or a._real is not None or not isinstance(dtype, torch.dtype):
        return NotImplemented
    if dtype == a.dtype:
        return a
    if dtype == torch.float32 and a._state == "s16":
        r = a._child("s32", torch.float32)
    elif dtype == a._dt16 and a._state == "p32":
        r = a._child("p16", a._dt16)
    else:
        return NotImplemented
    r._ops.append((func, tuple(_SELF if v is a else v for v in args), dict(kw)))
    return r


def _lz_float(func, args, kw):
    return _lz_cast(args[0], torch.float32, func, args, kw) if len(args) == 1 and not kw else NotImplemented


def _lz_half(func, args, kw):
    return _lz_cast(args[0], torch.float16, func, args, kw) if len(args) == 1 and not kw else NotImplemented


def _lz_bfloat16(func, args, kw):
    return _lz_cast(args[0], torch.bfloat16, func, args, kw) if len(args) == 1 and not kw else NotImplemented


def _lz_to(func, args, kw):
    if len(args) == 2 and not (set(kw) - {"non_blocking", "copy"}) and not kw.get("copy", False):
        return _lz_cast(args[0], args[1], func, args, kw)
    if len(args) == 1 and set(kw) <= {"dtype", "non_blocking", "copy"} and "dtype" in kw and not kw.get("copy", False):
        return _lz_cast(args[0], kw["dtype"], func, args, kw)
    return NotImplemented


def _lz_type_as(func, args, kw):
    if len(args) != 2 or kw or not isinstance(args[1], torch.Tensor) or type(args[0]) is not LazyScores:
        return NotImplemented
    a, other = args
    if other.device != a.device:
        return NotImplemented
    # (recorded as `.to(dtype)`: the other tensor may be a lazy one itself)
    return _lz_cast(a, other.dtype, torch.Tensor.to, (a, other.dtype), {})


def _lz_softmax(func, args, kw):
    """softmax(x, dim[, _stacklevel][, dtype]) in any of its spellings"""
    if not args or type(args[0]) is not LazyScores:
        return NotImplemented
    a = args[0]
    dim = args[1] if len(args) > 1 else kw.get("dim")
    dtype = kw.get("dtype")
    if a._real is not None or len(args) > 2 or (set(kw) - {"dim", "_stacklevel", "dtype"}) or type(dim) is not int or dim not in (-1, a.dim() - 1):
        return NotImplemented
    if dtype is None and a._state == "s16" and torch.is_autocast_enabled():
        dtype = torch.float32                                           # (autocast runs softmax in fp32 and returns fp32)
    if a._state == "s16" and dtype is None:
        r = a._child("p16")
    elif a._state in ("s16", "s32") and dtype in (None, torch.float32):
        r = a._child("p32", torch.float32)
    else:
        return NotImplemented
    r._ops.append((func, tuple(_SELF if v is a else v for v in args), dict(kw)))
    return r


def _lz_dropout(func, args, kw):
    """F.dropout(x, p, training, inplace) / torch.dropout(x, p, train): the identity in eval mode"""
    if not args or type(args[0]) is not LazyScores:
        return NotImplemented
    names = ("p", "training", "inplace") if func is F.dropout else ("p", "train")
    vals = dict(zip(names, args[1:]))
    vals.update(kw)
    training = vals.get("training", vals.get("train", True))
    if set(vals) - set(names) or (training and vals.get("p", 0.5) != 0):
        return NotImplemented
    return args[0]


def _lazy_handlers():
    T = torch.Tensor
    h = {T.__iadd__: _lz_iadd, T.add_: _lz_iadd, T.__add__: _lz_add, T.__radd__: _lz_add, T.add: _lz_add, torch.add: _lz_add,
         T.__truediv__: _lz_div, T.div: _lz_div, T.true_divide: _lz_div, torch.div: _lz_div, torch.true_divide: _lz_div,
         T.__mul__: _lz_mul, T.__rmul__: _lz_mul, T.mul: _lz_mul, torch.mul: _lz_mul,
         T.float: _lz_float, T.half: _lz_half, T.bfloat16: _lz_bfloat16, T.to: _lz_to, T.type_as: _lz_type_as,
         F.dropout: _lz_dropout, torch.dropout: _lz_dropout}
    for f in (T.size, T.dim, T.numel, T.nelement, T.element_size, T.is_floating_point, T.is_complex, T.get_device, T.ndimension,
              T.is_contiguous, T.stride):
        _LAZY_META_FUNCS.add(f)
    return h


_LAZY_HANDLERS = _lazy_handlers()

def _chain_signature(a, v):
    q = a._q

    def pat(t):
        sh = (1,) * (4 - t.dim()) + tuple(t.shape)
        return tuple(n != 1 for n in sh[:3])
    return (q.dtype, q.shape[-1], a._mul is not None, tuple(pat(t) for t in a._adds),
            tuple(f is _SELF or getattr(f, "__name__", "") for f, _a, _k in a._ops))


def _lazy_matmul(a, b, matmul):
    """`matmul(lazy, v)`: the fused kernel when the recorded chain is one it computes (and has reproduced the unfused bits once),
    else the realized tensor times v."""
    fusable = a._real is None and type(b) is torch.Tensor and attn_fused_enabled() and not torch.is_grad_enabled() and \
        (a._state == "p16" or (a._state == "p32" and torch.is_autocast_enabled() and torch.get_autocast_gpu_dtype() == a._dt16))
    if fusable:
        k = a._kt.transpose(-1, -2)
        plan = ops.attn_fused_plan(a._q, k, b, a._adds)
        if plan is not None:
            sig = _chain_signature(a, b)
            ok = _FUSED_OK.get(sig)
            ql, kl = _pad_lengths(a._q, k, a._adds) if _PAD_LENGTHS is not None else (None, None)
            if ok is None and not torch.cuda.is_current_stream_capturing():
                fused = ops.attn_fused(a._q, k, b, a._mul, a._adds, plan)
                ref = matmul(a._realize(), b)
                ok = _FUSED_OK[sig] = bool(fused.shape == ref.shape and torch.equal(fused.contiguous().view(torch.int16),
                                                                                  ref.contiguous().view(torch.int16)))
                stats["attn_fused_checks"] += 1
                if not ok:
                    import warnings
                    warnings.warn(f"vlmc.forward: the fused attention does not reproduce the unfused op sequence for {sig}; "
                                  "this chain stays unfused", RuntimeWarning)
                    return ref
                stats["attn_fused"] += 1
                a._consumed = True
                return fused
            if ok:
                stats["attn_fused"] += 1
                a._consumed = True
                if ql is not None or kl is not None:
                    stats["attn_fused_lens"] += 1
                    return ops.attn_fused(a._q, k, b, a._mul, a._adds, plan, ql, kl)
                return ops.attn_fused(a._q, k, b, a._mul, a._adds, plan)
    return matmul(a._realize(), b)


def _make_matmul(orig):
    Tensor, f32 = torch.Tensor, torch.float32
    lazy = attn_fused_enabled() and softmax_enabled()
    f32_on = f32_enabled()

    def matmul(a, b, *args, **kw):
        if _ident() != _mm_owner:
            return orig(a, b, *args, **kw)
        if type(a) is LazyScores:
            if args or kw or type(b) is LazyScores:
                return orig(a._realize(), _real_arg(b), *args, **kw)
            return _lazy_matmul(a, b, matmul)
        if type(b) is LazyScores:
            return orig(a, b._realize(), *args, **kw)
        if not args and not kw and type(a) is Tensor and type(b) is Tensor and a.dim() >= 3 and not torch.is_grad_enabled():
            if (torch.is_autocast_enabled() and torch.get_autocast_gpu_dtype() != a.dtype) or (a.dtype is f32 and not f32_on):
                return orig(a, b)                                   # (autocast would cast the operands: the library's call)
            if lazy and a.dim() == 4 and b.dim() == 4 and b.stride(2) == 1 and a.stride(3) == 1 and b.shape[3] > 1 and \
                    a.dtype is b.dtype and a.shape[:2] == b.shape[:2] and ops.attn_fused_plan(a, b.transpose(2, 3), b.transpose(2, 3)) is not None:
                # q @ k^T of an attention (k^T: a transposed view, keys K-contiguous): answered lazily
                return LazyScores(a, b, (*a.shape[:3], b.shape[3]), a.dtype)
            out = ops.attn_matmul(a, b, None, True)                   # (looked up per call: bench.py's probe wraps it); None: not a
            if out is not None:                                       # product the kernel computes
                stats["attn_kernel"] += 1
                return out
            stats["attn_library"] += 1
        return orig(a, b, *args, **kw)
    return matmul


def _make_sdpa(orig):
    """`F.scaled_dot_product_attention(q, k, v[, is_causal=True])` without mask or dropout on `vlmc_sdpa_fwd` (fused,
    batch-invariant, no [T, T] scores in HBM); every other call goes to the original."""
    Tensor = torch.Tensor

    def sdpa(q, k, v, attn_mask=None, dropout_p=0.0, is_causal=False, scale=None, **kw):
        if _ident() != _mm_owner:
            return orig(q, k, v, attn_mask=attn_mask, dropout_p=dropout_p, is_causal=is_causal, scale=scale, **kw)
        if attn_mask is None and dropout_p == 0.0 and type(is_causal) is bool and not kw and type(q) is Tensor and \
                not torch.is_grad_enabled() and not (torch.is_autocast_enabled() and torch.get_autocast_gpu_dtype() != q.dtype):
            out = ops.sdpa(q, k, v, scale, True, is_causal)
            if out is not None:
                stats["sdpa_kernel"] += 1
                return out
            stats["sdpa_library"] += 1
        return orig(q, k, v, attn_mask=attn_mask, dropout_p=dropout_p, is_causal=is_causal, scale=scale, **kw)
    return sdpa


def _make_mean(orig):
    """`x.mean(-1[, keepdim])` / `torch.mean(x, -1[, keepdim])` of an fp32 CUDA tensor on `vlmc_row_mean` (the fp32 mean of
    squares inside T5LayerNorm / LlamaRMSNorm: torch's reduction kernel is configured by the number of outputs, so a 4-token
    sample gets other last bits alone than in a group); every other call goes to the original."""
    run = ops.row_mean
    Tensor, f32 = torch.Tensor, torch.float32

    def mean(x, *args, **kw):
        if _ident() != _mm_owner:
            return orig(x, *args, **kw)
        if type(x) is Tensor and x.dtype is f32 and x.is_cuda and not torch.is_grad_enabled() and x.dim() >= 2:
            dim = args[0] if args else kw.get("dim")
            if type(dim) in (tuple, list) and len(dim) == 1:
                dim = dim[0]
            if type(dim) is int and (dim == -1 or dim == x.dim() - 1) and len(args) <= 2 and not (set(kw) - {"dim", "keepdim"}) \
                    and x.shape[-1] > 0:
                keepdim = args[1] if len(args) > 1 else kw.get("keepdim", False)
                stats["mean_kernel"] += 1
                return run(x, bool(keepdim))
        return orig(x, *args, **kw)
    return mean


def _make_softmax(orig, functional):
    """`F.softmax(x, dim=-1)` / `torch.softmax(x, -1)` / `x.softmax(-1)` of an fp32 or 16-bit CUDA tensor on `vlmc_softmax_rows`:
    masked-out entries behind a row's real ones leave its bits alone, so a calibration sample can be padded into a group of longer
    ones (calibration.py: padded groups), and the fused attention forms its softmax in the same order; 16-bit input: fp32
    arithmetic, rounded once (under autocast: fp32 out, as autocast's softmax returns it); every other call -- another dim,
    gradients -- goes to the original.  A lazy attention chain records the call."""
    Tensor, f32 = torch.Tensor, torch.float32

    def softmax(x, *args, **kw):
        if _ident() == _mm_owner:
            if type(x) is LazyScores:
                r = _lz_softmax(softmax, (x,) + args, kw)
                return r if r is not NotImplemented else orig(x._realize(), *args, **kw)
            if type(x) is Tensor and x.is_cuda and not torch.is_grad_enabled() and x.dim() >= 1 and (x.dtype is f32 or x.dtype in ops._16BIT):
                dim = args[0] if args else kw.get("dim")
                extra = set(kw) - {"dim", "_stacklevel", "dtype"}
                dtype = kw.get("dtype")
                if type(dim) is int and (dim == -1 or dim == x.dim() - 1) and len(args) <= 1 and not extra and x.shape[-1] > 0 and \
                        (dtype is None or dtype is f32):
                    if dtype is None and x.dtype is not f32 and torch.is_autocast_enabled():
                        dtype = f32
                    stats["softmax_kernel"] += 1
                    return ops.softmax_rows(x, dtype)
        return orig(x, *args, **kw)
    return softmax


def softmax_enabled():
    return os.environ.get("VLMC_SOFTMAX", "1") != "0"


def _make_gelu(orig):
    """`F.gelu(x[, approximate=...])` of a 16-bit CUDA tensor on `vlmc_gelu`: torch's elementwise kernel computes a tensor's last
    partial block with other instructions than its body (an fma contraction: ~20 % of all 16-bit inputs differ in the last bit), so
    which bits a sample's last rows get depends on how many samples share the forward -- the one elementwise op of the blocks that is
    not batch-invariant in torch.  The kernel uses the body's arithmetic for every element (`VLMC_GELU=0`: torch's)."""
    Tensor = torch.Tensor

    def gelu(x, *args, **kw):
        if _ident() == _mm_owner:
            if type(x) is Tensor and x.is_cuda and (x.dtype in ops._16BIT or (x.dtype is torch.float32 and not torch.is_autocast_enabled() and f32_enabled())) and \
                    not torch.is_grad_enabled() and not args and \
                    not (set(kw) - {"approximate"}) and kw.get("approximate", "none") in ("none", "tanh") and x.numel() > 0:
                stats["gelu_kernel"] += 1
                y = ops.gelu(x, kw.get("approximate", "none"))
                tag = getattr(x, "_vlmc_rows", None)
                if tag is not None and y.shape == x.shape:           # (a compact slice's output: GELU keeps rows, gelu(0) = 0)
                    y._vlmc_rows = tag
                return y
        return orig(x, *args, **kw)
    return gelu


def function_mode_enabled():
    return os.environ.get("VLMC_TORCH_FUNCTION_MODE", "0") == "1"


class _ReplayMode(torch.overrides.TorchFunctionMode):
    """The replay's routing as a torch-function mode: a call of one of the table's functions goes to its handler (the closures the
    attribute patches install), every other call straight through."""

    def __init__(self, table):
        super().__init__()
        self.table = table

    def __torch_function__(self, func, types, args=(), kwargs=None):
        h = self.table.get(func)
        if h is None:
            return func(*args, **kwargs) if kwargs else func(*args)
        return h(*args, **kwargs) if kwargs else h(*args)


def _function_table():
    """function object -> handler, for the switches in force (built per context: the closures read the switches when made)"""
    import torch.nn.functional as F_
    base = torch._C.TensorBase
    t = {}
    for f in (torch.matmul, torch.bmm, base.__matmul__, base.matmul, base.bmm):
        t[f] = _make_matmul(f)
    if os.environ.get("VLMC_SDPA", "1") != "0":
        t[F_.scaled_dot_product_attention] = _make_sdpa(F_.scaled_dot_product_attention)
    if softmax_enabled():
        t[F_.softmax] = _make_softmax(F_.softmax, True)
        t[torch.softmax] = _make_softmax(torch.softmax, False)
        t[base.softmax] = _make_softmax(base.softmax, False)
    if os.environ.get("VLMC_GELU", "1") != "0":
        t[F_.gelu] = _make_gelu(F_.gelu)
    if os.environ.get("VLMC_ROW_MEAN", "1") != "0":
        t[torch.mean] = _make_mean(torch.mean)
        t[base.mean] = _make_mean(base.mean)
    return t


@contextlib.contextmanager
def invariant_matmuls():
    """Batched 16-bit `matmul`s run on `vlmc_attn_matmul`, fp32 means over the last dimension on `vlmc_row_mean`, for the
    duration (nestable)."""
    global _mm_depth, _mm_owner
    if not (enabled() and attn_matmul_enabled()) or (_mm_depth > 0 and _mm_owner != _ident()):
        yield
        return
    if function_mode_enabled():
        # the same routing through a scoped `torch.overrides.TorchFunctionMode` instead of attribute patches: nothing
        # global is assigned, nothing to restore.  Measured slower -- every torch call of the forward then goes through Python once more
        # (profiles/r06_phase_timeline.md 6) -- so it is the cross-check route, not the default.
        _mm_depth += 1
        try:
            if _mm_depth > 1:                                    # nested: the outer context's mode is in force
                yield
            else:
                _mm_owner = _ident()
                with _ReplayMode(_function_table()):
                    yield
        finally:
            _mm_depth -= 1
            if _mm_depth == 0:
                _realize_escaped()
                _mm_owner = None
        return
    if _mm_depth == 0:
        _mm_owner = _ident()
        base = torch._C.TensorBase
        _mm_saved.update(matmul=torch.matmul, bmm=torch.bmm)
        torch.matmul = _make_matmul(torch.matmul)
        torch.bmm = _make_matmul(torch.bmm)
        for name in ("__matmul__", "matmul", "bmm"):              # (inherited from TensorBase: the patch shadows, `del` restores)
            if name not in torch.Tensor.__dict__:
                setattr(torch.Tensor, name, _make_matmul(getattr(base, name)))
                _mm_saved.setdefault("tensor", []).append(name)
        if os.environ.get("VLMC_SDPA", "1") != "0":               # fused attention written as F.scaled_dot_product_attention
            import torch.nn.functional as F_
            _mm_saved["sdpa"] = F_.scaled_dot_product_attention
            F_.scaled_dot_product_attention = _make_sdpa(F_.scaled_dot_product_attention)
        if softmax_enabled():                                     # the attention's fp32 softmax: padding-invariant
            import torch.nn.functional as F_
            _mm_saved["softmax"] = (F_.softmax, torch.softmax)
            F_.softmax = _make_softmax(F_.softmax, True)
            torch.softmax = _make_softmax(torch.softmax, False)
            if "softmax" not in torch.Tensor.__dict__:
                setattr(torch.Tensor, "softmax", _make_softmax(base.softmax, False))
                _mm_saved.setdefault("tensor", []).append("softmax")
        if os.environ.get("VLMC_GELU", "1") != "0":               # torch's GELU: another instruction sequence in a tensor's last block
            import torch.nn.functional as F_
            _mm_saved["gelu"] = F_.gelu
            F_.gelu = _make_gelu(F_.gelu)
        if os.environ.get("VLMC_ROW_MEAN", "1") != "0":           # the fp32 mean inside the norms (batch-variant in torch)
            _mm_saved["mean"] = torch.mean
            torch.mean = _make_mean(torch.mean)
            if "mean" not in torch.Tensor.__dict__:
                setattr(torch.Tensor, "mean", _make_mean(base.mean))
                _mm_saved.setdefault("tensor", []).append("mean")
    _mm_depth += 1
    try:
        yield
    finally:
        _mm_depth -= 1
        if _mm_depth == 0:
            _realize_escaped()
            _mm_owner = None
            torch.matmul, torch.bmm = _mm_saved.pop("matmul"), _mm_saved.pop("bmm")
            if "mean" in _mm_saved:
                torch.mean = _mm_saved.pop("mean")
            if "gelu" in _mm_saved:
                import torch.nn.functional as F_
                F_.gelu = _mm_saved.pop("gelu")
            if "softmax" in _mm_saved:
                import torch.nn.functional as F_
                F_.softmax, torch.softmax = _mm_saved.pop("softmax")
            if "sdpa" in _mm_saved:
                import torch.nn.functional as F_
                F_.scaled_dot_product_attention = _mm_saved.pop("sdpa")
            for name in _mm_saved.pop("tensor", []):
                delattr(torch.Tensor, name)


# ---- RMS norms of the language-model blocks: one launch instead of seven ------------------------------------------------------
# A module qualifies by what it IS -- no children, exactly one parameter `weight` [n] in fp16 / bf16, no buffers, an epsilon
# attribute -- and is only ever replaced after its OWN forward has been reproduced bit for bit by `vlmc_rms_norm` on random rows
# (once per class, dtype, width and epsilon, under the replay's patches; that comparison also picks the rsqrt flavour that is
# torch's).  A norm that scales by (1 + weight), keeps fp32, adds a bias or differs in any rounding fails the comparison and keeps
# its own forward.  `VLMC_RMS_NORM=0`: never.
_NORM_OK = {}                 # (class, dtype, n, eps) -> rsqrt_mode, or None: the module's forward is not this op sequence
_NORM_SCAN = weakref.WeakKeyDictionary()      # root module -> its norm candidates (the walk is done once per root)


def rms_norm_enabled():
    return os.environ.get("VLMC_RMS_NORM", "1") != "0"


def _norm_eps(m):
    for name in ("variance_epsilon", "eps", "epsilon"):
        v = m.__dict__.get(name)
        if isinstance(v, float) and 0.0 < v < 1.0:
            return v
    return None


def _norm_candidates(root):
    found = _NORM_SCAN.get(root)
    if found is None:
        found = []
        for m in root.modules():
            if m._modules or len(m._parameters) != 1 or any(b is not None for b in m._buffers.values()) or type(m) is nn.Linear:
                continue
            w = m._parameters.get("weight")
            if w is None or w.dim() != 1 or w.dtype not in (torch.float16, torch.bfloat16) or not w.is_cuda or _norm_eps(m) is None:
                continue
            found.append(m)
        _NORM_SCAN[root] = found
    return found


_RSQRT_MODES = {}             # device index -> the rsqrt flavours worth trying, torch's first


def _rsqrt_modes(device):
    """torch.rsqrt(float) on ROCm is 1 / sqrt in DOUBLE rounded to float (ATen's `::rsqrt(a)` picks the double overload): if that
    holds on this build -- checked on 2^18 values against torch.rsqrt itself -- it is the only flavour tried; otherwise the two
    fp32 ones, and the module-level comparison below decides on more rows."""
    modes = _RSQRT_MODES.get(device.index)
    if modes is None:
        g = torch.Generator(device=device).manual_seed(7)
        v = torch.cat([torch.rand(1 << 17, generator=g, device=device) * 4 + 1e-6, torch.randn(1 << 17, generator=g, device=device).abs() * 1e3 + 1e-8])
        modes = (0,) if torch.equal(torch.rsqrt(v), (1.0 / torch.sqrt(v.double())).float()) else (1, 2)
        _RSQRT_MODES[device.index] = modes
    return modes


def _norm_mode(m):
    """rsqrt flavour with which `vlmc_rms_norm` IS `m.forward` (bitwise, on random rows), or None.  Called inside the replay's
    patches (the module's own forward takes its mean from `vlmc_row_mean`, as it will in the replay), outside any graph capture."""
    w, eps = m.weight, _norm_eps(m)
    key = (type(m), w.dtype, w.shape[0], eps)
    if key in _NORM_OK:
        return _NORM_OK[key]
    mode = None
    try:
        g = torch.Generator(device=w.device).manual_seed(20240917)
        modes = _rsqrt_modes(w.device)
        rows = 37 if modes == (0,) else 1500                                  # (flavours that differ in 1 ulp of r show in ~1 element per 10 rows)
        x = (torch.randn(3, rows, w.shape[0], generator=g, device=w.device) * torch.tensor([0.02, 1.0, 30.0], device=w.device)[:, None, None]).to(w.dtype)
        # The verdict is cached per (class, dtype, width, eps) and then holds for EVERY module of the class: it must not be
        # taken with a weight that hides where the class rounds its product -- an all-ones weight (a fresh or synthetic norm)
        # makes `(w.float() * h32).to(dtype)` and `w * h32.to(dtype)` agree bit for bit.  The module is shown a
        # random, non-trivial weight for the comparison and gets its own back.
        own = w.data
        trial = (torch.randn(w.shape[0], generator=g, device=w.device) * 0.3 + 1.0).to(w.dtype)
        with torch.no_grad():
            try:
                w.data = trial
                want = m.forward(x)
            finally:
                w.data = own
            if isinstance(want, torch.Tensor) and want.dtype == w.dtype and want.shape == x.shape:
                for cand in modes:
                    if torch.equal(ops.rms_norm(x, trial, eps, cand), want):
                        mode = cand
                        break
    except Exception:
        mode = None
    _NORM_OK[key] = mode
    return mode


def _make_norm(m, mode, eps):
    own = m.forward

    def forward(x, *a, **kw):
        if not a and not kw and type(x) is torch.Tensor and x.is_cuda and x.dtype == m.weight.dtype and x.shape[-1] == m.weight.shape[0] \
                and not torch.is_grad_enabled() and not (torch.is_autocast_enabled() and torch.get_autocast_gpu_dtype() != x.dtype):
            stats["norm_kernel"] += 1
            return ops.rms_norm(x, m.weight, eps, mode)
        return own(x, *a, **kw)
    return forward


@contextlib.contextmanager
def invariant_linears(modules, roots=()):
    """Route the forward of the given `nn.Linear` modules (exact type) through `linear` for the duration -- and the batched
    matmuls of the blocks' attention through `vlmc_attn_matmul` (`invariant_matmuls`).  `roots`: the blocks those linears live
    in; their RMS norms run on `vlmc_rms_norm` where that reproduces the norm's own forward exactly (above)."""
    global _active
    if not enabled():
        yield
        return
    patched = [m for m in modules if type(m) is nn.Linear and "forward" not in m.__dict__]
    tracker = _Tracker(patched)
    for m in patched:
        m.forward = (lambda mod: (lambda x: tracker.call(mod, x)))(m)
    _active += 1
    norms = []
    try:
        with invariant_matmuls():
            if roots and rms_norm_enabled() and attn_matmul_enabled() and os.environ.get("VLMC_ROW_MEAN", "1") != "0" \
                    and torch.cuda.is_available() and not torch.cuda.is_current_stream_capturing():
                for root in roots:
                    for m in _norm_candidates(root):
                        if "forward" in m.__dict__ or m.training:
                            continue
                        mode = _norm_mode(m)
                        if mode is not None:
                            m.forward = _make_norm(m, mode, _norm_eps(m))
                            norms.append(m)
            yield tracker
    finally:
        _realize_escaped()
        _active -= 1
        tracker.close()
        for m in patched:
            m.__dict__.pop("forward", None)
        for m in norms:
            m.__dict__.pop("forward", None)
