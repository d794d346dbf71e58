"""Which calibration samples share a block forward, and how ragged ones are padded into one (`plan_groups`, `plan_padded`, the row maps the
linears skip padding rows by, the stacking / padding of the cached kwargs).  Used by the walk (`calibration.walk_blocks`) and by the stacked passes
of finished towers (`replay_towers.TowerGraph`).  Split out of `calibration.py` in round 6."""
from __future__ import annotations

import contextlib
import os
import threading

import torch
import torch.nn as nn

from vlmc import forward, phases

from lavis.compression.pruners.replay_state import (  # noqa: F401
    REPLAY_TOKEN_BUDGET,
    _CTX,
    _bits_equal,
    pad_ragged_enabled,
    replay_group_size,
)


def plan_groups(cur_in, caches, n_samples, group_max):
    """Which calibration samples go through the block together: samples whose input and cached kwargs have identical
    shapes and dtypes (they need not be neighbours: real calibration text is ragged), at most `group_max` per call and at
    most `VLMC_REPLAY_TOKENS` (default 65536) rows of activations per call.  Returns lists of sample indices, ordered by
    their first member; the statistics keep the reference's per-sample order whatever the grouping."""
    try:
        budget = max(1, int(os.environ.get("VLMC_REPLAY_TOKENS", str(REPLAY_TOKEN_BUDGET))))
    except ValueError:
        budget = REPLAY_TOKEN_BUDGET
    buckets = {}
    for j in range(n_samples):
        buckets.setdefault(_stack_key(cur_in[j], caches[j]), []).append(j)
    chunks = []
    for idx in buckets.values():
        x = cur_in[idx[0]]
        rows = max(1, x.numel() // max(1, x.shape[-1]))
        g = max(1, min(group_max, budget // rows))
        chunks += [idx[t:t + g] for t in range(0, len(idx), g)]
    chunks.sort(key=lambda c: c[0])
    return chunks


# While a stacked forward runs: (number of stacked calibration samples, their common batch size, their indices in the
# calibration set).  The statistics hooks read it to keep the reference's per-sample bookkeeping (one `add_batch` per
# sample, :304-314) in the reference's sample order.


def stacked_samples():
    return _CTX.stacked


def stacked_lengths(padded_tokens):
    """During the forward of a PADDED group of ragged samples: the int32 device tensor [samples] of the token rows that are each
    sample's own, for a hook input whose token dimension has `padded_tokens` rows; None otherwise (nothing is padded)."""
    ln = _CTX.stacked_lengths
    return None if ln is None else ln.get(int(padded_tokens))


# ---- ragged calibration text: ONE padded forward per block instead of one per distinct length ---------------------------------
# Real calibration prompts and answers are ragged (blip2_t5_instruct.py:49-53: up to 128 / 256 tokens); grouping the samples by
# shape makes 7 / 18 groups per encoder / decoder block pass on the bench's ragged set, each a walk of the block's Python.  A
# tower whose blocks are called with additive attention masks (the reference's T5 stack always is: extended masks,
# modeling_t5.py:1060-1115) can take all lengths at once: inputs padded with zero rows, masks padded with the dtype's minimum,
# the cross-attention's states padded with zero rows.  A sample's rows keep their bits because every op of the block is
# row-wise, or a product on the batch-invariant kernels (extra key columns do not touch the real ones; masked probabilities are
# exactly 0 in `attn @ v`), or the softmax -- which runs on `vlmc_softmax_rows` during a replay for exactly this reason.  The
# statistics hooks are told each sample's own token count (stacked_lengths).  `VLMC_PAD_RAGGED=0`: groups of equal shapes only.
PAD_MASK_KEYS = {"attention_mask": "self", "encoder_attention_mask": "cross"}


PAD_STATE_KEYS = {"encoder_hidden_states": "cross"}


def plan_padded(cur_in, caches, n_samples, group_max):
    """[(chunk, spec)] covering samples 0 .. n_samples - 1 with PADDED groups, or None when the samples are not ragged or cannot be
    padded (no mask kwarg to hide the padding behind, tensors this function does not know how to pad, mixed dtypes / widths)."""
    if n_samples < 2 or not pad_ragged_enabled():
        return None
    x0, c0 = cur_in[0], caches[0]
    if x0.dim() != 3 or x0.shape[0] != 1 or not x0.is_cuda:
        return None
    T, S = [], []
    for j in range(n_samples):
        x, c = cur_in[j], caches[j]
        if x.dim() != 3 or x.shape[0] != 1 or x.shape[2] != x0.shape[2] or x.dtype != x0.dtype or sorted(c) != sorted(c0):
            return None
        t, s_len = x.shape[1], None
        for k, v in c.items():
            v0 = c0[k]
            if not isinstance(v, torch.Tensor):
                if isinstance(v0, torch.Tensor) or (v is not v0 and v != v0):
                    return None
                continue
            if not isinstance(v0, torch.Tensor) or v.dtype != v0.dtype or v.dim() != v0.dim():
                return None
            if k in PAD_STATE_KEYS:
                if v.dim() != 3 or v.shape[0] != 1 or v.shape[2] != v0.shape[2]:
                    return None
                s_len = v.shape[1]
            elif k not in PAD_MASK_KEYS:
                return None                                           # a tensor kwarg nobody told us how to pad
        for k, kind in PAD_MASK_KEYS.items():
            v = c.get(k)
            if v is None:
                continue
            keys = t if kind == "self" else s_len
            if not (isinstance(v, torch.Tensor) and v.is_floating_point() and v.dim() == 4 and v.shape[0] == 1 and v.shape[1] == 1
                    and keys is not None and v.shape[3] == keys and v.shape[2] in (1, t)):
                return None
        T.append(t)
        S.append(s_len)
    ragged_t, ragged_s = len(set(T)) > 1, len(set(S)) > 1
    if not (ragged_t or ragged_s):
        return None
    if ragged_t and not isinstance(c0.get("attention_mask"), torch.Tensor):
        return None                                                   # nothing to hide padded keys behind
    if ragged_s and (None in S or not isinstance(c0.get("encoder_attention_mask"), torch.Tensor)):
        return None
    try:
        budget = max(1, int(os.environ.get("VLMC_REPLAY_TOKENS", str(REPLAY_TOKEN_BUDGET))))
    except ValueError:
        budget = REPLAY_TOKEN_BUDGET
    # Which samples share a padded forward: ONE group.  (Buckets of similar length -- up to three, 26-34 % fewer rows -- were measured
    # slower in round 5, 486 / 541 against 478 ms: a T5 block forward is ~45 launches whatever its rows; since round 6 the linears skip
    # the padding rows and the fused attention the padding keys, so the rows buckets would save are hardly computed any more.  Removed.)
    buckets = [list(range(n_samples))]
    out = []
    chunks = []
    for bucket in buckets:
        g = max(2, min(group_max, budget // max(T[j] for j in bucket)))
        chunks += [bucket[c_:c_ + g] for c_ in range(0, len(bucket), g)]
    for chunk in chunks:
        tp = max(T[j] for j in chunk)
        sp = max(S[j] for j in chunk) if S[chunk[0]] is not None else None
        if sp is not None and sp == tp:
            sp += 8                                                   # the hooks tell the two kinds of input apart by their padded length
        dev = x0.device
        lengths = {tp: int32_on([T[j] for j in chunk], dev)}
        rows = {(len(chunk), tp): row_map([T[j] for j in chunk], tp, dev)}
        if sp is not None:
            lengths[sp] = int32_on([S[j] for j in chunk], dev)
            rows[(len(chunk), sp)] = row_map([S[j] for j in chunk], sp, dev)
        out.append((chunk, {"T": [T[j] for j in chunk], "S": [S[j] for j in chunk], "tp": tp, "sp": sp, "lengths": lengths,
                            "rows": rows}))
    return out


def int32_on(values, device):
    """A small host list (token counts, a row map) as an int32 device tensor without draining the GPU (vlmc/forward.py: int32_on)."""
    return forward.int32_on(values, device)


def row_map(lengths, padded, device):
    """(int32 device tensor [len(lengths) * padded], number of real rows) for a [samples, padded, d] stack whose sample t owns
    its first lengths[t] token rows: the flattened indices of the real rows in order, then those of the padding rows
    (vlmc_linear_fwd_rows computes the former and clears the latter; vlmc/forward.py: padded_rows)."""
    import numpy as np
    ln = np.asarray(lengths, dtype=np.int64)
    tok = np.arange(padded, dtype=np.int64)[None, :]
    real = tok < ln[:, None]
    flat = (np.arange(len(ln), dtype=np.int64)[:, None] * padded + tok)
    order = np.concatenate([flat[real], flat[~real]]).astype(np.int32)
    return int32_on(order, device), int(real.sum())


def _ragged_to_padded(pieces, padded, fill):
    """[T_j, ...] tensors -> [n, padded, ...], `fill` behind each sample's own rows: ONE concatenation and ONE scatter of the real rows (a copy per
    sample -- `pad_sequence`, or a Python loop of slice assignments -- was ~550 copy launches at the start of every T5 walk)."""
    import numpy as np
    n, lengths = len(pieces), [int(p_.shape[0]) for p_ in pieces]
    flat = torch.cat(pieces, dim=0)
    shape = (n * padded,) + tuple(flat.shape[1:])
    out = flat.new_zeros(shape) if fill == 0 else torch.full(shape, fill, dtype=flat.dtype, device=flat.device)
    ln = np.asarray(lengths, dtype=np.int64)
    tok = np.arange(padded, dtype=np.int64)[None, :]
    rows = (np.arange(n, dtype=np.int64)[:, None] * padded + tok)[tok < ln[:, None]]
    idx = torch.from_numpy(rows)
    idx = idx.pin_memory().to(flat.device, non_blocking=True) if flat.is_cuda else idx
    out.index_copy_(0, idx, flat)
    return out.view((n, padded) + tuple(flat.shape[1:]))


def _pad_inputs(xs, tp):
    """[1, T_j, d] tensors -> [n, tp, d], zero rows behind each sample's own"""
    return _ragged_to_padded([x_[0] for x_ in xs], tp, 0)


def _pad_caches(group, spec):
    """The cached kwargs of a padded group as one set: masks padded with the dtype's minimum (keys that do not exist; the rows of
    queries that do not exist are never read), cross-attention states with zero rows, everything else as the first sample has it."""
    n, tp, sp = len(group), spec["tp"], spec["sp"]
    out = {}
    for k, v0 in group[0].items():
        if not isinstance(v0, torch.Tensor):
            out[k] = v0
        elif k in PAD_STATE_KEYS:
            out[k] = _ragged_to_padded([c[k][0] for c in group], sp, 0)
        else:
            keys = tp if PAD_MASK_KEYS[k] == "self" else sp
            q = tp if v0.shape[2] != 1 else 1
            lowest = torch.finfo(v0.dtype).min
            if q == 1:                                               # [1, 1, 1, keys_j]: one row of keys per sample
                out[k] = _ragged_to_padded([c[k].reshape(-1) for c in group], keys, lowest).view(n, 1, 1, keys)
                continue
            m = torch.full((n, 1, q, keys), lowest, dtype=v0.dtype, device=v0.device)
            for t, c in enumerate(group):
                v = c[k]
                m[t, :, :v.shape[2], :v.shape[3]] = v[0]
            out[k] = m
    return out


def _stack_key(x, cache):
    sig = [tuple(x.shape), x.dtype]
    for k in sorted(cache):
        v = cache[k]
        sig.append((k, tuple(v.shape), v.dtype) if isinstance(v, torch.Tensor) else (k, repr(v)))
    return tuple(sig)


def _stack_caches(group, b0):
    """The cached kwargs of a group of samples as ONE set of kwargs for a stacked forward, or None if they cannot be:
    tensors that carry the samples' batch dimension (`shape[0] == b0`: attention masks, encoder states, a per-sample
    position bias) are concatenated along it; a tensor WITHOUT it (a ViT `rel_pos_bias` [heads, N, N], a `layer_head_mask`
    [heads]) is passed once if every sample holds the same object or the same bits -- concatenating it would hand the
    block a wrong shape, or broadcast silently where the sizes happen to line up; anything else sends the group to the
    per-sample path."""
    out = {}
    for k in group[0]:
        v0 = group[0][k]
        if not isinstance(v0, torch.Tensor):
            out[k] = v0
        elif v0.dim() >= 1 and v0.shape[0] == b0 and v0.dim() >= 2:
            out[k] = torch.cat([c[k] for c in group], dim=0)
        elif all(c[k] is v0 for c in group[1:]) or all(_bits_equal(c[k], v0) for c in group[1:]):
            out[k] = v0
        else:
            return None
    return out
