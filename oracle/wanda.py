"""Oracle: Wanda activation statistics, score and mask selection (CPU, numpy/torch).

TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.  Every function cites the
reference lines it restates (paths relative to /root/reference).

Numerical contract (pinned against the reference by tests/golden, see
tests/test_oracle_golden.py):

* statistics  `lavis/compression/pruners/wanda_pruner.py:68-81`
    per hook call X[b, T, in]:
        s *= float32(n / (n + b));  n += b
        s += ( sqrtf( chain_t fmaf(x_t, x_t, acc) ) )**2 / float32(n)
    torch's CPU `norm(p=2, dim=1)` on the transposed fp32 view reduces each
    channel sequentially over tokens with one fused multiply-add per token
    (verified here bit-for-bit for bf16/fp16/fp32 inputs).
* score       `wanda_pruner.py:318` / `:666`   |W| (upcast) * sqrtf(s), fp32
* row select  `wanda_pruner.py:332-337`  stable sort per row, first int(in*ratio)
* ViT select  `wanda_pruner.py:682-683`  matrix-wide strict `<` threshold
* n:m select  `wanda_pruner.py:326-329`  n smallest in every m consecutive
  columns (ties: lowest column first -- torch.topk's tie order is
  implementation-defined)
* apply       `wanda_pruner.py:339-341`  mask = ~pruned (True = keep); W[pruned] = 0
"""
from __future__ import annotations

import numpy as np
import torch


# --------------------------------------------------------------------------- #
# exact fp32 fused multiply-add, vectorised (round-to-odd in float64)
# --------------------------------------------------------------------------- #
def _fma32(a: np.ndarray, b: np.ndarray, c: np.ndarray) -> np.ndarray:
    """Correctly rounded float32 fma(a, b, c) for float32 arrays.

    a*b is exact in float64 (24+24 <= 53 bits).  The float64 sum is made
    round-to-odd (via the TwoSum error term) so the final rounding to float32
    cannot double-round.
    """
    p = a.astype(np.float64) * b.astype(np.float64)
    c64 = c.astype(np.float64)
    s = p + c64
    bb = s - p
    err = (p - (s - bb)) + (c64 - bb)            # exact error of the float64 add
    bits = s.view(np.int64).copy()
    inexact = (err != 0) & np.isfinite(s)
    even = (bits & 1) == 0
    # true value lies between s and its neighbour in the direction of err
    toward_larger_mag = (err > 0) == (s > 0)
    adj = np.where(toward_larger_mag, 1, -1).astype(np.int64)
    # s == 0 with err != 0 cannot happen (sum of two doubles that cancels is exact)
    fix = inexact & even
    bits = np.where(fix, bits + adj, bits)
    return bits.view(np.float64).astype(np.float32)


def act_sqnorm(x: torch.Tensor) -> np.ndarray:
    """(||x[:, c]||_2)**2 per channel for one hook call.

    x: [tokens, in] (or [b, T, in], flattened like wanda_pruner.py:73-75),
    any float dtype.  Returns float32 [in] = square(sqrtf(sum_t x^2)) with the
    sequential fma chain torch's CPU kernel uses (wanda_pruner.py:80-81).
    """
    x = x.reshape(-1, x.shape[-1]).to(torch.float32).numpy()
    acc = np.zeros(x.shape[1], dtype=np.float32)
    for t in range(x.shape[0]):
        acc = _fma32(x[t], x[t], acc)
    r = np.sqrt(acc, dtype=np.float32)
    return (r * r).astype(np.float32)


def scaler_update(scaler_row: np.ndarray, nsamples: int, normsq: np.ndarray, batch: int = 1):
    """One `WrappedGPT.add_batch` recurrence step (wanda_pruner.py:77-81).

    Returns (new_scaler_row float32 [in], new_nsamples).
    """
    f = np.float32(nsamples / (nsamples + batch))      # python double -> fp32 scalar
    s = (scaler_row.astype(np.float32) * f).astype(np.float32)
    nsamples += batch
    q = (normsq.astype(np.float32) / np.float32(nsamples)).astype(np.float32)
    return (s + q).astype(np.float32), nsamples


def wanda_stats(xs) -> np.ndarray:
    """scaler_row after feeding the hook calls `xs` (list of [b,T,in] tensors)."""
    in_f = xs[0].shape[-1]
    s = np.zeros(in_f, dtype=np.float32)
    n = 0
    for x in xs:
        b = x.shape[0] if x.dim() == 3 else 1
        s, n = scaler_update(s, n, act_sqnorm(x), b)
    return s


# --------------------------------------------------------------------------- #
# score + selection
# --------------------------------------------------------------------------- #
def wanda_score(W: torch.Tensor, scaler_row: np.ndarray) -> np.ndarray:
    """|W| * sqrt(scaler_row) in fp32 (wanda_pruner.py:318)."""
    w = W.detach().to(torch.float32).abs().numpy()
    return (w * np.sqrt(scaler_row.astype(np.float32), dtype=np.float32)[None, :]).astype(np.float32)


def importance_score(score: np.ndarray) -> float:
    """mean(score) as a python float (wanda_pruner.py:320).  Compared with rtol
    1e-5: the fp32 summation order of torch.mean is not part of the contract."""
    return float(np.mean(score, dtype=np.float64))


def select_rows(score: np.ndarray, k: int) -> np.ndarray:
    """Per-row k smallest by stable sort (wanda_pruner.py:332-337).
    Returns pruned mask bool [out,in] (True = pruned)."""
    out_f, in_f = score.shape
    idx = np.argsort(score, axis=1, kind="stable")        # NaN sorts last, like torch
    pruned = np.zeros((out_f, in_f), dtype=bool)
    if k > 0:
        np.put_along_axis(pruned, idx[:, :k], True, axis=1)
    return pruned


def select_matrix(score: np.ndarray, k_index: int) -> np.ndarray:
    """Matrix-wide strict threshold (ViT rule, wanda_pruner.py:682-683):
    thr = sort(flatten)[k_index]; pruned = score < thr."""
    flat = np.sort(score.reshape(-1), kind="stable")
    thr = flat[k_index]
    return score < thr


def select_nm(score: np.ndarray, n: int, m: int, ties: str = "torch_cpu") -> np.ndarray:
    """n smallest of every m consecutive columns (wanda_pruner.py:326-329: `torch.topk(tmp, n, dim=1, largest=False)`).
    Groups whose n-th and (n+1)-th smallest scores are EQUAL are decided as the reference's CPU run decides them
    (`ties="torch_cpu"`: oracle/topk_order.py, the order libstdc++'s nth_element leaves equal keys in) or lowest column
    first (`ties="lowest"`, the rule of rounds 1-4).  A trailing group shorter than m (in % m != 0) is handled like
    torch.topk on the short slice would fail -> we require in % m == 0 (true for every model width)."""
    from . import topk_order
    out_f, in_f = score.shape
    assert in_f % m == 0, "n:m selection needs in_features % m == 0"
    g = score.reshape(out_f, in_f // m, m)
    idx = np.argsort(g, axis=2, kind="stable")[:, :, :n]
    pruned = np.zeros_like(g, dtype=bool)
    np.put_along_axis(pruned, idx, True, axis=2)
    if ties == "torch_cpu" and 0 < n < m:
        srt = np.sort(g, axis=2)
        tied = (srt[:, :, n - 1] == srt[:, :, n]) | (np.isnan(srt[:, :, n - 1]) & np.isnan(srt[:, :, n]))
        for r, c in zip(*np.nonzero(tied)):
            pruned[r, c, :] = False
            pruned[r, c, topk_order.smallest(g[r, c], n)] = True
    else:
        assert ties in ("torch_cpu", "lowest")
    return pruned.reshape(out_f, in_f)


def nm_tie_groups(score: np.ndarray, n: int, m: int) -> np.ndarray:
    """bool [out, in/m]: groups where the n-th and (n+1)-th smallest scores are
    equal, i.e. where torch.topk's answer is implementation-defined."""
    out_f, in_f = score.shape
    g = np.sort(score.reshape(out_f, in_f // m, m), axis=2)
    return g[:, :, n - 1] == g[:, :, n]


def prune_linear(W: torch.Tensor, scaler_row: np.ndarray, mode: str, *, ratio=None, n=0, m=0,
                 apply_zero=True):
    """Full per-linear step of the reference loop body (wanda_pruner.py:316-341 /
    :664-687).  mode in {"row", "matrix", "nm"}.

    Returns dict(mask=bool [out,in] True=keep, weight=tensor like W (zeroed unless
    apply_zero=False), importance_score=float).
    """
    score = wanda_score(W, scaler_row)
    if mode == "nm":
        pruned = select_nm(score, n, m)
    elif mode == "row":
        pruned = select_rows(score, int(score.shape[1] * ratio))
    elif mode == "matrix":
        pruned = select_matrix(score, int(score.size * ratio))
    else:
        raise ValueError(mode)
    Wn = W.detach().clone()
    if apply_zero:
        Wn[torch.from_numpy(pruned)] = 0
    return {"mask": ~pruned, "weight": Wn, "importance_score": importance_score(score)}
