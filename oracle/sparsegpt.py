"""Oracle: SparseGPT Hessian accumulation and blocked OBS pruning (PyTorch-CPU tensor algebra).

TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.  Restates
/root/reference/lavis/compression/pruners/sparsegpt_pruner.py:
    hessian_update    :68-79    H *= n/(n+b); n += b; x = sqrt(2/n) * float(X)^T; H += x x^T
    inverse_factor    :92-160   dead columns, inf clamps, damp-only-on-failure Cholesky,
                                cholesky_inverse, upper Cholesky of the inverse
    prune             :81-215   importance score, per-128-column mask (unstructured threshold
                                with `<=`, or n:m chosen on the compensated block), sequential
                                column sweep, trailing update
n:m ties: n smallest by stable order (lowest column first), like oracle/wanda.py.
Pinned against the reference by tests/golden/sparsegpt.npz.
"""
import math

import torch

from . import topk_order


def hessian_update(H: torch.Tensor, nsamples: int, x: torch.Tensor):
    """One `SparseGPT.add_batch` call; x is [b, T, in] or [T, in].  Returns new nsamples (H in place)."""
    if x.dim() == 2:
        x = x.unsqueeze(0)
    b = x.shape[0]
    xt = x.reshape(-1, x.shape[-1]).t()
    H *= nsamples / (nsamples + b)
    nsamples += b
    xt = math.sqrt(2 / nsamples) * xt.float()
    H += xt.matmul(xt.t())
    return nsamples


def _clamp_inf(H):
    pos = torch.isinf(H) & (H > 0)
    if pos.any():
        H[pos] = torch.quantile(H, 0.999)
    neg = torch.isinf(H) & (H < 0)
    if neg.any():
        H[neg] = torch.quantile(H, 0.001)


def _chol_with_damping(H, damp, upper, trace=None, which=None):
    idx = torch.arange(H.shape[0])
    steps = 0
    while True:
        L, info = torch.linalg.cholesky_ex(H, upper=upper)
        if int(info) == 0 and not torch.isnan(L).any():
            if trace is not None:
                trace[which] = steps            # how often this loop damped: the route the reference took
            return L
        H[idx, idx] += damp                     # damping is added ONLY after a failure (:114-128)
        steps += 1


def inverse_factor(H: torch.Tensor, W: torch.Tensor, percdamp=0.01, trace=None):
    """Upper Cholesky factor U of H^-1 (U^T U = H^-1); zeroes W's dead columns in place.  `trace` (a dict) receives the
    number of damping steps of the two loops (:112-128, :139-150) as "damp_H" / "damp_Hinv"."""
    H = H.clone()
    dead = torch.diag(H) == 0
    H[dead, dead] = 1
    W[:, dead] = 0
    _clamp_inf(H)
    L = _chol_with_damping(H, percdamp * torch.mean(torch.diag(H)), upper=False, trace=trace, which="damp_H")
    Hi = torch.cholesky_inverse(L)
    _clamp_inf(Hi)
    return _chol_with_damping(Hi, percdamp * torch.mean(torch.diag(Hi).abs()), upper=True, trace=trace, which="damp_Hinv")


def block_mask_unstructured(W1, diag1, sparsity):
    tmp = W1 ** 2 / diag1.reshape(1, -1) ** 2
    thresh = torch.sort(tmp.flatten())[0][int(tmp.numel() * sparsity)]
    return tmp <= thresh


def sweep_block(W1, U1, mask1, prune_n, prune_m):
    """Sequential OBS sweep over the columns of one block (:186-205).  W1 [out,c] is consumed;
    returns (Q1, Err1, mask1)."""
    count = W1.shape[1]
    Q1 = torch.zeros_like(W1)
    Err1 = torch.zeros_like(W1)
    d_all = torch.diag(U1)
    for i in range(count):
        w = W1[:, i]
        d = U1[i, i]
        if prune_n != 0 and i % prune_m == 0:
            tmp = W1[:, i:i + prune_m] ** 2 / d_all[i:i + prune_m].reshape(1, -1) ** 2
            srt, order = torch.sort(tmp, dim=1, stable=True)
            idx = order[:, :prune_n]
            # :191 is `torch.topk(tmp, prune_n, dim=1, largest=False)`: WHICH of several equal scores it returns on the CPU is
            # libstdc++'s nth_element order (oracle/topk_order.py); only rows whose n-th and (n+1)-th smallest are equal can differ
            # from the stable order (exact zeros of an already pruned weight)
            if 0 < prune_n < tmp.shape[1]:
                for r in torch.nonzero(srt[:, prune_n - 1] == srt[:, prune_n]).flatten().tolist():
                    idx[r] = torch.tensor(topk_order.smallest(tmp[r].tolist(), prune_n), dtype=idx.dtype)
            mask1.scatter_(1, i + idx, True)
        q = w.clone()
        q[mask1[:, i]] = 0
        Q1[:, i] = q
        err = (w - q) / d
        W1[:, i:] -= err.unsqueeze(1).matmul(U1[i, i:].unsqueeze(0))
        Err1[:, i] = err
    return Q1, Err1, mask1


@torch.no_grad()
def prune(weight: torch.Tensor, H: torch.Tensor, sparsity, prune_n=0, prune_m=0, blocksize=128, percdamp=0.01, trace=None):
    """`fasterprune`: returns (new weight in weight.dtype, importance_score, pruned mask bool)."""
    W = weight.detach().clone().float()
    U = inverse_factor(H, W, percdamp, trace)
    score = W ** 2 / torch.diag(U).reshape(1, -1) ** 2
    importance = score.abs().mean().item()
    cols = W.shape[1]
    pruned = torch.zeros_like(W, dtype=torch.bool)
    for i1 in range(0, cols, blocksize):
        i2 = min(i1 + blocksize, cols)
        W1 = W[:, i1:i2].clone()
        U1 = U[i1:i2, i1:i2]
        if prune_n == 0:
            mask1 = block_mask_unstructured(W1, torch.diag(U1), sparsity)
        else:
            mask1 = torch.zeros_like(W1) == 1
        Q1, Err1, mask1 = sweep_block(W1, U1, mask1, prune_n, prune_m)
        W[:, i1:i2] = Q1
        pruned[:, i1:i2] = mask1
        W[:, i2:] -= Err1.matmul(U[i1:i2, i2:])
    return W.to(weight.dtype), importance, pruned
