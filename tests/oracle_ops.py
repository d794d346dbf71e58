"""Test-only stand-ins for `vlmc.ops` built on the CPU oracle, with the same signatures.

Used by the CPU host-logic tests to drive the drop-in pruners' orchestration (capture,
block walk, hook de-duplication, statistics exchange, importance read-back) without a
GPU: the tests monkeypatch `vlmc.ops.*` with these.  The product never imports this."""
import numpy as np
import torch

from oracle import wanda as OW

_PARTS = {"row": lambda o, i: o, "matrix": lambda o, i: 512, "nm": lambda o, i: 2048}


def act_sqnorm(x, out=None):
    if x.dim() == 2:
        x = x.unsqueeze(0)
    rows = torch.from_numpy(np.stack([OW.act_sqnorm(x[c]) for c in range(x.shape[0])]))
    if out is not None:
        out.copy_(rows)
        return out
    return rows


def wanda_scaler_update(scaler_row, nsamples_before, normsq, batch=1, sqrt_out=None):
    s = scaler_row.numpy().copy()
    n = nsamples_before
    if n == 0:
        s[:] = 0
    if normsq is not None:
        ns = normsq.reshape(-1, scaler_row.numel()).numpy()
        for c in range(ns.shape[0]):
            s, n = OW.scaler_update(s, n, ns[c], batch)
    scaler_row.copy_(torch.from_numpy(s))
    if sqrt_out is not None:
        sqrt_out.copy_(torch.from_numpy(np.sqrt(s, dtype=np.float32)))
    return n


def sqrt_scaler(scaler_row):
    return torch.from_numpy(np.sqrt(scaler_row.numpy(), dtype=np.float32))


def select_partials(mode, out_f, in_f):
    return _PARTS[mode](out_f, in_f)


def wanda_select(weight, sqrt_scaler_row, mode, *, k=0, n=0, m=0, apply_zero=True, mask=None, partials=None):
    w = weight.detach().to(torch.float32).abs().numpy()
    score = (w * sqrt_scaler_row.numpy()[None, :]).astype(np.float32)
    if mode == "row":
        pruned = OW.select_rows(score, k)
    elif mode == "matrix":
        pruned = OW.select_matrix(score, k)
    else:
        pruned = OW.select_nm(score, n, m)
    keep = torch.from_numpy(~pruned)
    if mask is None:
        mask = keep
    else:
        mask.copy_(keep)
    if apply_zero:
        weight[torch.from_numpy(pruned)] = 0
    nparts = select_partials(mode, *weight.shape)
    if partials is None:
        partials = torch.zeros(nparts, dtype=torch.float64)
    partials[:nparts] = 0
    partials[0] = float(score.astype(np.float64).sum())
    return mask, partials[:nparts]


def install(monkeypatch):
    from vlmc import ops
    for name in ("act_sqnorm", "wanda_scaler_update", "sqrt_scaler", "select_partials", "wanda_select"):
        monkeypatch.setattr(ops, name, globals()[name])
