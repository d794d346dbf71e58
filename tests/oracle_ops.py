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


# ---- SparseGPT stand-ins (for vlmc.sparsegpt) ----------------------------------------------
class OracleSparseGPT:
    def __init__(self, layer):
        from oracle import sparsegpt as OS
        self._os = OS
        self.layer = layer
        self.rows, self.columns = layer.weight.shape
        self.H = torch.zeros((self.columns, self.columns))
        self.nsamples = 0

    def add_batch(self, inp, out=None):
        self.nsamples = self._os.hessian_update(self.H, self.nsamples, inp)

    def free(self):
        self.H = None


def oracle_fasterprune(layer, H, sparsity, prune_n=0, prune_m=0, blocksize=128, percdamp=0.01, return_mask=False):
    from oracle import sparsegpt as OS
    Wn, imp, pruned = OS.prune(layer.weight.data, H, sparsity, prune_n, prune_m, blocksize, percdamp)
    setattr(layer.weight, "importance_score", imp)
    layer.weight.data = Wn
    return pruned if return_mask else None


def install_sparsegpt(monkeypatch):
    from vlmc import sparsegpt
    monkeypatch.setattr(sparsegpt, "SparseGPT", OracleSparseGPT)
    monkeypatch.setattr(sparsegpt, "fasterprune", oracle_fasterprune)
