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


def act_sqnorm_batch(xs, outs=None, call_tokens=None):
    if call_tokens is not None and any(ct is not None for ct in call_tokens):      # padded groups: each call's own rows only
        rows = []
        for x, ct in zip(xs, call_tokens):
            if ct is None:
                rows.append(act_sqnorm(x))
            else:
                rows.append(torch.cat([act_sqnorm(x[c:c + 1, :int(n)]) for c, n in enumerate(ct.tolist())]))
    else:
        rows = [act_sqnorm(x) for x in xs]
    if outs is not None:
        for o, r in zip(outs, rows):
            o.copy_(r)
        return outs
    return rows


def wanda_scaler_update_batch(scalers, nsamples_before, normsqs, batch=1, sqrt_outs=None):
    sqrt_outs = [None] * len(scalers) if sqrt_outs is None else sqrt_outs
    n = nsamples_before
    for s, nsq, sq in zip(scalers, normsqs, sqrt_outs):
        n = wanda_scaler_update(s, nsamples_before, nsq.contiguous(), batch, sqrt_out=sq)
    return n


def wanda_select_batch(weights, sqrt_rows, mode, *, ks=None, n=0, m=0, apply_zero=True, masks=None, partials=None):
    ks = [0] * len(weights) if ks is None else ks
    out_m, out_p = [], []
    for i, (w, sq) in enumerate(zip(weights, sqrt_rows)):
        mk, pt = wanda_select(w, sq, mode, k=ks[i], n=n, m=m, apply_zero=apply_zero,
                              mask=None if masks is None else masks[i], partials=None if partials is None else partials[i])
        out_m.append(mk)
        out_p.append(pt)
    return out_m, out_p


def install(monkeypatch):
    from vlmc import ops
    for name in ("act_sqnorm", "wanda_scaler_update", "sqrt_scaler", "select_partials", "wanda_select", "act_sqnorm_batch",
                 "wanda_scaler_update_batch", "wanda_select_batch"):
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
        self.factor_cache = {}

    def add_batch(self, inp, out=None):
        self.nsamples = self._os.hessian_update(self.H, self.nsamples, inp)

    def free(self):
        self.H = None


def oracle_fasterprune(layer, H, sparsity, prune_n=0, prune_m=0, blocksize=128, percdamp=0.01, return_mask=False,
                       factor_cache=None, score_sink=None):
    from oracle import sparsegpt as OS
    # the reference factorizes per linear; a shared Hessian must survive for the next linear that uses it
    Wn, imp, pruned = OS.prune(layer.weight.data, H.clone(), sparsity, prune_n, prune_m, blocksize, percdamp)
    setattr(layer.weight, "importance_score", imp)
    layer.weight.data = Wn
    return pruned if return_mask else None


def install_sparsegpt(monkeypatch):
    from vlmc import sparsegpt
    monkeypatch.setattr(sparsegpt, "SparseGPT", OracleSparseGPT)
    monkeypatch.setattr(sparsegpt, "fasterprune", oracle_fasterprune)
    monkeypatch.setattr(sparsegpt, "factorize_many", lambda items, percdamp=0.01, **kw: None)     # the oracle factorizes per linear


# ---- DSnoT stand-ins (for vlmc.dsnot) -------------------------------------------------------
def dsnot_act_moments(x):
    """Per-call moments exactly as the reference's hook forms them (dsnot_pruner.py:88-100)."""
    xt = x.reshape(-1, x.shape[-1]).t().type(torch.float32)
    return torch.stack([torch.norm(xt, p=2, dim=1) ** 2, torch.sum(xt, dim=1), torch.var(xt, dim=1, unbiased=False)])


def dsnot_stats_update(in_features, device, normsq, sums, vars_, tokens, batch):
    """oracle.dsnot.DSnoTStat.add_batch restated over per-call moments (same op order and scalar types)."""
    scaler, sum_row, var, n, ntok = torch.zeros(in_features), torch.zeros(in_features), None, 0, 0
    for c, num in enumerate(tokens):
        var = vars_[c].clone() if ntok == 0 else (var * ntok + vars_[c] * num) / (ntok + num)
        ntok += num
        scaler *= n / (n + batch)
        sum_row *= n / (n + batch)
        n += batch
        scaler += normsq[c] / n
        sum_row += sums[c] / n
    if var is None:
        var = torch.zeros(in_features)
    return scaler, sum_row, var, torch.sqrt(scaler), None


def _ostat(stat):
    from types import SimpleNamespace
    return SimpleNamespace(scaler_row=stat.scaler_row, sum_metric_row=stat.sum_row, var=stat.var_row.reshape(-1, 1))


def oracle_dsnot_prune_linear(weight, stat, ratio, *, prune_n=0, prune_m=0, initial_method="wanda", without_DSnoT=False,
                              max_cycle_time=100, update_threshold=0.1, pow_of_var_regrowing=1.0, without_same_sign=True,
                              apply_zero=True):
    from oracle import dsnot as OD
    if prune_n != 0:
        pruned = OD.prune_nm(weight.data, _ostat(stat), prune_n, prune_m, initial_method=initial_method,
                             max_cycle_time=max_cycle_time, update_threshold=update_threshold,
                             pow_of_var_regrowing=pow_of_var_regrowing)
    else:
        pruned = OD.prune_unstructured(weight.data, _ostat(stat), ratio, initial_method=initial_method,
                                       without_DSnoT=without_DSnoT, max_cycle_time=max_cycle_time,
                                       update_threshold=update_threshold, pow_of_var_regrowing=pow_of_var_regrowing,
                                       without_same_sign=without_same_sign)
        if pruned is None:
            return None
    if apply_zero:
        weight[pruned] = 0
    return ~pruned


def install_dsnot(monkeypatch):
    """vlmc.dsnot's three C-ABI entry points replaced; DsnotInputStat / gather_stats stay the product's."""
    from vlmc import dsnot
    monkeypatch.setattr(dsnot, "act_moments", dsnot_act_moments)
    monkeypatch.setattr(dsnot, "act_moments_calls", lambda x, calls: torch.stack(
        [dsnot_act_moments(c) for c in x.reshape(calls, -1, x.shape[-1])], dim=1))
    monkeypatch.setattr(dsnot, "stats_update", dsnot_stats_update)
    monkeypatch.setattr(dsnot, "prune_linear", oracle_dsnot_prune_linear)


# ---- global pruners: oracle stand-in for ops.score_select (CPU tensors) ---------------------------------
def score_select(weights, mode, *, scopes, scope_ks, scores=None, prev_keeps=None, protect_ks=None, apply_weights=True, keeps=None):
    from oracle import global_select as OG
    n = len(scopes)
    weights = list(weights) if weights is not None else [None] * n
    scores = list(scores) if scores is not None else [None] * n
    prev_keeps = list(prev_keeps) if prev_keeps is not None else [None] * n
    protect_ks = list(protect_ks) if protect_ks is not None else [0] * n
    sc = []
    for w, s, pk, prot in zip(weights, scores, prev_keeps, protect_ks):
        v = {"weight": lambda: OG.score_magnitude(w), "score": lambda: s.float().clone(),
             "absw_score": lambda: OG.score_aobd(w, s)}[mode]()
        if pk is not None:
            v = v * pk.to(v.dtype)
        if prot > 0:
            thr = torch.sort(v.flatten(), descending=True)[0][prot - 1]
            v = v.clone()
            v[v >= thr] = torch.finfo(v.dtype).max
        sc.append(v)
    out = []
    thr_of = {}
    for sid, k in enumerate(scope_ks):
        flat = torch.cat([v.flatten() for v, s_ in zip(sc, scopes) if s_ == sid])
        assert 1 <= k <= flat.numel()
        thr_of[sid] = OG.kth_smallest(flat, int(k))
    for i, (v, sid) in enumerate(zip(sc, scopes)):
        keep = v > thr_of[sid]
        if keeps is not None:
            keeps[i].copy_(keep)
            keep = keeps[i]
        if apply_weights and weights[i] is not None:
            weights[i].mul_(keep.to(weights[i].dtype))
        out.append(keep)
    return out


def install_global(monkeypatch):
    from vlmc import ops
    monkeypatch.setattr(ops, "score_select", score_select)
