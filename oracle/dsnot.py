"""Oracle: DSnoT (dynamic sparse no training) statistics and mask refinement, PyTorch-CPU.

TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.  Restates
/root/reference/lavis/compression/pruners/dsnot_pruner.py:
    DSnoTStat.add_batch      :79-101   scaler_row / sum_metric_row running means (per sample),
                                       token-weighted running variance
    reorder_indices          :1881-1925 negatives in order, zero-valued entries -> index 0,
                                       positives reversed at the tail
    prune_unstructured       :553-751  (T5/LLM) and :1285-1482 (ViT, per row there too)
    prune_nm                 :407-552
Literal semantics are kept, including the mask "patch" of :734-740 whose net effect is
`mask[p] = False; mask[r] = True` in EVERY cycle for EVERY row while any row still updates
(SURVEY.md F7), `round()` instead of `int()` for the row budget, and the non-consuming head/tail
pointers.  Pinned against the reference by tests/golden/dsnot.npz.
"""
import torch


class DSnoTStat:
    """State of the reference's DSnoT `WrappedGPT` (the unused `mean` is not tracked)."""

    def __init__(self, in_features):
        self.scaler_row = torch.zeros(in_features)
        self.sum_metric_row = torch.zeros(in_features)
        self.var = torch.zeros(in_features, 1)
        self.nsamples = 0
        self.ntokens = 0

    def add_batch(self, inp):
        if inp.dim() == 2:
            inp = inp.unsqueeze(0)
        b = inp.shape[0]
        x = inp.reshape(-1, inp.shape[-1]).t().type(torch.float32)          # [in, tokens]
        var_inp = torch.var(x, dim=1, unbiased=False, keepdim=True)
        num = x.shape[1]
        self.var = var_inp if self.ntokens == 0 else (self.var * self.ntokens + var_inp * num) / (self.ntokens + num)
        self.ntokens += num
        self.scaler_row *= self.nsamples / (self.nsamples + b)
        self.sum_metric_row *= self.nsamples / (self.nsamples + b)
        self.nsamples += b
        self.scaler_row += torch.norm(x, p=2, dim=1) ** 2 / self.nsamples
        self.sum_metric_row += torch.sum(x, dim=1) / self.nsamples


def reorder_indices(v: torch.Tensor) -> torch.Tensor:
    """For every row of v: positions of the negative entries in order, then (for every zero entry)
    position 0, then positions of the positive entries in reverse order."""
    out = torch.zeros(v.shape, dtype=torch.int64)
    for r in range(v.shape[0]):
        neg = torch.nonzero(v[r] < 0).flatten()
        pos = torch.nonzero(v[r] > 0).flatten()
        out[r, :neg.numel()] = neg
        if pos.numel():
            out[r, v.shape[1] - pos.numel():] = pos.flip(0)
    return out


def _metrics(W, stat, initial_method):
    D = W * stat.sum_metric_row.reshape(1, -1)                              # signed weight * mean activation
    wanda = torch.abs(W) * torch.sqrt(stat.scaler_row.reshape(1, -1))
    if initial_method == "wanda":
        init = wanda.clone()
    elif initial_method == "magnitude":
        init = torch.abs(W)
    else:
        raise ValueError("initial_method must be 'wanda' or 'magnitude' (the reference's 'sparsegpt' branch cannot run)")
    return D, wanda, init


@torch.no_grad()
def prune_unstructured(W, stat, ratio, *, initial_method="wanda", without_DSnoT=False, max_cycle_time=100,
                       update_threshold=0.1, pow_of_var_regrowing=1.0, without_same_sign=True):
    """Returns the pruned mask (True = pruned) [out, in], or None when ratio == 0 (the reference skips the linear)."""
    D, wanda, init = _metrics(W, stat, initial_method)
    out_f, in_f = init.shape
    if ratio == 0.:
        return None
    k = round(in_f * ratio)
    res = in_f - k
    order = torch.sort(init, dim=-1, stable=True)[1]
    P0, R0 = order[:, :k], order[:, k:]
    m = torch.zeros((out_f, in_f), dtype=torch.bool)
    m.scatter_(1, P0, True)
    if without_DSnoT:
        return m
    wanda.scatter_(1, P0, float("inf"))
    kept_sorted = torch.sort(wanda, dim=1, stable=True)[1][:, :res]
    prune_list = torch.gather(kept_sorted, 1, reorder_indices(torch.gather(D, 1, kept_sorted)))
    G = D.clone()
    G.scatter_(1, R0, 0)
    err = G.sum(dim=1, keepdim=True)
    sign0 = torch.sign(err)
    if pow_of_var_regrowing:
        G = G / torch.pow(stat.var.reshape(1, -1), pow_of_var_regrowing)
    regrow_list = torch.sort(G, dim=1, stable=True)[1]
    rptr = torch.zeros((out_f, 2), dtype=torch.int64)
    rptr[:, 1] = in_f - 1
    pptr = torch.zeros((out_f, 2), dtype=torch.int64)
    pptr[:, 1] = res - 1
    step = torch.tensor([1, -1])
    u = torch.ones((out_f, 1), dtype=torch.bool)
    rows = torch.arange(out_f)
    cycle = 0
    while bool(u.any()) and cycle < max_cycle_time:
        cycle += 1
        rs = (err > 0).long().flatten()                       # tail when the error is positive
        r = regrow_list[rows, rptr[rows, rs]]
        rptr[rows, rs] += step[rs]
        ps = (err < 0).long().flatten()
        p = prune_list[rows, pptr[rows, ps]]
        pptr[rows, ps] += step[ps]
        Dp, Dr = D[rows, p].unsqueeze(1), D[rows, r].unsqueeze(1)
        after = err + Dp - Dr
        u = u & (err.abs() > update_threshold)
        if not without_same_sign:
            u = u & (sign0 == torch.sign(after))
        m[rows, p] = False                                    # net effect of the four scatters (:727-740)
        m[rows, r] = True
        err = err + torch.where(u, Dp, torch.zeros_like(Dp))
        err = err - torch.where(u, Dr, torch.zeros_like(Dr))
    return m


@torch.no_grad()
def prune_nm(W, stat, n, m_, *, initial_method="wanda", max_cycle_time=100, update_threshold=0.1,
             pow_of_var_regrowing=1.0, trace=None, ties="torch_cpu"):
    """n:m branch (:407-552).  Group order by stable sort (the reference's torch.sort is unstable on ties).
    `trace` (a dict): receives "tie_rows", the rows whose walk met an m-group whose two smallest metrics were EQUAL when
    `torch.topk(pruning_block, 1, largest=False)` (:517-519) had to pick one -- an exhausted group, both kept entries already
    at rowmax + 1 = +inf.  `ties="torch_cpu"` (default): the entry the reference's CPU run picks (oracle/topk_order.py);
    `ties="lowest"`: the lowest column (the rule of rounds 1-4)."""
    D, _, init = _metrics(W, stat, initial_method)
    init = init.clone().float()
    out_f, in_f = init.shape
    g = torch.sort(init.reshape(out_f, in_f // m_, m_), dim=2, stable=True)[1] + \
        (torch.arange(in_f // m_) * m_).reshape(1, -1, 1)
    P0 = g[:, :, :n].reshape(out_f, -1)
    R0 = g[:, :, n:].reshape(out_f, -1)
    mask = torch.zeros((out_f, in_f), dtype=torch.bool)
    mask.scatter_(1, P0, True)
    G = D.clone()
    G.scatter_(1, R0, 0)
    err = G.sum(dim=1, keepdim=True)
    sign0 = torch.sign(err)
    if pow_of_var_regrowing:
        G = G / torch.pow(stat.var.reshape(1, -1), pow_of_var_regrowing)
    regrow_list = torch.sort(G, dim=1, stable=True)[1]
    rptr = torch.zeros((out_f, 2), dtype=torch.int64)
    rptr[:, 1] = in_f - 1
    step = torch.tensor([1, -1])
    init.scatter_(1, P0, float("inf"))
    big = init.max(dim=1, keepdim=True)[0] + 1
    u = torch.ones((out_f, 1), dtype=torch.bool)
    rows = torch.arange(out_f)
    cycle = 1
    while bool(u.any()) and not cycle > max_cycle_time:
        cycle += 1
        rs = (err > 0).long().flatten()
        pos = rptr[rows, rs]
        r = regrow_list[rows, pos]
        Dr = D[rows, r].unsqueeze(1)
        start = r - r % m_
        block = torch.stack([init[rows, start + a] for a in range(m_)], dim=1)
        srt = torch.sort(block, dim=1, stable=True)
        if trace is not None:
            trace.setdefault("tie_rows", set()).update(torch.nonzero(srt[0][:, 0] == srt[0][:, 1]).flatten().tolist())
        pick = srt[1][:, 0].clone()                                         # smallest kept metric of r's group
        if ties == "torch_cpu":
            from . import topk_order
            for rr in torch.nonzero(srt[0][:, 0] == srt[0][:, 1]).flatten().tolist():
                pick[rr] = topk_order.smallest(block[rr].tolist(), 1)[0]
        p = start + pick
        Dp = D[rows, p].unsqueeze(1)
        after = err + Dp - Dr
        u = u & (sign0 == torch.sign(after)) & (err.abs() > update_threshold)
        init[rows, p] = big.flatten()
        mask[rows, p] = u.flatten()
        mask[rows, r] = ~u.flatten()
        err = err + torch.where(u, Dp, torch.zeros_like(Dp))
        err = err - torch.where(u, Dr, torch.zeros_like(Dr))
        rptr[rows, rs] = pos + step[rs]
    return mask
