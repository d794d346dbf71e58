"""Oracle (fast form): the Wanda loop body as the SAME PyTorch CPU op sequence the
reference executes (torch.norm, torch.sort(stable), scatter_, boolean-index zeroing).

TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.  This is the CPU baseline that
bench.py times ("cpu_baseline", kind "port"): it is what the reference's pruner does
per linear when run on the host cores, restated here without the model/hook
plumbing.  tests/test_oracle_golden.py pins it against the reference's golden
vectors and against the explicit restatement in oracle/wanda.py.
"""
from __future__ import annotations

import torch


class WandaStat:
    """Running activation statistic of one linear (WrappedGPT, wanda_pruner.py:51-81)."""

    def __init__(self, in_features: int):
        self.scaler_row = torch.zeros(in_features)
        self.nsamples = 0

    def add_batch(self, inp: torch.Tensor):
        if inp.dim() == 2:
            inp = inp.unsqueeze(0)
        b = inp.shape[0]
        inp = inp.reshape(-1, inp.shape[-1]).t()                      # :73-75
        self.scaler_row *= self.nsamples / (self.nsamples + b)        # :77
        self.nsamples += b                                            # :78
        inp = inp.type(torch.float32)                                 # :80
        self.scaler_row += torch.norm(inp, p=2, dim=1) ** 2 / self.nsamples   # :81


@torch.no_grad()
def prune_linear(weight: torch.Tensor, scaler_row: torch.Tensor, mode: str, *, ratio=None, n=0, m=0, apply_zero=True):
    """wanda_pruner.py:318-341 (mode "row"), :666-687 (mode "matrix"), :326-329 (mode "nm").
    Zeroes `weight` in place when apply_zero.  Returns (mask True=keep, importance_score)."""
    W_metric = torch.abs(weight) * torch.sqrt(scaler_row.reshape((1, -1)))          # :318
    importance = W_metric.abs().mean().item()                                       # :320
    W_mask = torch.zeros_like(W_metric) == 1
    if mode == "nm":
        for ii in range(0, W_metric.shape[1], m):                                   # :326-329
            tmp = W_metric[:, ii:ii + m].float()
            # stable ascending order = lowest column first on ties (oracle/wanda.py: select_nm)
            idx = torch.sort(tmp, dim=1, stable=True)[1][:, :n]
            W_mask.scatter_(1, ii + idx, True)
    elif mode == "row":
        sort_res = torch.sort(W_metric, dim=-1, stable=True)                        # :332
        indices = sort_res[1][:, :int(W_metric.shape[1] * ratio)]                   # :336
        W_mask.scatter_(1, indices, True)                                           # :337
    elif mode == "matrix":
        thres = torch.sort(W_metric.flatten())[0][int(W_metric.numel() * ratio)]    # :682
        W_mask = W_metric < thres                                                   # :683
    else:
        raise ValueError(mode)
    if apply_zero:
        weight[W_mask] = 0                                                          # :341
    return ~W_mask, importance
