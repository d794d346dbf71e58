"""n:m selection when scores TIE inside an m-group (wanda_pruner.py:326-329, `torch.topk(tmp, n, dim=1, largest=False)`;
dsnot_pruner.py:517-519 with n = 1 on an m-group whose kept entries are exhausted).

Which of several equal scores `torch.topk` returns is not specified by PyTorch; on the CPU -- the reference path this build
is held against -- it is whatever libstdc++'s `std::nth_element` leaves in front (ATen TopKImpl.h).  Rounds 1-4 decided ties
"lowest column first" and only BOUNDED the divergence; round 5 restates the CPU order itself (oracle/topk_order.py for the
oracle, csrc/topk_order.hpp for the kernels), so the masks the reference recorded on crafted ties
(`tests/golden/nm_ties.npz`: Wanda 2:4 / 4:8 in three dtypes, DSnoT walks that return to exhausted groups) are reproduced
EXACTLY, by the oracle and by the kernels.  Also here: the restated order against this container's own `torch.topk` on
random tie patterns, and how many groups of a re-pruned (already half-zero) weight the old rule decided differently.
"""
import math
import random

import numpy as np
import pytest
import torch

import golden_io
from oracle import topk_order
from oracle import wanda as OW

TIES = golden_io.load("nm_ties")
CASES = sorted({k.split("/")[0] for k in TIES if not k.startswith("dsnot_")})
DSNOT_CASES = sorted({k.split("/")[0] for k in TIES if k.startswith("dsnot_")})


def _split(name):
    W, xs = TIES[f"{name}/W"], TIES[f"{name}/xs"]
    n, m = int(TIES[f"{name}/n"]), int(TIES[f"{name}/m"])
    s = OW.wanda_stats([x[None] for x in xs])
    score = OW.wanda_score(W, s)                                        # fp32 [out, in]: what both sides rank
    tied = OW.nm_tie_groups(score, n, m)                                # a tie across the selection boundary
    return W, s, score, n, m, tied


def test_restated_order_is_this_containers_torch_topk():
    """oracle/topk_order.py against `torch.topk` on the CPU: 20 000 random tie patterns, m in {4, 8}, several n, NaN included."""
    rng = random.Random(0)
    for m, k in [(4, 2), (8, 4), (4, 1), (4, 3), (8, 2), (8, 6), (8, 1), (2, 1)]:
        for _ in range(2500):
            nv = rng.randint(1, 4)
            vals = [float(rng.randint(0, nv - 1)) for _ in range(m)]
            if rng.random() < 0.1:
                vals[rng.randrange(m)] = float("nan")
            if rng.random() < 0.1:
                vals[rng.randrange(m)] = math.inf
            want = sorted(torch.topk(torch.tensor([vals, vals]), k, dim=1, largest=False)[1][0].tolist())
            assert topk_order.smallest(vals, k) == want, (m, k, vals)


@pytest.mark.parametrize("name", CASES)
def test_oracle_reproduces_the_recorded_reference_masks_on_ties(name):
    W, s, score, n, m, tied = _split(name)
    got = OW.prune_linear(W, s, "nm", n=n, m=m)
    assert int(tied.sum()) > 20, "the fixture is meant to tie"
    assert np.array_equal(got["mask"], TIES[f"{name}/mask"].numpy())
    assert torch.equal(got["weight"], TIES[f"{name}/Wn"])
    assert got["importance_score"] == pytest.approx(float(TIES[f"{name}/imp"]), rel=1e-5)
    # the rule of rounds 1-4 differs on these fixtures (so they do pin the order), and only among equal scores
    old = OW.select_nm(score, n, m, ties="lowest")
    ref_pruned = ~TIES[f"{name}/mask"].numpy()
    differ = (old != ref_pruned).reshape(old.shape[0], -1, m).any(axis=2)
    assert differ.any() and not differ[~tied].any()
    print(f"{name}: {int(tied.sum())} tied groups, {int(differ.sum())} of them decided differently by lowest-column-first")


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_kernel_reproduces_the_recorded_reference_masks_on_ties(name):
    from vlmc import ops
    dev = "cuda:0"
    W, s, score, n, m, tied = _split(name)
    Wd = W.clone().to(dev)
    sq = ops.sqrt_scaler(torch.from_numpy(s).to(dev))
    mask, parts = ops.wanda_select(Wd, sq, "nm", n=n, m=m)
    assert np.array_equal(mask.cpu().numpy(), TIES[f"{name}/mask"].numpy()), "kernel and reference disagree on a tie"
    assert torch.equal(Wd.cpu(), TIES[f"{name}/Wn"])


def test_how_much_of_a_repruned_weight_the_old_rule_decided_differently():
    """re-pruning an already 50 %-sparse 5120 x 2048 weight at 2:4 -- exact-zero scores tie in every group
    with three or four zeros (5 / 16 of the groups).  Fraction of GROUPS whose mask lowest-column-first decided otherwise than
    the reference's CPU run (now reproduced): reported, and bounded so that a change of either rule is noticed."""
    g = torch.Generator().manual_seed(0)
    W = torch.randn(5120, 2048, generator=g) * 0.02
    W[torch.rand(5120, 2048, generator=g) < 0.5] = 0
    score = (W.abs() * (torch.rand(2048, generator=g) + 0.5)).numpy()[:512]            # 512 rows are plenty for a fraction
    tied = OW.nm_tie_groups(score, 2, 4)
    new, old = OW.select_nm(score, 2, 4), OW.select_nm(score, 2, 4, ties="lowest")
    differ = (new != old).reshape(512, -1, 4).any(axis=2)
    assert not differ[~tied].any()
    frac_tied, frac_diff = tied.mean(), differ.mean()
    print(f"2:4 re-prune of a 50 %-zero weight: {frac_tied:.4f} of the groups tie at the boundary, {frac_diff:.4f} were decided differently")
    assert 0.28 < frac_tied < 0.34                                        # P(>= 3 zeros of 4) = 5 / 16
    assert 0.20 < frac_diff <= frac_tied                                  # (measured: every tied group -- the CPU order never takes the two lowest of >= 3 equal zeros)
    # the pruned WEIGHTS are the same either way (only zeros change sides); what differs is which zero positions keep a mask bit
    assert np.array_equal(np.where(new, 0, score), np.where(old, 0, score))


# ---- DSnoT n:m: an exhausted m-group (dsnot_pruner.py:517-519) -----------------------------------------------------------
def _dsnot_inputs(name):
    from oracle import dsnot as OD
    W, xs = TIES[f"{name}/W"], TIES[f"{name}/xs"]
    st = OD.DSnoTStat(W.shape[1])
    for x in xs:
        st.add_batch(x[None])
    kw = {k.split("/")[-1]: (v.item() if hasattr(v, "item") else v) for k, v in TIES.items() if k.startswith(f"{name}/kw/")}
    return W, xs, st, kw


@pytest.mark.parametrize("name", DSNOT_CASES)
def test_dsnot_oracle_reproduces_the_recorded_walk_through_exhausted_groups(name):
    """When the regrow walk returns to an m-group whose two kept entries were both swapped out already, every entry of the
    group sits at +inf and `torch.topk(pruning_block, 1, largest=False)` picks one of four equal values; each pick changes
    the rest of the row's walk.  Every row equals the reference's recorded mask; the old rule did not."""
    from oracle import dsnot as OD
    W, xs, st, kw = _dsnot_inputs(name)
    trace = {}
    pruned = OD.prune_nm(W, st, 2, 4, trace=trace, **kw)
    ref_keep = TIES[f"{name}/mask"]
    tie_rows = sorted(trace.get("tie_rows", ()))
    assert len(tie_rows) >= 3
    assert torch.equal(~pruned, ref_keep)
    old = OD.prune_nm(W, st, 2, 4, ties="lowest", **kw)
    differing = [r for r in tie_rows if not torch.equal(~old[r], ref_keep[r])]
    clean = [r for r in range(W.shape[0]) if r not in set(tie_rows)]
    assert differing and torch.equal(~old[clean], ref_keep[clean])
    print(f"{name}: {len(tie_rows)} rows met an exhausted group; lowest-column-first ended {len(differing)} of them with another mask")


@pytest.mark.gpu
@pytest.mark.parametrize("name", DSNOT_CASES)
@pytest.mark.parametrize("lists", ["1", "0"])
def test_dsnot_kernels_reproduce_the_recorded_walk(name, lists, monkeypatch):
    """both kernels: the list kernel (default) and the per-cycle arg-min kernel (VLMC_DSNOT_LISTS=0)"""
    from vlmc import dsnot
    monkeypatch.setenv("VLMC_DSNOT_LISTS", lists)
    dev = "cuda:0"
    W, xs, ost, kw = _dsnot_inputs(name)
    st = dsnot.DsnotInputStat(W.shape[1], dev)
    for x in xs:
        st.add_call(x[None].to(dev))
    st.finalize()
    Wd = W.clone().to(dev)
    keep = dsnot.prune_linear(Wd, st, 0.5, prune_n=2, prune_m=4, **kw)
    assert torch.equal(keep.cpu(), TIES[f"{name}/mask"])
