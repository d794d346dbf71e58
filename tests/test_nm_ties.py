"""n:m selection when scores TIE inside an m-group (wanda_pruner.py:326-329, `torch.topk(tmp, n, dim=1, largest=False)`).

Which of several equal scores `torch.topk` returns is implementation-defined (CPU: whatever libstdc++'s nth_element leaves
in front; CUDA: its radix select's order).  `tests/golden/nm_ties.npz` RECORDS the reference's answer on this container's CPU;
the build's rule is "lowest column first" (the stable order of the per-row rule).  What is asserted, for the CPU oracle and for
the GPU kernel:
  * every m-group WITHOUT a tie at the selection boundary has exactly the reference's mask;
  * in every tied group both prune exactly n columns, and the columns they disagree on carry EQUAL scores -- the two masks
    are permutations of each other among equal-score columns, so the pruned score mass, the importance score and the sparsity
    pattern per group are identical;
  * the divergence exists (the goldens do exercise ties) and is counted.
"""
import numpy as np
import pytest
import torch

import golden_io
from oracle import wanda as OW

TIES = golden_io.load("nm_ties")
CASES = sorted({k.split("/")[0] for k in TIES if not k.startswith("dsnot_")})
DSNOT_CASES = sorted({k.split("/")[0] for k in TIES if k.startswith("dsnot_")})


def _split(name):
    W, xs = TIES[f"{name}/W"], TIES[f"{name}/xs"]
    n, m = int(TIES[f"{name}/n"]), int(TIES[f"{name}/m"])
    s = OW.wanda_stats([x[None] for x in xs])
    score = OW.wanda_score(W, s)                                        # fp32 [out, in]: what both sides rank
    g = score.reshape(score.shape[0], -1, m)
    srt = np.sort(g, axis=2)
    tied = srt[:, :, n - 1] == srt[:, :, n]                             # a tie across the selection boundary
    return W, s, score, n, m, tied


def _hold_against_reference(name, keep, score, n, m, tied):
    ref = TIES[f"{name}/mask"].numpy()
    out_f, in_f = ref.shape
    kg, rg, sg = keep.reshape(out_f, -1, m), ref.reshape(out_f, -1, m), score.reshape(out_f, -1, m)
    assert ((~kg).sum(axis=2) == n).all() and ((~rg).sum(axis=2) == n).all()          # n pruned per group, both
    assert np.array_equal(kg[~tied], rg[~tied]), "a group without ties differs from the reference"
    differ = kg != rg
    assert not differ[~tied].any()
    # where they differ, the columns one side prunes and the other keeps carry the same score
    for r, c in zip(*np.nonzero(differ.any(axis=2))):
        mine, theirs = sg[r, c][~kg[r, c]], sg[r, c][~rg[r, c]]
        assert np.array_equal(np.sort(mine), np.sort(theirs)), (name, r, c)
    return int(tied.sum()), int(differ.any(axis=2).sum())


@pytest.mark.parametrize("name", CASES)
def test_oracle_policy_against_the_recorded_torch_answer(name):
    W, s, score, n, m, tied = _split(name)
    got = OW.prune_linear(W, s, "nm", n=n, m=m)
    n_tied, n_diff = _hold_against_reference(name, got["mask"], score, n, m, tied)
    assert n_tied > 20, "the fixture is meant to tie"
    assert n_diff > 0, "torch.topk happened to agree with lowest-column-first everywhere: the fixture pins nothing"
    assert got["importance_score"] == pytest.approx(float(TIES[f"{name}/imp"]), rel=1e-5)
    # the pruned weights are the reference's wherever the masks agree; the zeroed mass is the same
    agree = torch.from_numpy(got["mask"] == TIES[f"{name}/mask"].numpy())
    assert torch.equal(got["weight"][agree], TIES[f"{name}/Wn"][agree])
    print(f"{name}: {n_tied} tied groups, {n_diff} decided differently from torch's CPU topk (equal scores)")


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_kernel_policy_is_the_oracles_and_is_held_against_torch(name):
    from vlmc import ops
    dev = "cuda:0"
    W, s, score, n, m, tied = _split(name)
    want = OW.prune_linear(W, s, "nm", n=n, m=m)
    Wd = W.clone().to(dev)
    sq = ops.sqrt_scaler(torch.from_numpy(s).to(dev))
    mask, parts = ops.wanda_select(Wd, sq, "nm", n=n, m=m)
    keep = mask.cpu().numpy()
    assert np.array_equal(keep, want["mask"]), "kernel and oracle disagree on a tie"      # lowest column first, both
    assert torch.equal(Wd.cpu(), want["weight"])
    _hold_against_reference(name, keep, score, n, m, tied)


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
def test_dsnot_exhausted_group_policy_against_the_recorded_reference_walk(name):
    """When the regrow walk returns to an m-group whose two kept entries were both swapped out already, both sit at
    rowmax + 1 and `torch.topk(pruning_block, 1, largest=False)` picks one of two equal values -- implementation-defined.
    Rows that never meet such a group reproduce the reference's recorded mask exactly; rows that do are decided lowest
    column first here, and the fixture shows the reference's CPU run deciding some of them otherwise."""
    from oracle import dsnot as OD
    W, xs, st, kw = _dsnot_inputs(name)
    trace = {}
    pruned = OD.prune_nm(W, st, 2, 4, trace=trace, **kw)
    ref_keep = TIES[f"{name}/mask"]
    tie_rows = sorted(trace.get("tie_rows", ()))
    clean = [r for r in range(W.shape[0]) if r not in set(tie_rows)]
    assert len(tie_rows) >= 3 and len(clean) >= 3, (len(tie_rows), len(clean))
    assert torch.equal(~pruned[clean], ref_keep[clean]), "a row that never met an exhausted group differs from the reference"
    differing = [r for r in tie_rows if not torch.equal(~pruned[r], ref_keep[r])]
    print(f"{name}: {len(tie_rows)} rows met an exhausted group, {len(differing)} of them end with another mask than torch's CPU topk gives")
    assert differing, "torch.topk happened to pick the lowest column in every exhausted group: the fixture pins nothing"


@pytest.mark.gpu
@pytest.mark.parametrize("name", DSNOT_CASES)
def test_dsnot_kernel_decides_exhausted_groups_like_the_oracle(name):
    from oracle import dsnot as OD
    from vlmc import dsnot
    dev = "cuda:0"
    W, xs, ost, kw = _dsnot_inputs(name)
    want = OD.prune_nm(W, ost, 2, 4, **kw)
    st = dsnot.DsnotInputStat(W.shape[1], dev)
    for x in xs:
        st.add_call(x[None].to(dev))
    st.finalize()
    Wd = W.clone().to(dev)
    keep = dsnot.prune_linear(Wd, st, 0.5, prune_n=2, prune_m=4, **kw)
    assert torch.equal(keep.cpu(), ~want)
