"""GPU parity of SparseGPT: the fused column sweep (C ABI vlmc_sparsegpt_sweep) is bit-exact
against the oracle given the same factor; the whole `fasterprune` (library Cholesky / GEMMs on the
GPU) matches the reference's golden vectors within BASELINE.json's bar: identical masks up to
near-ties, updated fp32 weights within 1e-3 relative."""
import numpy as np
import pytest
import torch
import torch.nn as nn

import golden_io
from oracle import sparsegpt as OS

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
G = golden_io.load("sparsegpt")
CASES = sorted({k.split("/")[0] for k in G})


def _rel_err(got, ref):
    got, ref = got.float(), ref.float()
    return float((got - ref).norm() / ref.norm().clamp_min(1e-30))


@pytest.mark.parametrize("count,rows,nm", [(128, 37, (0, 0)), (128, 300, (2, 4)), (96, 64, (4, 8)), (128, 5, (1, 4)),
                                           (40, 33, (0, 0)), (128, 1030, (0, 0))])
def test_sweep_kernel_bit_exact_vs_oracle(count, rows, nm):
    from vlmc import sparsegpt as SG
    g = torch.Generator().manual_seed(count + rows)
    W1 = torch.randn(rows, count, generator=g) * 0.05
    W1[torch.rand(rows, count, generator=g) < 0.1] = 0
    A = torch.randn(count, count * 2, generator=g)
    U = torch.linalg.cholesky(A @ A.t() / count + 0.1 * torch.eye(count), upper=True)
    n, m = nm
    mask1 = OS.block_mask_unstructured(W1, torch.diag(U), 0.5) if n == 0 else torch.zeros_like(W1) == 1
    Q, Err, mask_ref = OS.sweep_block(W1.clone(), U, mask1.clone(), n, m)
    Wd = torch.zeros(rows, count + 7, device=DEV)              # strided: block inside a wider matrix
    Wd[:, 3:3 + count] = W1.to(DEV)
    Ud = torch.zeros(count + 5, count + 5, device=DEV)
    Ud[2:2 + count, 2:2 + count] = U.to(DEV)
    err = torch.empty(rows, count, device=DEV)
    mout = torch.zeros(rows, count + 7, dtype=torch.bool, device=DEV)
    # pointers at the block's first column / diagonal element, like fasterprune does
    SG.sweep_block(Wd[:, 3:], 0, count, Ud[2:, 2:], mask1.to(DEV).contiguous() if n == 0 else None, n, m, err, mout[:, 3:])
    assert torch.equal(Wd[:, 3:3 + count].cpu(), Q)
    assert torch.equal(err.cpu(), Err)
    assert torch.equal(mout[:, 3:3 + count].cpu(), mask_ref)


@pytest.mark.parametrize("nm", [(2, 4), (4, 8), (1, 4), (3, 4)])
def test_sweep_kernel_decides_tied_groups_like_torch_topk_on_the_cpu(nm):
    """sparsegpt_pruner.py:190-192 picks the n columns of a group with `torch.topk(tmp, prune_n, dim=1, largest=False)`; equal scores
    (exact zeros of an already sparse weight, before the first compensation touches them: the first group of a block) are decided
    by the CPU kernel's nth_element order, restated in oracle/topk_order.py and csrc/topk_order.hpp.  60 % zeros: about half of the
    rows tie in their first group."""
    from vlmc import sparsegpt as SG
    n, m = nm
    rows, count = 2000, 128
    g = torch.Generator().manual_seed(7 + n + m)
    W1 = torch.randn(rows, count, generator=g) * 0.05
    W1[torch.rand(rows, count, generator=g) < 0.6] = 0
    A = torch.randn(count, count * 2, generator=g)
    U = torch.linalg.cholesky(A @ A.t() / count + 0.1 * torch.eye(count), upper=True)
    tmp = W1[:, :m] ** 2 / torch.diag(U)[:m].reshape(1, -1) ** 2
    srt = torch.sort(tmp, dim=1)[0]
    tied = srt[:, n - 1] == srt[:, n]
    assert int(tied.sum()) > 200
    want = torch.zeros(rows, m, dtype=torch.bool).scatter_(1, torch.topk(tmp, n, dim=1, largest=False)[1], True)    # the reference's op
    Q, Err, mask_ref = OS.sweep_block(W1.clone(), U, torch.zeros_like(W1) == 1, n, m)
    assert torch.equal(mask_ref[:, :m], want), "the oracle's tie order is not this container's torch.topk"
    stable = torch.zeros(rows, m, dtype=torch.bool).scatter_(1, torch.sort(tmp, dim=1, stable=True)[1][:, :n], True)
    assert (stable != want).any(dim=1).sum() > 50                          # (the old rule decided these rows differently)
    Wd = W1.to(DEV).contiguous()
    err = torch.empty(rows, count, device=DEV)
    mout = torch.zeros(rows, count, dtype=torch.bool, device=DEV)
    SG.sweep_block(Wd, 0, count, U.contiguous().to(DEV), None, n, m, err, mout)
    assert torch.equal(mout.cpu(), mask_ref) and torch.equal(Wd.cpu(), Q) and torch.equal(err.cpu(), Err)


@pytest.mark.parametrize("name", CASES)
def test_fasterprune_matches_reference_golden(name):
    from vlmc import sparsegpt as SG
    W, xs = G[f"{name}/W"], G[f"{name}/xs"]
    xs = xs.to(W.dtype)                                      # fp32 cases are stored as (exactly representable) fp16
    lin = nn.Linear(W.shape[1], W.shape[0], bias=False)
    lin.weight.data = W.clone()
    lin = lin.to(DEV)
    sg = SG.SparseGPT(lin)
    for x in xs:
        sg.add_batch(x[None].to(DEV), None)
    assert sg.nsamples == xs.shape[0]
    if f"{name}/H" in G:
        assert _rel_err(sg.H.cpu(), G[f"{name}/H"]) < 1e-5
    pruned = SG.fasterprune(lin, sg.H, float(G[f"{name}/sparsity"]), int(G[f"{name}/n"]), int(G[f"{name}/m"]), return_mask=True)
    ref = G[f"{name}/Wn"]
    got = lin.weight.data.cpu()
    assert got.dtype == ref.dtype
    zero_agree = ((got == 0) == (ref == 0)).float().mean().item()
    assert zero_agree >= 0.995, f"mask agreement {zero_agree}"
    tol = 2e-2 if name.endswith("rankdef") else 1e-3        # a rank-3 Hessian lives on its damping
    lowp = 1.0 if ref.dtype == torch.float32 else 8.0       # + rounding of the stored dtype
    assert _rel_err(got, ref) < tol * lowp, _rel_err(got, ref)
    assert lin.weight.importance_score == pytest.approx(float(G[f"{name}/imp"]), rel=1e-2 if name.endswith("rankdef") else 1e-4)
    assert bool((got[pruned.cpu()] == 0).all())
    if int(G[f"{name}/n"]):
        n, m = int(G[f"{name}/n"]), int(G[f"{name}/m"])
        assert bool((pruned.view(W.shape[0], -1, m).sum(-1) == n).all())


def test_fasterprune_larger_layer_vs_oracle():
    """[256, 512] fp32, 4 blocks, 2:4 and unstructured, against the CPU oracle on the same Hessian."""
    from vlmc import sparsegpt as SG
    g = torch.Generator().manual_seed(3)
    W = torch.randn(256, 512, generator=g) * 0.05
    X = torch.randn(2048, 512, generator=g) + 0.1
    H = (2 / 2048) * X.t() @ X
    for n, m, sp in [(0, 0, 0.5), (2, 4, 0.5)]:
        want, imp, pruned = OS.prune(W, H.clone(), sp, n, m)
        lin = nn.Linear(512, 256, bias=False)
        lin.weight.data = W.clone()
        lin = lin.to(DEV)
        got_mask = SG.fasterprune(lin, H.clone().to(DEV), sp, n, m, return_mask=True)
        agree = got_mask.cpu() == pruned
        assert agree.float().mean().item() >= 0.999                     # near-ties at the block thresholds may flip
        got = lin.weight.data.cpu()
        # rows without a flipped entry must agree to 1e-3; a flip legitimately re-routes the
        # compensation of its whole row
        clean = agree.all(dim=1)
        assert clean.float().mean().item() > 0.8
        assert _rel_err(got[clean], want[clean]) < 1e-3
        assert lin.weight.importance_score == pytest.approx(imp, rel=1e-4)


# ---- blocked Cholesky (vlmc_chol_block + library GEMMs) --------------------------------------------------
@pytest.mark.parametrize("n", [1, 7, 128, 129, 200, 1408, 2048])
@pytest.mark.parametrize("upper", [False, True])
def test_blocked_cholesky_matches_library_factorization(n, upper):
    from vlmc import sparsegpt
    g = torch.Generator().manual_seed(n)
    X = torch.randn(max(2 * n, 64), n, generator=g)
    H = (X.t() @ X / X.shape[0] + 0.05 * torch.eye(n)).to("cuda:0")
    F, info = sparsegpt.blocked_cholesky(H, upper=upper)
    assert int(info.item()) == 0
    ref = torch.linalg.cholesky(H.double(), upper=upper)
    assert float((F.double() - ref).abs().max() / ref.abs().max()) < 2e-5
    rec = (F.t() @ F) if upper else (F @ F.t())
    assert float((rec - H).abs().max() / H.abs().max()) < 1e-5
    tri = torch.triu(F, 1) if not upper else torch.tril(F, -1)
    assert float(tri.abs().max()) == 0.0 if n > 1 else True


def test_blocked_cholesky_reports_the_failing_column_like_lapack():
    from vlmc import sparsegpt
    n = 300
    g = torch.Generator().manual_seed(5)
    X = torch.randn(2 * n, n, generator=g)
    H = (X.t() @ X / X.shape[0] + 0.05 * torch.eye(n))
    H[200, 200] = -1.0                                       # breaks positive definiteness at column 200 (0-based)
    _, info = sparsegpt.blocked_cholesky(H.to("cuda:0"))
    _, ref_info = torch.linalg.cholesky_ex(H)
    assert int(info.item()) == int(ref_info.item()) == 201
    Hn = H.clone()
    Hn[5, 5] = float("nan")
    _, info = sparsegpt.blocked_cholesky(Hn.to("cuda:0"))
    assert int(info.item()) == 6


@pytest.mark.parametrize("n", [64, 128, 200, 1408, 2048])
def test_direct_inverse_factor_equals_the_reference_chain(n):
    """U with U^T U = H^-1 from one factorization of the index-reversed Hessian vs the reference's
    cholesky -> cholesky_inverse -> cholesky(upper) in fp64."""
    from vlmc import sparsegpt
    g = torch.Generator().manual_seed(n)
    X = torch.randn(max(4 * n, 256), n, generator=g) + 0.2
    H = (2 * X.t() @ X / X.shape[0]).to("cuda:0")
    U, info = sparsegpt.inverse_upper_factor(H.clone())
    assert int(info.item()) == 0
    Hd = H.double()
    ref = torch.linalg.cholesky(torch.cholesky_inverse(torch.linalg.cholesky(Hd)), upper=True)
    assert float(torch.tril(U, -1).abs().max()) == 0.0 if n > 1 else True
    assert float((U.double() - ref).abs().max() / ref.abs().max()) < 5e-4
    eye = U.double().t() @ U.double() @ Hd
    assert float((eye - torch.eye(n, dtype=torch.float64, device="cuda:0")).abs().max()) < 5e-3
    # a second call reuses the captured graph and static buffers
    U2, _ = sparsegpt.inverse_upper_factor(H.clone())
    assert torch.equal(U, U2)


def test_direct_inverse_factor_flags_a_non_pd_hessian_and_the_damping_loop_recovers():
    from vlmc import sparsegpt
    n = 256
    g = torch.Generator().manual_seed(1)
    X = torch.randn(3, n, generator=g)
    H = (X.t() @ X).to("cuda:0")                                # rank 3
    _, info = sparsegpt.inverse_upper_factor(H.clone())
    assert int(info.item()) != 0
    U, dead = sparsegpt.factorize(H.clone())                   # falls back to the reference's chain and its damping loops
    assert not bool(torch.isnan(U).any()) and not bool(dead.any())
    Hd = H.clone()
    ref_L = sparsegpt._chol_with_damping(Hd, 0.01 * torch.mean(torch.diag(Hd)), upper=False)
    Hi = torch.cholesky_inverse(ref_L)
    ref_U = sparsegpt._chol_with_damping(Hi, 0.01 * torch.mean(torch.diag(Hi).abs()), upper=True)
    assert torch.equal(U, ref_U.contiguous())


def test_hessians_of_a_block_factorized_together_equal_one_by_one(monkeypatch):
    """`factorize_many`: the 4-7 independent Cholesky chains of a block on streams of their own, equal sizes in separate
    buffer slots, one host check for all (sparsegpt_pruner.py:112-150 runs them one after the other).  Clean Hessians: the
    very factors and dead masks of `factorize`, bit for bit.  Hessians the reference would damp (not positive definite; fewer
    rows than columns): the attempts H + k damp I, k = 0, 1, 2 run side by side and the first clean one is the upper factor
    of (H + k damp I)^-1 -- the matrix the reference's damp / invert / factorize chain arrives at (same k, other roundings)."""
    from vlmc import sparsegpt as SG
    g = torch.Generator(device=DEV).manual_seed(3)

    def hess(n, rows, dead=None):
        x = torch.randn(rows, n, generator=g, device=DEV)
        if dead is not None:
            x[:, dead] = 0
        return (x.t() @ x) * (2.0 / rows)
    specs = [(256, 1024, None), (384, 2048, None), (256, 700, 5), (256, 1024, None), (640, 4096, None), (256, 1024, None),
             (256, 100, None)]                                                  # the last one: rows < columns (singular)
    Hs = [hess(*s) for s in specs]
    bad = hess(256, 1024)
    bad[7, 7] = -1.0                                                            # not positive definite
    Hs.append(bad)
    rows_seen = [s[1] for s in specs] + [1024]
    one = []
    s0 = dict(SG.factor_stats)
    for H, r in zip(Hs, rows_seen):
        one.append(SG.factorize(H.clone(), 0.01, rows_seen=r))
    s1 = dict(SG.factor_stats)
    assert s1["chain"] - s0["chain"] == 2 and s1["direct"] - s0["direct"] == 6
    caches = [{"rows_seen": r} for r in rows_seen]
    history = {}
    SG.factorize_many([(H.clone(), c) for H, c in zip(Hs, caches)], percdamp=0.01, history=history)
    s2 = dict(SG.factor_stats)
    # the singular one comes out clean after ONE damping step; the one with a -1 on its diagonal needs ~50 (0.01 mean(diag)
    # each): after four attempts it is left to the reference's own loop
    assert s2["direct"] - s1["direct"] == 6 and s2["damped"] - s1["damped"] == 1 and s2["chain"] - s1["chain"] == 1
    assert [history[i] for i in range(8)] == [0, 0, 0, 0, 0, 0, 1, 3]
    for i, ((U, dead), c) in enumerate(zip(one, caches)):
        assert torch.equal(c["dead"], dead)
        if i != 6:
            assert torch.equal(c["U"], U)                                        # clean, or left to the reference's loop: bit-identical
        else:
            Hd = Hs[i].double().clone()
            Hd.diagonal().add_(0.01 * float(torch.diag(Hs[i]).double().mean()))   # k = 1
            got = c["U"].double()
            assert float((got.t() @ got @ Hd - torch.eye(Hd.shape[0], device=DEV, dtype=torch.float64)).abs().max()) < 2e-3
            assert float((c["U"] - U).norm() / U.norm()) < 2e-3                  # and it tracks the three-step chain
            assert bool((torch.tril(c["U"], -1) == 0).all())
    assert int(one[2][1].sum()) == 1
    # the next block: inputs that needed no damping are tried undamped only; same results
    caches3 = [{"rows_seen": r} for r in rows_seen]
    SG.factorize_many([(H.clone(), c) for H, c in zip(Hs, caches3)], percdamp=0.01, history=history)
    for a, b in zip(caches, caches3):
        assert torch.equal(a["U"], b["U"])
    # an input that was clean before and is not now: the damped attempts follow in a second round
    caches4 = [{"rows_seen": 100}]
    SG.factorize_many([(Hs[6].clone(), caches4[0])], percdamp=0.01, history={0: 0})
    assert torch.equal(caches4[0]["U"], caches[6]["U"])
    monkeypatch.setenv("VLMC_SGPT_CONCURRENT", "0")
    caches2 = [{"rows_seen": r} for r in rows_seen]
    SG.factorize_many([(H.clone(), c) for H, c in zip(Hs, caches2)], percdamp=0.01)
    for (U, dead), b in zip(one, caches2):
        assert torch.equal(U, b["U"])                                            # switched off: the one-by-one route


@pytest.mark.parametrize("nm", [(0, 0), (2, 4)])
def test_linears_sharing_a_factor_swept_together_equal_their_own_sweeps(nm):
    """`fasterprune_group`: q / k / v (one Hessian, one factor) stacked along the rows and swept in ONE loop, with the per-block
    threshold of the unstructured rule kept per linear (sparsegpt_pruner.py:183-185).  Sweep, compensation and trailing update
    are row-wise: every linear ends with the weights of its own `fasterprune` (the trailing GEMM runs on more rows: allowed to
    differ in the last bit of a few elements, nothing else), the zero pattern and importance scores included."""
    import torch.nn as nn
    from vlmc import sparsegpt as SG
    g = torch.Generator(device=DEV).manual_seed(11)
    n, rows_x = 512, 4096
    x = torch.randn(rows_x, n, generator=g, device=DEV)
    H = (x.t() @ x) * (2.0 / rows_x)
    outs, spars = (384, 256, 640), (0.5, 0.5, 0.3)
    def fresh():
        torch.manual_seed(5)
        return [nn.Linear(n, o, bias=False).to(DEV).to(torch.bfloat16) for o in outs]
    single, grouped = fresh(), fresh()
    cache = {"rows_seen": rows_x}
    SG.factorize_many([(H.clone(), cache)])
    s1, s2 = [], []
    for lin, sp in zip(single, spars):
        SG.fasterprune(lin, None, sp, prune_n=nm[0], prune_m=nm[1], factor_cache=cache, score_sink=s1)
    SG.fasterprune_group(grouped, list(spars), cache, prune_n=nm[0], prune_m=nm[1], score_sink=s2)
    for a, b, sp in zip(single, grouped, spars):
        wa, wb = a.weight.data.float(), b.weight.data.float()
        assert wa.shape == wb.shape and a.weight.dtype == b.weight.dtype == torch.bfloat16
        assert float(((wa == 0) != (wb == 0)).float().mean()) < 1e-4                       # the same zero pattern
        assert float((wa - wb).norm() / wa.norm()) < 1e-3
        if nm[0]:
            assert bool(((wb == 0).view(wb.shape[0], -1, 4).sum(-1) >= 2).all())
        else:
            assert abs(float((wb == 0).float().mean()) - sp) < 0.01
    for (_, x1), (_, x2) in zip(s1, s2):
        assert float((x1 - x2).abs() / x1.abs()) < 1e-6


def _select_sweep_case(rows_per_scope, count, spars, seed, ties=False, nans=False):
    from vlmc import sparsegpt as SG
    g = torch.Generator().manual_seed(seed)
    rows = sum(rows_per_scope)
    W1 = torch.randn(rows, count, generator=g) * 0.05
    if ties:                                                     # a few distinct magnitudes: the threshold sits in a flood of ties
        W1 = (W1 * 40).round() / 40
    W1[torch.rand(rows, count, generator=g) < 0.1] = 0
    if nans:
        W1[1, 3] = float("nan")
        W1[rows - 1, count - 1] = float("inf")
    A = torch.randn(count, count * 2, generator=g)
    U = torch.linalg.cholesky(A @ A.t() / count + 0.1 * torch.eye(count), upper=True)
    Wd = torch.zeros(rows, count + 9, device=DEV)
    Wd[:, 5:5 + count] = W1.to(DEV)
    Ud = torch.zeros(count + 5, count + 5, device=DEV)
    Ud[2:2 + count, 2:2 + count] = U.to(DEV)
    # the route it replaces: scores and library sort on the GPU (bit-identical scores), the sweep kernel with the mask handed in
    Wref = Wd.clone()
    d = torch.diag(Ud[2:2 + count, 2:2 + count])
    masks, ranks, r0 = [], [], 0
    for r, sp in zip(rows_per_scope, spars):
        tmp = Wref[r0:r0 + r, 5:5 + count] ** 2 / d.reshape(1, -1) ** 2
        rank = min(int(tmp.numel() * sp), tmp.numel() - 1)
        thresh = torch.sort(tmp.flatten())[0][rank]
        masks.append(tmp <= thresh)
        ranks.append(rank)
        r0 += r
    mask1 = torch.cat(masks, 0).contiguous()
    err_ref = torch.empty(rows, count, device=DEV)
    mref = torch.zeros(rows, count + 9, dtype=torch.bool, device=DEV)
    SG.sweep_block(Wref[:, 5:], 0, count, Ud[2:, 2:], mask1, 0, 0, err_ref, mref[:, 5:])
    return SG, Wd, Ud, ranks, Wref, err_ref, mref


@pytest.mark.parametrize("rows_per_scope,count,spars", [
    ((8,), 128, (0.5,)), ((37,), 96, (0.3,)), ((2052,), 128, (0.5,)), ((2051, 33), 128, (0.5, 0.5)), ((6144,), 128, (0.5,)),
    ((1408, 1408, 1408), 128, (0.5, 0.5, 0.5)), ((384, 256, 640), 128, (0.5, 0.5, 0.3)), ((5120, 5120), 128, (0.5, 0.4)),
    ((4, 4, 4, 4), 40, (0.0, 0.5, 0.99, 1.0))])
@pytest.mark.parametrize("kind", ["random", "ties", "nan"])
def test_select_sweep_one_launch_equals_threshold_then_sweep(rows_per_scope, count, spars, kind):
    """`vlmc_sparsegpt_select_sweep` (threshold of sparsegpt_pruner.py:183-185 per stacked linear + the sweep of :186-205 in
    one launch of co-resident workgroups) against the route it replaces: scores, `torch.sort`, `tmp <= thresh`, then
    `vlmc_sparsegpt_sweep` with that mask.  Weights, errors and masks bit for bit; the workspace comes back zero."""
    SG, Wd, Ud, ranks, Wref, err_ref, mref = _select_sweep_case(rows_per_scope, count, spars, seed=count + sum(rows_per_scope),
                                                                 ties=kind == "ties", nans=kind == "nan")
    rows = sum(rows_per_scope)
    err = torch.empty(rows, count, device=DEV)
    mout = torch.zeros(rows, count + 9, dtype=torch.bool, device=DEV)
    SG.select_sweep_block(Wd[:, 5:], 0, count, Ud[2:, 2:], list(rows_per_scope), ranks, err, mout[:, 5:])
    torch.cuda.synchronize()
    assert torch.equal(mout, mref)
    assert torch.equal(Wd.view(torch.int32), Wref.view(torch.int32))
    assert torch.equal(err.view(torch.int32), err_ref.view(torch.int32))
    assert all(int(ws.abs().sum()) == 0 for ws in SG._select_ws.values()) and SG._select_ws      # (one per device and stream)


@pytest.mark.parametrize("level", [1, 2, 3])
def test_select_sweep_failed_grid_barrier_is_finished_by_the_last_workgroup(level, monkeypatch):
    """A grid barrier that fails (workgroups not all resident: CUs held by another stream) fails for every workgroup, nothing
    has been written, and the launch's last workgroup does thresholds and sweeps alone: same bits, workspace zero again."""
    monkeypatch.setenv("VLMC_SGPT_SELECT_FORCE_FAIL", str(level))
    SG, Wd, Ud, ranks, Wref, err_ref, mref = _select_sweep_case((384, 256, 640), 128, (0.5, 0.5, 0.3), seed=level, ties=level == 2)
    err = torch.empty(1280, 128, device=DEV)
    mout = torch.zeros(1280, 128 + 9, dtype=torch.bool, device=DEV)
    SG.select_sweep_block(Wd[:, 5:], 0, 128, Ud[2:, 2:], [384, 256, 640], ranks, err, mout[:, 5:])
    torch.cuda.synchronize()
    assert torch.equal(mout, mref)
    assert torch.equal(Wd.view(torch.int32), Wref.view(torch.int32))
    assert torch.equal(err.view(torch.int32), err_ref.view(torch.int32))
    assert all(int(ws.abs().sum()) == 0 for ws in SG._select_ws.values()) and SG._select_ws      # (one per device and stream)
    monkeypatch.delenv("VLMC_SGPT_SELECT_FORCE_FAIL")
    SG2, Wd2, Ud2, ranks2, Wref2, err_ref2, mref2 = _select_sweep_case((384, 256, 640), 128, (0.5, 0.5, 0.3), seed=level)
    SG.select_sweep_block(Wd2[:, 5:], 0, 128, Ud2[2:, 2:], [384, 256, 640], ranks2, err, None)        # the next call finds it usable
    assert torch.equal(Wd2.view(torch.int32), Wref2.view(torch.int32))


def test_select_sweep_rejects_what_it_cannot_hold():
    from vlmc import _lib, sparsegpt as SG
    W = torch.zeros(20000, 128, device=DEV)
    U = torch.eye(128, device=DEV)
    err = torch.empty(20000, 128, device=DEV)
    assert not SG.select_sweep_usable([20000]) and not SG.select_sweep_usable([4] * 5) and SG.select_sweep_usable([6])
    with pytest.raises(_lib.VlmcError):
        SG.select_sweep_block(W, 0, 128, U, [20000], [5], err)
    with pytest.raises(_lib.VlmcError):
        SG.select_sweep_block(W[:6], 0, 128, U, [6], [6 * 128], err[:6])                  # a rank outside the scope


@pytest.mark.parametrize("grouped", [False, True])
def test_fasterprune_with_one_launch_blocks_equals_the_multi_launch_route(grouped, monkeypatch):
    """`fasterprune` / `fasterprune_group` through `vlmc_sparsegpt_select_sweep` (default) against `VLMC_SGPT_SELECT_SWEEP=0`
    (scores, multi-tensor radix select, logical_not, sweep with the mask handed in): the same weights bit for bit."""
    import torch.nn as nn
    from vlmc import sparsegpt as SG
    g = torch.Generator(device=DEV).manual_seed(23)
    n, rows_x = 640, 4096
    x = torch.randn(rows_x, n, generator=g, device=DEV)
    H = (x.t() @ x) * (2.0 / rows_x)
    outs, spars = ((384, 256, 644), (0.5, 0.5, 0.3)) if grouped else ((1030,), (0.45,))

    def run(fused):
        monkeypatch.setattr(SG, "_SELECT_SWEEP", fused)
        torch.manual_seed(5)
        lins = [nn.Linear(n, o, bias=False).to(DEV).to(torch.bfloat16) for o in outs]
        cache = {"rows_seen": rows_x}
        SG.factorize_many([(H.clone(), cache)])
        sink = []
        if grouped:
            SG.fasterprune_group(lins, list(spars), cache, score_sink=sink)
        else:
            SG.fasterprune(lins[0], None, spars[0], factor_cache=cache, score_sink=sink)
        return [l.weight.data.clone() for l in lins]
    a, b = run(True), run(False)
    for wa, wb, sp in zip(a, b, spars):
        assert torch.equal(wa.view(torch.int16), wb.view(torch.int16))
        assert abs(float((wa == 0).float().mean()) - sp) < 0.01


# ---- round 4: the damping ROUTE held against the reference's own decision and against fp64 ---------------------------------
def test_damping_route_on_the_goldens_is_the_references_and_fp64s():
    """sparsegpt_pruner.py:112-150 damps H only while its Cholesky fails.  On the reference's golden Hessians the route is
    known (tests/test_oracle_golden.py::test_sparsegpt_damping_route_of_the_reference_on_the_rank_deficient_golden: the oracle
    that reproduces the reference's pruned weights bit for bit damps the rank-deficient H ONCE, the clean one never; fp64
    arithmetic decides the same).  `factorize_many` -- attempts k = 0, 1 side by side, the first clean one in k order -- must take
    the same route, through the chain of launches and through the persistent kernel alike, and its factor must be the factor
    of (H + k damp I)^-1."""
    from vlmc import sparsegpt as SG
    for name, want_k in (("fp32_rankdef", 1), ("fp32_u50", 0)):
        H0 = G[f"{name}/H"].clone()
        # fp64: the smallest k for which H + k damp I has a Cholesky factor
        Hd = H0.double()
        damp = 0.01 * Hd.diagonal().mean()
        k64 = 0
        while int(torch.linalg.cholesky_ex(Hd + k64 * damp * torch.eye(Hd.shape[0], dtype=torch.float64))[1]) != 0:
            k64 += 1
        assert k64 == want_k
        trace = {}
        OS.inverse_factor(H0.clone(), G[f"{name}/W"].clone().float(), 0.01, trace)
        assert trace == {"damp_H": want_k, "damp_Hinv": 0}                         # the reference's own route (CPU oracle)
        for persistent in (True, False):
            SG._PERSISTENT = persistent
            try:
                hist, cache = {}, {"rows_seen": 18 if want_k else 288}
                before = dict(SG.factor_stats)
                SG.factorize_many([(H0.clone().to(DEV), cache)], percdamp=0.01, history=hist)
            finally:
                SG._PERSISTENT = True
            assert hist == {0: want_k}, (name, persistent, hist)
            assert SG.factor_stats["chain"] == before["chain"]                       # decided without the fallback chain
            U = cache["U"].double().cpu()
            target = Hd + want_k * damp * torch.eye(Hd.shape[0], dtype=torch.float64)
            assert float((U.t() @ U @ target - torch.eye(Hd.shape[0], dtype=torch.float64)).abs().max()) < 5e-3
            assert bool((torch.tril(cache["U"], -1) == 0).all()) and not bool(cache["dead"].any())


@pytest.mark.parametrize("n", [256, 1408, 2048])
def test_persistent_factorization_equals_the_chain_and_any_workgroup_count(n):
    """`vlmc_chol_inverse` (ONE persistent launch, csrc/chol_persistent.hip) against the chain of launches per 128 columns:
    the same factor up to fp32 summation order, the same accuracy against fp64, bit-identical for any number of workgroups;
    a matrix that is not positive definite is flagged with LAPACK's column (and every workgroup leaves)."""
    from vlmc import sparsegpt as SG
    g = torch.Generator(device=DEV).manual_seed(n)
    X = torch.randn(3 * n, n, generator=g, device=DEV)
    H = (X.t() @ X) / (3 * n) + 0.01 * torch.eye(n, device=DEV)
    outs = {}
    for wgs in (None, 7, 1):
        U, info = SG.inverse_upper_factor(H, max_workgroups=wgs)
        assert int(info) == 0
        outs[wgs] = U
    assert torch.equal(outs[7], outs[None]) and torch.equal(outs[1], outs[None])
    SG._PERSISTENT = False
    try:
        Uc, info = SG.inverse_upper_factor(H)
    finally:
        SG._PERSISTENT = True
    assert float((outs[None] - Uc).norm() / Uc.norm()) < 2e-6
    Hd, Ud = H.double(), outs[None].double()
    assert float((Ud.t() @ Ud @ Hd - torch.eye(n, device=DEV, dtype=torch.float64)).abs().max()) < 5e-5
    bad = H.clone()
    bad[200, 200] = -1.0                          # (reversed inside: the failing column is reported in the reversed order)
    U, info = SG.inverse_upper_factor(bad)
    assert int(info) > 0
    SG._PERSISTENT = False
    try:
        _, info_c = SG.inverse_upper_factor(bad)
    finally:
        SG._PERSISTENT = True
    assert int(info) == int(info_c)


# ---- K10 on fp32 matrix cores (round 6): vlmc_sparsegpt_trailing_update and the look-ahead around it -------------------------------
@pytest.mark.parametrize("rows,ncols,count", [(1, 1, 1), (37, 5, 128), (128, 128, 128), (300, 1000, 96), (2048, 4992, 128), (10240, 1920, 128),
                                              (130, 257, 127)])
def test_trailing_update_matches_fp64(rows, ncols, count):
    """`W[:, i2:] -= Err1.matmul(Hinv[i1:i2, i2:])` (sparsegpt_pruner.py:210): every element one fp32 accumulator over the block's k and one
    subtraction -- against the fp64 product within fp32 accumulation error (north_star: 1e-3 relative on updated fp32 weights; here ~1e-6)."""
    from vlmc import sparsegpt as SG
    g = torch.Generator(device=DEV).manual_seed(rows + ncols + count)
    i1, extra = 64, 40                                                         # the block sits inside a larger factor; W has columns to the left
    U = torch.randn(i1 + count + 8, extra + ncols, generator=g, device=DEV) * 0.3
    W = torch.randn(rows, extra + ncols, generator=g, device=DEV) * 0.05
    err = torch.randn(rows, 128, generator=g, device=DEV) * 0.02
    want = W.double().clone()
    want[:, extra:] -= err[:, :count].double() @ U[i1:i1 + count, extra:].double()
    W0 = W.clone()
    SG.trailing_update(W, extra, extra + ncols, err, U, i1, i1 + count)
    assert torch.equal(W[:, :extra], W0[:, :extra]), "columns left of the update were touched"
    scale = (err[:, :count].double().abs() @ U[i1:i1 + count, extra:].double().abs()) + W0[:, extra:].double().abs()
    assert bool(((W[:, extra:].double() - want[:, extra:]).abs() <= 2e-6 * scale + 1e-30).all())
    # splitting the columns over launches changes nothing: an element's arithmetic does not depend on what shares its launch
    W2 = W0.clone()
    cut = extra + min(ncols, 128)
    SG.trailing_update(W2, extra, cut, err, U, i1, i1 + count)
    SG.trailing_update(W2, cut, extra + ncols, err, U, i1, i1 + count)
    assert torch.equal(W2, W)


@pytest.mark.parametrize("grouped,nm", [(False, (0, 0)), (True, (0, 0)), (True, (2, 4)), (False, (2, 4))])
def test_block_loop_from_one_call_equals_the_loop_issued_from_python(grouped, nm, monkeypatch):
    """`vlmc_sparsegpt_prune_blocks` (every block's sweep + trailing update from one call) against the same entry points called block
    by block from Python (`VLMC_SGPT_BLOCK_LOOP=0`): identical weights and masks, bit for bit, at a T5 width."""
    from vlmc import sparsegpt as SG
    g = torch.Generator().manual_seed(4)
    in_f = 2048
    layers0 = [nn.Linear(in_f, n, bias=False) for n in ((512, 384) if grouped else (640,))]
    for l in layers0:
        l.weight.data = (torch.randn(l.weight.shape, generator=g) * 0.05)
    X = torch.randn(4096, in_f, generator=g) * (torch.rand(in_f, generator=g) + 0.2)
    res = []
    for loop in (True, False):
        monkeypatch.setattr(SG, "_BLOCK_LOOP", loop)
        layers = [nn.Linear(in_f, l.weight.shape[0], bias=False).to(DEV) for l in layers0]
        for l, l0 in zip(layers, layers0):
            l.weight.data = l0.weight.data.clone().to(DEV)
        H = (X.to(DEV).t() @ X.to(DEV)) * (2.0 / X.shape[0])
        cache = {}
        U, dead = SG.factorize(H, 0.01)
        cache["U"], cache["dead"] = U, dead
        masks = None
        if grouped:
            SG.fasterprune_group(layers, [0.5] * len(layers), cache, prune_n=nm[0], prune_m=nm[1])
        else:
            masks = SG.fasterprune(layers[0], None, 0.5, prune_n=nm[0], prune_m=nm[1], factor_cache=cache, return_mask=True)
        torch.cuda.synchronize()
        res.append(([l.weight.data.clone() for l in layers], masks))
    for a, b in zip(res[0][0], res[1][0]):
        assert torch.equal(a, b)
        assert abs(float((a == 0).float().mean()) - 0.5) < 0.02
    if res[0][1] is not None:
        assert torch.equal(res[0][1], res[1][1])
