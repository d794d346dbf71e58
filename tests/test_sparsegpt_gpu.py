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
