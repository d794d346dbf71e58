"""GPU parity at BASELINE.json's full layer sizes (FlanT5-XL / ViT-g / Vicuna-7B shapes).

The checker here is the reference's own op sequence (torch.sort(stable) / flatten-sort
threshold / scatter, wanda_pruner.py:318-341, :666-687) executed by PyTorch ON THE GPU on the
same fp32 scores -- bit-exact by construction -- plus the C oracle on row/sample subsets, plus
size-independent properties (exact per-row counts, idempotence, zeroed == ~mask)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _mk(out_f, in_f, dtype, seed, zero_frac=0.0):
    g = torch.Generator(device=DEV).manual_seed(seed)
    W = (torch.randn(out_f, in_f, device=DEV, generator=g) * 0.02).to(dtype)
    if zero_frac:
        W[torch.rand(out_f, in_f, device=DEV, generator=g) < zero_frac] = 0
    s = torch.rand(in_f, device=DEV, generator=g) * 4 + 0.01
    return W, s


def _torch_reference_mask(W, sq, mode, k):
    score = W.abs().float() * sq[None, :]                    # exact fp32 products, same bits as the kernel's
    if mode == "row":
        idx = torch.sort(score, dim=-1, stable=True)[1][:, :k]
        pruned = torch.zeros_like(score, dtype=torch.bool)
        pruned.scatter_(1, idx, True)
    else:
        thr = torch.sort(score.flatten())[0][k]
        pruned = score < thr
    return ~pruned, float(score.double().mean().item())


ROW_SHAPES = [(5120, 2048), (2048, 5120), (2048, 2048), (4096, 4096), (11008, 4096), (4096, 11008)]


@pytest.mark.parametrize("shape", ROW_SHAPES)
@pytest.mark.parametrize("dtype,zero_frac", [(torch.bfloat16, 0.0), (torch.float16, 0.5)])
def test_row_select_full_size_exact(shape, dtype, zero_frac):
    from vlmc import ops
    W, s = _mk(*shape, dtype, seed=shape[0] + shape[1], zero_frac=zero_frac)
    sq = ops.sqrt_scaler(s)
    k = int(shape[1] * 0.5)
    want_mask, want_imp = _torch_reference_mask(W, sq, "row", k)
    W0 = W.clone()
    mask, parts = ops.wanda_select(W, sq, "row", k=k)
    assert torch.equal(mask, want_mask), f"{int((mask != want_mask).sum())} mask entries differ"
    assert torch.equal(W, torch.where(mask, W0, torch.zeros_like(W0)))
    assert int((~mask).sum(1).min()) == k and int((~mask).sum(1).max()) == k       # exactly k pruned per row
    assert float(parts.sum().item()) / W.numel() == pytest.approx(want_imp, rel=1e-5)
    m2, _ = ops.wanda_select(W, sq, "row", k=k)                                     # idempotent
    assert torch.equal(m2, mask)


@pytest.mark.parametrize("shape", [(4224, 1408), (1408, 1408), (6144, 1408), (1408, 6144)])
@pytest.mark.parametrize("ratio", [0.5, 0.3])
def test_matrix_select_full_size_exact(shape, ratio):
    from vlmc import ops
    W, s = _mk(*shape, torch.float16, seed=shape[0] * 3 + shape[1])
    sq = ops.sqrt_scaler(s)
    k = int(W.numel() * ratio)
    want_mask, want_imp = _torch_reference_mask(W, sq, "matrix", k)
    W0 = W.clone()
    mask, parts = ops.wanda_select(W, sq, "matrix", k=k)
    assert torch.equal(mask, want_mask)
    assert torch.equal(W, torch.where(mask, W0, torch.zeros_like(W0)))
    assert float(parts.sum().item()) / W.numel() == pytest.approx(want_imp, rel=1e-5)


def test_nm_select_full_size_properties_and_oracle_rows():
    from vlmc import ops
    from oracle import wanda_c as OC
    W, s = _mk(5120, 2048, torch.bfloat16, seed=5, zero_frac=0.2)
    sq = ops.sqrt_scaler(s)
    W0 = W.clone()
    mask, _ = ops.wanda_select(W, sq, "nm", n=2, m=4)
    assert bool(((~mask).view(5120, 512, 4).sum(-1) == 2).all())                    # exactly 2 of every 4
    assert torch.equal(W, torch.where(mask, W0, torch.zeros_like(W0)))
    if OC.available():
        rows = slice(1000, 1064)
        m_c, W_c, _ = OC.select(W0[rows].cpu(), s.cpu().numpy(), "nm", n=2, m=4)
        assert np.array_equal(mask[rows].cpu().numpy(), m_c)
        assert torch.equal(W[rows].cpu(), W_c)


@pytest.mark.parametrize("shape,dtype", [((128, 257, 1408), torch.float16), ((128, 64, 2048), torch.bfloat16),
                                         ((128, 16, 5120), torch.bfloat16), ((128, 257, 6144), torch.float16)])
def test_stats_full_size_vs_c_oracle_samples(shape, dtype):
    """All 128 calibration samples in one launch; three of them re-checked bit-for-bit by the C
    oracle, and the running mean over all 128 against the oracle's recurrence on the kernel's rows."""
    from vlmc import ops
    from oracle import wanda_c as OC
    if not OC.available():
        pytest.skip("C oracle not built")
    g = torch.Generator(device=DEV).manual_seed(shape[1])
    x = (torch.randn(shape, device=DEV, generator=g) + 0.1).to(dtype)
    nsq = ops.act_sqnorm(x)
    for j in (0, 63, 127):
        assert np.array_equal(nsq[j].cpu().numpy().view(np.uint32), OC.act_sqnorm(x[j].cpu()).view(np.uint32)), j
    s = torch.zeros(shape[2], device=DEV)
    assert ops.wanda_scaler_update(s, 0, nsq, 1) == 128
    want, _ = OC.scaler_update(np.zeros(shape[2], np.float32), 0, nsq.cpu().numpy(), 1)
    assert np.array_equal(s.cpu().numpy().view(np.uint32), want.view(np.uint32))


@pytest.mark.parametrize("name,shape", [("wi_0", (5120, 2048)), ("wo", (2048, 5120))])
def test_baseline_config_0_end_to_end_vs_cpu_oracle(name, shape):
    """BASELINE.json configs[0] exactly as SURVEY.md §8(d) writes it: one FlanT5-XL encoder FFN linear, bf16,
    W ~ N(0, 0.02) seed 0, 8 calibration samples [1, 64, in] ~ N(0.1, 1) seeds 1..8, 50 % per row -- `scaler_row`, mask,
    zeroed weights and importance score against the CPU oracle, bit for bit."""
    from oracle import wanda as OW
    from vlmc import wanda
    out_f, in_f = shape
    W = (torch.randn(out_f, in_f, generator=torch.Generator().manual_seed(0)) * 0.02).to(torch.bfloat16)
    xs = [(torch.randn(1, 64, in_f, generator=torch.Generator().manual_seed(j)) + 0.1).to(torch.bfloat16) for j in range(1, 9)]
    st = wanda.InputStat(in_f, DEV)
    for x in xs:
        st.add_call(x.to(DEV))
    st.finalize()
    s_ref = OW.wanda_stats(xs)
    assert np.array_equal(st.scaler_row.cpu().numpy().view(np.uint32), s_ref.view(np.uint32))
    Wd = W.clone().to(DEV)
    mask, parts = wanda.prune_linear(Wd, st, "row", ratio=0.5)
    want = OW.prune_linear(W, s_ref, "row", ratio=0.5)
    assert np.array_equal(mask.cpu().numpy(), want["mask"])
    assert torch.equal(Wd.cpu(), want["weight"])
    assert float(parts.sum().item()) / W.numel() == pytest.approx(want["importance_score"], rel=1e-6)
