"""GPU: seeded random shapes / ranks / tie patterns for the fused select kernels against the CPU oracle
(bit-exact masks and pruned weights), single and batched entry points."""
import numpy as np
import pytest
import torch

from oracle import wanda as OW

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _case(rng, dtype, max_in=3000):
    out_f = int(rng.integers(1, 70))
    in_f = int(rng.choice([int(rng.integers(1, 40)) * 8, int(rng.integers(8, max_in)), 8, 16, 1024, 2048]))
    g = torch.Generator().manual_seed(int(rng.integers(0, 2**31)))
    W = (torch.randn(out_f, in_f, generator=g) * 0.02).to(dtype)
    style = rng.integers(0, 4)
    if style == 1:
        W[torch.rand(out_f, in_f, generator=g) < 0.3] = 0                       # exact-zero scores
    elif style == 2:
        W = W[:, :1].expand(out_f, in_f).clone()                                # every score of a row ties (before scaling)
    elif style == 3:
        W = (torch.randint(-3, 4, (out_f, in_f), generator=g).float() * 0.01).to(dtype)   # few distinct values
    s = (torch.rand(in_f, generator=g) * 4 + 0.01).numpy().astype(np.float32)
    if rng.integers(0, 3) == 0:
        s[:] = s[0]                                                             # constant scale: ties survive scaling
    return W, s


@pytest.mark.parametrize("seed", range(6))
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
def test_row_and_matrix_select_random_cases(seed, dtype):
    from vlmc import ops
    rng = np.random.default_rng(1000 * seed + {torch.bfloat16: 1, torch.float16: 2, torch.float32: 3}[dtype])
    for _ in range(6):
        W, s = _case(rng, dtype)
        out_f, in_f = W.shape
        sq = ops.sqrt_scaler(torch.from_numpy(s).to(DEV))
        for mode in ("row", "matrix"):
            if mode == "row":
                k = int(rng.integers(0, in_f + 1))
                ratio_kw = dict(ratio=k / in_f)
                # the oracle takes a ratio: make int(in * ratio) == k exactly
                while int(in_f * ratio_kw["ratio"]) != k:
                    ratio_kw["ratio"] = np.nextafter(ratio_kw["ratio"], 2.0)
            else:
                k = int(rng.integers(0, W.numel()))
                ratio_kw = dict(ratio=k / W.numel())
                while int(W.numel() * ratio_kw["ratio"]) != k:
                    ratio_kw["ratio"] = np.nextafter(ratio_kw["ratio"], 2.0)
            Wd = W.clone().to(DEV)
            mask, _ = ops.wanda_select(Wd, sq, mode, k=k)
            want = OW.prune_linear(W, s, mode, **ratio_kw)
            assert np.array_equal(mask.cpu().numpy(), want["mask"]), (mode, W.shape, k)
            assert torch.equal(Wd.cpu().view(torch.uint8), want["weight"].view(torch.uint8)), (mode, W.shape, k)


@pytest.mark.parametrize("seed", range(4))
def test_batched_select_random_job_mixes(seed):
    from vlmc import ops
    rng = np.random.default_rng(77 + seed)
    dtype = [torch.bfloat16, torch.float16][seed % 2]
    cases = [_case(rng, dtype, max_in=1200) for _ in range(int(rng.integers(2, 15)))]
    sqs = [ops.sqrt_scaler(torch.from_numpy(s).to(DEV)) for _, s in cases]
    for mode in ("row", "matrix"):
        Ws = [W.clone().to(DEV) for W, _ in cases]
        ks = [int(W.shape[1] * 0.4) if mode == "row" else int(W.numel() * 0.4) for W, _ in cases]
        masks, _ = ops.wanda_select_batch(Ws, sqs, mode, ks=ks)
        for (W, s), Wd, mk in zip(cases, Ws, masks):
            want = OW.prune_linear(W, s, mode, ratio=0.4)
            assert np.array_equal(mk.cpu().numpy(), want["mask"]), (mode, W.shape)
            assert torch.equal(Wd.cpu().view(torch.uint8), want["weight"].view(torch.uint8))


def _wide_case(rng, dtype, in_f):
    """Like `_case` with a given (8-aligned) row width and few rows."""
    out_f = int(rng.integers(1, 24))
    g = torch.Generator().manual_seed(int(rng.integers(0, 2**31)))
    W = (torch.randn(out_f, in_f, generator=g) * 0.02).to(dtype)
    style = rng.integers(0, 4)
    if style == 1:
        W[torch.rand(out_f, in_f, generator=g) < 0.3] = 0
    elif style == 2:
        W = W[:, :1].expand(out_f, in_f).clone()
    elif style == 3:
        W = (torch.randint(-3, 4, (out_f, in_f), generator=g).float() * 0.01).to(dtype)
    s = (torch.rand(in_f, generator=g) * 4 + 0.01).numpy().astype(np.float32)
    if rng.integers(0, 3) == 0:
        s[:] = s[0]
    return W, s


@pytest.mark.parametrize("seed", range(4))
def test_mixed_width_row_select_random_job_mixes(seed):
    """Random mixes of narrow (<= 2048) and wide (<= 8192) 8-aligned rows with per-job ranks: one mixed launch
    (select_rows_mixed_kernel), bit-exact against the oracle."""
    from vlmc import ops
    rng = np.random.default_rng(4242 + seed)
    dtype = [torch.bfloat16, torch.float16][seed % 2]
    n_jobs = int(rng.integers(2, 13))
    widths = [int(rng.integers(1, 257)) * 8 for _ in range(n_jobs)]
    widths[0] = int(rng.integers(257, 1025)) * 8                 # at least one wide and one narrow job
    widths[1] = min(widths[1], 2048)
    cases = [_wide_case(rng, dtype, w) for w in widths]
    sqs = [ops.sqrt_scaler(torch.from_numpy(s).to(DEV)) for _, s in cases]
    Ws = [W.clone().to(DEV) for W, _ in cases]
    ks = [int(rng.integers(0, W.shape[1] + 1)) for W, _ in cases]
    masks, _ = ops.wanda_select_batch(Ws, sqs, "row", ks=ks)
    for (W, s), Wd, mk, k in zip(cases, Ws, masks, ks):
        pruned = OW.select_rows(OW.wanda_score(W, s), k)
        assert np.array_equal(mk.cpu().numpy(), ~pruned), (W.shape, k)
        want = W.clone()
        want[torch.from_numpy(pruned)] = 0
        assert torch.equal(Wd.cpu().view(torch.uint8), want.view(torch.uint8)), (W.shape, k)
