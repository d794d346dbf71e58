"""GPU parity: HIP kernels (through the C ABI) vs the CPU oracle and the reference's
golden vectors.  Bit-exact for masks, statistics and pruned weights; importance_score
(a mean of fp32 scores) at rtol 1e-5."""
import numpy as np
import pytest
import torch

import golden_io
from oracle import wanda as OW

pytestmark = pytest.mark.gpu
UNIT = golden_io.load("wanda_unit")
DEV = "cuda:0"


def _ops():
    from vlmc import ops
    return ops


def _cases(prefix):
    return sorted({k.split("/")[1] for k in UNIT if k.startswith(prefix + "/")})


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


# ------------------------------------------------------------------ statistics --
def test_ieee_sqrt_div_fma_on_device():
    """The contract rests on correctly rounded sqrt/div/fma on the GPU: check the three
    through the stats kernels on adversarial magnitudes (incl. subnormals)."""
    ops = _ops()
    rng = np.random.default_rng(1)
    x = (rng.standard_normal((1, 1, 1 << 16)) * 10.0 ** rng.integers(-22, 18, (1, 1, 1 << 16))).astype(np.float32)
    got = ops.act_sqnorm(torch.from_numpy(x).to(DEV)).cpu().numpy()[0]
    r = np.sqrt((x[0, 0] * x[0, 0]).astype(np.float32), dtype=np.float32)
    assert np.array_equal(_bits(got), _bits(r * r))
    s = torch.from_numpy(np.abs(x[0, 0]).copy()).to(DEV)
    nsq = torch.from_numpy(np.abs(x[0, 0, ::-1]).copy()).to(DEV)[None]
    ops.wanda_scaler_update(s, 6, nsq, 1)
    want, _ = OW.scaler_update(np.abs(x[0, 0]), 6, np.abs(x[0, 0, ::-1]), 1)
    assert np.array_equal(_bits(s.cpu().numpy()), _bits(want))


@pytest.mark.parametrize("name", _cases("g1"))
def test_stats_match_reference_golden(name):
    ops = _ops()
    n, b = int(UNIT[f"g1/{name}/n"]), int(UNIT[f"g1/{name}/b"])
    states = UNIT[f"g1/{name}/states"].numpy()
    s = torch.zeros(states.shape[1], dtype=torch.float32, device=DEV)
    ns = 0
    for j in range(n):                                   # hook-style: one call at a time
        x = UNIT[f"g1/{name}/x{j}"].to(DEV)
        nsq = ops.act_sqnorm(x.reshape(1, -1, x.shape[-1]))
        ns = ops.wanda_scaler_update(s, ns, nsq, b)
        assert np.array_equal(_bits(s.cpu().numpy()), _bits(states[j])), f"call {j}"
    # batched: all calls in one launch, recurrence in one launch
    xs = torch.stack([UNIT[f"g1/{name}/x{j}"].reshape(-1, states.shape[1]) for j in range(n)]).to(DEV)
    s2 = torch.zeros_like(s)
    assert ops.wanda_scaler_update(s2, 0, ops.act_sqnorm(xs), b) == n * b
    assert np.array_equal(_bits(s2.cpu().numpy()), _bits(states[-1]))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
@pytest.mark.parametrize("shape", [(3, 17, 40), (2, 64, 2048), (5, 257, 1408), (1, 1, 8), (4, 33, 1001), (130, 16, 256)])
def test_sqnorm_vs_oracle(dtype, shape):
    ops = _ops()
    g = torch.Generator().manual_seed(hash((str(dtype), shape)) % 1000)
    x = ((torch.randn(shape, generator=g) + 0.1) * 3).to(dtype)
    got = ops.act_sqnorm(x.to(DEV)).cpu().numpy()
    for c in range(shape[0]):
        assert np.array_equal(_bits(got[c]), _bits(OW.act_sqnorm(x[c]))), f"call {c}"


def test_sqnorm_strided_rows_and_misaligned_base():
    ops = _ops()
    g = torch.Generator().manual_seed(5)
    big = (torch.randn(4, 20, 200, generator=g)).to(torch.bfloat16).to(DEV)
    view = big[:, :, 3:131]                              # row stride 200, base offset 6 bytes
    got = ops.act_sqnorm(view).cpu().numpy()
    for c in range(4):
        assert np.array_equal(_bits(got[c]), _bits(OW.act_sqnorm(view[c].cpu())))


# ------------------------------------------------------------------- selection --
def _run_select(W, s, mode, **kw):
    ops = _ops()
    Wd = W.clone().to(DEV)
    mask, ssum = ops.wanda_select(Wd, ops.sqrt_scaler(torch.from_numpy(s).to(DEV)), mode, **kw)
    return mask.cpu().numpy(), Wd.cpu(), float(ssum.sum().item()) / W.numel()


def _check_vs_oracle(W, s, mode, ratio=None, n=0, m=0, apply_zero=True):
    if mode == "row":
        kw = dict(k=int(W.shape[1] * ratio))
    elif mode == "matrix":
        kw = dict(k=int(W.numel() * ratio))
    else:
        kw = dict(n=n, m=m)
    mask, Wn, imp = _run_select(W, s, mode, apply_zero=apply_zero, **kw)
    want = OW.prune_linear(W, s, mode, ratio=ratio, n=n, m=m, apply_zero=apply_zero)
    bad = int((mask != want["mask"]).sum())
    assert bad == 0, f"{bad} mask entries differ"
    assert torch.equal(Wn.view(torch.uint8), want["weight"].view(torch.uint8)), "pruned weights differ"
    if np.isfinite(want["importance_score"]):
        assert imp == pytest.approx(want["importance_score"], rel=1e-5)
    return mask


@pytest.mark.parametrize("group,mode", [("g2", "row"), ("g3", "matrix"), ("g4", "nm")])
def test_select_matches_reference_golden(group, mode):
    for name in _cases(group):
        W, xs = UNIT[f"{group}/{name}/W"], UNIT[f"{group}/{name}/xs"]
        ops = _ops()
        s = torch.zeros(W.shape[1], dtype=torch.float32, device=DEV)
        ops.wanda_scaler_update(s, 0, ops.act_sqnorm(xs.to(DEV)), 1)
        s = s.cpu().numpy()
        if mode == "nm":
            kw = dict(n=int(UNIT[f"{group}/{name}/n"]), m=int(UNIT[f"{group}/{name}/m"]))
        elif mode == "row":
            kw = dict(k=int(W.shape[1] * float(UNIT[f"{group}/{name}/ratio"])))
        else:
            kw = dict(k=int(W.numel() * float(UNIT[f"{group}/{name}/ratio"])))
        mask, Wn, imp = _run_select(W, s, mode, **kw)
        assert np.array_equal(mask, UNIT[f"{group}/{name}/mask"].numpy()), name
        assert torch.equal(Wn, UNIT[f"{group}/{name}/Wn"]), name
        assert imp == pytest.approx(float(UNIT[f"{group}/{name}/imp"]), rel=1e-5), name


def _rand_case(out_f, in_f, dtype, seed, zero_frac=0.0, dup=False):
    g = torch.Generator().manual_seed(seed)
    W = (torch.randn(out_f, in_f, generator=g) * 0.02).to(dtype)
    if zero_frac:
        W[torch.rand(out_f, in_f, generator=g) < zero_frac] = 0
    if dup:
        W[:, : in_f // 2] = W[:, :1]
    s = (torch.rand(in_f, generator=g) * 4 + 0.01).numpy().astype(np.float32)
    if dup:
        s[: in_f // 2] = s[0]
    return W, s


ROW_SHAPES = [(64, 2048), (48, 1408), (16, 4096), (12, 5120), (10, 6144), (6, 11008), (33, 104), (7, 100), (5, 2050),
              (300, 512), (3, 16384)]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
@pytest.mark.parametrize("shape", ROW_SHAPES)
def test_row_select_vs_oracle(dtype, shape):
    W, s = _rand_case(*shape, dtype, seed=shape[0] * 7 + shape[1])
    _check_vs_oracle(W, s, "row", ratio=0.5)


@pytest.mark.parametrize("ratio", [0.0, 0.001, 0.3, 0.77, 0.999, 1.0])
def test_row_select_ratios(ratio):
    W, s = _rand_case(40, 2048, torch.bfloat16, seed=11)
    _check_vs_oracle(W, s, "row", ratio=ratio)


@pytest.mark.parametrize("shape", [(32, 2048), (9, 5120), (20, 1408), (11, 100)])
def test_row_select_heavy_ties(shape):
    """Already-pruned weights (half zeros), duplicated columns, all-equal rows: the stable
    tie rule (lowest column first) decides most of the mask."""
    W, s = _rand_case(*shape, torch.bfloat16, seed=3, zero_frac=0.6)
    _check_vs_oracle(W, s, "row", ratio=0.5)
    W, s = _rand_case(*shape, torch.float16, seed=4, dup=True)
    _check_vs_oracle(W, s, "row", ratio=0.5)
    W = torch.full(shape, 0.5, dtype=torch.bfloat16)
    _check_vs_oracle(W, np.ones(shape[1], dtype=np.float32), "row", ratio=0.37)


def test_row_select_nan_and_inf_scores():
    W, s = _rand_case(24, 2048, torch.bfloat16, seed=8)
    s[5] = np.inf
    s[9] = np.nan
    W[3, 5] = 0            # 0 * inf = nan
    W[:, 100:1500] = 0
    _check_vs_oracle(W, s, "row", ratio=0.9)
    _check_vs_oracle(W, s, "row", ratio=0.5)


def test_row_select_is_idempotent_and_lora_mode_keeps_weights():
    W, s = _rand_case(128, 2048, torch.bfloat16, seed=21)
    m1, W1, _ = _run_select(W, s, "row", k=1024)
    m2, W2, _ = _run_select(W1, s, "row", k=1024)
    assert np.array_equal(m1, m2) and torch.equal(W1, W2)
    m3, W3, _ = _run_select(W, s, "row", k=1024, apply_zero=False)
    assert np.array_equal(m1, m3) and torch.equal(W3, W)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16, torch.float32])
@pytest.mark.parametrize("shape", [(96, 1408), (128, 352), (37, 100), (4224, 176), (8, 8)])
@pytest.mark.parametrize("ratio", [0.5, 0.05])
def test_matrix_select_vs_oracle(dtype, shape, ratio):
    W, s = _rand_case(*shape, dtype, seed=shape[0] + shape[1])
    _check_vs_oracle(W, s, "matrix", ratio=ratio)


def test_matrix_select_ties_at_threshold_are_kept():
    W, s = _rand_case(64, 256, torch.float16, seed=2, zero_frac=0.7)   # threshold == 0 -> nothing < 0
    mask = _check_vs_oracle(W, s, "matrix", ratio=0.5)
    assert mask.all()
    W, s = _rand_case(64, 256, torch.float16, seed=2, dup=True)
    _check_vs_oracle(W, s, "matrix", ratio=0.3)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
@pytest.mark.parametrize("nm", [(2, 4), (4, 8), (1, 2), (1, 4), (3, 4)])
@pytest.mark.parametrize("shape", [(64, 2048), (31, 104), (5, 12), (16, 1408)])
def test_nm_select_vs_oracle(dtype, nm, shape):
    if shape[1] % nm[1]:
        pytest.skip("in % m != 0")
    W, s = _rand_case(*shape, dtype, seed=shape[1] + nm[0], zero_frac=0.2)
    _check_vs_oracle(W, s, "nm", n=nm[0], m=nm[1])


def test_bad_arguments_raise():
    from vlmc import _lib
    ops = _ops()
    W = torch.zeros(4, 16, dtype=torch.bfloat16, device=DEV)
    s = torch.ones(16, device=DEV)
    s = ops.sqrt_scaler(s)
    with pytest.raises(_lib.VlmcError):
        ops.wanda_select(W, s, "row", k=17)
    with pytest.raises(_lib.VlmcError):
        ops.wanda_select(W, s, "nm", n=2, m=3)
    with pytest.raises(_lib.VlmcError):
        ops.wanda_select(W, s, "matrix", k=64)


def test_launch_events_are_carried_by_the_next_timed_kernel():
    """vlmc_set_launch_events (the benchmark's timing hook): the next statistics launch records its own begin / end
    into the caller's HIP events (hipExtLaunchKernel); the pair is consumed by that one launch; results are unchanged."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from vlmc import _lib, ops
    lib = _lib.load()
    hipev = bench.HipEvents()
    x = (torch.randn(64, 257, 1408, device="cuda:0") + 0.1).half()              # 46 MB: a launch of ~10 us
    want = ops.act_sqnorm(x)
    torch.cuda.synchronize()
    a, b = hipev.new(), hipev.new()
    lib.vlmc_set_launch_events(a, b)
    got = ops.act_sqnorm(x)
    again = ops.act_sqnorm(x)                                                    # (no events pending any more)
    torch.cuda.synchronize()
    ms = hipev.elapsed_ms(a, b)
    assert 0.002 < ms < 5.0, ms
    assert torch.equal(got, want) and torch.equal(again, want)
    # a select launch takes them as well
    W = (torch.randn(256, 2048, device="cuda:0") * 0.02).bfloat16()
    sq = ops.sqrt_scaler(torch.rand(2048, device="cuda:0") + 0.1)
    c, d = hipev.new(), hipev.new()
    lib.vlmc_set_launch_events(c, d)
    ops.wanda_select(W, sq, "row", k=1024)
    torch.cuda.synchronize()
    assert 0.001 < hipev.elapsed_ms(c, d) < 5.0
    hipev.free(a, b, c, d)
