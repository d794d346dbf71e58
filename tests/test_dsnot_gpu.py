"""GPU parity of DSnoT: statistics (vlmc_act_moments + vlmc_dsnot_stats_update) and the fused
refinement (vlmc_dsnot_refine + vlmc_dsnot_apply) against the reference's golden vectors and the
CPU oracle.  Masks are compared exactly; statistics to 1e-6 relative (torch's CPU sum/var reduce
in a vectorised order that is not part of the contract -- for 16-bit activations the sums are
exact in fp32 and match bit for bit in practice)."""
import numpy as np
import pytest
import torch

import golden_io
from oracle import dsnot as OD

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
G = golden_io.load("dsnot")
CASES = sorted({k.split("/")[0] for k in G if not k.startswith("stat/")})


def _stat(xs, in_f):
    from vlmc import dsnot
    st = dsnot.DsnotInputStat(in_f, DEV)
    for x in xs:
        st.add_call(x.to(DEV))
    return st.finalize()


@pytest.mark.parametrize("name", sorted({k.split("/")[1] for k in G if k.startswith("stat/")}))
def test_stats_match_reference_golden(name):
    n = int(G[f"stat/{name}/n"])
    xs = [G[f"stat/{name}/x{j}"] for j in range(n)]
    st = _stat(xs, xs[0].shape[-1])
    torch.testing.assert_close(st.scaler_row.cpu(), G[f"stat/{name}/scaler{n - 1}"], rtol=1e-6, atol=0)
    torch.testing.assert_close(st.sum_row.cpu(), G[f"stat/{name}/sum{n - 1}"], rtol=2e-6, atol=1e-7)
    torch.testing.assert_close(st.var_row.cpu(), G[f"stat/{name}/var{n - 1}"], rtol=2e-6, atol=0)
    assert st.nsamples == sum(x.shape[0] for x in xs)


def _kw(name):
    kw = {}
    for k in G:
        if k.startswith(f"{name}/kw/"):
            v = G[k]
            kw[k.split("/")[-1]] = v.item() if hasattr(v, "item") else v
    return kw


@pytest.mark.parametrize("name", CASES)
def test_refinement_matches_reference_golden(name):
    from vlmc import dsnot
    W, xs = G[f"{name}/W"], G[f"{name}/xs"]
    st = _stat([x[None] for x in xs], W.shape[1])
    n, m = int(G[f"{name}/n"]), int(G[f"{name}/m"])
    Wd = W.clone().to(DEV)
    keep = dsnot.prune_linear(Wd, st, float(G[f"{name}/ratio"]), prune_n=n, prune_m=m, **_kw(name))
    ref = G[f"{name}/mask"]
    diff = int((keep.cpu() != ref).sum())
    assert diff == 0, f"{diff} mask entries differ from the reference"
    assert torch.equal(Wd.cpu(), G[f"{name}/Wn"])


@pytest.mark.parametrize("shape,nm", [((64, 2048), (0, 0)), ((48, 4096), (0, 0)), ((16, 11008), (0, 0)), ((40, 2048), (2, 4)),
                                      ((24, 5120), (4, 8)), ((32, 1408), (0, 0))])
def test_refinement_vs_oracle_model_widths(shape, nm):
    """Real layer widths (multi-wave rows), skewed activations so that the update rule is exercised."""
    from vlmc import dsnot
    out_f, in_f = shape
    g = torch.Generator().manual_seed(in_f + out_f)
    W = (torch.randn(out_f, in_f, generator=g) * 0.02).to(torch.bfloat16)
    xs = [((torch.randn(1, 9, in_f, generator=g) * 0.5) + 0.2).to(torch.bfloat16) for _ in range(4)]
    ost = OD.DSnoTStat(in_f)
    for x in xs:
        ost.add_batch(x)
    st = _stat(xs, in_f)
    # feed the oracle the device statistics so that only the refinement itself is compared
    ost.scaler_row, ost.sum_metric_row, ost.var = st.scaler_row.cpu(), st.sum_row.cpu(), st.var_row.cpu().reshape(-1, 1)
    n, m = nm
    Wd = W.clone().to(DEV)
    keep = dsnot.prune_linear(Wd, st, 0.5, prune_n=n, prune_m=m, max_cycle_time=60, update_threshold=0.05)
    want = OD.prune_nm(W, ost, n, m, max_cycle_time=60, update_threshold=0.05) if n else \
        OD.prune_unstructured(W, ost, 0.5, max_cycle_time=60, update_threshold=0.05)
    agree = (keep.cpu() == ~want).float().mean().item()
    assert agree == 1.0, f"mask agreement {agree}"


def test_lora_mode_and_zero_ratio():
    from vlmc import dsnot
    W, xs = G["t5_bf16_r50/W"], G["t5_bf16_r50/xs"]
    st = _stat([x[None] for x in xs], W.shape[1])
    Wd = W.clone().to(DEV)
    keep = dsnot.prune_linear(Wd, st, 0.5, apply_zero=False)
    assert torch.equal(Wd.cpu(), W) and torch.equal(keep.cpu(), G["t5_bf16_r50/mask"])
    assert dsnot.prune_linear(Wd, st, 0.0) is None


# ---- list-based fast path vs the per-cycle reference kernel: identical events -------------------------
def _refine_events(W, st, keep, n, m, mc, thr, lists, without_same_sign=1, wanda_init=1, pow_var=1.0, radix_only=False):
    import os
    from vlmc import _lib
    from vlmc.ops import _dtype_code, _stream
    out_f, in_f = W.shape
    events = torch.zeros((out_f, mc), dtype=torch.int32, device=DEV)
    stop = torch.zeros(out_f, dtype=torch.int32, device=DEV)
    os.environ["VLMC_DSNOT_LISTS"] = "1" if lists else "0"
    os.environ["VLMC_DSNOT_RADIX_ONLY"] = "1" if radix_only else "0"
    try:
        _lib.check(_lib.load().vlmc_dsnot_refine(W.data_ptr(), _dtype_code(W), out_f, in_f, W.stride(0), keep.data_ptr(),
                                                 st.sqrt_row.data_ptr(), st.sum_row.data_ptr(), st.var_row.data_ptr(), wanda_init,
                                                 n, m, mc, thr, pow_var, without_same_sign, events.data_ptr(), stop.data_ptr(),
                                                 _stream()))
    finally:
        os.environ.pop("VLMC_DSNOT_LISTS", None)
        os.environ.pop("VLMC_DSNOT_RADIX_ONLY", None)
    torch.cuda.synchronize()
    return events.cpu(), stop.cpu()


@pytest.mark.parametrize("shape", [(24, 1408), (16, 2048), (12, 4096), (6, 5120), (5, 11008), (8, 256), (7, 64), (9, 16384)])
@pytest.mark.parametrize("nm", [(0, 0), (2, 4), (4, 8), (1, 2)])
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_list_kernel_emits_the_same_events_as_the_cycle_kernel(shape, nm, dtype):
    from vlmc import dsnot, ops
    out_f, in_f = shape
    n, m = nm
    g = torch.Generator().manual_seed(in_f + out_f + 7 * n)
    W = (torch.randn(out_f, in_f, generator=g) * 0.02).to(dtype)
    W[torch.rand(out_f, in_f, generator=g) < 0.02] = 0                        # exact zeros: D == 0 columns, metric ties
    W[:, : in_f // 16] = W[:, :1]                                             # duplicated columns: G and metric ties
    xs = [((torch.randn(1, 9, in_f, generator=g) * 0.5) + 0.2).to(dtype) for _ in range(4)]
    for x in xs:
        x[..., : in_f // 16] = x[..., :1]
    st = dsnot.DsnotInputStat(in_f, DEV)
    for x in xs:
        st.add_call(x.to(DEV))
    st.finalize()
    Wd = W.to(DEV)
    if n:
        keep, _ = ops.wanda_select(Wd, st.sqrt_row, "nm", n=n, m=m, apply_zero=False)
    else:
        keep, _ = ops.wanda_select(Wd, st.sqrt_row, "row", k=round(in_f * 0.5), apply_zero=False)
    mc = min(100, in_f - 1)
    for thr, wss in ((0.05, 1), (0.0, 0), (1e9, 1)):
        ev_old, stop_old = _refine_events(Wd, st, keep, n, m, mc, thr, lists=False, without_same_sign=wss)
        ev_new, stop_new = _refine_events(Wd, st, keep, n, m, mc, thr, lists=True, without_same_sign=wss)
        assert torch.equal(stop_old, stop_new), (thr, wss)
        bad = (ev_old != ev_new).nonzero()
        assert bad.numel() == 0, (thr, wss, bad[:5].tolist(), ev_old[bad[0][0], bad[0][1]].item(), ev_new[bad[0][0], bad[0][1]].item())
        # the list heads by the exact radix route alone (the counting-sort route's fallback for heavy ties)
        ev_rdx, stop_rdx = _refine_events(Wd, st, keep, n, m, mc, thr, lists=True, without_same_sign=wss, radix_only=True)
        assert torch.equal(stop_old, stop_rdx) and torch.equal(ev_old, ev_rdx), (thr, wss)


@pytest.mark.parametrize("case", ["constant_rows", "few_values", "nan_inf", "one_outlier"])
@pytest.mark.parametrize("in_f", [1408, 4096])
def test_list_heads_with_degenerate_keys(case, in_f):
    """Inputs that defeat the counting sort's bins (one value everywhere, a handful of values, NaN / Inf keys stretching the
    key range, one huge outlier squeezing everything else into one bin) take the radix route: same events as the per-cycle
    kernel either way."""
    from vlmc import dsnot, ops
    g = torch.Generator().manual_seed(in_f)
    out_f = 12
    W = (torch.randn(out_f, in_f, generator=g) * 0.02).half()
    if case == "constant_rows":
        W[:] = 0.0123
        W[1::2] = -0.5
    elif case == "few_values":
        W = (torch.randint(-2, 3, (out_f, in_f), generator=g).float() * 0.01).half()
    elif case == "nan_inf":
        W[:, 5] = float("nan")
        W[:, 77] = float("inf")
        W[::2, 300] = float("-inf")
    else:
        W[:, 9] = 6e4
    xs = [((torch.randn(1, 9, in_f, generator=g) * 0.5) + 0.2).half() for _ in range(3)]
    st = dsnot.DsnotInputStat(in_f, DEV)
    for x in xs:
        st.add_call(x.to(DEV))
    st.finalize()
    Wd = W.to(DEV)
    keep, _ = ops.wanda_select(Wd, st.sqrt_row, "row", k=in_f // 2, apply_zero=False)
    for n, m in ((0, 0), (2, 4)):
        kp = keep if not n else ops.wanda_select(Wd, st.sqrt_row, "nm", n=n, m=m, apply_zero=False)[0]
        ev_old, stop_old = _refine_events(Wd, st, kp, n, m, 100, 0.01, lists=False)
        ev_new, stop_new = _refine_events(Wd, st, kp, n, m, 100, 0.01, lists=True)
        assert torch.equal(stop_old, stop_new) and torch.equal(ev_old, ev_new), (case, n, m)


def test_list_kernel_capacity_falls_back_to_cycle_kernel():
    """max_cycle > 128 does not fit the list heads: vlmc_dsnot_refine silently uses the per-cycle kernel."""
    from vlmc import dsnot, ops
    g = torch.Generator().manual_seed(3)
    W = (torch.randn(6, 1024, generator=g) * 0.02).to(torch.float16)
    st = dsnot.DsnotInputStat(1024, DEV)
    st.add_call(((torch.randn(1, 9, 1024, generator=g) * 0.5) + 0.2).to(torch.float16).to(DEV))
    st.finalize()
    Wd = W.to(DEV)
    keep, _ = ops.wanda_select(Wd, st.sqrt_row, "row", k=512, apply_zero=False)
    ev_a, stop_a = _refine_events(Wd, st, keep, 0, 0, 200, 0.01, lists=True)
    ev_b, stop_b = _refine_events(Wd, st, keep, 0, 0, 200, 0.01, lists=False)
    assert torch.equal(ev_a, ev_b) and torch.equal(stop_a, stop_b)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16, torch.float32])
def test_moments_of_stacked_calls_equal_the_per_call_launches(dtype):
    """One launch for the samples of a grouped replay forward (`act_moments_calls`) gives each sample the bits of its own
    launch; `DsnotInputStat.ordered()` puts out-of-order groups back into sample order."""
    from vlmc import dsnot
    g = torch.Generator().manual_seed(5)
    x = ((torch.randn(6 * 2, 9, 1408, generator=g) * 0.5) + 0.2).to(dtype).to(DEV)       # 6 calls of batch 2
    one = torch.stack([dsnot.act_moments(x[2 * c:2 * c + 2].reshape(1, -1, 1408)) for c in range(6)], dim=1)
    assert torch.equal(dsnot.act_moments_calls(x, 6), one)
    a, b = dsnot.DsnotInputStat(1408, DEV), dsnot.DsnotInputStat(1408, DEV)
    for c in range(6):
        a.add_call(x[2 * c:2 * c + 2])
    b.add_calls(torch.cat([x[8:12], x[0:4]]), 4, (4, 5, 0, 1))        # groups of non-neighbouring samples, late group first
    b.add_calls(x[4:8], 2, (2, 3))
    a.finalize(); b.finalize()
    for k in ("scaler_row", "sum_row", "var_row", "sqrt_row"):
        assert torch.equal(getattr(a, k), getattr(b, k)), k
    assert a.nsamples == b.nsamples == 12 and a.ntokens == b.ntokens


# ---- return_reorder_indice as a callable of the drop-in module (dsnot_pruner.py:1881-1925) ------------------------------------------
_FG = golden_io.load("formats")
_REORDER = sorted({k.split("/")[1] for k in _FG if k.startswith("reorder/")})


@pytest.mark.parametrize("name", _REORDER)
def test_return_reorder_indice_on_gpu_matches_reference_fixture(name):
    """The HIP ordering (vlmc_reorder_indices) through the drop-in module's own symbol, against the reference's recorded answers."""
    from lavis.compression.pruners.dsnot_pruner import return_reorder_indice
    x, want = _FG[f"reorder/{name}/in"], _FG[f"reorder/{name}/out"]
    got = return_reorder_indice(x.to(DEV))
    assert got.dtype == torch.int64 and got.device.type == "cuda"
    assert torch.equal(got.cpu(), want), name


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16])
@pytest.mark.parametrize("shape", [(1, 1), (3, 255), (5, 256), (7, 257), (4, 5120), (2, 11008), (300, 70)])
def test_return_reorder_indice_on_gpu_matches_oracle(shape, dtype):
    """Widths around the 256-column chunk, model widths, zeros / NaN / -0.0 among the entries; strided input."""
    from lavis.compression.pruners.dsnot_pruner import return_reorder_indice
    g = torch.Generator().manual_seed(shape[0] * 131 + shape[1])
    x = torch.randn(shape, generator=g)
    x[torch.rand(shape, generator=g) < 0.2] = 0.0
    x[torch.rand(shape, generator=g) < 0.02] = float("nan")
    x[torch.rand(shape, generator=g) < 0.02] = -0.0
    x = x.to(dtype)
    assert torch.equal(return_reorder_indice(x.to(DEV)).cpu(), OD.reorder_indices(x.float()))
    wide = torch.cat([x, x], dim=1).to(DEV)                                  # a row-strided view of a wider tensor
    assert torch.equal(return_reorder_indice(wide[:, :shape[1]]).cpu(), OD.reorder_indices(x.float()))
    if shape[1] > 1:                                                          # all negative / all positive rows
        assert torch.equal(return_reorder_indice(-x.abs().nan_to_num(1.0).clamp_min(1e-3).to(DEV)).cpu(),
                           torch.arange(shape[1]).repeat(shape[0], 1))
        assert torch.equal(return_reorder_indice(x.abs().nan_to_num(1.0).clamp_min(1e-3).to(DEV)).cpu(),
                           torch.arange(shape[1] - 1, -1, -1).repeat(shape[0], 1))


def test_return_reorder_indice_refuses_cpu_tensors():
    from lavis.compression.pruners.dsnot_pruner import return_reorder_indice
    with pytest.raises(RuntimeError, match="GPU only"):
        return_reorder_indice(torch.randn(2, 3))
