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
