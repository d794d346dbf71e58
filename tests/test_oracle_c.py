"""CPU: the C restatement (oracle/wanda_oracle.c) against the reference's golden vectors and
against the numpy/torch oracle -- two independent restatements must agree bit for bit."""
import numpy as np
import pytest
import torch

import golden_io
from oracle import wanda as OW
from oracle import wanda_c as OC

pytestmark = pytest.mark.skipif(not OC.available(), reason="oracle/_build not built (run __graft_entry__.build())")
UNIT = golden_io.load("wanda_unit")


def _cases(prefix):
    return sorted({k.split("/")[1] for k in UNIT if k.startswith(prefix + "/")})


@pytest.mark.parametrize("name", _cases("g1"))
def test_c_stats_match_reference_golden(name):
    n, b = int(UNIT[f"g1/{name}/n"]), int(UNIT[f"g1/{name}/b"])
    s = np.zeros(UNIT[f"g1/{name}/states"].shape[1], dtype=np.float32)
    ns = 0
    for j in range(n):
        s, ns = OC.scaler_update(s, ns, OC.act_sqnorm(UNIT[f"g1/{name}/x{j}"]), b)
        assert np.array_equal(s.view(np.uint32), UNIT[f"g1/{name}/states"][j].numpy().view(np.uint32)), j


@pytest.mark.parametrize("group,mode", [("g2", "row"), ("g3", "matrix"), ("g4", "nm")])
def test_c_select_matches_reference_golden(group, mode):
    for name in _cases(group):
        W, xs = UNIT[f"{group}/{name}/W"], UNIT[f"{group}/{name}/xs"]
        s = np.zeros(W.shape[1], dtype=np.float32)
        ns = 0
        for x in xs:
            s, ns = OC.scaler_update(s, ns, OC.act_sqnorm(x), 1)
        if mode == "nm":
            kw = dict(n=int(UNIT[f"{group}/{name}/n"]), m=int(UNIT[f"{group}/{name}/m"]))
        elif mode == "row":
            kw = dict(k=int(W.shape[1] * float(UNIT[f"{group}/{name}/ratio"])))
        else:
            kw = dict(k=int(W.numel() * float(UNIT[f"{group}/{name}/ratio"])))
        mask, Wn, imp = OC.select(W, s, mode, **kw)
        assert np.array_equal(mask, UNIT[f"{group}/{name}/mask"].numpy()), name
        assert torch.equal(Wn, UNIT[f"{group}/{name}/Wn"]), name
        assert imp == pytest.approx(float(UNIT[f"{group}/{name}/imp"]), rel=1e-6)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
def test_c_and_python_oracles_agree_on_larger_cases(dtype):
    g = torch.Generator().manual_seed(9)
    xs = [((torch.randn(1, 33, 512, generator=g) + 0.2) * 2).to(dtype) for _ in range(5)]
    s = OW.wanda_stats(xs)
    sc = np.zeros(512, dtype=np.float32)
    n = 0
    for x in xs:
        sc, n = OC.scaler_update(sc, n, OC.act_sqnorm(x), 1)
    assert np.array_equal(s.view(np.uint32), sc.view(np.uint32))
    W = (torch.randn(96, 512, generator=g) * 0.02).to(dtype)
    W[torch.rand(96, 512, generator=g) < 0.3] = 0
    for mode, kw_c, kw_p in [("row", dict(k=256), dict(ratio=0.5)), ("matrix", dict(k=96 * 512 // 3), dict(ratio=1 / 3)),
                             ("nm", dict(n=2, m=4), dict(n=2, m=4))]:
        mask, Wn, imp = OC.select(W, s, mode, **kw_c)
        want = OW.prune_linear(W, s, mode, **kw_p)
        if mode == "matrix":
            assert int(W.numel() * (1 / 3)) == 96 * 512 // 3
        assert np.array_equal(mask, want["mask"]), mode
        assert torch.equal(Wn, want["weight"]), mode
