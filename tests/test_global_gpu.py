"""GPU parity of K17 (vlmc_score_select, the global pruners' threshold selection) against the CPU oracle:
bit-identical keep masks and weights for every score mode, scope layout, per-layer protection, previous
masks, mixed dtypes, ties, signed zeros and ragged tensor sizes."""
import numpy as np
import pytest
import torch

import oracle_ops

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _tensors(rng, n, dtypes, ragged=True, big=False):
    out = []
    for i in range(n):
        dt = dtypes[i % len(dtypes)]
        if big:
            shape = (int(rng.integers(64, 400)), int(rng.integers(8, 64)) * 8)
        elif ragged and rng.integers(0, 3) == 0:
            shape = (int(rng.integers(1, 9)), int(rng.integers(1, 50)))
        else:
            shape = (int(rng.integers(1, 40)), int(rng.integers(1, 30)) * 8)
        g = torch.Generator().manual_seed(int(rng.integers(0, 2**31)))
        w = (torch.randn(shape, generator=g) * 0.05)
        style = rng.integers(0, 4)
        if style == 1:
            w[torch.rand(shape, generator=g) < 0.3] = 0
            w[torch.rand(shape, generator=g) < 0.1] = -0.0
        elif style == 2:
            w = torch.randint(-3, 4, shape, generator=g).float() * 0.01
        out.append(w.to(dt))
    return out


def _run_both(mode, ws, S, prev, scopes, ks, protect):
    from vlmc import ops
    wd = [w.clone().to(DEV) for w in ws] if ws is not None else None
    Sd = [s.to(DEV) for s in S] if S is not None else None
    pd = [p.to(DEV) for p in prev] if prev is not None else None
    got = ops.score_select(wd, mode, scopes=scopes, scope_ks=ks, scores=Sd, prev_keeps=pd, protect_ks=protect)
    wc = [w.clone() for w in ws] if ws is not None else None
    want = oracle_ops.score_select(wc, mode, scopes=scopes, scope_ks=ks, scores=S, prev_keeps=prev, protect_ks=protect)
    for i, (g, w_) in enumerate(zip(got, want)):
        assert torch.equal(g.cpu(), w_), (mode, i, tuple(w_.shape), int((g.cpu() != w_).sum()))
    if ws is not None:
        for a, b in zip(wd, wc):
            assert torch.equal(a.cpu().view(torch.uint8), b.view(torch.uint8))      # including the sign of pruned zeros
    return got


@pytest.mark.parametrize("seed", range(5))
@pytest.mark.parametrize("mode", ["weight", "score", "absw_score"])
@pytest.mark.parametrize("layout", ["global", "per_model", "layerwise"])
def test_score_select_matches_oracle(seed, mode, layout):
    rng = np.random.default_rng(seed * 31 + len(mode) + len(layout))
    n = int(rng.integers(1, 12))
    ws = _tensors(rng, n, [torch.bfloat16, torch.float16, torch.float32])
    g = torch.Generator().manual_seed(seed)
    S = None
    if mode != "weight":
        S = [torch.randn(w.shape, generator=g) if mode == "score" else torch.rand(w.shape, generator=g) for w in ws]
        if seed % 2:
            S = [(s * 4).round() / 4 for s in S]                                  # ties between tensors
    scopes = {"global": [0] * n, "per_model": [0 if i < (n + 1) // 2 else 1 for i in range(n)], "layerwise": list(range(n))}[layout]
    nsc = max(scopes) + 1
    sizes = [sum(w.numel() for w, s in zip(ws, scopes) if s == sid) for sid in range(nsc)]
    ks = [max(1, int(float(rng.uniform(0.05, 0.95)) * sz)) for sz in sizes]
    prev = [torch.rand(w.shape, generator=g) > 0.3 for w in ws] if seed % 3 == 1 else None
    protect = [int(w.numel() * 0.2) for w in ws] if (seed % 2 == 0 and layout != "layerwise") else None
    _run_both(mode, ws, S, prev, scopes, ks, protect)


def test_score_select_extreme_ranks_and_special_values():
    rng = np.random.default_rng(5)
    ws = _tensors(rng, 4, [torch.float32], ragged=False)
    total = sum(w.numel() for w in ws)
    for k in (1, 2, total - 1, total):
        _run_both("weight", ws, None, None, [0] * 4, [k], None)
    S = [torch.randn(w.shape) for w in ws]
    S[1][0, 0] = float("nan")
    S[2][0, :3] = float("inf")
    S[3][0, :3] = -float("inf")
    for k in (1, 5, total // 2, total - 2, total):
        _run_both("score", ws, S, None, [0] * 4, [k], None)
        _run_both("score", None, S, None, [0, 0, 1, 1], [min(k, ws[0].numel() + ws[1].numel()), min(k, ws[2].numel() + ws[3].numel())], None)


def test_score_select_per_layer_scalars_and_errors():
    from vlmc import ops
    from vlmc._lib import VlmcError
    S = [torch.tensor([float(v)]) for v in (0.3, 0.1, 0.7, 0.1, 0.5)]
    got = _run_both("score", None, S, None, [0] * 5, [2], None)
    assert [bool(g.item()) for g in got] == [True, False, True, False, True]
    w = torch.randn(8, 8, device=DEV)
    with pytest.raises(VlmcError):
        ops.score_select([w], "weight", scopes=[0], scope_ks=[0])                  # the reference's IndexError on k == 0
    with pytest.raises(VlmcError):
        ops.score_select([w], "weight", scopes=[0], scope_ks=[65])
    with pytest.raises(VlmcError):
        ops.score_select([w], "absw_score", scopes=[0], scope_ks=[3])              # S missing


def test_score_select_large_mixed_model_like_job_table():
    """~60 tensors, a few million elements, fp16 'vision' + bf16 'language' scopes; thresholds checked by counting."""
    rng = np.random.default_rng(11)
    ws = _tensors(rng, 60, [torch.float16, torch.bfloat16], big=True)
    scopes = [i % 2 for i in range(60)]
    sizes = [sum(w.numel() for w, s in zip(ws, scopes) if s == sid) for sid in range(2)]
    ks = [int(0.5 * s) for s in sizes]
    got = _run_both("weight", ws, None, None, scopes, ks, None)
    for sid in range(2):
        pruned = sum(int((~g).sum()) for g, s in zip(got, scopes) if s == sid)
        assert pruned >= ks[sid]                                                   # ties at the threshold are pruned too


# ---- the drop-in pruners end to end on the GPU ------------------------------------------------------------
from test_global_host_logic import VARIANTS, golden_weights, run_global  # noqa: E402


@pytest.mark.parametrize("name", ["mag_global", "mag_per_model_it2", "mag_layerwise_mixed"])
def test_magnitude_pruners_reproduce_reference_golden_exactly(name):
    pruned = run_global(name, DEV)
    got = dict(pruned.named_parameters())
    for k, ref in golden_weights(name).items():
        assert got[k].dtype == ref.dtype
        assert torch.equal(got[k].data.cpu().view(torch.uint8), ref.view(torch.uint8)), k


@pytest.mark.parametrize("name", ["aobd_global", "aobd_layerwise_it2"])
def test_aobd_pruner_matches_reference_golden_up_to_gradient_rounding(name):
    """The gradients come from the GPU's GEMMs (different summation order than the CPU's): scores next to the
    threshold may flip; everything else is identical."""
    pruned = run_global(name, DEV)
    got = dict(pruned.named_parameters())
    agree = total = 0
    for k, ref in golden_weights(name).items():
        g = got[k].data.cpu()
        same = (g == 0) == (ref == 0)
        agree += int(same.sum())
        total += same.numel()
        assert torch.equal(g[same & (ref != 0)], ref[same & (ref != 0)])
    assert agree / total > 0.995, agree / total


def test_aobd_fused_score_equals_selection_on_materialised_scores():
    """|w| * |G| formed inside the kernel == get_mask on compute_importance_scores() (same GPU gradients)."""
    from lavis.compression.pruners.global_pruner import BLIPT5AOBDPruner
    from lavis.compression.pruners.utils import loss_vision_language
    model = toy_models_init().to(DEV)
    batches = [{k: t.to(DEV) for k, t in b.items()} for b in __import__("toy_models").make_batches(4, seed=11)]
    pr = BLIPT5AOBDPruner(model=model, data_loader=batches, num_samples=4, is_global=True)
    for p in model.parameters():
        p.requires_grad = True
    layers = {k: v for k, v in model.named_parameters() if v.dim() == 2 and ".block" in k}
    scores = pr.compute_importance_scores(model, batches, layers, loss_vision_language)
    masks = pr.get_mask(scores, 0.4, 1.0)
    before = {k: v.data.clone() for k, v in layers.items()}
    pr.global_iterative_pruning(0.4, layers, iteratation=1)
    for k, v in layers.items():
        assert torch.equal(v.data, before[k] * masks[k].to(v.dtype)), k
        assert torch.equal(pr.masks[k], masks[k].bool())


def toy_models_init():
    import toy_models
    return toy_models.init_toy(toy_models.ToyBlipT5(), seed=7).eval()


def test_random_and_mezo_pruners_run_on_the_gpu():
    pruned = run_global("rand_global", DEV)
    layers = {k: v for k, v in pruned.named_parameters() if v.dim() == 2 and ".block" in k}
    total = sum(v.numel() for v in layers.values())
    zeros = sum(int((v == 0).sum()) for v in layers.values())
    assert zeros == int((1 - 0.6) * total)                         # continuous random scores: no ties, exactly k pruned
    pruned = run_global("mezo_global", DEV)
    layers = {k: v for k, v in pruned.named_parameters() if v.dim() == 2 and ".block" in k}
    ref = dict(toy_models_init().named_parameters())
    dropped = 0
    for k, v in layers.items():
        if bool((v == 0).all()):
            dropped += 1
        else:
            assert torch.allclose(v.data.cpu(), ref[k].data, atol=1e-2)    # +-z*eps round trips leave fp32 noise
    assert dropped == int(0.4 * len(layers))                       # int(p * number of layers) layers are removed whole
