"""CPU: the drop-in global pruners (lavis/compression/pruners/global_pruner.py) driven through the oracle
stand-in of `ops.score_select` reproduce the reference's goldens (tests/golden/global.npz) bit for bit; the
oracle's own get_mask / get_layerwise_mask are pinned to the reference's outputs on explicit scores."""
import numpy.random as npr
import pytest
import torch

import golden_io
import oracle_ops
import toy_models
from oracle import global_select as OG

G = golden_io.load("global")

VARIANTS = {
    "mag_global": ("blipt5_mag_pruner", dict(is_global=True), torch.float32),
    "mag_per_model_it2": ("blipt5_mag_pruner", dict(is_global=True, prune_per_model=True, iteration=2), torch.float32),
    "mag_layerwise_mixed": ("blipt5_mag_pruner", dict(is_global=False), torch.bfloat16),
    "rand_global": ("blipt5_rand_pruner", dict(is_global=True), torch.float32),
    "aobd_global": ("blipt5_aobd_pruner", dict(is_global=True), torch.float32),
    "aobd_layerwise_it2": ("blipt5_aobd_pruner", dict(is_global=False, iteration=2), torch.float32),
    "mezo_global": ("blipt5_mezo_pruner", dict(is_global=True, num_noise=2), torch.float32),
}


def run_global(name, device="cpu"):
    from lavis.compression import load_pruner
    pruner_name, kw, t5_dtype = VARIANTS[name]
    torch.manual_seed(0)
    npr.seed(1234)
    model = toy_models.init_toy(toy_models.ToyBlipT5(t5_dtype=t5_dtype), seed=7).eval().to(device)
    batches = [{k: t.to(device) for k, t in b.items()} for b in toy_models.make_batches(6, seed=11)]
    spec = "2-0.6-1.0-1.0"
    pruner = load_pruner(pruner_name, model, batches, cfg=dict(t5_prune_spec=spec, vit_prune_spec=spec, num_samples=4, **kw))
    pruned, extra = pruner.prune()
    assert extra is None
    return pruned


def golden_weights(name):
    return {k[len(name) + 1:]: v for k, v in G.items() if k.startswith(name + "/")}


@pytest.mark.parametrize("name", list(VARIANTS))
def test_global_pruner_matches_reference_golden(name, monkeypatch):
    oracle_ops.install_global(monkeypatch)
    pruned = run_global(name)
    want = golden_weights(name)
    assert want
    got = dict(pruned.named_parameters())
    for k, ref in want.items():
        assert got[k].dtype == ref.dtype
        assert torch.equal(got[k].data.view(torch.uint8), ref.view(torch.uint8)), k     # signed zeros included
        assert got[k].requires_grad                                                       # model_reset restored the flags


def test_oracle_get_mask_pinned_to_reference():
    scores = {k.split("/")[-1]: v for k, v in G.items() if k.startswith("get_mask/scores/")}
    for tag, fn in [("capped", lambda s: OG.get_mask(s, 0.5, 0.6)), ("uncapped", lambda s: OG.get_mask(s, 0.3, 1.0)),
                    ("layerwise", lambda s: OG.get_layerwise_mask(s, 0.45))]:
        got = fn({k: v.clone() for k, v in scores.items()})
        for k, m in got.items():
            assert torch.equal(m, G[f"get_mask/{tag}/{k}"]), (tag, k)


def test_pruner_get_mask_api_through_stand_in(monkeypatch):
    oracle_ops.install_global(monkeypatch)
    from lavis.compression.pruners.global_pruner import BLIPT5MagPruner
    pr = BLIPT5MagPruner(model=toy_models.ToyBlipT5(), data_loader=[])
    scores = {k.split("/")[-1]: v for k, v in G.items() if k.startswith("get_mask/scores/")}
    for tag, got in [("capped", pr.get_mask(scores, 0.5, 0.6)), ("uncapped", pr.get_mask(scores, 0.3, 1.0)),
                     ("layerwise", pr.get_layerwise_mask(scores, 0.45))]:
        for k, m in got.items():
            assert m.dtype == torch.float32 and torch.equal(m, G[f"get_mask/{tag}/{k}"]), (tag, k)
    with pytest.raises(IndexError):
        pr.get_mask(scores, 0.0, 1.0)                     # int(p * N) == 0: the reference indexes an empty topk


def test_oracle_iterative_prune_equals_pruner_goldens():
    """The oracle's own restatement of the loop (:153-201) on the magnitude score."""
    for name, kw in [("mag_global", dict(is_global=True, prune_per_model=False, iteration=1)),
                     ("mag_per_model_it2", dict(is_global=True, prune_per_model=True, iteration=2))]:
        model = toy_models.init_toy(toy_models.ToyBlipT5(), seed=7)
        params = {k: v.data for k, v in model.named_parameters() if v.dim() == 2 and ".block" in k}
        OG.iterative_prune(params, lambda ps: {k: OG.score_magnitude(v) for k, v in ps.items()}, 1 - 0.6, **kw)
        for k, ref in golden_weights(name).items():
            assert torch.equal(params[k].view(torch.uint8), ref.view(torch.uint8)), (name, k)


def test_registry_has_the_global_pruners():
    import lavis.compression  # noqa: F401
    from lavis.common.registry import registry
    for n in ("blipt5_mag_pruner", "blipt5_rand_pruner", "blipt5_aobd_pruner", "blipt5_mezo_pruner"):
        assert registry.get_pruner_class(n).pruner_name == n
