"""ECoFLaP first stage: `LayerSparsity` (score-proportional per-layer / per-block sparsity) and the pruners'
`sparsity_ratio_granularity` plumbing, against runs of the REFERENCE on the toy InstructBLIP
(tests/golden/ecoflap.npz).  CPU: kernels replaced by the oracle stand-ins, everything else (autograd scores,
MeZO perturbations, budget solver, grouping) is this repo's code and must match bit for bit.
GPU: the same through the HIP kernels; scores are summed on the device, so allocations agree to ~1e-6."""
import numpy as np
import pytest
import torch

import golden_io
import oracle_ops
import toy_models

G = golden_io.load("ecoflap")
VARIANTS = {
    "wanda_block_aobd_sum": dict(cls="wanda", gran="block", score="aobd_sum", kw={}),
    "wanda_layer_obd_avg": dict(cls="wanda", gran="layer", score="obd_avg", kw={}),
    "wanda_model_gradient_sum": dict(cls="wanda", gran="model", score="gradient_sum", kw={}),
    "wanda_block_olmezo": dict(cls="wanda", gran="block", score="olmezo-gradient_sum", kw=dict(num_noise=2)),
    "wanda_block_olmezo_aobd": dict(cls="wanda", gran="block", score="olmezo-aobd_sum", kw=dict(num_noise=2)),
    "wanda_layer_lmezo_obd": dict(cls="wanda", gran="layer", score="lmezo-obd_sum", kw={}),
    "dsnot_block_per_model": dict(cls="dsnot", gran="block", score="aobd_sum", kw=dict(prune_per_model=True, max_cycle_time=4)),
}


def _run(name, device):
    from lavis.compression import load_pruner
    v = VARIANTS[name]
    torch.manual_seed(0)
    np.random.seed(1234)
    model = toy_models.init_toy(toy_models.ToyBlipT5(), seed=7).eval().to(device)
    batches = [{k: t.to(device) for k, t in b.items()} for b in toy_models.make_batches(6, seed=11)]
    spec = "2-0.5-1.0-1.0"
    cfg = dict(t5_prune_spec=spec, vit_prune_spec=spec, t5_pruning_method=v["cls"], vit_pruning_method=v["cls"], num_samples=6,
               max_sparsity_per_layer=0.8, score_method=v["score"], sparsity_ratio_granularity=v["gran"],
               num_data_first_stage=4, **v["kw"])
    return load_pruner(f"blipt5_{v['cls']}_pruner", model, batches, cfg=cfg).prune()


@pytest.mark.parametrize("name", list(VARIANTS))
def test_allocation_and_masks_match_reference_run_on_cpu(name, monkeypatch):
    oracle_ops.install(monkeypatch)
    oracle_ops.install_dsnot(monkeypatch)
    pruned, sd = _run(name, "cpu")
    keys = [str(k) for k in G[f"{name}/keys"]]
    assert sorted(sd.keys()) == keys
    got = torch.tensor([float(sd[k]) for k in keys], dtype=torch.float64)
    assert torch.equal(got, G[f"{name}/sparsity"]), (got - G[f"{name}/sparsity"]).abs().max()
    assert 0.0 <= float(got.min()) and float(got.max()) <= 0.8 + 1e-6 and float(got.max() - got.min()) > 1e-3   # non-uniform
    n = 0
    for mn, mod in pruned.named_modules():
        if f"{name}/mask/{mn}" in G:
            assert torch.equal(mod.mask, G[f"{name}/mask/{mn}"]), mn
            n += 1
    assert n == 2 * 4 + 2 * 7 + 2 * 11


def test_budget_solver_edge_cases():
    from lavis.compression.pruners.layer_single_base_pruner import LayerSparsity
    f = LayerSparsity._keep_budget_per_group
    # one dominant group saturates, the rest share what is left; nobody exceeds the max sparsity
    sp = f(600, {"a": torch.tensor(100.0), "b": torch.tensor(1.0), "c": torch.tensor(1.0)}, {"a": 400, "b": 400, "c": 400}, 0.8)
    assert sp["a"] == 0.0 and all(0.0 <= v <= 0.8 + 1e-6 for v in sp.values())
    # budget below the guaranteed floor: the loop is not entered, everyone sits at the max sparsity
    sp = f(10, {"a": torch.tensor(1.0), "b": torch.tensor(2.0)}, {"a": 100, "b": 100}, 0.8)
    assert all(abs(v - 0.8) < 1e-6 for v in sp.values())


def test_uniform_without_grouping_and_granularity_names():
    from lavis.compression.pruners.layer_single_base_pruner import LayerSparsity, UniformSparsity
    ls = LayerSparsity(None, None, None, 4, 0.5, layer_to_group_mapping={})
    sd = ls.return_sparsity()
    assert isinstance(sd, UniformSparsity) and sd["anything"] == 0.5
    with pytest.raises(AssertionError):
        LayerSparsity(None, None, None, 4, 0.9, max_sparsity_per_layer=0.8)                 # :147


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["wanda_block_aobd_sum", "wanda_layer_obd_avg", "dsnot_block_per_model"])
def test_allocation_on_gpu_tracks_reference_run(name):
    pruned, sd = _run(name, "cuda:0")
    keys = [str(k) for k in G[f"{name}/keys"]]
    got = torch.tensor([float(sd[k]) for k in keys], dtype=torch.float64)
    assert torch.allclose(got, G[f"{name}/sparsity"], rtol=0, atol=2e-3), (got - G[f"{name}/sparsity"]).abs().max()
    tot = diff = 0
    for mn, mod in pruned.named_modules():
        if f"{name}/mask/{mn}" in G:
            ref = G[f"{name}/mask/{mn}"]
            tot += ref.numel()
            diff += int((mod.mask.cpu() != ref).sum())
    assert tot > 0 and diff / tot < 0.02, diff / tot


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["wanda_block_olmezo", "wanda_block_olmezo_aobd", "wanda_layer_lmezo_obd"])
def test_zeroth_order_scores_run_on_gpu(name):
    """The per-layer zeroth-order variants keep ONE host scalar per layer and combine it with the weights
    (layer_single_base_pruner.py:566-571, :645-650, :722-727); the weights live on the GPU here.  The perturbations come
    from the device's random stream, so the allocation is checked for what any seed must give, not against the CPU run."""
    pruned, sd = _run(name, "cuda:0")
    keys = [str(k) for k in G[f"{name}/keys"]]
    assert sorted(sd.keys()) == keys
    got = torch.tensor([float(sd[k]) for k in keys], dtype=torch.float64)
    assert bool(torch.isfinite(got).all()) and 0.0 <= float(got.min()) and float(got.max()) <= 0.8 + 1e-6
    assert float(got.max() - got.min()) > 1e-3                                           # non-uniform
    n = 0
    for mn, mod in pruned.named_modules():
        if f"{name}/mask/{mn}" in G:
            assert mod.mask.is_cuda and mod.mask.shape == mod.weight.shape
            n += 1
    assert n == 2 * 4 + 2 * 7 + 2 * 11
