"""The `llm_model...model.layers` branch of the BLIP pruners (InstructBLIP-Vicuna, BASELINE.json configs 3-4): drop-in Wanda
and DSnoT pruners with `t5_model_prefix="llm_model"` against the reference's own runs on the toy Vicuna model
(tests/golden/vicuna_e2e.npz) -- exact on CPU through the oracle stand-ins, bit-identical masks on the GPU."""
import pytest
import torch

import golden_io
import oracle_ops
import toy_models

G = golden_io.load("vicuna_e2e")
VARIANTS = {
    "wanda_r50": ("blipt5_wanda_pruner", "wanda", torch.float32, {}),
    "wanda_2_4_bf16": ("blipt5_wanda_pruner", "wanda", torch.bfloat16, dict(prune_n=2, prune_m=4)),
    "dsnot_r50": ("blipt5_dsnot_pruner", "dsnot", torch.float32, dict(max_cycle_time=12)),
}


def run(name, device):
    from lavis.compression import load_pruner
    pruner_name, method, llm_dtype, kw = VARIANTS[name]
    torch.manual_seed(0)
    model = toy_models.init_toy(toy_models.ToyBlipVicuna(llm_dtype=llm_dtype), seed=5).eval().to(device)
    batches = [{k: t.to(device) for k, t in b.items()} for b in toy_models.make_batches(6, seed=13)]
    spec = "2-0.5-1.0-1.0"
    cfg = dict(t5_prune_spec=spec, vit_prune_spec=spec, t5_pruning_method=method, vit_pruning_method=method, num_samples=6,
               t5_model_prefix="llm_model", max_sparsity_per_layer=1.01, **kw)
    pruned, sd = load_pruner(pruner_name, model, batches, cfg=cfg).prune()
    assert sd is None
    return pruned


def compare(name, pruned):
    got = pruned.state_dict()
    n_masks = 0
    for key, ref in G.items():
        if key.startswith(f"{name}/sd/"):
            k = key[len(name) + 4:]
            assert got[k].dtype == ref.dtype and torch.equal(got[k].cpu(), ref), k
        elif key.startswith(f"{name}/mask/"):
            mod = dict(pruned.named_modules())[key[len(name) + 6:]]
            assert torch.equal(mod.mask.cpu(), ref), key
            n_masks += 1
    assert n_masks == 2 * 4 + 3 * 7                      # every ViT and LLaMA-layer linear carries its mask
    assert pruned.llm_model.config.use_cache is True     # restored after the capture (:271)


@pytest.mark.parametrize("name", list(VARIANTS))
def test_llm_branch_matches_reference_golden_on_cpu(name, monkeypatch):
    oracle_ops.install(monkeypatch)
    oracle_ops.install_dsnot(monkeypatch)
    compare(name, run(name, "cpu"))


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(VARIANTS))
def test_llm_branch_on_gpu(name):
    """Activations come from the GPU's GEMMs, so statistics differ from the CPU golden in the last bits: masks must agree
    up to near-ties, and every kept weight is untouched."""
    pruned = run(name, "cuda:0")
    mods = dict(pruned.named_modules())
    agree = total = 0
    for key, ref in G.items():
        if key.startswith(f"{name}/mask/"):
            m = mods[key[len(name) + 6:]]
            same = m.mask.cpu() == ref
            agree += int(same.sum())
            total += same.numel()
            assert bool((m.weight.data[~m.mask] == 0).all())
    assert total > 0 and agree / total > 0.985, agree / total
