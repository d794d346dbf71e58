"""CPU: the on-disk artefacts of a pruning run (train.py:677-714) and the tower-wise reload with the reference's key
rewrites (evaluate_new.py:226-276), on the toy InstructBLIP pruned through the drop-in Wanda pruner."""
import os

import pytest
import torch
import yaml

import oracle_ops
import pruner_helpers
import toy_models


def _pruned(monkeypatch, name):
    oracle_ops.install(monkeypatch)
    return pruner_helpers.run_pruner(name)


def test_save_writes_the_reference_layout(monkeypatch, tmp_path):
    from vlmc import formats
    model, _ = _pruned(monkeypatch, "fp32_r40_lora")
    sd = {"t5_model.encoder.block.0.layer.0.SelfAttention.q.weight": 0.4, "visual_encoder.blocks.0.attn.qkv.weight": 0.35}
    paths = formats.save_pruned_model(model, "job7", "blipt5_wanda_pruner", sparsity_dict=sd, start_time=0.0, root=str(tmp_path))
    assert paths["checkpoint"] == str(tmp_path / "pruned_checkpoint/V+L/blipt5_wanda_pruner/job7.pth")
    assert sorted(os.listdir(tmp_path)) == ["importance_scores", "pruned_checkpoint", "sparsity_dict", "training_statistics"]
    state = torch.load(paths["checkpoint"])
    assert set(state) == set(model.state_dict())
    assert any(k.endswith(".mask") for k in state) and any("lora_A" in k for k in state)       # the full state dict, as saved there
    assert yaml.safe_load(open(paths["sparsity_dict"])) == sd
    stats = yaml.safe_load(open(paths["training_statistics"]))
    assert set(stats) == {"memory", "time"} and stats["time"] > 0
    scores = torch.load(paths["importance_scores"])
    want = {k: v.importance_score for k, v in model.named_parameters() if getattr(v, "importance_score", None) is not None}
    assert scores == want and len(want) > 0 and all(isinstance(v, float) for v in scores.values())
    # a non-dict "sparsity" (the pruners return None without ECoFLaP) writes no yaml
    paths = formats.save_pruned_model(model, "job8", "m", sparsity_dict=None, root=str(tmp_path))
    assert "sparsity_dict" not in paths and not os.path.exists(tmp_path / "sparsity_dict" / "job8.yaml")


def test_reload_per_tower_drops_lora_and_masks_and_wrapper_prefixes(monkeypatch, tmp_path):
    from vlmc import formats
    model, _ = _pruned(monkeypatch, "fp32_r50")
    state = dict(model.state_dict())
    state["t5_model.encoder.block.0.SelfAttention.q.lora_A.weight"] = torch.zeros(4, 32)      # must not reach load_state_dict
    state["visual_encoder.blocks.0.attn.qkv.mask"] = torch.ones(96, 32, dtype=torch.bool)
    # what a PEFT-wrapped run saves: `base_model.model.` inside the tower's keys
    wrapped = {}
    for k, v in state.items():
        if k.startswith("t5_model."):
            k = "t5_model.base_model.model." + k[len("t5_model."):]
        elif k.startswith("visual_encoder."):
            k = "visual_encoder.base_model.model." + k[len("visual_encoder."):]
        wrapped[k] = v
    wrapped["visual_encoder.not_in_the_model.weight"] = torch.zeros(3)
    path = str(tmp_path / "ckpt.pth")
    torch.save(wrapped, path)
    fresh = toy_models.init_toy(toy_models.ToyBlipT5(), seed=99)
    assert formats.load_pruned_language_model(fresh, path) == "t5_model"
    assert formats.load_pruned_vit(fresh, path) == "visual_encoder."
    plain = {k: v for k, v in state.items() if "lora" not in k and "mask" not in k}
    got = fresh.state_dict()
    for k, v in plain.items():
        if k.startswith(("t5_model.", "visual_encoder.")):
            assert torch.equal(got[k], v), k
    assert not torch.equal(got["t5_proj.weight"], state["t5_proj.weight"])        # outside both towers: untouched
    orig = sum(p.numel() for p in fresh.parameters())
    assert 45 < formats.remaining_proportion(fresh, orig) < 75                     # half of the prunable weights are gone


def test_language_tower_probe_order_and_missing_tower(tmp_path):
    from vlmc import formats
    m = torch.nn.Module()
    m.llm_model = torch.nn.Linear(4, 4)
    ref = torch.nn.Linear(4, 4)
    path = str(tmp_path / "c.pth")
    torch.save({"llm_model.base_model.model.weight": ref.weight.data, "llm_model.bias": ref.bias.data,
                "llm_model.lora_A.weight": torch.zeros(2, 4), "llm_model.mask": torch.ones(4, 4)}, path)
    assert formats.load_pruned_language_model(m, path) == "llm_model"
    assert torch.equal(m.llm_model.weight.data, ref.weight.data)
    assert formats.load_pruned_language_model(torch.nn.Module(), path) is None


# ---- pinned to what the REFERENCE's own statements wrote and read (tests/golden/formats.npz; make_golden.py: gen_formats drives the save block of
# ---- train.py:677-714 and the reload blocks of evaluate_new.py:226-276 on the PEFT-wrapped toy) ----------------------------------------------------
def _wrapped_toy():
    """The model of gen_formats, built with the DROP-IN peft package (same seeds, same order of draws)."""
    from lavis.peft.src.peft import LoraConfig, get_peft_model
    model = toy_models.init_toy(toy_models.ToyBlipT5(), seed=3)
    model.t5_model = get_peft_model(model.t5_model, LoraConfig(r=4, lora_alpha=16, target_modules=[".q", ".k", ".v", ".o", ".wi_0", "wi_1", "wo"],
                                                               lora_dropout=0.0, bias="none", task_type="CAUSAL_LM"))
    model.visual_encoder = get_peft_model(model.visual_encoder, LoraConfig(r=2, lora_alpha=16, target_modules=[".qkv", ".proj", ".fc1", ".fc2"],
                                                                           lora_dropout=0.0, bias="none", task_type="ViT"))
    g = torch.Generator().manual_seed(11)
    k = 0
    for name, mod in model.named_modules():
        if hasattr(mod, "lora_A") and hasattr(mod, "mask"):
            mod.mask = torch.rand(mod.weight.shape, generator=g) > 0.5
            mod.weight.importance_score = float(k) * 0.25 + 0.125
            with torch.no_grad():
                mod.lora_B.weight.copy_(torch.randn(mod.lora_B.weight.shape, generator=g) * 0.01)
            k += 1
    return model


def test_artefacts_equal_what_the_references_save_block_wrote(tmp_path):
    import golden_io
    from vlmc import formats
    G = golden_io.load("formats")
    model = _wrapped_toy()
    sd = {"t5_model.base_model.model.encoder.block.0.layer.0.SelfAttention.q.weight": 0.5, "visual_encoder.blocks.1.mlp.fc2.weight": 0.25}
    paths = formats.save_pruned_model(model, "job42", "blipt5_wanda_pruner", sparsity_dict=sd, start_time=0.0, root=str(tmp_path))
    listing = sorted(os.path.relpath(os.path.join(d, f), tmp_path) for d, _, fs in os.walk(tmp_path) for f in fs)
    assert listing == [str(x) for x in G["files"]]
    state = torch.load(paths["checkpoint"])
    assert list(state.keys()) == [str(k) for k in G["ckpt/keys"]]                  # names AND order: the wrappers' module tree
    assert [str(v.dtype) for v in state.values()] == [str(x) for x in G["ckpt/dtypes"]]
    assert [",".join(map(str, v.shape)) for v in state.values()] == [str(x) for x in G["ckpt/shapes"]]
    for k, v in state.items():
        if "lora_A" in k:                                                          # kaiming draws from the global RNG stream on both sides
            continue
        assert torch.equal(v, G["ckpt/tensor/" + k]), k
    assert open(paths["sparsity_dict"]).read() == str(G["sparsity_yaml"])
    stats = yaml.safe_load(open(paths["training_statistics"]))
    assert sorted(stats) == [str(x) for x in G["stats/keys"]]
    scores = torch.load(paths["importance_scores"])
    assert list(scores.keys()) == [str(k) for k in G["scores/keys"]]
    assert [float(v) for v in scores.values()] == [float(v) for v in G["scores/values"]]


def test_reload_equals_what_the_references_reload_blocks_loaded(tmp_path):
    import golden_io
    from vlmc import formats
    G = golden_io.load("formats")
    state = {str(k): G["ckpt/tensor/" + str(k)] for k in G["ckpt/keys"]}           # the checkpoint the reference wrote
    path = str(tmp_path / "ref.pth")
    torch.save(state, path)
    fresh = toy_models.init_toy(toy_models.ToyBlipT5(), seed=99)
    assert formats.load_pruned_language_model(fresh, path) == "t5_model"
    assert formats.load_pruned_vit(fresh, path) == "visual_encoder."
    got = fresh.state_dict()
    want = {k[len("reloaded/"):]: v for k, v in G.items() if k.startswith("reloaded/")}
    assert list(got.keys()) == list(want.keys())
    for k, v in want.items():
        assert torch.equal(got[k], v), k
    # and it really took the pruned towers' tensors, stripped of `base_model.model.`
    assert torch.equal(got["t5_model.encoder.block.0.SelfAttention.q.weight"],
                       state["t5_model.base_model.model.encoder.block.0.SelfAttention.q.weight"])


def test_return_reorder_indice_fixture():
    """dsnot_pruner.py:1881-1925 stand-alone: the docstring's example and edge rows, the oracle's restatement."""
    import golden_io
    from oracle import dsnot as OD
    G = golden_io.load("formats")
    names = sorted({k.split("/")[1] for k in G if k.startswith("reorder/")})
    assert "docstring" in names and len(names) >= 5
    for n in names:
        assert torch.equal(OD.reorder_indices(G[f"reorder/{n}/in"]), G[f"reorder/{n}/out"]), n
    assert G["reorder/docstring/out"].tolist() == [[1, 2, 0], [0, 2, 1], [2, 1, 0], [0, 1, 2]]


# ---- PEFT surface: LoraModel.enable / disable_adapter_layers, get_peft_config_as_dict (lora.py:221-236), default targets ----------------
def test_lora_model_adapter_switches_and_config_dict():
    """`PeftModel.disable_adapter()` (peft_model.py:312-316) goes through LoraModel.disable_adapter_layers / enable_adapter_layers;
    with the adapters disabled a SparseLoRA layer is the plain linear on W (lora.py:361-362), on CPU as well."""
    from lavis.peft.src.peft.tuners.lora import LoraLayer
    model = _wrapped_toy()
    t5 = model.t5_model
    layers = [m for m in t5.modules() if isinstance(m, LoraLayer)]
    assert layers and all(not m.disable_adapters for m in layers)
    x = torch.randn(2, 3, layers[0].in_features)
    with t5.disable_adapter():
        assert all(m.disable_adapters for m in layers)
        assert torch.equal(layers[0](x), torch.nn.functional.linear(x, layers[0].weight, layers[0].bias))
    assert all(not m.disable_adapters for m in layers)
    t5.base_model.disable_adapter_layers()
    assert all(m.disable_adapters for m in layers)
    t5.base_model.enable_adapter_layers()
    cfg = t5.base_model.get_peft_config_as_dict()
    assert cfg["peft_type"] == "LORA" and cfg["task_type"] == "CAUSAL_LM" and cfg["r"] == 4 and cfg["lora_alpha"] == 16
    assert cfg["inference_mode"] is False and t5.base_model.get_peft_config_as_dict(inference=True)["inference_mode"] is True
    assert t5.base_model.modules_to_save is None


def test_get_peft_model_default_targets_follow_the_model_type():
    """mapping.py:152-158: `target_modules=None` takes the model type's default linears; an unknown type raises as the reference does."""
    import types

    from lavis.peft.src.peft import LoraConfig, get_peft_model
    from lavis.peft.src.peft.tuners.lora import Linear

    class Attn(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.q_proj, self.k_proj, self.v_proj = torch.nn.Linear(8, 8), torch.nn.Linear(8, 8), torch.nn.Linear(8, 8)

    class Net(torch.nn.Module):
        def __init__(self, model_type):
            super().__init__()
            self.config = types.SimpleNamespace(model_type=model_type)
            self.attn = Attn()

    wrapped = get_peft_model(Net("llama"), LoraConfig(r=2, lora_alpha=4, target_modules=None, task_type="CAUSAL_LM"))
    attn = wrapped.base_model.model.attn
    assert type(attn.q_proj) is Linear and type(attn.v_proj) is Linear and type(attn.k_proj) is torch.nn.Linear
    with pytest.raises(ValueError, match="target_modules"):
        get_peft_model(Net("no-such-model"), LoraConfig(r=2, lora_alpha=4, target_modules=None, task_type="CAUSAL_LM"))


def test_qformer_wrapper_takes_positional_input_ids_and_hands_on_the_reference_defaults():
    from lavis.peft.src.peft import LoraConfig, get_peft_model
    seen = {}

    class QF(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.query = torch.nn.Linear(4, 4)

        def forward(self, **kw):
            seen.update(kw)
            return 7

    w = get_peft_model(QF(), LoraConfig(r=2, lora_alpha=4, target_modules=[".query", "query"], task_type="Qformer"))
    assert w(torch.zeros(1, 2, dtype=torch.long), query_embeds="q") == 7
    assert seen["input_ids"].shape == (1, 2) and seen["query_embeds"] == "q"
    assert seen["use_cache"] is True and seen["is_decoder"] is True and seen["reduction"] == "mean" and seen["return_logits"] is False


def test_pack_state_dict_24_takes_a_non_contiguous_mask():
    """ `m.view` raised on a transposed mask before the `.contiguous()` fallback; CPU: such a pair simply stays unpacked."""
    from vlmc import formats
    w = torch.randn(8, 16).half()
    m = (torch.arange(16 * 8).reshape(16, 8) % 4 < 2).t()                 # [8, 16], stride (1, 8)
    assert not m.is_contiguous()
    out = formats.pack_state_dict_24({"l.weight": w, "l.mask": m})
    assert set(out) == {"l.weight", "l.mask"}
