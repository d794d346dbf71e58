"""CPU: the on-disk artefacts of a pruning run (train.py:677-714) and the tower-wise reload with the reference's key
rewrites (evaluate_new.py:226-276), on the toy InstructBLIP pruned through the drop-in Wanda pruner."""
import os

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
