"""GPU round trip of the on-disk formats: prune on the GPU through the drop-in pruners ->
`save_pruned_model` (train.py:677-714) -> `load_pruned_language_model` / `load_pruned_vit` (evaluate_new.py:226-276)
into a fresh model -> identical weights, the zero pattern of the masks, and the reference's key sets.

(The reference keeps both sides inline in its drivers' `main()`, which cannot run in this snapshot -- SURVEY.md F3 -- so
there is no reference-generated file to compare with; the layout itself is pinned literally in tests/test_formats.py.)"""
import os

import pytest
import torch
import yaml

import pruner_helpers as H
import toy_models

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("variant,lora", [("fp32_r50", False), ("mixed_2_4", False), ("fp32_r40_lora", True)])
def test_prune_save_reload_round_trip(variant, lora, tmp_path):
    from vlmc import formats
    pruned, sparsity = H.run_pruner(variant, DEV)
    masks = {n: m.mask.clone() for n, m in pruned.named_modules() if hasattr(m, "mask") and torch.is_tensor(m.mask)}
    assert len(masks) == 2 * 4 + 2 * 7 + 2 * 11
    paths = formats.save_pruned_model(pruned, "job", "blipt5_wanda_pruner", sparsity_dict=sparsity, start_time=0.0, root=str(tmp_path))
    state = torch.load(paths["checkpoint"], map_location="cpu")
    live = pruned.state_dict()
    assert list(state) == list(live)                                        # train.py:684: the full state dict, same order
    for k, v in live.items():
        assert torch.equal(state[k], v.cpu()), k
    mask_keys = [k for k in state if k.endswith(".mask")]
    if lora:
        # SparseLoRA layers carry `mask` as a persistent buffer (lora.py:317): it travels with the checkpoint
        assert len(mask_keys) == len(masks)
        for n, m in masks.items():
            assert torch.equal(state[n + ".mask"], m.cpu()), n
        assert any("lora_A" in k for k in state)
    else:
        # a prune-only run attaches `module.mask` as a plain attribute (wanda_pruner.py:339): not in the file, the zeros are
        assert not mask_keys
        for n, m in masks.items():
            assert bool((state[n + ".weight"][~m.cpu()] == 0).all()), n
    scores = torch.load(paths["importance_scores"])
    assert len(scores) == len(masks) and all(isinstance(v, float) and v > 0 for v in scores.values())
    stats = yaml.safe_load(open(paths["training_statistics"]))
    assert set(stats) == {"memory", "time"} and stats["memory"] > 0       # peak GPU memory in GB, as the reference records
    assert ("sparsity_dict" in paths) == isinstance(sparsity, dict)
    # ---- reload, tower by tower, into a fresh model on the GPU ----------------------------------------------------
    v = H.VARIANTS[variant]
    fresh = toy_models.init_toy(toy_models.ToyBlipT5(vit_dtype=v["vit_dtype"], t5_dtype=v["t5_dtype"]), seed=99).to(DEV)
    assert formats.load_pruned_language_model(fresh, paths["checkpoint"]) == "t5_model"
    assert formats.load_pruned_vit(fresh, paths["checkpoint"]) == "visual_encoder."
    got = fresh.state_dict()
    n_checked = 0
    for k, t in live.items():
        if "lora" in k or "mask" in k or not k.startswith(("t5_model.", "visual_encoder.")):
            continue
        assert got[k].is_cuda and torch.equal(got[k], t), k
        n_checked += 1
    assert n_checked >= len(masks)
    if not lora:
        for n, m in masks.items():
            assert bool((got[n + ".weight"][~m] == 0).all()), n
    assert sorted(os.listdir(tmp_path)) == sorted(["importance_scores", "pruned_checkpoint", "training_statistics"] +
                                                  (["sparsity_dict"] if isinstance(sparsity, dict) else []))
