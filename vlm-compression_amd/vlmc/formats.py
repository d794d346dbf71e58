"""On-disk formats either side of the pruning path, as the reference's drivers write and read them.

Writing (train.py:677-714, after `pruner.prune()` / the RESSA loop):
    pruned_checkpoint/V+L/<pruning_method>/<job_id>.pth   full `model.state_dict()` (incl. the `mask` buffers and LoRA tensors)
    sparsity_dict/<job_id>.yaml                           the dict `prune()` returned, when it is a dict
    training_statistics/<job_id>.yaml                     {"memory": peak GB, "time": seconds}
    importance_scores/<job_id>.pth                        {param name: weight.importance_score}
Reading (evaluate_new.py:226-276, evaluate_old.py:246-290): one tower at a time, dropping LoRA and mask entries and
the PEFT wrapper prefixes, then `load_state_dict` into the tower.

Host-side file formats only: the tensors are whatever the pruners left on the GPU; nothing above touches the kernels.

Packed 2:4 (SURVEY.md §8(f)3 "an optional packed 2:4 format for inference"; the reference has none): `pack_state_dict_24` /
`unpack_state_dict_24` turn the `weight` + `mask` pair of every linear that a 2:4 prune left behind into
`weight_packed24` [out, in / 2] + `weight_meta24` [out, in / 8] (include/vlmc.h: vlmc_pack_24) and back, bit for bit --
9 / 16 of the bytes of the 16-bit weight, 3 / 8 of weight + mask.
"""
from __future__ import annotations

import os
import time

import torch
import yaml

LANGUAGE_TOWERS = ("t5_model", "opt_model", "llm_model")          # the order evaluate_new.py probes them in
VIT_PREFIXES = ("visual.", "visual_encoder.")


def save_pruned_model(model, job_id: str, pruning_method: str, sparsity_dict=None, start_time: float | None = None,
                      root: str = ".") -> dict:
    """train.py:677-714.  Returns the paths written."""
    out = {}
    folder = os.path.join(root, "pruned_checkpoint/V+L", pruning_method)
    os.makedirs(folder, exist_ok=True)
    out["checkpoint"] = os.path.join(folder, job_id + ".pth")
    torch.save(model.state_dict(), out["checkpoint"])
    if sparsity_dict is not None and isinstance(sparsity_dict, dict):
        folder = os.path.join(root, "sparsity_dict")
        os.makedirs(folder, exist_ok=True)
        out["sparsity_dict"] = os.path.join(folder, job_id + ".yaml")
        with open(out["sparsity_dict"], "w") as f:
            yaml.dump(sparsity_dict, f)
    peak_memory = (torch.cuda.max_memory_allocated() / 1024 ** 2) / 1000 if torch.cuda.is_available() else 0.0
    training_dict = {"memory": peak_memory, "time": time.time() - (start_time if start_time is not None else time.time())}
    folder = os.path.join(root, "training_statistics")
    os.makedirs(folder, exist_ok=True)
    out["training_statistics"] = os.path.join(folder, job_id + ".yaml")
    with open(out["training_statistics"], "w") as f:
        yaml.dump(training_dict, f)
    folder = os.path.join(root, "importance_scores")
    os.makedirs(folder, exist_ok=True)
    out["importance_scores"] = os.path.join(folder, job_id + ".pth")
    torch.save({k: v.importance_score for k, v in model.named_parameters() if getattr(v, "importance_score", None) is not None},
               out["importance_scores"])
    return out


def _strip(state, prefix, wrappers):
    """Entries of one tower without LoRA tensors and masks, tower prefix and PEFT wrapper prefixes removed
    (evaluate_new.py:229-231; `str.replace`, i.e. every occurrence, as there)."""
    state = {k: v for k, v in state.items() if k.startswith(prefix) and "lora" not in k and "mask" not in k}
    state = {k.replace(prefix + ".", ""): v for k, v in state.items()}
    for w in wrappers:
        state = {k.replace(w, ""): v for k, v in state.items()}
    return state


def load_pruned_language_model(model, checkpoint: str):
    """evaluate_new.py:226-249: the first of t5_model / opt_model / llm_model the model has receives the pruned weights
    (strict load).  Returns the tower's attribute name, or None when the model has none of them."""
    for tower in LANGUAGE_TOWERS:
        if getattr(model, tower, None) is None:
            continue
        state = torch.load(checkpoint, map_location="cpu")
        wrappers = ("base_model.model.", "base_model.Model.") if tower == "t5_model" else ("base_model.model.",)
        getattr(model, tower).load_state_dict(_strip(state, tower, wrappers))
        return tower
    return None


def load_pruned_vit(model, checkpoint: str, interpolate_pos_embed=None):
    """evaluate_new.py:251-276: the checkpoint's vision entries (prefix `visual.` or `visual_encoder.`) overwrite the
    matching entries of the current vision tower's state dict; unknown keys are ignored, missing ones keep their
    current values.  `interpolate_pos_embed(visual_encoder, state)` is the reference's eva_vit helper, if needed."""
    state = torch.load(checkpoint, map_location="cpu")
    prefix = None
    for cand in VIT_PREFIXES:
        if any(k.startswith(cand) for k in state.keys()):
            prefix = cand
            break
    assert prefix is not None
    state = {k: v for k, v in state.items() if k.startswith(prefix) and "lora" not in k and "mask" not in k}
    state = {k.replace(prefix, ""): v for k, v in state.items()}
    state = {k.replace("base_model.model.", ""): v for k, v in state.items()}
    current = model.visual_encoder.state_dict()
    for k, v in state.items():
        if k in current:
            current[k] = v
    if interpolate_pos_embed is not None:
        interpolate_pos_embed(model.visual_encoder, current)
    model.visual_encoder.load_state_dict(current)
    return prefix


def remaining_proportion(model, orig_total_size) -> float:
    """evaluate_new.py:279-283: non-zero parameters over the original parameter count, in percent."""
    kept = sum((p != 0).float().sum() for p in model.parameters())
    return float(kept / orig_total_size * 100)


PACKED_VALUES, PACKED_META = "weight_packed24", "weight_meta24"


def pack_state_dict_24(state: dict, keep_masks: bool = False) -> dict:
    """`model.state_dict()` of a 2:4-pruned model with every (`<name>.weight`, `<name>.mask`) pair whose mask keeps exactly two of
    every four input columns replaced by `<name>.weight_packed24` / `<name>.weight_meta24` (CUDA tensors; the kernels pack).
    Pairs that are not 2:4 (unstructured masks, fp32 weights, widths that are not multiples of 8) stay as they are.
    `keep_masks`: keep the `mask` entries of packed linears too (they are recoverable from the metadata)."""
    from . import ops
    out = {}
    for k, v in state.items():
        if k.endswith(".weight") and k[:-6] + "mask" in state and isinstance(v, torch.Tensor) and v.is_cuda and v.dim() == 2 and \
                v.dtype in (torch.float16, torch.bfloat16) and v.shape[1] % 8 == 0:
            m = state[k[:-6] + "mask"]
            if m.shape == v.shape and m.dtype == torch.bool and bool((m.reshape(v.shape[0], -1, 4).sum(-1) == 2).all()):
                w = v if v.stride(1) == 1 else v.contiguous()
                # (what is stored is W . mask: under SparseLoRA the pruned positions of `weight` are zero already, lora.py:362)
                values, meta = ops.pack_24(w, m if m.stride(1) == 1 else m.contiguous())
                out[k[:-6] + PACKED_VALUES], out[k[:-6] + PACKED_META] = values, meta
                continue
        out[k] = v
    if not keep_masks:
        for k in [k for k in out if k.endswith(".mask") and k[:-4] + PACKED_VALUES in out]:
            del out[k]
    return out


def unpack_state_dict_24(state: dict, device=None) -> dict:
    """The dense state dict back: `<name>.weight` (zeros where nothing was kept) and `<name>.mask` for every packed pair."""
    from . import ops
    out = {}
    for k, v in state.items():
        if k.endswith("." + PACKED_VALUES):
            base = k[:-len(PACKED_VALUES)]
            values, meta = v, state[base + PACKED_META]
            if device is not None:
                values, meta = values.to(device), meta.to(device)
            w, m = ops.unpack_24(values.contiguous(), meta.contiguous())
            out[base + "weight"], out[base + "mask"] = w, m
        elif not k.endswith("." + PACKED_META):
            out.setdefault(k, v)
    return out
