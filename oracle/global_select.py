"""CPU oracle of the global pruners' mask selection -- TEST INFRASTRUCTURE ONLY (tests/, smoke(),
bench cpu_baseline); the product path never imports this.

Restates lavis/compression/pruners/global_pruner.py of the reference:
  get_mask               :107-133
  get_layerwise_mask     :135-148
  global_iterative_pruning :153-201
  scores                 :255 (magnitude: the SIGNED fp32 weight), :262 (random), :311 (aobd: |w| * |mean|g||)
Pinned by tests/golden/global.npz, generated from the reference's own classes
(tests/golden/make_golden.py::gen_global).
"""
from __future__ import annotations

import torch


def kth_smallest(flat: torch.Tensor, k: int) -> torch.Tensor:
    """`torch.topk(flat, k, largest=False)[0][-1]`: the k-th smallest value (NaN sorts last).
    k == 0 indexes an empty tensor in the reference (:125-126) -> IndexError."""
    if k <= 0:
        raise IndexError("index -1 is out of bounds for dimension 0 with size 0")
    return torch.sort(flat)[0][k - 1]


def protect_top(scores: dict, max_sparsity_per_layer: float) -> dict:
    """:111-118 -- the top (1 - max_sparsity) share of every layer gets FLT_MAX (in place in the
    reference; returned here)."""
    out = {}
    for k, v in scores.items():
        v = v.clone()
        num = int(v.numel() * (1 - max_sparsity_per_layer))
        if num > 0:
            thr = torch.sort(v.flatten(), descending=True)[0][num - 1]
            v[v >= thr] = torch.finfo(v.dtype).max
        out[k] = v
    return out


def get_mask(scores: dict, p: float, max_sparsity_per_layer: float) -> dict:
    scores = protect_top(scores, max_sparsity_per_layer)
    flat = torch.cat([t.flatten() for t in scores.values()])
    thr = kth_smallest(flat, int(p * flat.numel()))
    return {k: (v > thr).to(v.dtype) for k, v in scores.items()}


def get_layerwise_mask(scores: dict, p: float) -> dict:
    out = {}
    for k, v in scores.items():
        flat = v.flatten()
        out[k] = (v > kth_smallest(flat, int(p * flat.numel()))).to(v.dtype)
    return out


def score_magnitude(w):
    return w.float()


def score_aobd(w, mean_abs_grad):
    return w.float().abs() * mean_abs_grad.float().abs()


def iterative_prune(params: dict, score_fn, target_sparsity: float, iteration: int = 1, *, is_global: bool, prune_per_model: bool,
                    max_sparsity_per_layer: float = 1.0, vit_prefix="visual_encoder", t5_prefix="t5_model"):
    """:153-201 on a dict name -> weight tensor (modified in place).  `score_fn(params)` returns
    name -> fp32 score.  Returns the last masks."""
    masks = None
    for i in range(1, iteration + 1):
        p_i = target_sparsity ** (iteration / i)
        scores = {k: v for k, v in score_fn(params).items() if k in params}
        if masks is not None:
            scores = {k: v * masks[k] for k, v in scores.items()}
        if is_global and not prune_per_model:
            masks = get_mask(scores, p_i, max_sparsity_per_layer)
        elif is_global:
            vis = {k: v for k, v in scores.items() if k.startswith(vit_prefix)}
            lang = {k: v for k, v in scores.items() if k.startswith(t5_prefix)}
            masks = get_mask(vis, p_i, max_sparsity_per_layer)
            masks.update(get_mask(lang, p_i, max_sparsity_per_layer))
        else:
            masks = get_layerwise_mask(scores, p_i)
        for k, w in params.items():
            if k in masks:
                w.mul_(masks[k].to(w.dtype))
    return masks
