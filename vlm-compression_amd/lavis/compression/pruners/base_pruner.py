"""BasePruner (reference: lavis/compression/pruners/base_pruner.py:7-82)."""
import torch

from lavis.compression.pruners.utils import print_time


class BasePruner:
    def __init__(self, model, data_loader, is_strct_pruning, keep_indices_or_masks_cache, importance_scores_cache,
                 is_global, num_samples):
        self.model = model
        self.data_loader = data_loader
        self.is_strct_pruning = is_strct_pruning
        self.is_global = is_global
        self.num_samples = num_samples
        self.keep_indices_or_masks_cache = keep_indices_or_masks_cache
        self.importance_scores_cache = importance_scores_cache

    def compute_importance_scores(self, model, data_loader, loss_func):
        raise NotImplementedError

    def get_params(self, model):
        named = list(model.named_parameters())
        return [n for n, _ in named], [p for _, p in named]

    def model_setup_and_record_attributes(self, model):
        """base_pruner.py:38-54: cast every state tensor to bf16, enable grads, remember both."""
        dtype_record, requires_grad_record = {}, {}
        for n, p in model.state_dict().items():
            dtype_record[n] = p.data.dtype
            p.data = p.data.type(torch.bfloat16)
        for n, p in model.named_parameters():
            requires_grad_record[n] = p.requires_grad
            p.requires_grad = True
        device = list(self.model.parameters())[0].device
        return dtype_record, requires_grad_record, device

    def model_reset(self, model, dtype_record, requires_grad_record, device):
        for n, p in model.named_parameters():
            p.requires_grad = requires_grad_record[n]
        for n, p in model.state_dict().items():
            p.data = p.data.type(dtype_record[n])
        model.to(device)

    def convert_spec_to_list(self, spec):
        num_layers, res, attn, ffn = spec.split("-")
        return int(num_layers), float(res), float(attn), float(ffn)

    def create_pruned_arch(self, *args, **kwargs):
        return NotImplementedError

    @print_time
    def _prune(self, model, importance_scores, keep_indices_or_masks, prune_spec, ignore_layers, is_global):
        raise NotImplementedError

    @print_time
    def prune(self, importance_scores=None, keep_indices_or_masks=None):
        raise NotImplementedError
