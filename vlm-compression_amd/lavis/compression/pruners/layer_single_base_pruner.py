"""LayerWiseBasePruner + the uniform path of LayerSparsity
(reference: lavis/compression/pruners/layer_single_base_pruner.py:10-108, 240-255).

The non-uniform (ECoFLaP first-order / MeZO) allocation of LayerSparsity
(:257-729) is listed as a next row in SURVEY.md §8(f) and is not built yet: asking for
a `sparsity_ratio_granularity` other than None/"none" raises NotImplementedError.
"""
from lavis.compression.pruners.base_pruner import BasePruner
from lavis.compression.pruners.utils import print_time


class LayerWiseBasePruner(BasePruner):
    def __init__(self, model, data_loader, prune_spec=None, importance_scores_cache=None,
                 keep_indices_or_masks_cache=None, is_strct_pruning=False, num_samples=64, is_global=False,
                 model_prefix="t5_model", sparsity_ratio_granularity=None, max_sparsity_per_layer=0.8,
                 score_method="obd_avg", num_data_first_stage=128, num_noise=1, sparsity_dict=None, noise_eps=1e-3,
                 prune_per_model=False, prune_n=0, prune_m=0, **kwargs):
        super().__init__(model=model, data_loader=data_loader, is_strct_pruning=is_strct_pruning,
                         importance_scores_cache=importance_scores_cache,
                         keep_indices_or_masks_cache=keep_indices_or_masks_cache, is_global=is_global,
                         num_samples=num_samples)
        self.sparsity_ratio_granularity = sparsity_ratio_granularity
        self.max_sparsity_per_layer = max_sparsity_per_layer
        self.score_method = score_method
        self.num_data_first_stage = num_data_first_stage
        self.num_noise = num_noise
        self.sparsity_dict = sparsity_dict
        self.noise_eps = noise_eps
        self.prune_per_model = prune_per_model
        self.prune_spec = prune_spec
        self.model_prefix = model_prefix
        self.prune_n, self.prune_m = prune_n, prune_m
        self.model_stem = getattr(self.model, model_prefix, None)

    def model_setup_and_record_attributes(self, model):
        """layer_single_base_pruner.py:72-87: dtypes are only recorded (no bf16 cast here)."""
        dtype_record, requires_grad_record = {}, {}
        for n, p in model.named_parameters():
            dtype_record[n] = p.data.dtype
        for n, p in model.named_parameters():
            requires_grad_record[n] = p.requires_grad
            p.requires_grad = True
        device = next(iter(model.parameters())).device
        return dtype_record, requires_grad_record, device

    def model_reset(self, model, dtype_record, requires_grad_record, device):
        for n, p in model.named_parameters():
            p.requires_grad = requires_grad_record[n]
        for n, p in model.named_parameters():
            p.data = p.data.type(dtype_record[n])
        model.to(device)


class UniformSparsity:
    """What `return_sparsity()` hands back when no grouping is requested
    (layer_single_base_pruner.py:251-255): every key maps to the same ratio."""

    def __init__(self, ratio):
        self.ratio = ratio

    def __getitem__(self, key):
        return self.ratio


class LayerSparsity:
    def __init__(self, model, data_loader, loss_func, num_samples, original_sparsity, max_sparsity_per_layer=0.8,
                 score_method="obd_avg", num_noise=1, noise_eps=1e-3, layer_to_group_mapping={}, prune_per_model=False,
                 per_model_group=("t5_model", "visual"), per_model_sparsity=()):
        self.model, self.data_loader, self.loss_func = model, data_loader, loss_func
        self.num_samples = num_samples
        self.original_sparsity = original_sparsity
        self.layer_to_group_mapping = layer_to_group_mapping
        self.max_sparsity_per_layer = max_sparsity_per_layer
        self.num_noise, self.noise_eps = num_noise, noise_eps
        self.prune_per_model = prune_per_model
        self.score_method = score_method
        if score_method is not None:
            self.score_compute, self.score_aggregate = score_method.split("_")   # exactly two parts (:144-145)
        assert self.max_sparsity_per_layer >= self.original_sparsity              # (:147)

    @print_time
    def return_sparsity(self):
        mapping = self.layer_to_group_mapping
        print(f"layer_to_group_mapping: {mapping}")
        if mapping is None or len(mapping) == 0:
            return UniformSparsity(self.original_sparsity)
        raise NotImplementedError(
            "non-uniform LayerSparsity (ECoFLaP / MeZO allocation, layer_single_base_pruner.py:257-729) "
            "is outside the built hot path (SURVEY.md §8f)")
