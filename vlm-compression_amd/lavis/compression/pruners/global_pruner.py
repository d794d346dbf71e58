"""Global (whole-model) pruners, drop-in for lavis/compression/pruners/global_pruner.py:49-383:
`blipt5_mag_pruner`, `blipt5_rand_pruner`, `blipt5_aobd_pruner`, `blipt5_mezo_pruner`.

Same constructor, same `prune()` result (model, None), same selection rule; what changes is where
it runs.  The reference moves every score to the CPU, concatenates them and calls torch.topk
(:120-127); here the weights (and gradient statistics) stay on the GPU and ONE call of the C ABI
(`vlmc_score_select`, K17) finds the threshold(s) by radix selection over all layers at once and
multiplies the weights by the masks -- nothing is concatenated, sorted or copied to the host.
"""
import numpy as np
import torch

from lavis.common.registry import registry
from lavis.compression.pruners.layer_single_base_pruner import LayerWiseBasePruner
from lavis.compression.pruners.utils import loss_vision_language, print_time
from vlmc import ops


def _device_flag(device):
    # the reference passes `device != "cpu"` (a torch.device against a str: always True, :289);
    # what it means is "move the batch to the GPU", which is what is done here
    return torch.device(device).type != "cpu"


class BLIPT5GlobalPruner(LayerWiseBasePruner):
    # how the C ABI forms the score from (weight, S): subclasses set it together with score_operands()
    score_mode = "score"

    def __init__(self, model, data_loader, t5_prune_spec=None, vit_prune_spec=None, t5_pruning_method=None,
                 vit_pruning_method=None, t5_importance_scores_cache=None, t5_keep_indices_or_masks_cache=None,
                 vit_importance_scores_cache=None, vit_keep_indices_or_masks_cache=None, importance_scores_cache=None,
                 keep_indices_or_masks_cache=None, is_strct_pruning=False, num_samples=64, is_global=False,
                 t5_model_prefix="t5_model", vit_model_prefix="visual_encoder", sparsity_ratio_granularity=None,
                 max_sparsity_per_layer=0.8, score_method="obd_avg", num_data_first_stage=128, num_noise=1,
                 sparsity_dict=None, prune_per_model=False, iteration=1, **kwargs):
        super().__init__(model=model, data_loader=data_loader, prune_spec=None, is_strct_pruning=is_strct_pruning,
                         importance_scores_cache=importance_scores_cache,
                         keep_indices_or_masks_cache=keep_indices_or_masks_cache, is_global=is_global,
                         num_samples=num_samples, model_prefix="tmp", sparsity_ratio_granularity=sparsity_ratio_granularity,
                         max_sparsity_per_layer=max_sparsity_per_layer, score_method=score_method,
                         num_data_first_stage=num_data_first_stage, num_noise=num_noise, sparsity_dict=sparsity_dict)
        self.t5_prune_spec = t5_prune_spec
        self.vit_prune_spec = vit_prune_spec
        self.t5_model_prefix = t5_model_prefix
        self.vit_model_prefix = vit_model_prefix
        self.prune_per_model = prune_per_model
        self.iteration = iteration

    # ---- scores ------------------------------------------------------------------------------------
    def compute_importance_scores(self, model, data_loader=None, dict_layers_to_prune={}, loss_func=None):
        raise NotImplementedError

    def score_operands(self, model, data_loader, dict_layers_to_prune, loss_func):
        """name -> fp32 operand S of the score (`score_mode` says how it combines with the weight).
        Default: the subclass only provides finished scores."""
        scores = self.compute_importance_scores(model, data_loader, dict_layers_to_prune, loss_func)
        return {k: v for k, v in scores.items() if k in dict_layers_to_prune}

    # ---- selection ---------------------------------------------------------------------------------
    @staticmethod
    def _rank(p, numel):
        k = int(p * numel)
        if k <= 0:      # the reference indexes an empty topk result here (:125-126, :141-142)
            raise IndexError("index -1 is out of bounds for dimension 0 with size 0")
        return k

    def _select(self, names, mode, weights, operands, prev, p, scope_of, max_sparsity_per_layer, apply_weights):
        """One `vlmc_score_select` over the named tensors; `scope_of(name)` -> hashable scope label."""
        labels, scopes = [], []
        for n in names:
            lab = scope_of(n)
            if lab not in labels:
                labels.append(lab)
            scopes.append(labels.index(lab))
        sizes = [0] * len(labels)
        ref = [weights[i] if weights is not None else operands[i] for i in range(len(names))]
        for s, t in zip(scopes, ref):
            sizes[s] += t.numel()
        ks = [self._rank(p, n) for n in sizes]
        protect = [int(t.numel() * (1 - max_sparsity_per_layer)) if max_sparsity_per_layer is not None else 0 for t in ref]
        protect = [max(k, 0) for k in protect]
        return ops.score_select(weights, mode, scopes=scopes, scope_ks=ks, scores=operands, prev_keeps=prev,
                                protect_ks=protect, apply_weights=apply_weights)

    def get_mask(self, importance_scores, p, max_sparsity_per_layer):
        """:107-133 on finished scores (fp32 tensors on the GPU): masks of the scores' dtype.
        (The reference also overwrites the protected scores with FLT_MAX in place; the scores are
        left untouched here.)"""
        names = list(importance_scores)
        vals = [importance_scores[k].contiguous() for k in names]
        keeps = self._select(names, "score", None, vals, None, p, lambda n: 0, max_sparsity_per_layer, False)
        return {k: kp.to(v.dtype) for k, kp, v in zip(names, keeps, vals)}

    def get_layerwise_mask(self, importance_scores, p):
        """:135-148."""
        names = list(importance_scores)
        vals = [importance_scores[k].contiguous() for k in names]
        keeps = self._select(names, "score", None, vals, None, p, lambda n: n, None, False)
        return {k: kp.to(v.dtype) for k, kp, v in zip(names, keeps, vals)}

    def forward_to_cache(self, model, batch, device):
        return model(batch)

    def _scope_fn(self):
        if self.is_global and not self.prune_per_model:
            print("global")
            return lambda n: 0
        if self.is_global and self.prune_per_model:
            print("model-level global")
            return lambda n: 0 if n.startswith(self.vit_model_prefix) else 1
        print("layer-wise")
        return lambda n: n

    def global_iterative_pruning(self, target_sparsity, dict_layers_to_prune, iteratation=1, max_sparsity_per_layer=1.0):
        """:153-201.  Scores, thresholds, masks and the weight update of one iteration are one
        `vlmc_score_select` call; the masks of the previous iteration ride along as `prev_keep`."""
        keeps = None
        for i in range(1, iteratation + 1):
            p_i = target_sparsity ** (iteratation / i)
            operands = self.score_operands(self.model, self.data_loader, dict_layers_to_prune, loss_vision_language)
            names = [k for k in dict_layers_to_prune if self.score_mode == "weight" or k in operands]
            if self.is_global and self.prune_per_model:
                # vision masks first, then language (:176-184); other prefixes are dropped as in the reference
                names = ([k for k in names if k.startswith(self.vit_model_prefix)] +
                         [k for k in names if k.startswith(self.t5_model_prefix) and not k.startswith(self.vit_model_prefix)])
            scope_of = self._scope_fn()
            params = [dict_layers_to_prune[k].data for k in names]
            S = [operands[k].contiguous() if self.score_mode != "weight" else None for k in names]
            per_layer_scalar = self.score_mode == "score" and any(s.numel() != w.numel() for s, w in zip(S, params))
            prev = [keeps[k] for k in names] if keeps is not None else None
            # global masks honour the per-layer cap, layer-wise ones do not (:171-186)
            cap = max_sparsity_per_layer if self.is_global else None
            if per_layer_scalar:
                # one score per layer (MeZO, :379-381): the mask broadcasts over the layer (:188-190)
                got = self._select(names, "score", None, S, prev, p_i, scope_of, cap, False)
                for w, kp in zip(params, got):
                    w.mul_(kp.to(w.dtype))
            else:
                for w in params:
                    if not w.is_contiguous():
                        raise ValueError("global pruners expect contiguous parameters")
                got = self._select(names, self.score_mode, params, S, prev, p_i, scope_of, cap, True)
            keeps = dict(zip(names, got))
            print(f"Step {i}, target sparsity: {p_i:.4f}")
        for k, v in self.model.named_parameters():
            print(k, " sparsity: ", (v == 0).float().sum() / v.numel())
        self.masks = keeps
        return self.model

    @print_time
    def prune(self, importance_scores=None, keep_indices_or_masks=None):
        """:203-243."""
        print("In: ", self.pruner_name)
        dtype_record, requires_grad_record, device = self.model_setup_and_record_attributes(self.model)
        if self.t5_prune_spec is None or self.vit_prune_spec is None:
            return self.model, None
        _, vit_keep_ratio, _, _ = self.convert_spec_to_list(self.vit_prune_spec)
        _, t5_keep_ratio, _, _ = self.convert_spec_to_list(self.t5_prune_spec)
        vit_keep_ratio = min(t5_keep_ratio, vit_keep_ratio)

        def check(name, v):
            return (len(v.shape) == 2 and ".block" in name and "relative_attention_bias.weight" not in name and
                    (name.startswith(self.t5_model_prefix) or name.startswith(self.vit_model_prefix)))

        parameters_to_prune = {k: v for k, v in self.model.named_parameters() if check(k, v)}
        self.model = self.global_iterative_pruning(1 - vit_keep_ratio, parameters_to_prune, iteratation=self.iteration,
                                                   max_sparsity_per_layer=1.0)
        self.model_reset(self.model, dtype_record, requires_grad_record, device)
        return self.model, None


@registry.register_pruner("blipt5_mag_pruner")
class BLIPT5MagPruner(BLIPT5GlobalPruner):
    """Score = the fp32 value of the weight, SIGNED as in the reference (:255)."""
    pruner_name = "blipt5_mag_pruner"
    score_mode = "weight"

    def compute_importance_scores(self, model, data_loader=None, dict_layers_to_prune={}, loss_func=None):
        return {k: v.data.float() for k, v in model.named_parameters()}

    def score_operands(self, model, data_loader, dict_layers_to_prune, loss_func):
        return {}                                           # the kernel reads the weights themselves


@registry.register_pruner("blipt5_rand_pruner")
class BLIPT5RandPruner(BLIPT5GlobalPruner):
    pruner_name = "blipt5_rand_pruner"
    score_mode = "score"

    def compute_importance_scores(self, model, data_loader=None, dict_layers_to_prune={}, loss_func=None):
        return {k: torch.randn_like(v.data).float() for k, v in model.named_parameters()}


@registry.register_pruner("blipt5_aobd_pruner")
class BLIPT5AOBDPruner(BLIPT5GlobalPruner):
    """Score = |w| * |mean over batches of |dL/dw|| (:266-313); the product is formed inside the kernel."""
    pruner_name = "blipt5_aobd_pruner"
    score_mode = "absw_score"

    @print_time
    def mean_abs_gradients(self, model, data_loader, dict_layers_to_prune, loss_func):
        names, params = [], []
        for k, v in model.named_parameters():
            if k in dict_layers_to_prune:
                names.append(k)
                params.append(v)
        grads = {k: 0 for k in names}
        device = next(iter(model.parameters())).device
        accum_samples = 0
        current_batch_index = 0
        for d in data_loader:
            if accum_samples >= self.num_samples:
                break
            loss, batch_len = loss_func(model, d, _device_flag(device))
            accum_samples += batch_len
            current_batch_index += 1
            gs = torch.autograd.grad(loss, params)
            for k, g in zip(names, gs):
                grads[k] = grads[k] + g.data.float().abs()
        for k in names:
            grads[k] = grads[k] / current_batch_index      # normalised by batches, not samples (:305-309)
        return names, params, grads

    def score_operands(self, model, data_loader, dict_layers_to_prune, loss_func):
        return self.mean_abs_gradients(model, data_loader, dict_layers_to_prune, loss_func)[2]

    def compute_importance_scores(self, model, data_loader=None, dict_layers_to_prune={}, loss_func=None):
        names, params, grads = self.mean_abs_gradients(model, data_loader, dict_layers_to_prune, loss_func)
        return {k: v.data.float().abs() * grads[k].abs() for k, v in zip(names, params)}


@registry.register_pruner("blipt5_mezo_pruner")
class BLIPT5AMeZoPruner(BLIPT5GlobalPruner):
    """One zeroth-order score per LAYER (:316-383): whole layers are kept or dropped."""
    pruner_name = "blipt5_mezo_pruner"
    score_mode = "score"

    def zo_perturb_parameters(self, params, random_seed=1, scaling_factor=1, zo_eps=1e-3):
        torch.manual_seed(random_seed)
        for param in params:
            z = torch.normal(mean=0, std=1, size=param.data.size(), device=param.data.device, dtype=param.data.dtype)
            param.data = param.data + scaling_factor * z * zo_eps

    @print_time
    def compute_importance_scores(self, model, data_loader=None, dict_layers_to_prune={}, loss_func=None):
        names, params = [], []
        model.eval()
        for k, v in model.named_parameters():
            if k in dict_layers_to_prune:
                names.append(k)
                params.append(v)
        device = next(iter(model.parameters())).device
        totals = {k: 0.0 for k in names}
        zo_eps = 1e-3
        for i, (name, param) in enumerate(zip(names, params)):
            print(i, name)
            accum_samples = 0
            for d in data_loader:
                if accum_samples >= self.num_samples:
                    break
                per_batch = 0
                for _ in range(self.num_noise):
                    if accum_samples >= self.num_samples:
                        break
                    seed = np.random.randint(1000000000)
                    self.zo_perturb_parameters([param], random_seed=seed, scaling_factor=1, zo_eps=zo_eps)
                    with torch.no_grad():
                        loss1, batch_len = loss_func(model, d, _device_flag(device))
                    self.zo_perturb_parameters([param], random_seed=seed, scaling_factor=-2, zo_eps=zo_eps)
                    with torch.no_grad():
                        loss2, batch_len = loss_func(model, d, _device_flag(device))
                    self.zo_perturb_parameters([param], random_seed=seed, scaling_factor=1, zo_eps=zo_eps)
                    accum_samples += batch_len
                    projected_grad = ((loss1 - loss2) / (2 * zo_eps)).item()
                    torch.manual_seed(seed)
                    per_batch += abs(projected_grad)
                # the reference accumulates through a 1-element fp32 tensor (:377)
                totals[name] = totals[name] + torch.tensor([per_batch], dtype=torch.float32).abs()
        return {k: (totals[k] if torch.is_tensor(totals[k]) else torch.zeros(1)).abs().to(device) for k in names}
