"""LayerWiseBasePruner and LayerSparsity, the ECoFLaP first-stage allocation of per-layer sparsity
(reference: lavis/compression/pruners/layer_single_base_pruner.py:10-108, 111-728).
"""
import torch

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
        """layer_single_base_pruner.py:72-87: dtypes are only recorded (no bf16 cast here).  One walk over the parameters
        instead of the reference's three: on one rank's share of the calibration set the host, not the GPU, sets the pace
        (profiles/r04_scaling_floor.md)."""
        dtype_record, requires_grad_record, device = {}, {}, None
        for n, p in model.named_parameters():
            dtype_record[n] = p.data.dtype
            requires_grad_record[n] = p.requires_grad
            p.requires_grad = True
            if device is None:
                device = p.device
        if device is None:
            raise StopIteration("model without parameters")       # (what the reference's next(iter(...)) raises)
        return dtype_record, requires_grad_record, device

    def model_reset(self, model, dtype_record, requires_grad_record, device):
        """layer_single_base_pruner.py:89-97: requires_grad and dtypes back to what was recorded, the model to its device.
        `p.data.type(dtype)` returns the tensor itself when the dtype is unchanged and `model.to(device)` is a walk that moves
        nothing when every tensor is there already: both are skipped in exactly those cases."""
        moved = False
        for n, p in model.named_parameters():
            p.requires_grad = requires_grad_record[n]
            if p.data.dtype != dtype_record[n]:
                p.data = p.data.type(dtype_record[n])
            moved = moved or p.device != device
        if moved or any(b.device != device for b in model.buffers()):
            model.to(device)
        # what the replay engine kept for THIS prune (graph proxies of finished towers, their recorded outputs, the list of
        # finished towers): released with it, so a later prune() -- or a training stage -- starts from the model alone
        self.__dict__.pop("_proxy_cache", None)
        self.__dict__.pop("_done_towers", None)


class UniformSparsity:
    """What `return_sparsity()` hands back when no grouping is requested
    (layer_single_base_pruner.py:251-255): every key maps to the same ratio."""

    def __init__(self, ratio):
        self.ratio = ratio

    def __getitem__(self, key):
        return self.ratio


class LayerSparsity:
    """Per-layer / per-block sparsity allocation (ECoFLaP's first stage) --
    layer_single_base_pruner.py:111-728.  `return_sparsity()` maps every parameter name of
    `layer_to_group_mapping` to the sparsity of its group such that the kept-parameter budget
    `int(total * (1 - original_sparsity))` is shared out in proportion to the groups' importance
    scores, no group exceeding `max_sparsity_per_layer`.

    Score methods (`score_method = "<compute>_<aggregate>"`, aggregate `avg` divides a group's score
    by its parameter count, :299-300):
      obd / aobd / gradient          first-order, one backward per calibration batch (:423-477)
      mezo-* / lmezo-* / olmezo-*    zeroth-order (MeZO): two forwards per perturbation (:495-728)
      real*                          iterative global magnitude-style pruning on the scores (:192-239)

    Literal quirks kept because they decide the numbers (SURVEY.md Appendix B.1 style):
      * `"obd" in score_compute` is tested first (:466), so `aobd` takes the obd formula
        w^2 * mean|g| (its own branch is dead code);
      * the budget solver runs in float32 tensors and its over-budget correction ADDS where it means
        to subtract (:381) -- both reproduced, they change the allocation in the last digits;
      * `lmezo` overwrites `num_samples` with 8 and uses 4 perturbations (:597-599).

    What differs: gradients and scores stay on the parameters' device (the reference copies every
    gradient to the host, 15 GB per batch for FlanT5-XL); the arithmetic is the same elementwise fp32.
    """

    def __init__(self, model, data_loader, loss_func, num_samples, original_sparsity, max_sparsity_per_layer=0.8,
                 score_method="obd_avg", num_noise=1, noise_eps=1e-3, layer_to_group_mapping={}, prune_per_model=False,
                 per_model_group=("t5_model", "visual"), per_model_sparsity=()):
        self.importance_measure = {}
        self.model, self.data_loader, self.loss_func = model, data_loader, loss_func
        self.num_samples = num_samples
        self.original_sparsity = original_sparsity
        self.layer_to_group_mapping = layer_to_group_mapping
        self.max_sparsity_per_layer = max_sparsity_per_layer
        self.num_noise, self.noise_eps = num_noise, noise_eps
        self.prune_per_model = prune_per_model
        self.score_method = score_method
        self.per_model_group = list(per_model_group)
        self.per_model_sparsity = list(per_model_sparsity)
        if score_method is not None:
            self.score_compute, self.score_aggregate = score_method.split("_")   # exactly two parts (:144-145)
        assert self.max_sparsity_per_layer >= self.original_sparsity              # (:147)

    # ---- masks from scores (used by the "real*" iterative variant) ------------------------------------
    def get_mask(self, importance_scores, p, max_sparsity_per_layer):
        """Global threshold at fraction p over all scores, after protecting each layer's top
        (1 - max_sparsity) share with the dtype's max (:149-175)."""
        for k, v in importance_scores.items():
            protect = int(v.numel() * (1 - max_sparsity_per_layer))
            if protect > 0:
                kth = torch.topk(v.flatten(), protect, largest=True)[0][-1]
                v[torch.where(v >= kth)] = torch.finfo(v.dtype).max
        flat = torch.cat([t.flatten() for t in importance_scores.values()])
        thr = torch.topk(flat, int(p * flat.numel()), largest=False)[0][-1]
        return {k: (v > thr).type(v.dtype) for k, v in importance_scores.items()}

    def get_layerwise_mask(self, importance_scores, p):
        masks = {}
        for k, v in importance_scores.items():
            flat = v.flatten()
            thr = torch.topk(flat, int(p * flat.numel()), largest=False)[0][-1]
            masks[k] = (v > thr).type(v.dtype)
        return masks

    def global_iterative_pruning(self, target_sparsity, dict_layers_to_prune, iteratation=1, max_sparsity_per_layer=1.0):
        """`real*` scores (:192-239): prune on the scores in `iteratation` steps of growing sparsity, read off
        the per-layer zero fraction, restore the weights."""
        named = [(k, v) for k, v in self.model.named_parameters() if k in dict_layers_to_prune]
        backup = {k: v.data.clone() for k, v in named}
        masks = None
        for i in range(1, iteratation + 1):
            p_i = target_sparsity ** (iteratation / i)
            measure = {k: v for k, v in self.compute_importance_scores(dict_layers_to_prune).items()
                       if k in dict_layers_to_prune}
            if masks is not None:
                for k in measure:
                    measure[k] *= masks[k]
            print("global")
            masks = self.get_mask(measure, p_i, max_sparsity_per_layer)
            for k, v in self.model.named_parameters():
                if k in masks:
                    v.data *= masks[k].type(v.dtype).to(v.device)
            print(f"Step {i}, target sparsity: {p_i:.4f}")
        sparsity = {k: ((v == 0).float().sum() / v.numel()).item() for k, v in self.model.named_parameters()}
        for k, v in named:
            v.data = backup[k].to(v.device)
        return sparsity

    # ---- the allocation --------------------------------------------------------------------------------
    @staticmethod
    def _keep_budget_per_group(total_to_keep, group_scores, group_numel, max_sparsity_per_layer):
        """The proportional hand-out of :305-399, in the reference's tensor dtypes (float32 scores,
        int64 -> float32 budgets) and operation order."""
        scores = torch.FloatTensor(list(group_scores.values()))
        numel = torch.LongTensor(list(group_numel.values()))
        keep = torch.zeros_like(scores, dtype=int)
        keep += torch.ceil(numel * (1 - max_sparsity_per_layer)).int()          # floor guaranteed by max sparsity
        while keep.sum() < total_to_keep:
            total_ratio = torch.sum(scores)
            rest = total_to_keep - keep.sum()
            add = torch.ceil((scores / total_ratio) * rest)
            keep = keep + add
            scores[keep >= numel] = 0                                            # full groups stop receiving
            keep = torch.clamp(keep, max=numel)
            if add.sum() == 0:                                                   # stuck: hand the rest out in order
                cur = keep.sum()
                if cur < total_to_keep:
                    need = total_to_keep - cur
                    while need > 0:
                        for idx in torch.where(scores > 0)[0]:
                            can = min(need, numel[idx] - keep[idx])
                            keep[idx] += can
                            need -= can
                            if need == 0:
                                break
            if keep.sum() > total_to_keep:                                       # over budget after the ceil
                excess = keep.sum() - total_to_keep
                while excess > 0:
                    for idx in torch.argsort(keep, descending=True, stable=True):
                        can = min(excess, keep[idx] - (numel[idx] * (1 - max_sparsity_per_layer)).int())
                        keep[idx] += can                                         # sic (:381): adds
                        excess -= can
                        if excess == 0:
                            break
        return {name: torch.clamp(1 - k / n, min=0, max=1).item() for name, k, n in zip(group_numel.keys(), keep, numel)}

    @print_time
    def return_sparsity(self):
        original_sparsity, mapping = self.original_sparsity, self.layer_to_group_mapping
        print(f"layer_to_group_mapping: {mapping}")
        if self.score_compute.startswith("real"):
            return self.global_iterative_pruning(original_sparsity, mapping, iteratation=3, max_sparsity_per_layer=1.0)
        if mapping is None or len(mapping) == 0:
            return UniformSparsity(original_sparsity)
        if len(self.importance_measure) == 0:
            if self.score_compute.startswith("mezo"):
                self.importance_measure = self.compute_importance_scores_mezo_diff(mapping)
            elif self.score_compute.startswith("lmezo"):
                self.importance_measure = self.compute_importance_scores_mezo_layer(mapping)
            elif self.score_compute.startswith("olmezo"):
                self.importance_measure = self.compute_importance_scores_mezo_layer_one(mapping)
            else:
                self.importance_measure = self.compute_importance_scores(mapping)

        groups = {}
        for layer, group in mapping.items():
            groups.setdefault(group, []).append(layer)
        numel = {k: v.numel() for k, v in self.model.named_parameters() if k in mapping}
        total = sum(numel.values())
        total_to_keep = int(total * (1 - original_sparsity))
        group_scores, group_numel = {}, {}
        for group, layers in groups.items():
            score, n = 0, 0
            for layer in layers:
                score += self.importance_measure[layer].sum()
                n += numel[layer]
            if self.score_aggregate == "avg":
                score /= n
            group_scores[group], group_numel[group] = score, n

        if self.prune_per_model:
            group_sparsity = {}
            for prefix, sparsity in zip(self.per_model_group, self.per_model_sparsity):
                print(prefix)
                sub_scores = {k: v for k, v in group_scores.items() if k.startswith(prefix)}
                sub_numel = {k: v for k, v in group_numel.items() if k.startswith(prefix)}
                sub_keep = int(sum(sub_numel.values()) * (1 - sparsity))
                group_sparsity.update(self._keep_budget_per_group(sub_keep, sub_scores, sub_numel, self.max_sparsity_per_layer))
        else:
            group_sparsity = self._keep_budget_per_group(total_to_keep, group_scores, group_numel, self.max_sparsity_per_layer)

        kept = sum((1 - group_sparsity[k]) * group_numel[k] for k in group_numel)
        print(f"compute_total_keep_parameters: {kept}, total_parameters_to_keep: {total_to_keep}")
        layer_sparsity = {layer: group_sparsity[group] for layer, group in mapping.items()}
        print(f"layer_sparsity: {layer_sparsity}")
        return layer_sparsity

    # ---- importance scores -----------------------------------------------------------------------------
    def _selected(self, mapping):
        names, params = [], []
        for k, v in self.model.named_parameters():
            if k in mapping:
                names.append(k)
                params.append(v)
        return names, params

    def _from_gradient_measure(self, prefix, names, params, grads):
        """Final formula shared by the zeroth-order variants (:566-571, :645-650, :722-727)."""
        if self.score_compute == prefix + "-gradient":
            return {k: grads[k].abs() for k in names}
        # the per-layer variants keep ONE host scalar per layer (shape [1], :642,:719); the reference brings the weights
        # to the host to combine (`v.cpu().data.float()`), here the scalar goes to the weights' device instead
        g = {k: (grads[k].to(v.device) if torch.is_tensor(grads[k]) else grads[k]) for k, v in zip(names, params)}
        if self.score_compute == prefix + "-aobd":
            return {k: v.data.float().abs() * g[k].abs() for k, v in zip(names, params)}
        if self.score_compute == prefix + "-obd":
            return {k: v.data.float() ** 2 * g[k] ** 2 for k, v in zip(names, params)}
        raise UnboundLocalError(f"score method {self.score_compute!r} defines no importance measure")   # as the reference fails

    @print_time
    def compute_importance_scores(self, layer_to_group_mapping):
        """First-order scores (:423-477): mean over the calibration batches of g^2 (`obd`) or |g| (otherwise),
        combined with the weights."""
        names, params = self._selected(layer_to_group_mapping)
        acc = {k: 0 for k in names}
        device = next(iter(self.model.parameters())).device
        seen, n_batches = 0, 0
        for batch in self.data_loader:
            if seen >= self.num_samples:
                break
            loss, batch_len = self.loss_func(self.model, batch, device.type != "cpu")
            seen += batch_len
            n_batches += 1
            grads = torch.autograd.grad(loss, params)
            assert len(grads) == len(names) == len(params)
            for k, g in zip(names, grads):
                g = g.data.float()
                acc[k] += g ** 2 if self.score_compute == "obd" else g.abs()
        for k in names:
            acc[k] /= n_batches                     # batches, not samples: the loss is already a batch mean (:456-460)
        if "obd" in self.score_compute:             # also true for "aobd" (:466)
            return {k: (v.data.float() ** 2) * acc[k] for k, v in zip(names, params)}
        if "aobd" in self.score_compute:            # unreachable, kept for the record (:469)
            return {k: v.data.float().abs() * acc[k].abs() for k, v in zip(names, params)}
        if "gradient" in self.score_compute:
            return {k: acc[k].abs() for k in names}
        raise UnboundLocalError(f"score method {self.score_compute!r} defines no importance measure")

    def zo_perturb_parameters(self, params, random_seed=1, scaling_factor=1, zo_eps=1e-3):
        """theta += scaling_factor * z * eps with z ~ N(0, 1) re-drawn from `random_seed` (MeZO, :480-493)."""
        torch.manual_seed(random_seed)
        for param in params:
            z = torch.normal(mean=0, std=1, size=param.data.size(), device=param.data.device, dtype=param.data.dtype)
            param.data = param.data + scaling_factor * z * zo_eps

    def _projected_grad(self, params, batch, device, seed, zo_eps):
        """(loss(theta + eps z) - loss(theta - eps z)) / (2 eps); theta restored afterwards."""
        self.zo_perturb_parameters(params, random_seed=seed, scaling_factor=1, zo_eps=zo_eps)
        with torch.no_grad():
            loss1, batch_len = self.loss_func(self.model, batch, device.type != "cpu")
        self.zo_perturb_parameters(params, random_seed=seed, scaling_factor=-2, zo_eps=zo_eps)
        with torch.no_grad():
            loss2, batch_len = self.loss_func(self.model, batch, device.type != "cpu")
        self.zo_perturb_parameters(params, random_seed=seed, scaling_factor=1, zo_eps=zo_eps)
        return ((loss1 - loss2) / (2 * zo_eps)).item(), batch_len

    def compute_importance_scores_mezo_diff(self, layer_to_group_mapping):
        """`mezo-*` (:495-575): MeZO-SGD steps on all selected parameters at once; the score basis is the mean
        absolute weight drift."""
        import numpy as np
        self.model.eval()
        names, params = self._selected(layer_to_group_mapping)
        backup = {k: v.data.clone() for k, v in zip(names, params)}
        total = sum(v.numel() for v in params)
        device = next(iter(self.model.parameters())).device
        zo_eps, lr = self.noise_eps, 1 / total * 1e-3
        seen, n_batches = 0, 0
        for batch in self.data_loader:
            if seen >= self.num_samples:
                break
            print(seen)
            seed = np.random.randint(1000000000)
            projected, batch_len = self._projected_grad(params, batch, device, seed, zo_eps)
            seen += batch_len
            n_batches += 1
            torch.manual_seed(seed)
            for param in params:
                z = torch.normal(mean=0, std=1, size=param.data.size(), device=param.data.device, dtype=param.data.dtype)
                param.data = param.data - projected * z * lr
        grads = {}
        for k, p_ in zip(names, params):
            grads[k] = (p_.data - backup[k]).float().abs() / n_batches
            p_.data = backup[k]
        return self._from_gradient_measure("mezo", names, params, grads)

    def _per_layer_mezo(self, layer_to_group_mapping, n_mezo, absolute):
        import numpy as np
        self.model.eval()
        names, params = self._selected(layer_to_group_mapping)
        device = next(iter(self.model.parameters())).device
        grads = {k: 0 for k in names}
        for i, (name, param) in enumerate(zip(names, params)):
            print(i, name)
            seen = 0
            for batch in self.data_loader:
                if seen >= self.num_samples:
                    break
                per_batch = 0
                for _ in range(n_mezo):
                    if seen >= self.num_samples:
                        break
                    seed = np.random.randint(1000000000)
                    projected, batch_len = self._projected_grad([param], batch, device, seed, self.noise_eps)
                    seen += batch_len
                    torch.manual_seed(seed)
                    per_batch += abs(projected) if absolute else projected
                grads[name] += torch.FloatTensor([per_batch]).abs()
        print(grads)
        return names, params, grads

    def compute_importance_scores_mezo_layer(self, layer_to_group_mapping):
        """`lmezo-*` (:577-655): one scalar per layer, 4 perturbations per batch, 8 samples (both hard-coded there)."""
        self.num_samples = 8
        names, params, grads = self._per_layer_mezo(layer_to_group_mapping, 4, absolute=False)
        return self._from_gradient_measure("lmezo", names, params, grads)

    def compute_importance_scores_mezo_layer_one(self, layer_to_group_mapping):
        """`olmezo-*` (:657-728, ECoFLaP's zeroth-order score): one scalar per layer, `num_noise` perturbations per
        batch, absolute projected gradients."""
        names, params, grads = self._per_layer_mezo(layer_to_group_mapping, self.num_noise, absolute=True)
        return self._from_gradient_measure("olmezo", names, params, grads)
