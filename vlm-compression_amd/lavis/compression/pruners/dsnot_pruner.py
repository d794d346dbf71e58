"""DSnoT pruners behind the reference's `lavis.compression` API, running on the gfx950 kernels
(`vlmc.dsnot`): `t5_dsnot_pruner`, `vit_dsnot_pruner`, `blipt5_dsnot_pruner`.

Reference: lavis/compression/pruners/dsnot_pruner.py (T5LayerDSnoTPruner :107-858,
VITLayerDSnoTPruner :861-1596, BLIPT5LayerDSnoTPruner :1599-1878).  Same registry names,
constructor kwargs, `prune(...) -> (model, sparsity_dict | None)` contract and side effects
(`module.mask` bool [out,in] True = keep; weights zeroed in place unless `lora_model`; no
`importance_score` -- the reference's assignment is commented out, :365).

  reference op sequence (per linear)                        here
  -------------------------------------------------------  -----------------------------------
  hook: var / norm**2 / sum over tokens, running means       vlmc_act_moments (one launch per
  (WrappedGPT.add_batch :79-101)                             DISTINCT input tensor) +
                                                             vlmc_dsnot_stats_update
  three full sorts, reorder_indices (python loop over        vlmc_wanda_select (initial mask) +
  rows), <= max_cycle_time cycles of ~20 small kernels       vlmc_dsnot_refine (one launch, one
  (:553-751 / n:m :407-552; the ViT copy is identical and    workgroup per row) + vlmc_dsnot_apply
  per-row as well, :1285-1482)

Differences that are visible to a caller:
* the standalone tower pruners accept `initial_method` (the reference reads
  `self.initial_method` in `_prune` but only the BLIP pruner sets it, :1675, so its tower
  classes cannot run on their own);
* `initial_method="sparsegpt"` raises (the reference's branch needs a Hessian that its hook
  never accumulates, :85-86 commented out);
* `max_cycle_time >= in_features` raises instead of indexing out of range.

There is no CPU fallback: without the HIP library or a GPU model the pruner raises.
"""
from __future__ import annotations

import os

import torch

from lavis.common.registry import registry
from lavis.compression.pruners import calibration as cal
from lavis.compression.pruners.layer_single_base_pruner import LayerSparsity, LayerWiseBasePruner
from lavis.compression.pruners.utils import print_time

_VERBOSE = os.environ.get("VLMC_VERBOSE", "0") != "0"


class DsnotStatCollector:
    """Forward hooks on the linears of one block (dsnot_pruner.py:338-356).  Linears fed by the
    very same tensor (q/k/v, wi_0/wi_1) share one moments launch and one statistic."""

    def __init__(self, subset):
        from vlmc import dsnot
        self._dsnot = dsnot
        self.subset = subset
        self.calls = {n: [] for n in subset}              # per linear: one record (of one or more samples) per hook call
        self._cache = {}
        self._cur = -1
        self.handles = [m.register_forward_hook(self._make_hook(n)) for n, m in subset.items()]

    def _make_hook(self, name):
        def hook(_module, inp, _out):
            x = inp[0].data
            if x.dim() == 2:
                x = x.unsqueeze(0)
            key = (x.data_ptr(), tuple(x.shape), tuple(x.stride()), x.dtype, x._version)
            hit = self._cache.get(key)
            if hit is None:
                stacked = cal.stacked_samples()              # batched replay: one record per stacked calibration sample
                calls, _b0, idx = stacked if stacked and stacked[0] * stacked[1] == x.shape[0] else (1, x.shape[0], (self._cur,))
                st = self._dsnot.DsnotInputStat(x.shape[-1], x.device)
                st.add_calls(x, calls, idx)                  # ONE launch: sum of squares, sum, variance over tokens, per sample
                hit = (x, st)                                # `x` stays referenced until the next sample (see wanda collector)
                self._cache[key] = hit
            self.calls[name].append(hit[1])
        return hook

    def next_sample(self, j=None):
        self._cache.clear()
        self._cur = self._cur + 1 if j is None else j

    def close(self):
        for h in self.handles:
            h.remove()
        self.handles = []
        self._cache.clear()

    def finalize(self):
        """{name: DsnotInputStat}; linears whose hooks saw identical tensors share the object."""
        shared, out, order = {}, {}, []
        for name, calls in self.calls.items():
            sig = tuple(id(c) for c in calls)                # (DsnotInputStat.ordered() restores the reference's sample order)
            st = shared.get(sig)
            if st is None:
                mod = self.subset[name]
                st = self._dsnot.DsnotInputStat(mod.weight.shape[1], mod.weight.device)
                for c in calls:
                    st.extend(c)
                shared[sig] = st
                order.append(st)
            out[name] = st
        self._dsnot.gather_stats(order)
        return out


class _DsnotBlockMixin:
    """Per-block DSnoT step; the reference's T5/LLM and ViT `_prune` bodies are the same code."""

    def _dsnot_block(self, i, subset, run_pass, n_inps, batch0, *, module_to_process, model_prefix, sparsity_ratio,
                     lora_model):
        from vlmc import dsnot
        col = DsnotStatCollector(subset)
        try:
            run_pass(col.next_sample, outputs=False)
        finally:
            col.close()
        stats = col.finalize()
        for name, mod in subset.items():
            st = stats[name]
            assert st.nsamples == n_inps * batch0                               # dsnot_pruner.py:359
            W = mod.weight.data
            if not W.is_contiguous():
                raise RuntimeError(f"{name}: weight must be contiguous")
            if self.prune_n != 0:
                ratio = None
                if _VERBOSE:
                    print(f"pruning {model_prefix} layer {i} {name} at structured {self.prune_n}:{self.prune_m} sparsity")
            else:
                ratio = sparsity_ratio[f"{module_to_process}.{i}.{name}.weight"]
                if _VERBOSE:
                    print(f"pruning {model_prefix} layer {i} {name} at unstructured {ratio} sparsity")
            keep = dsnot.prune_linear(W, st, ratio, prune_n=self.prune_n, prune_m=self.prune_m,
                                      initial_method=self.initial_method, without_DSnoT=self.without_DSnoT,
                                      max_cycle_time=self.max_cycle_time, update_threshold=self.update_threshold,
                                      pow_of_var_regrowing=self.pow_of_var_regrowing,
                                      without_same_sign=self.without_same_sign, apply_zero=not lora_model)
            if keep is None:                                                     # ratio == 0: `continue` (:560-561)
                continue
            setattr(mod, "mask", keep)                                           # True = keep (:753)

    def _set_dsnot_options(self, initial_method, skip_layer, skip_sub_layer, pow_of_var_regrowing, max_cycle_time,
                           update_threshold, without_same_sign, without_DSnoT):
        self.initial_method = initial_method
        self.pow_of_var_regrowing = pow_of_var_regrowing
        self.without_same_sign = without_same_sign
        self.without_DSnoT = without_DSnoT
        self.update_threshold = update_threshold
        self.skip_layer = skip_layer
        self.skip_sub_layer = skip_sub_layer
        self.max_cycle_time = max_cycle_time


def _base_kwargs(loc):
    keys = ["model", "data_loader", "prune_spec", "is_strct_pruning", "importance_scores_cache",
            "keep_indices_or_masks_cache", "is_global", "num_samples", "model_prefix", "sparsity_ratio_granularity",
            "max_sparsity_per_layer", "score_method", "num_data_first_stage", "num_noise", "sparsity_dict", "noise_eps",
            "prune_per_model", "prune_n", "prune_m"]
    return {k: loc[k] for k in keys}


@registry.register_pruner("t5_dsnot_pruner")
class T5LayerDSnoTPruner(LayerWiseBasePruner, _DsnotBlockMixin):
    """T5 / OPT / LLaMA tower (dsnot_pruner.py:107-858)."""
    pruner_name = "t5_dsnot_pruner"

    def __init__(self, model, data_loader, prune_spec=None, importance_scores_cache=None,
                 keep_indices_or_masks_cache=None, is_strct_pruning=False, num_samples=64, is_global=False,
                 model_prefix="t5_model", sparsity_ratio_granularity=None, max_sparsity_per_layer=0.8,
                 score_method="obd_avg", num_data_first_stage=128, num_noise=1, sparsity_dict=None, noise_eps=1e-3,
                 prune_per_model=False, skip_layer=None, skip_sub_layer=None, pow_of_var_regrowing=1.,
                 max_cycle_time=1e2, update_threshold=0.1, without_same_sign=True, without_DSnoT=False, prune_n=0,
                 prune_m=0, initial_method="wanda", **kwargs):
        super().__init__(**_base_kwargs(locals()))
        self._set_dsnot_options(initial_method, skip_layer, skip_sub_layer, pow_of_var_regrowing, max_cycle_time,
                                update_threshold, without_same_sign, without_DSnoT)

    def forward_to_cache(self, model, batch, lora_model=False):
        return model(batch)

    def get_sparsity(self, original_sparsity, sparsity_ratio_granularity=None):
        from lavis.compression.pruners.wanda_pruner import uniform_or_layer_sparsity
        return uniform_or_layer_sparsity(self, original_sparsity, sparsity_ratio_granularity)

    def check_sparsity(self, model, module_to_process="encoder.block"):
        layers = cal.get_module_recursive(model, module_to_process)
        zeros = total = 0
        for layer in layers:
            for mod in cal.find_layers(layer).values():
                zeros += (mod.weight.data == 0).sum().item()
                total += mod.weight.numel()
        return float(zeros) / total

    def prepare_calibration_input_encoder(self, model, dataloader, model_prefix, n_samples,
                                          module_to_process="encoder.block", lora_model=False):
        cfg = getattr(model, model_prefix).config
        use_cache, cfg.use_cache = cfg.use_cache, False
        try:
            return cal.capture_block_inputs(model, dataloader, n_samples, module_to_process, self.forward_to_cache,
                                            lora_model, vit=False, model_prefix=self.model_prefix, done_towers=getattr(self, "_done_towers", None),
                                        proxy_cache=self.__dict__.setdefault("_proxy_cache", {}))
        finally:
            cfg.use_cache = use_cache

    @print_time
    def _prune(self, model, dataloader, model_prefix, module_to_process="encoder.block", n_samples=64,
               sparsity_ratio=0.5, lora_model=False):
        cfg = getattr(model, model_prefix).config
        use_cache, cfg.use_cache = cfg.use_cache, False
        with torch.no_grad():
            inps, outs, caches = self.prepare_calibration_input_encoder(model, dataloader, model_prefix, n_samples,
                                                                        module_to_process, lora_model)
        n_inps, batch0 = len(inps) * cal.calibration_shard()[1], inps[0].shape[0]

        def prune_block(i, layer, subset, run_pass, state):
            self._dsnot_block(i, subset, run_pass, n_inps, batch0, module_to_process=module_to_process,
                              model_prefix=model_prefix, sparsity_ratio=sparsity_ratio, lora_model=lora_model)

        cal.walk_blocks(model, inps, outs, caches, module_to_process, n_samples,
                        lambda: model.maybe_autocast(dtype=torch.bfloat16), prune_block, tuple_output=True)
        cfg.use_cache = use_cache
        cal.release_tower_memory()
        return model

    @cal.quiet_gc
    @print_time
    def prune(self, importance_scores=None, keep_indices_or_masks=None, lora_model=False):
        dtype_record, requires_grad_record, device = self.model_setup_and_record_attributes(self.model)
        if self.prune_spec is None:
            return self.model, None
        _, keep_ratio, _, _ = self.convert_spec_to_list(self.prune_spec)
        sparsity_dict = self.get_sparsity(1 - keep_ratio, sparsity_ratio_granularity=self.sparsity_ratio_granularity)
        for side in ("encoder", "decoder"):                                     # :842-857
            self.model = self._prune(self.model, self.data_loader, model_prefix=self.model_prefix,
                                     module_to_process=f"{self.model_prefix}.{side}.block", n_samples=self.num_samples,
                                     sparsity_ratio=sparsity_dict, lora_model=lora_model)
        self.model_reset(self.model, dtype_record, requires_grad_record, device)
        return self.model, sparsity_dict


@registry.register_pruner("vit_dsnot_pruner")
class VITLayerDSnoTPruner(LayerWiseBasePruner, _DsnotBlockMixin):
    """EVA ViT tower (dsnot_pruner.py:861-1596): the same per-row procedure; blocks return a
    tensor, and the block forward runs under bf16 autocast like the T5 one (:1086, :1493)."""
    pruner_name = "vit_dsnot_pruner"

    def __init__(self, model, data_loader, prune_spec=None, importance_scores_cache=None,
                 keep_indices_or_masks_cache=None, is_strct_pruning=False, num_samples=64, is_global=False,
                 model_prefix="visual", sparsity_ratio_granularity=None, max_sparsity_per_layer=0.8,
                 score_method="obd_avg", num_data_first_stage=128, num_noise=1, sparsity_dict=None, noise_eps=1e-3,
                 prune_per_model=False, prune_n=0, prune_m=0, initial_method="wanda", skip_layer=None,
                 skip_sub_layer=None, pow_of_var_regrowing=1., max_cycle_time=1e2, update_threshold=0.1,
                 without_same_sign=True, without_DSnoT=False, **kwargs):
        super().__init__(**_base_kwargs(locals()))
        self._set_dsnot_options(initial_method, skip_layer, skip_sub_layer, pow_of_var_regrowing, max_cycle_time,
                                update_threshold, without_same_sign, without_DSnoT)

    def forward_to_cache(self, model, batch, lora_model=False):
        return model.encode_image(batch["image"])

    get_sparsity = T5LayerDSnoTPruner.get_sparsity
    check_sparsity = T5LayerDSnoTPruner.check_sparsity

    def prepare_calibration_input_encoder(self, model, dataloader, model_prefix, n_samples,
                                          module_to_process="encoder.block", lora_model=False):
        return cal.capture_block_inputs(model, dataloader, n_samples, module_to_process, self.forward_to_cache,
                                        lora_model, vit=True, done_towers=getattr(self, "_done_towers", None),
                                        proxy_cache=self.__dict__.setdefault("_proxy_cache", {}))

    @print_time
    def _prune(self, model, dataloader, model_prefix, module_to_process="encoder.block", n_samples=64,
               sparsity_ratio=0.5, lora_model=False):
        with torch.no_grad():
            inps, outs, caches = self.prepare_calibration_input_encoder(model, dataloader, model_prefix, n_samples,
                                                                        module_to_process, lora_model)
        n_inps, batch0 = len(inps) * cal.calibration_shard()[1], inps[0].shape[0]

        def prune_block(i, layer, subset, run_pass, state):
            self._dsnot_block(i, subset, run_pass, n_inps, batch0, module_to_process=module_to_process,
                              model_prefix=model_prefix, sparsity_ratio=sparsity_ratio, lora_model=lora_model)

        cal.walk_blocks(model, inps, outs, caches, module_to_process, n_samples,
                        lambda: model.maybe_autocast(dtype=torch.bfloat16), prune_block, tuple_output=False,
                        memo_cache=self.__dict__.get("_proxy_cache"))
        cal.release_tower_memory()
        return model

    @cal.quiet_gc
    @print_time
    def prune(self, importance_scores=None, keep_indices_or_masks=None, lora_model=False):
        dtype_record, requires_grad_record, device = self.model_setup_and_record_attributes(self.model)
        if self.prune_spec is None:
            return self.model, None
        _, keep_ratio, _, _ = self.convert_spec_to_list(self.prune_spec)
        sparsity_dict = self.get_sparsity(1 - keep_ratio, sparsity_ratio_granularity=self.sparsity_ratio_granularity)
        self.model = self._prune(self.model, self.data_loader, model_prefix=self.model_prefix,
                                 module_to_process=f"{self.model_prefix}.blocks", n_samples=self.num_samples,
                                 sparsity_ratio=sparsity_dict, lora_model=lora_model)
        self.model_reset(self.model, dtype_record, requires_grad_record, device)
        return self.model, sparsity_dict


@registry.register_pruner("blipt5_dsnot_pruner")
class BLIPT5LayerDSnoTPruner(LayerWiseBasePruner, _DsnotBlockMixin):
    """ViT tower, then T5 encoder, then T5 decoder (or the LLaMA/OPT stack) -- dsnot_pruner.py:1599-1878."""
    pruner_name = "blipt5_dsnot_pruner"

    def __init__(self, model, data_loader, t5_prune_spec=None, vit_prune_spec=None, t5_pruning_method=None,
                 vit_pruning_method=None, t5_importance_scores_cache=None, t5_keep_indices_or_masks_cache=None,
                 vit_importance_scores_cache=None, vit_keep_indices_or_masks_cache=None, importance_scores_cache=None,
                 keep_indices_or_masks_cache=None, is_strct_pruning=False, num_samples=64, is_global=False,
                 t5_model_prefix="t5_model", vit_model_prefix="visual_encoder", sparsity_ratio_granularity=None,
                 max_sparsity_per_layer=0.8, score_method="obd_avg", num_data_first_stage=128, num_noise=1,
                 sparsity_dict=None, noise_eps=1e-3, prune_per_model=False, initial_method="wanda", skip_layer=None,
                 skip_sub_layer=None, pow_of_var_regrowing=1., max_cycle_time=1e2, update_threshold=0.1,
                 without_same_sign=True, without_DSnoT=False, peft_postfix="", prune_n=0, prune_m=0, **kwargs):
        prune_spec, model_prefix = None, f"{vit_model_prefix}+{t5_model_prefix}"
        super().__init__(**_base_kwargs(locals()))
        self._set_dsnot_options(initial_method, skip_layer, skip_sub_layer, pow_of_var_regrowing, max_cycle_time,
                                update_threshold, without_same_sign, without_DSnoT)
        self.t5_prune_spec = t5_prune_spec
        self.vit_prune_spec = vit_prune_spec
        self.peft_postfix = peft_postfix
        assert t5_pruning_method is not None
        assert vit_pruning_method is not None
        self.t5_model_prefix = t5_model_prefix
        self.vit_model_prefix = vit_model_prefix

    def get_sparsity(self, t5_sparsity, vit_sparsity, sparsity_ratio_granularity=None):
        """(:1683-1757): a yaml override, else LayerSparsity at the MEAN of both sparsities (per-tower budgets
        under `prune_per_model`)."""
        original_sparsity = 0.5 * (t5_sparsity + vit_sparsity)
        if self.sparsity_dict is not None:
            import yaml
            with open(self.sparsity_dict, "r") as f:
                return yaml.load(f, Loader=yaml.FullLoader)
        from lavis.compression.pruners.utils import loss_vision_language
        from lavis.compression.pruners.wanda_pruner import layer_to_group_mapping
        # same grouping rule as the Wanda pruner's get_sparsity (:1693-1740)
        mapping = layer_to_group_mapping(self, sparsity_ratio_granularity)
        return LayerSparsity(self.model, self.data_loader, loss_vision_language, self.num_data_first_stage, original_sparsity,
                             self.max_sparsity_per_layer, self.score_method, self.num_noise, self.noise_eps, mapping,
                             prune_per_model=self.prune_per_model,
                             per_model_group=[self.t5_model_prefix, self.vit_model_prefix],
                             per_model_sparsity=[t5_sparsity, vit_sparsity]).return_sparsity()

    def forward_to_cache(self, model, batch, lora_model=False):
        if lora_model:
            return model(batch, vit_dense=True, llm_dense=True)                 # always dense here (:1759-1763)
        return model(batch)

    def _tower(self, cls, **kw):
        self.prepare_calibration_input_encoder = lambda *a, **k: cls.prepare_calibration_input_encoder(self, *a, **k)
        out = cls._prune(self, self.model, self.data_loader, **kw)
        # the finished tower's blocks may replay from HIP graphs while the next tower's inputs are captured
        self._done_towers = getattr(self, "_done_towers", []) + [kw["module_to_process"]]
        return out

    @cal.quiet_gc
    @print_time
    def prune(self, importance_scores=None, keep_indices_or_masks=None, lora_model=False):
        dtype_record, requires_grad_record, device = self.model_setup_and_record_attributes(self.model)
        global_sparsity_dict = None
        _, vit_keep_ratio, _, _ = self.convert_spec_to_list(self.vit_prune_spec)
        _, t5_keep_ratio, _, _ = self.convert_spec_to_list(self.t5_prune_spec)
        if self.sparsity_ratio_granularity not in [None, "none"]:
            global_sparsity_dict = self.get_sparsity(1 - t5_keep_ratio, 1 - vit_keep_ratio,
                                                     sparsity_ratio_granularity=self.sparsity_ratio_granularity)

        if self.vit_prune_spec is not None and float(vit_keep_ratio) < 1.:
            s = 1 - vit_keep_ratio
            sd = global_sparsity_dict if global_sparsity_dict is not None else \
                self.get_sparsity(s, s, sparsity_ratio_granularity=None)
            self.model = self._tower(VITLayerDSnoTPruner, model_prefix=self.vit_model_prefix,
                                     module_to_process=f"{self.vit_model_prefix}.blocks", n_samples=self.num_samples,
                                     sparsity_ratio=sd, lora_model=lora_model)

        if self.t5_prune_spec is not None and float(t5_keep_ratio) < 1.:
            s = 1 - t5_keep_ratio
            sd = global_sparsity_dict if global_sparsity_dict is not None else \
                self.get_sparsity(s, s, sparsity_ratio_granularity=None)
            if "t5_model" in self.t5_model_prefix:
                for side in ("encoder", "decoder"):
                    self.model = self._tower(T5LayerDSnoTPruner, model_prefix=self.t5_model_prefix,
                                             module_to_process=f"{self.t5_model_prefix}.{side}.block",
                                             n_samples=self.num_samples, sparsity_ratio=sd, lora_model=lora_model)
            else:
                self.model = self._tower(T5LayerDSnoTPruner, model_prefix=self.t5_model_prefix,
                                         module_to_process=f"{self.t5_model_prefix}{self.peft_postfix}.model.layers",
                                         n_samples=self.num_samples, sparsity_ratio=sd, lora_model=lora_model)

        self.model_reset(self.model, dtype_record, requires_grad_record, device)
        return self.model, global_sparsity_dict

    def check(self, name, v, model_prefix):
        return len(v.shape) == 2 and ".block" in name and "relative_attention_bias.weight" not in name \
            and name.startswith(model_prefix)

    def trans_sparsity(self, vit_params, t5_params, vit_keep_ratio, t5_keep_ratio):
        """(:1873-1878) equalise the kept parameter budgets of the two towers."""
        vit_keep = (vit_params + t5_params) * vit_keep_ratio / 2
        t5_keep = (vit_params + t5_params) * t5_keep_ratio / 2
        return min(vit_keep / vit_params, 1.0), min(t5_keep / t5_params, 1.0)


def return_reorder_indice(input_tensor):
    """The reference's module-level helper (dsnot_pruner.py:1881-1925), same name and contract: for
    [[1., -2., 3.], [-2, 2., -4], [5., 6., -7], [-6, -7, -4]] the int64 indices [[1, 2, 0], [0, 2, 1], [2, 1, 0], [0, 1, 2]] --
    negative entries keep their relative order at the head, positive entries are flipped at the tail, an entry that is
    neither contributes a 0 between them.  One launch of `vlmc_reorder_indices` instead of two fp64 [rows, cols] index
    matrices, two sorts, a flip and a sum; GPU tensors only (no CPU fallback, like every op of this build)."""
    from vlmc import dsnot
    return dsnot.reorder_indices(input_tensor)
