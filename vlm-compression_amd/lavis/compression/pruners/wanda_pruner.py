"""Wanda pruners behind the reference's `lavis.compression` API, running on the
gfx950 kernels (`vlmc.ops`): `t5_wanda_pruner`, `vit_wanda_pruner`, `blipt5_wanda_pruner`.

Reference: lavis/compression/pruners/wanda_pruner.py (T5LayerWandaPruner :84-494,
VITLayerWandaPruner :497-793, BLIPT5LayerWandaPruner :796-1052).  Same registry names,
constructor kwargs, `prune(importance_scores=None, keep_indices_or_masks=None,
lora_model=False) -> (model, sparsity_dict | None)` contract and side effects
(`module.mask` bool [out,in] True = keep, `weight.importance_score` float, weights
zeroed in place unless `lora_model`).  What differs is where the arithmetic runs:

  reference op sequence (per linear)                    here
  ---------------------------------------------------  ---------------------------------
  hook: norm(x.t().float(), dim=1)**2, running mean     vlmc_act_sqnorm (per hook call,
  (:68-81)                                               shared by linears with the same
                                                         input) + vlmc_wanda_scaler_update
  |W| * sqrt(scaler_row); .cpu().abs().mean().item()    fused in vlmc_wanda_select; the
  (:318-320, full [out,in] D2H copy)                     score is never materialised, one
  sort / topk loop / threshold; scatter_; W[mask]=0      8-byte readback per BLOCK
  (:322-341, :670-687)

There is no CPU fallback: without the HIP library or a GPU model the pruner raises.
"""
from __future__ import annotations

import os

import torch

from lavis.common.registry import registry
from lavis.compression.pruners import calibration as cal
from lavis.compression.pruners.layer_single_base_pruner import LayerSparsity, LayerWiseBasePruner
from lavis.compression.pruners.utils import print_time
from vlmc import phases

_VERBOSE = os.environ.get("VLMC_VERBOSE", "0") != "0"


# --------------------------------------------------------------------------------------
# statistics collection through forward hooks
# --------------------------------------------------------------------------------------
class WandaStatCollector:
    """Forward hooks on the linears of one block (wanda_pruner.py:295-314).  Linears that
    receive the very same tensor (q/k/v, wi_0/wi_1) share one statistic, and all distinct
    inputs of one sample's block forward are reduced by ONE batched launch (issued when the
    next sample starts, or at close)."""

    def __init__(self, subset):
        from vlmc import ops
        self._ops = ops
        self.subset = subset
        self.rows = {n: [] for n in subset}          # per linear and hook call: (sample indices, holder [rows | None], batch)
        self._cur = -1                               # index of the sample an unstacked forward belongs to
        self._cache = {}                              # input signature -> (x kept alive, holder)
        self._pending = []                            # (x [calls, tokens, in], holder) not yet reduced
        self.handles = [m.register_forward_hook(self._make_hook(n)) for n, m in subset.items()]

    def _make_hook(self, name):
        def hook(_module, inp, _out):
            x = inp[0].data
            if x.dim() == 2:
                x = x.unsqueeze(0)
            key = (x.data_ptr(), tuple(x.shape), tuple(x.stride()), x.dtype, x._version)
            hit = self._cache.get(key)
            stacked = cal.stacked_samples()          # (samples, batch per sample, sample indices) of a batched replay call
            calls, b0, idx = stacked if stacked and stacked[0] * stacked[1] == x.shape[0] else (1, x.shape[0], (self._cur,))
            if hit is None:
                # `x` stays referenced until it has been reduced, so its memory cannot be recycled for a
                # different activation with the same address/shape in the meantime
                hit = (x, [None])                    # the holder receives the [calls, in] statistic rows of this tensor
                self._cache[key] = hit
                # a padded group of ragged samples: the token rows of sample c that are its own (the rest is padding)
                lens = cal.stacked_lengths(x.shape[1]) if calls > 1 and x.dim() == 3 else None
                self._pending.append((x.reshape(calls, -1, x.shape[-1]), hit[1], lens))
            # one record per hook call: the calibration samples it stands for (grouped replay visits samples out of order;
            # `finalize` puts the per-sample rows back into the reference's order)
            self.rows[name].append((idx, hit[1], b0))
        return hook

    def _flush(self):
        if self._pending:
            with phases.phase("stat"):
                self._flush_pending()

    def _flush_pending(self):
        by_dtype = {}
        for x, holder, lens in self._pending:
            by_dtype.setdefault(x.dtype, []).append((x, holder, lens))
        for items in by_dtype.values():
            by_calls = {}
            for x, holder, lens in items:
                by_calls.setdefault(x.shape[0], []).append((x, holder, lens))
            for same in by_calls.values():                     # one launch per group of inputs with equally many calls
                lens = [l for _, _, l in same]
                if any(l is not None for l in lens):           # a padded group of ragged samples
                    outs = self._ops.act_sqnorm_batch([x for x, _, _ in same], call_tokens=lens)
                else:
                    outs = self._ops.act_sqnorm_batch([x for x, _, _ in same])
                for (_, holder, _l), rows in zip(same, outs):
                    holder[0] = rows                           # row c = the c-th sample of the call, sliced only where needed
        self._pending = []

    def next_sample(self, j=None):
        self._flush()
        self._cache.clear()
        self._cur = self._cur + 1 if j is None else j

    def close(self):
        for h in self.handles:
            h.remove()
        self.handles = []
        self._flush()
        self._cache.clear()

    def finalize(self):
        """{name: InputStat}; linears whose hooks saw identical tensors share the object."""
        from vlmc import wanda
        shared, out, order = {}, {}, []
        for name, recs in self.rows.items():
            sig = tuple(id(h) for _, h, _ in recs)             # linears fed by the same tensors share the statistic
            st = shared.get(sig)
            if st is None:
                in_f = self.subset[name].weight.shape[1]
                st = wanda.InputStat(in_f, self.subset[name].weight.device)
                if len(recs) == 1 and all(a < b for a, b in zip(recs[0][0], recs[0][0][1:])):
                    idx, holder, b0 = recs[0]                  # one grouped call, samples in order: its rows as they are
                    st.rows, st.batches = [holder[0][:len(idx)]], [b0] * len(idx)
                else:                                          # the reference's sample order (stable: calls within a sample)
                    flat = sorted(((j, h, c, b) for idx, h, b in recs for c, j in enumerate(idx)), key=lambda r: r[0])
                    st.rows = _row_runs([(h[0], c) for _, h, c, _ in flat])
                    st.batches = [b for _, _, _, b in flat]
                shared[sig] = st
                order.append(st)
            out[name] = st
        wanda.gather_stats(order)
        return out


def _row_runs(refs):
    """[(rows tensor, row index)] in sample order -> the fewest [k, in] tensors with the same rows in the same order: runs
    of consecutive rows of one launch's output stay ONE slice (all 128 samples of a grouped replay: the tensor itself)."""
    out, i = [], 0
    while i < len(refs):
        base, c0 = refs[i]
        j = i + 1
        while j < len(refs) and refs[j][0] is base and refs[j][1] == c0 + (j - i):
            j += 1
        out.append(base[c0:c0 + (j - i)])
        i = j
    return out


def _importance_backlog(backlog, subset, names, partial_rows, numels):
    """weight.importance_score = mean score of the linear (the reference copies each fp32 [out,in] metric to the host for
    it, wanda_pruner.py:320).  The sums stay on the device until the tower is done: a host copy per block would drain the
    GPU's queue 87 times per FlanT5-XL prune (~0.1 s)."""
    backlog.append(([subset[n].weight for n in names], partial_rows.sum(dim=1), numels))


def _importance_readback(backlog, owner=None):
    """ONE device->host copy per tower for the importance scores of all its linears -- or, when the towers are pruned by
    a pruner that says so (`_defer_score_readback`, the three-tower BLIP pruner), ONE per prune: the copy waits for
    everything the GPU has queued, and the host has the next tower's capture sweeps to issue meanwhile."""
    if not backlog or (owner is not None and getattr(owner, "_defer_score_readback", False)):
        return
    sums = torch.cat([s for _, s, _ in backlog]).cpu().tolist()
    k = 0
    for weights, _, numels in backlog:
        for w, numel in zip(weights, numels):
            setattr(w, "importance_score", sums[k] / numel)
            k += 1
    backlog.clear()


class _WandaBlockMixin:
    """Per-block Wanda step shared by the T5/LLM and ViT variants."""

    def _wanda_block(self, i, subset, run_pass, n_inps, batch0, *, unstructured_mode, module_to_process, model_prefix,
                     sparsity_ratio, lora_model):
        from vlmc import ops, wanda
        col = WandaStatCollector(subset)
        try:
            run_pass(col.next_sample, outputs=False)
        finally:
            col.close()
        with phases.phase("stat"):
            stats = col.finalize()
        with phases.phase("select"):
            self._wanda_select_block(i, subset, stats, n_inps, batch0, unstructured_mode=unstructured_mode,
                                     module_to_process=module_to_process, model_prefix=model_prefix,
                                     sparsity_ratio=sparsity_ratio, lora_model=lora_model)

    def _wanda_select_block(self, i, subset, stats, n_inps, batch0, *, unstructured_mode, module_to_process, model_prefix,
                            sparsity_ratio, lora_model):
        from vlmc import ops, wanda

        names = list(subset)
        mode = unstructured_mode if self.prune_n == 0 else "nm"
        nparts = [ops.select_partials(mode, *subset[n].weight.shape) for n in names]
        dev = subset[names[0]].weight.device
        partial_rows = torch.zeros((len(names), max(nparts)), dtype=torch.float64, device=dev)
        weights, ratios = [], []
        for name in names:
            mod, st = subset[name], stats[name]
            assert st.nsamples == n_inps * batch0                      # wanda_pruner.py:317
            if not mod.weight.data.is_contiguous():
                raise RuntimeError(f"{name}: weight must be contiguous")
            weights.append(mod.weight.data)
            if self.prune_n != 0:
                ratios.append(None)
                if _VERBOSE:
                    print(f"pruning {model_prefix} layer {i} {name} at structured {self.prune_n}:{self.prune_m} sparsity")
            else:
                ratios.append(sparsity_ratio[f"{module_to_process}.{i}.{name}.weight"])
                if _VERBOSE:
                    print(f"pruning {model_prefix} layer {i} {name} at unstructured {ratios[-1]} sparsity")
        # every linear of the block in batched launches (the reference's per-linear loop, :316-341)
        masks = wanda.prune_block(weights, [stats[n] for n in names], mode, ratios=ratios, n=self.prune_n, m=self.prune_m,
                                  apply_zero=not lora_model, partials=[partial_rows[li] for li in range(len(names))])
        for name, mask in zip(names, masks):
            setattr(subset[name], "mask", mask)                         # True = keep (:339)
        _importance_backlog(self.__dict__.setdefault("_score_backlog", []), subset, names, partial_rows,
                            [subset[n].weight.numel() for n in names])


# --------------------------------------------------------------------------------------
@registry.register_pruner("t5_wanda_pruner")
class T5LayerWandaPruner(LayerWiseBasePruner, _WandaBlockMixin):
    """T5 / OPT / LLaMA tower: per-output-row selection (wanda_pruner.py:84-494)."""
    pruner_name = "t5_wanda_pruner"

    def __init__(self, model, data_loader, prune_spec=None, importance_scores_cache=None,
                 keep_indices_or_masks_cache=None, is_strct_pruning=False, num_samples=64, is_global=False,
                 model_prefix="t5_model", sparsity_ratio_granularity=None, max_sparsity_per_layer=0.8,
                 score_method="obd_avg", num_data_first_stage=128, num_noise=1, sparsity_dict=None, noise_eps=1e-3,
                 prune_per_model=False, prune_n=0, prune_m=0, **kwargs):
        super().__init__(model=model, data_loader=data_loader, prune_spec=prune_spec,
                         is_strct_pruning=is_strct_pruning, importance_scores_cache=importance_scores_cache,
                         keep_indices_or_masks_cache=keep_indices_or_masks_cache, is_global=is_global,
                         num_samples=num_samples, model_prefix=model_prefix,
                         sparsity_ratio_granularity=sparsity_ratio_granularity,
                         max_sparsity_per_layer=max_sparsity_per_layer, score_method=score_method,
                         num_data_first_stage=num_data_first_stage, num_noise=num_noise, sparsity_dict=sparsity_dict,
                         noise_eps=noise_eps, prune_per_model=prune_per_model, prune_n=prune_n, prune_m=prune_m)

    def forward_to_cache(self, model, batch, lora_model=False):
        return model(batch)

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
        # under calibration sharding every rank holds 1/world of the samples; statistics are global
        n_inps, batch0 = len(inps) * cal.calibration_shard()[1], inps[0].shape[0]

        def prune_block(i, layer, subset, run_pass, state):
            self._wanda_block(i, subset, run_pass, n_inps, batch0, unstructured_mode="row",
                              module_to_process=module_to_process, model_prefix=model_prefix,
                              sparsity_ratio=sparsity_ratio, lora_model=lora_model)

        cal.walk_blocks(model, inps, outs, caches, module_to_process, n_samples,
                        lambda: model.maybe_autocast(dtype=torch.bfloat16), prune_block, tuple_output=True,
                        pad_ragged=True)                       # (the Wanda statistic takes per-sample token counts: WandaStatCollector)
        _importance_readback(self.__dict__.setdefault("_score_backlog", []), self)
        cfg.use_cache = use_cache
        cal.release_tower_memory()
        return model


@registry.register_pruner("vit_wanda_pruner")
class VITLayerWandaPruner(LayerWiseBasePruner, _WandaBlockMixin):
    """EVA ViT tower: ONE matrix-wide threshold per linear (wanda_pruner.py:497-793)."""
    pruner_name = "vit_wanda_pruner"

    def __init__(self, model, data_loader, prune_spec=None, importance_scores_cache=None,
                 keep_indices_or_masks_cache=None, is_strct_pruning=False, num_samples=64, is_global=False,
                 model_prefix="visual", sparsity_ratio_granularity=None, max_sparsity_per_layer=0.8,
                 score_method="obd_avg", num_data_first_stage=128, num_noise=1, sparsity_dict=None, noise_eps=1e-3,
                 prune_per_model=False, prune_n=0, prune_m=0, **kwargs):
        super().__init__(model=model, data_loader=data_loader, prune_spec=prune_spec,
                         is_strct_pruning=is_strct_pruning, importance_scores_cache=importance_scores_cache,
                         keep_indices_or_masks_cache=keep_indices_or_masks_cache, is_global=is_global,
                         num_samples=num_samples, model_prefix=model_prefix,
                         sparsity_ratio_granularity=sparsity_ratio_granularity,
                         max_sparsity_per_layer=max_sparsity_per_layer, score_method=score_method,
                         num_data_first_stage=num_data_first_stage, num_noise=num_noise, sparsity_dict=sparsity_dict,
                         noise_eps=noise_eps, prune_per_model=prune_per_model, prune_n=prune_n, prune_m=prune_m)

    def forward_to_cache(self, model, batch, lora_model=False):
        return model.encode_image(batch["image"])

    check_sparsity = T5LayerWandaPruner.check_sparsity

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
        # under calibration sharding every rank holds 1/world of the samples; statistics are global
        n_inps, batch0 = len(inps) * cal.calibration_shard()[1], inps[0].shape[0]

        def prune_block(i, layer, subset, run_pass, state):
            self._wanda_block(i, subset, run_pass, n_inps, batch0, unstructured_mode="matrix",
                              module_to_process=module_to_process, model_prefix=model_prefix,
                              sparsity_ratio=sparsity_ratio, lora_model=lora_model)

        cal.walk_blocks(model, inps, outs, caches, module_to_process, n_samples, lambda: model.maybe_autocast(),
                        prune_block, tuple_output=False, memo_cache=self.__dict__.get("_proxy_cache"))
        _importance_readback(self.__dict__.setdefault("_score_backlog", []), self)
        cal.release_tower_memory()
        return model


def layer_to_group_mapping(pruner, granularity):
    """Parameter name -> group name for `sparsity_ratio_granularity` in {None, "none", "model", "layer", "block"}
    (wanda_pruner.py:872-919): 2-D weights of the transformer blocks of both towers."""
    if granularity is None or granularity == "none":
        return {}
    t5, vit = pruner.t5_model_prefix, pruner.vit_model_prefix

    def prunable(name, v):
        return len(v.shape) == 2 and ".block" in name and "relative_attention_bias.weight" not in name and \
            (name.startswith(t5) or name.startswith(vit))
    names = [k for k, v in pruner.model.named_parameters() if prunable(k, v)]
    if granularity == "layer":
        return {k: k for k in names}
    if granularity == "model":
        return {k: (t5 if k.startswith(t5) else vit if k.startswith(vit) else "other") for k in names}
    if granularity == "block":
        # "<t5 prefix>.encoder.block.<i>" (4 name parts) / "<vit prefix>.blocks.<i>" (3 parts)
        return {k: ".".join(k.split(".")[:4 if k.startswith(t5) else 3]) if (k.startswith(t5) or k.startswith(vit)) else "other"
                for k in names}
    raise NotImplementedError


def uniform_or_layer_sparsity(pruner, original_sparsity, granularity, loss_func=None, **per_model):
    """`get_sparsity` of the BLIP pruners (wanda_pruner.py:866-939): a yaml override, else LayerSparsity --
    uniform without a grouping, ECoFLaP's score-proportional allocation with one."""
    if pruner.sparsity_dict is not None:
        import yaml
        with open(pruner.sparsity_dict, "r") as f:
            return yaml.load(f, Loader=yaml.FullLoader)
    if loss_func is None:
        from lavis.compression.pruners.utils import loss_vision_language as loss_func
    mapping = layer_to_group_mapping(pruner, granularity) if hasattr(pruner, "t5_model_prefix") else {}
    return LayerSparsity(pruner.model, pruner.data_loader, loss_func, pruner.num_data_first_stage, original_sparsity,
                         pruner.max_sparsity_per_layer, pruner.score_method, pruner.num_noise, pruner.noise_eps,
                         mapping, **per_model).return_sparsity()


@registry.register_pruner("blipt5_wanda_pruner")
class BLIPT5LayerWandaPruner(LayerWiseBasePruner):
    """ViT tower, then T5 encoder, then T5 decoder (or the LLaMA/OPT stack) --
    wanda_pruner.py:796-1052."""
    pruner_name = "blipt5_wanda_pruner"

    def __init__(self, model, data_loader, t5_prune_spec=None, vit_prune_spec=None, t5_pruning_method=None,
                 vit_pruning_method=None, t5_importance_scores_cache=None, t5_keep_indices_or_masks_cache=None,
                 vit_importance_scores_cache=None, vit_keep_indices_or_masks_cache=None, importance_scores_cache=None,
                 keep_indices_or_masks_cache=None, is_strct_pruning=False, num_samples=64, is_global=False,
                 t5_model_prefix="t5_model", vit_model_prefix="visual_encoder", sparsity_ratio_granularity=None,
                 max_sparsity_per_layer=0.8, score_method="obd_avg", num_data_first_stage=128, num_noise=1,
                 sparsity_dict=None, noise_eps=1e-3, prune_per_model=False, peft_postfix="", prune_n=0, prune_m=0,
                 **kwargs):
        super().__init__(model=model, data_loader=data_loader, prune_spec=None, is_strct_pruning=is_strct_pruning,
                         importance_scores_cache=importance_scores_cache,
                         keep_indices_or_masks_cache=keep_indices_or_masks_cache, is_global=is_global,
                         num_samples=num_samples, model_prefix=f"{vit_model_prefix}+{t5_model_prefix}",
                         sparsity_ratio_granularity=sparsity_ratio_granularity,
                         max_sparsity_per_layer=max_sparsity_per_layer, score_method=score_method,
                         num_data_first_stage=num_data_first_stage, num_noise=num_noise, sparsity_dict=sparsity_dict,
                         noise_eps=noise_eps, prune_per_model=prune_per_model, prune_n=prune_n, prune_m=prune_m)
        self.t5_prune_spec = t5_prune_spec
        self.vit_prune_spec = vit_prune_spec
        self.peft_postfix = peft_postfix
        self.vit_dense = True
        self.llm_dense = True
        assert t5_pruning_method is not None
        assert vit_pruning_method is not None
        self.t5_model_prefix = t5_model_prefix
        self.vit_model_prefix = vit_model_prefix

    # the tower implementations are borrowed as unbound methods, like the reference does with
    # functools.partial (:983-988, :1011-1015)
    _wanda_block = _WandaBlockMixin._wanda_block
    _wanda_select_block = _WandaBlockMixin._wanda_select_block

    def get_sparsity(self, original_sparsity, sparsity_ratio_granularity=None):
        return uniform_or_layer_sparsity(self, original_sparsity, sparsity_ratio_granularity)

    def forward_to_cache(self, model, batch, lora_model=False):
        if lora_model:
            return model(batch, vit_dense=self.vit_dense, llm_dense=self.llm_dense)      # :941-945
        return model(batch)

    def _tower(self, cls, last=False, **kw):
        self.prepare_calibration_input_encoder = lambda *a, **k: cls.prepare_calibration_input_encoder(self, *a, **k)
        if last:          # nothing is captured after this tower: what a capture phase records for the NEXT one is not recorded
            self.__dict__.setdefault("_proxy_cache", {})[("last_tower",)] = kw["module_to_process"]
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
            global_sparsity_dict = self.get_sparsity(1 - t5_keep_ratio,
                                                     sparsity_ratio_granularity=self.sparsity_ratio_granularity)
        # the calibration forward of a tower is DENSE exactly when that tower is pruned (:966-967)
        self.vit_dense = True if float(vit_keep_ratio) < 1. else False
        self.llm_dense = True if float(t5_keep_ratio) < 1. else False
        self._defer_score_readback = True           # importance scores are read back once, below
        try:
            prune_t5 = self.t5_prune_spec is not None and float(t5_keep_ratio) < 1.
            if self.vit_prune_spec is not None and float(vit_keep_ratio) < 1.:
                sd = global_sparsity_dict if global_sparsity_dict not in [None, "none"] else \
                    self.get_sparsity(1 - vit_keep_ratio, sparsity_ratio_granularity=None)
                self.model = self._tower(VITLayerWandaPruner, last=not prune_t5, model_prefix=self.vit_model_prefix,
                                         module_to_process=f"{self.vit_model_prefix}.blocks",
                                         n_samples=self.num_samples, sparsity_ratio=sd, lora_model=lora_model)

            if prune_t5:
                sd = global_sparsity_dict if global_sparsity_dict is not None else \
                    self.get_sparsity(1 - t5_keep_ratio, sparsity_ratio_granularity=None)
                if "t5_model" in self.t5_model_prefix:
                    for side in ("encoder", "decoder"):
                        self.model = self._tower(T5LayerWandaPruner, last=side == "decoder", model_prefix=self.t5_model_prefix,
                                                 module_to_process=f"{self.t5_model_prefix}.{side}.block",
                                                 n_samples=self.num_samples, sparsity_ratio=sd, lora_model=lora_model)
                else:
                    self.model = self._tower(T5LayerWandaPruner, last=True, model_prefix=self.t5_model_prefix,
                                             module_to_process=f"{self.t5_model_prefix}{self.peft_postfix}.model.layers",
                                             n_samples=self.num_samples, sparsity_ratio=sd, lora_model=lora_model)
        finally:
            # also when a tower raises: the towers done so far get their importance scores, the flag does not outlive
            # the call (a later stand-alone tower prune on this object reads its scores back itself)
            self._defer_score_readback = False
            _importance_readback(self.__dict__.setdefault("_score_backlog", []))  # the towers' importance scores, one copy
        self.model_reset(self.model, dtype_record, requires_grad_record, device)
        return self.model, global_sparsity_dict

    def check(self, name, v, model_prefix):
        return len(v.shape) == 2 and ".block" in name and "relative_attention_bias.weight" not in name \
            and name.startswith(model_prefix)
