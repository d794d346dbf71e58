"""SparseGPT pruners behind the reference's `lavis.compression` API: `t5_sparsegpt_pruner`,
`vit_sparsegpt_pruner`, `blipt5_sparsegpt_pruner`
(reference: lavis/compression/pruners/sparsegpt_pruner.py:222-497, :500-867, :870-1090).

Same registry names, constructor kwargs and `prune(importance_scores=None,
keep_indices_or_masks=None) -> (model, sparsity_dict | None)` contract (no `lora_model`
argument, like the reference).  Quirks kept: calibration stops by BATCH index (`i >= n_samples`,
:391-393), both towers are pruned whenever their spec is not None (no keep-ratio < 1 test,
:1021,:1044), the LLM branch walks `...model.decoder.layers` (OPT layout, :1084), no `module.mask`
is attached (SparseGPT only rewrites the weights).  The arithmetic lives in `vlmc.sparsegpt`.
"""
from __future__ import annotations

import os

import torch

from lavis.common.registry import registry
from lavis.compression.pruners import calibration as cal
from lavis.compression.pruners.layer_single_base_pruner import LayerWiseBasePruner
from lavis.compression.pruners.utils import print_time
from lavis.compression.pruners.wanda_pruner import uniform_or_layer_sparsity

_KW = dict(prune_spec=None, importance_scores_cache=None, keep_indices_or_masks_cache=None, is_strct_pruning=False,
           num_samples=64, is_global=False, sparsity_ratio_granularity=None, max_sparsity_per_layer=0.8,
           score_method="obd_avg", num_data_first_stage=128, num_noise=1, sparsity_dict=None, noise_eps=1e-3,
           prune_per_model=False, prune_n=0, prune_m=0)


class _SparseGPTBlockMixin:
    def _capture(self, model, dataloader, n_samples, module_to_process, vit, model_prefix=None):
        return cal.capture_block_inputs(model, dataloader, n_samples, module_to_process,
                                        lambda m, b, _lora=False: self.forward_to_cache(m, b), False, vit=vit,
                                        model_prefix=model_prefix, count_batches=True, done_towers=getattr(self, "_done_towers", None),
                                        proxy_cache=self.__dict__.setdefault("_proxy_cache", {}))

    def _sparsegpt_block(self, i, subset, run_pass, n_inps, module_to_process, sparsity_ratio):
        """Hooks -> one dense pass -> prune every linear (sparsegpt_pruner.py:405-459).  Linears that receive the
        very same tensor (q/k/v, wi_0/wi_1, cross-attention k/v) have bit-identical Hessians: they share ONE
        accumulator (one GEMM per hook input instead of one per linear) and ONE Cholesky chain."""
        from vlmc import sparsegpt
        wrapped, fed = {}, {}            # linear name -> accumulator; input signature -> accumulator fed in this forward

        def make_hook(name):
            def hook(_m, inp, out):
                x = inp[0].data
                key = (x.data_ptr(), tuple(x.shape), tuple(x.stride()), x.dtype, x._version)
                acc = wrapped.get(name)
                if acc is None:                                       # first sample: discover who shares what
                    acc = fed.get(key)
                    if acc is None:
                        acc = sparsegpt.SparseGPT(subset[name])
                        acc.add_batch(x, None)     # (`out` is None where the statistics pass cut the dead tail; add_batch never reads it, sparsegpt_pruner.py:68-79)
                        fed[key] = acc
                    wrapped[name] = acc
                elif fed.get(key) is not acc:
                    if key in fed:
                        raise RuntimeError(f"{name}: the linears sharing an input changed between calibration samples")
                    acc.add_batch(x, None)     # (`out` is None where the statistics pass cut the dead tail; add_batch never reads it, sparsegpt_pruner.py:68-79)
                    fed[key] = acc
                acc._keep_alive = x                                   # the signature holds only while `x` lives
            return hook

        handles = [mod.register_forward_hook(make_hook(name)) for name, mod in subset.items()]
        try:
            run_pass(lambda _j: fed.clear(), outputs=False)
        finally:
            for h in handles:
                h.remove()
        unique = list({id(a): a for a in wrapped.values()}.values())
        from vlmc.shard import require_real_exchange
        require_real_exchange("blipt5_sparsegpt_pruner")
        _allreduce_hessians(unique)
        owner = _shard_linears(subset, wrapped)
        rank = cal.calibration_shard()[0]
        mine = [name for name in subset if owner is None or owner[name] == rank]
        # the Hessians this rank prunes with: factorized together, each chain on a stream of its own, one host check for all
        sparsegpt.factorize_many(list({id(wrapped[n]): (wrapped[n].H, wrapped[n].factor_cache) for n in mine}.values()), percdamp=0.01,
                                 history=self.__dict__.setdefault("_damping_history", {}).setdefault(module_to_process, {}))
        # importance scores (:165: a Python float per linear) are queued on the pruner and read back once per tower: a host
        # copy per block made the host wait for the block's sweeps before it could issue the next block's replay
        scores = self.__dict__.setdefault("_score_backlog", [])
        by_acc = {}
        for name in mine:
            assert wrapped[name].nsamples == n_inps                                # :442
            by_acc.setdefault(id(wrapped[name]), []).append(name)
        def sweep(names):
            acc = wrapped[names[0]]
            keys = [f"{module_to_process}.{i}.{n}.weight" for n in names]
            same = len({(subset[n].weight.dtype, subset[n].weight.shape[1]) for n in names}) == 1
            if len(names) > 1 and same and "U" in acc.factor_cache and sparsegpt.stacked_sweeps_enabled():
                # linears fed the same tensor share the factor: ONE column sweep over their stacked rows
                sparsegpt.fasterprune_group([subset[n] for n in names], [sparsity_ratio[k] for k in keys], acc.factor_cache,
                                            prune_n=self.prune_n, prune_m=self.prune_m, blocksize=128, score_sink=scores)
                return
            for name, key in zip(names, keys):
                sparsegpt.fasterprune(subset[name], acc.H, sparsity_ratio[key], prune_n=self.prune_n, prune_m=self.prune_m,
                                      percdamp=0.01, blocksize=128, factor_cache=acc.factor_cache, score_sink=scores)

        # The sweeps of linears with DIFFERENT inputs are independent chains of (cols / 128) x [latency-bound sweep kernel +
        # trailing GEMM] (the reference prunes them one after the other, :430-446): each chain on a stream of its own, the
        # longest first (`VLMC_SGPT_SWEEP_STREAMS=1`: one after the other).  Same kernels on the same data: bit-identical.
        # configs[2] (2:4): 2.10 -> 1.90 s per prune.
        groups = sorted(by_acc.values(), key=lambda g: -(subset[g[0]].weight.shape[1] * sum(subset[n].weight.shape[0] for n in g)))
        dev = subset[groups[0][0]].weight.device if groups else None
        # (n:m mode only: the unstructured block threshold is agreed on through grid barriers between co-resident workgroups --
        # vlmc_sparsegpt_select_sweep -- and two such launches side by side strand each other: measured 8.6 s against 2.5 s)
        streams = sparsegpt.sweep_streams(dev) if groups and dev.type == "cuda" and len(groups) > 1 and self.prune_n != 0 else []
        if not streams:
            for names in groups:
                sweep(names)
        else:
            main = torch.cuda.current_stream(dev)
            n0 = len(scores)
            for g, names in enumerate(groups):
                st = streams[g % len(streams)]
                st.wait_stream(main)
                for t in (wrapped[names[0]].factor_cache.get("U"), wrapped[names[0]].factor_cache.get("dead")):
                    if isinstance(t, torch.Tensor):
                        t.record_stream(st)
                with torch.cuda.stream(st):
                    sweep(names)
            for st in streams:
                main.wait_stream(st)
            for name in mine:                                   # made on a side stream, used on the caller's from here on
                subset[name].weight.data.record_stream(main)
            for _w, sc in scores[n0:]:
                sc.record_stream(main)
        if owner is not None:
            _flush_score_backlog(self)                                             # (the exchange sends the scores along)
            _exchange_pruned(subset, owner, rank)
        for acc in unique:
            acc.free()


def _flush_score_backlog(pruner):
    """`weight.importance_score` for every linear pruned since the last flush: one host copy (end of a tower, also on error)."""
    from vlmc import sparsegpt
    sparsegpt.flush_scores(pruner.__dict__.get("_score_backlog", []))


def _shard_linears(subset, wrapped):
    """Multi-GPU (BASELINE.json config 3, "layers sharded"): after the Hessian all-reduce every rank holds the same H, and
    pruning a linear is deterministic, so each linear needs to be pruned by ONE rank only.  Linears sharing an accumulator
    (one factorization) stay together; groups go to the least loaded rank, heaviest first (n^3 for the factorization +
    rows * n per sweep).  Returns {linear name: rank}, or None when running single / as replicas
    (`VLMC_SGPT_SHARD_LAYERS=0`: every rank prunes everything)."""
    rank, world = cal.calibration_shard()
    if world == 1 or os.environ.get("VLMC_SGPT_SHARD_LAYERS", "1") == "0":
        return None
    groups = {}
    for name in subset:
        groups.setdefault(id(wrapped[name]), []).append(name)

    def cost(names):
        n = subset[names[0]].weight.shape[1]
        return n ** 3 / 3 + sum(subset[m].weight.shape[0] * n * 128.0 for m in names)
    load = [0.0] * world
    owner = {}
    for names in sorted(groups.values(), key=lambda g: (-cost(g), g[0])):
        r = min(range(world), key=lambda k: (load[k], k))
        load[r] += cost(names)
        for m in names:
            owner[m] = r
    return owner


def _exchange_pruned(subset, owner, rank):
    """Every linear's pruned weights (and its importance score) travel from the rank that pruned it to all others."""
    import torch.distributed as dist
    names = list(subset)
    dev = subset[names[0]].weight.device
    scores = torch.zeros(len(names), dtype=torch.float64, device=dev)
    for k, name in enumerate(names):
        w = subset[name].weight
        if owner[name] == rank:
            scores[k] = float(getattr(w, "importance_score", 0.0))
        data = w.data if w.data.is_contiguous() else w.data.contiguous()
        dist.broadcast(data, src=owner[name])
        if data is not w.data:
            w.data.copy_(data)
    dist.all_reduce(scores)
    for k, name in enumerate(names):
        setattr(subset[name].weight, "importance_score", float(scores[k].item()))


def _allreduce_hessians(wrapped):
    """Multi-GPU: every rank accumulated H over its share of the samples; the global running mean
    is the sample-weighted average.  (Not in the reference, which runs replicas.)"""
    import torch.distributed as dist
    rank, world = cal.calibration_shard()
    if world == 1:
        return
    for w in wrapped:
        total = torch.tensor([float(w.nsamples)], device=w.H.device)
        w.H.mul_(w.nsamples)
        dist.all_reduce(w.H)
        dist.all_reduce(total)
        w.nsamples = int(total.item())
        w.H.div_(w.nsamples)
        if hasattr(w, "_folded"):
            w._folded = w.nsamples


@registry.register_pruner("t5_sparsegpt_pruner")
class T5LayerSparseGPTPruner(LayerWiseBasePruner, _SparseGPTBlockMixin):
    pruner_name = "t5_sparsegpt_pruner"

    def __init__(self, model, data_loader, model_prefix="t5_model", **kwargs):
        kw = dict(_KW); kw.update({k: v for k, v in kwargs.items() if k in _KW})
        super().__init__(model=model, data_loader=data_loader, model_prefix=model_prefix, **kw)

    def forward_to_cache(self, model, batch):
        return model(batch)

    @print_time
    def _prune(self, model, dataloader, device, model_prefix, module_to_process="encoder.block", n_samples=64,
               sparsity_ratio=0.5):
        cfg = getattr(model, model_prefix).config
        use_cache, cfg.use_cache = cfg.use_cache, False
        print("loading calibdation data")
        with torch.no_grad():
            inps, outs, caches = self._capture(model, dataloader, n_samples, module_to_process, vit=False,
                                               model_prefix=self.model_prefix)
        n_inps = len(inps) * cal.calibration_shard()[1]

        def prune_block(i, layer, subset, run_pass, state):
            self._sparsegpt_block(i, subset, run_pass, n_inps, module_to_process, sparsity_ratio)

        try:
            cal.walk_blocks(model, inps, outs, caches, module_to_process, n_samples,
                            lambda: model.maybe_autocast(dtype=torch.bfloat16), prune_block, tuple_output=True)
        finally:
            _flush_score_backlog(self)
        cfg.use_cache = use_cache
        cal.release_tower_memory()
        return model


@registry.register_pruner("vit_sparsegpt_pruner")
class VITLayerSparseGPTPruner(LayerWiseBasePruner, _SparseGPTBlockMixin):
    pruner_name = "vit_sparsegpt_pruner"

    def __init__(self, model, data_loader, model_prefix="visual", **kwargs):
        kw = dict(_KW); kw.update({k: v for k, v in kwargs.items() if k in _KW})
        super().__init__(model=model, data_loader=data_loader, model_prefix=model_prefix, **kw)

    def forward_to_cache(self, model, batch):
        return model.encode_image(batch["image"])

    @print_time
    def _prune(self, model, dataloader, device, model_prefix, module_to_process="encoder.block", n_samples=64,
               sparsity_ratio=0.5):
        with torch.no_grad():
            inps, outs, caches = self._capture(model, dataloader, n_samples, module_to_process, vit=True)
        n_inps = len(inps) * cal.calibration_shard()[1]

        def prune_block(i, layer, subset, run_pass, state):
            self._sparsegpt_block(i, subset, run_pass, n_inps, module_to_process, sparsity_ratio)

        try:
            cal.walk_blocks(model, inps, outs, caches, module_to_process, n_samples, lambda: model.maybe_autocast(),
                            prune_block, tuple_output=False, memo_cache=self.__dict__.get("_proxy_cache"))
        finally:
            _flush_score_backlog(self)
        cal.release_tower_memory()
        return model


@registry.register_pruner("blipt5_sparsegpt_pruner")
class BLIPT5LayerSparseGPTPruner(LayerWiseBasePruner, _SparseGPTBlockMixin):
    pruner_name = "blipt5_sparsegpt_pruner"

    def __init__(self, model, data_loader, t5_prune_spec=None, vit_prune_spec=None, t5_pruning_method=None,
                 vit_pruning_method=None, t5_model_prefix="t5_model", vit_model_prefix="visual_encoder", **kwargs):
        kw = dict(_KW); kw.update({k: v for k, v in kwargs.items() if k in _KW})
        kw["prune_spec"] = None
        super().__init__(model=model, data_loader=data_loader, model_prefix=f"{vit_model_prefix}+{t5_model_prefix}", **kw)
        self.t5_prune_spec = t5_prune_spec
        self.vit_prune_spec = vit_prune_spec
        assert t5_pruning_method is not None
        assert vit_pruning_method is not None
        self.t5_model_prefix = t5_model_prefix
        self.vit_model_prefix = vit_model_prefix

    def get_sparsity(self, original_sparsity, sparsity_ratio_granularity=None):
        return uniform_or_layer_sparsity(self, original_sparsity, sparsity_ratio_granularity)

    def forward_to_cache(self, model, batch):
        return model(batch)

    @cal.quiet_gc
    @print_time
    def prune(self, importance_scores=None, keep_indices_or_masks=None):
        print("In: ", self.pruner_name)
        dtype_record, requires_grad_record, device = self.model_setup_and_record_attributes(self.model)
        global_sparsity_dict = None
        if self.sparsity_ratio_granularity is not None:
            _, vit_keep_ratio, _, _ = self.convert_spec_to_list(self.vit_prune_spec)
            _, t5_keep_ratio, _, _ = self.convert_spec_to_list(self.t5_prune_spec)
            assert vit_keep_ratio == t5_keep_ratio
            global_sparsity_dict = self.get_sparsity(1 - vit_keep_ratio,
                                                     sparsity_ratio_granularity=self.sparsity_ratio_granularity)
        if self.vit_prune_spec is not None:
            _, keep_ratio, _, _ = self.convert_spec_to_list(self.vit_prune_spec)
            sd = global_sparsity_dict if global_sparsity_dict is not None else self.get_sparsity(1 - keep_ratio, None)
            self.model = VITLayerSparseGPTPruner._prune(self, self.model, self.data_loader, device,
                                                        model_prefix=self.vit_model_prefix,
                                                        module_to_process=f"{self.vit_model_prefix}.blocks",
                                                        n_samples=self.num_samples, sparsity_ratio=sd)
            self._done_towers = getattr(self, "_done_towers", []) + [f"{self.vit_model_prefix}.blocks"]
        if self.t5_prune_spec is not None:
            _, keep_ratio, _, _ = self.convert_spec_to_list(self.t5_prune_spec)
            sd = global_sparsity_dict if global_sparsity_dict is not None else self.get_sparsity(1 - keep_ratio, None)
            if "t5_model" in self.t5_model_prefix:
                for side in ("encoder", "decoder"):
                    self.model = T5LayerSparseGPTPruner._prune(self, self.model, self.data_loader, device,
                                                               model_prefix=self.t5_model_prefix,
                                                               module_to_process=f"{self.t5_model_prefix}.{side}.block",
                                                               n_samples=self.num_samples, sparsity_ratio=sd)
                    self._done_towers = getattr(self, "_done_towers", []) + [f"{self.t5_model_prefix}.{side}.block"]
            else:
                self.model = T5LayerSparseGPTPruner._prune(self, self.model, self.data_loader, device,
                                                           model_prefix=self.t5_model_prefix,
                                                           module_to_process=f"{self.t5_model_prefix}.model.decoder.layers",
                                                           n_samples=self.num_samples, sparsity_ratio=sd)
        self.model_reset(self.model, dtype_record, requires_grad_record, device)
        from vlmc import sparsegpt as _sg
        _sg.release_caches()          # factorization graphs + n x n work buffers are not kept past the prune
        return self.model, global_sparsity_dict
