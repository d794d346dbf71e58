"""Calibration capture and the per-block walk shared by the layer-wise pruners.

Behavioural restatement (not a copy) of the machinery every reference pruner repeats:
`get_module_recursive` / `find_layers` (wanda_pruner.py:16-48),
`prepare_calibration_input_encoder` with its `Catcher` (T5/LLM :213-273, ViT :583-625)
and the block loop skeleton of `_prune` (:275-354, :627-699): run the block on every
captured sample with forward hooks on its linears, let the method prune the linears,
run the block again with the pruned weights, swap inputs/outputs.

Quirks kept on purpose:
* the block-0 kwargs are replayed for every block, so T5 blocks 1.. run with
  `position_bias=None`;
* `find_layers` matches types exactly (a LoRA-wrapped linear is found as ONE module);
* the sample count is taken from `batch["image"]` / `batch["text_input"]`, and the
  loop stops only when a batch BEGINS at or past `n_samples`.

Multi-GPU extension (not in the reference): under an initialised
`torch.distributed` group each rank captures and replays only its contiguous share
of the calibration samples; the pruning method exchanges statistics
(`vlmc.wanda.gather_stats`) so that results are bit-identical on every rank and for
every world size.  `VLMC_SHARD_CALIB=0` restores the reference's replica behaviour.
"""
from __future__ import annotations

import contextlib
import os
import threading

import torch
import torch.nn as nn

from vlmc import forward, phases

# The engine's parts (round 6: this file was 2 700 lines): what they share, the grouping / padding of samples, finished towers, the capture
# phases.  Their names are re-exported here -- the pruners, bench.py and the tests import this module.
from lavis.compression.pruners import replay_capture, replay_padding, replay_state, replay_towers  # noqa: F401
from lavis.compression.pruners.replay_state import (  # noqa: F401
    GRAPH_MIN_SAMPLES,
    LLM_KEYS,
    MEMO_MAX_BYTES,
    OPT_KEYS,
    REPLAY_GROUP_DEFAULT,
    REPLAY_TOKEN_BUDGET,
    T5_KEYS,
    _CTX,
    _PruneContext,
    _Stop,
    _bits_equal,
    _keys_for,
    block_tensors,
    calibration_shard,
    capture_graph,
    capture_streams,
    find_layers,
    get_module_recursive,
    graph_replay_enabled,
    graph_stats,
    later_check_enabled,
    pad_ragged_enabled,
    prunable_layer_types,
    quiet_gc,
    release_tower_memory,
    replay_group_size,
    storage_signature,
    tower_batch_enabled,
    tower_graph_enabled,
    tower_memo_enabled,
    tower_pad_enabled,
    tower_predict_enabled,
)
from lavis.compression.pruners.replay_padding import (  # noqa: F401
    PAD_MASK_KEYS,
    PAD_STATE_KEYS,
    _pad_caches,
    _pad_inputs,
    _stack_caches,
    _stack_key,
    int32_on,
    plan_groups,
    plan_padded,
    row_map,
    stacked_lengths,
    stacked_samples,
)
from lavis.compression.pruners.replay_towers import (  # noqa: F401
    FROZEN_TOWERS,
    GraphedModule,
    TowerGraph,
    TowerMemo,
    _Defer,
    _HiddenOnly,
    _LaterEqual,
    _memo_kind,
    _wrap_towers,
    seed_tower_memo,
    with_frozen_towers,
)
from lavis.compression.pruners.replay_capture import (  # noqa: F401
    MERGED_CAPTURE_MIN,
    _batch_signature,
    _capture_block_inputs,
    _capture_merged,
    _capture_once,
    _merge_batches,
    all_linears,
    capture_block_inputs,
    merged_capture_enabled,
)


class _TailStop(ValueError):
    """Raised by `statistics_only` when the block's last linear has handed its input to the statistics hooks."""


def tail_skip_enabled():
    """`VLMC_SKIP_DEAD_TAIL=0`: run the statistics pass to the end of the block like the reference (:308-311 writes
    `outs[j]` in that pass too, and overwrites every one of them in the second pass before anything reads them)."""
    return os.environ.get("VLMC_SKIP_DEAD_TAIL", "1") != "0"


class statistics_only:
    """The first pass over a block exists for the hooks on its linears' INPUTS (wanda_pruner.py:303-311,
    sparsegpt_pruner.py add_batch(inp, out) never reads `out`): what the block computes after its last linear has been
    given its input -- that linear's own product (fc2 of a ViT block is 27 % of the block's flops), the residual add
    behind it -- is dead, the second pass overwrites `outs`.  Inside this context the block's forward ends there.

    Which linear is the last one is LEARNED, not assumed: the first forward of a tower's first block runs to the end
    while the call order of its linears is recorded; afterwards the tail is cut only in a forward whose calls so far
    are exactly that order (each linear once, same sequence) -- a block that calls a linear twice, or in a different
    order, or has other linears than the first block had, runs to the end.  The statistics are the same bits either
    way (tests/test_pruner_host_logic.py, tests/test_replay_invariance_gpu.py)."""

    def __init__(self, subset, learned):
        self.subset, self.learned = subset, learned          # learned: dict shared by the tower's blocks
        self.seen, self.handles, self.last = [], [], None

    def __enter__(self):
        if not tail_skip_enabled() or self.learned.get("order") is False or any(m.training for m in self.subset.values()):
            return self                                      # (a training-mode block may draw random numbers behind the cut)
        names = {id(m): n for n, m in self.subset.items()}

        def note(mod, args):
            self.seen.append(names[id(mod)])

        self.handles = [m.register_forward_pre_hook(note) for m in self.subset.values()]
        order = self.learned.get("order")
        if order and set(order) == set(self.subset) and len(order) == len(self.subset):
            last = self.subset[order[-1]]
            inner = last.forward                             # (possibly forward.invariant_linears' patch)

            def dead_product(x, *a, **kw):
                if tuple(self.seen) != order:                # not the sequence that was learned: compute
                    return inner(x, *a, **kw)
                return None                                  # the product is never formed: a hook that reads `out` fails loudly

            def stop(mod, args, out):
                if tuple(self.seen) == order:
                    raise _TailStop
            stop._vlmc_engine_hook = True                    # (not a statistics hook: graph replays do not call it)

            self._had = "forward" in last.__dict__
            self._inner = inner
            last.forward = dead_product
            self.handles.append(last.register_forward_hook(stop))     # after the pruner's hooks: they have seen the input
            self.last = last
        return self

    def new_forward(self):
        self.seen = []

    def end_forward(self, completed):
        """Called after every forward of the block; `completed` = it ran to the end (no _TailStop)."""
        if completed and self.learned.get("order") is None:
            ok = len(set(self.seen)) == len(self.seen) == len(self.subset) and len(self.seen) > 1
            self.learned["order"] = tuple(self.seen) if ok else False

    def __exit__(self, *exc):
        for h in self.handles:
            h.remove()
        if self.last is not None:
            if self._had:
                self.last.forward = self._inner
            else:
                del self.last.__dict__["forward"]
        return False


class BlockGraph:
    """One block forward captured in a HIP graph and replayed for every calibration sample of the same shape.

    A batch-1 forward of a T5 / ViT block is ~30 small kernels whose launch and Python dispatch cost dwarfs their
    GPU time; the 2 x 128 x 87 of them are what a prune spends its time on once the statistics and select kernels
    take milliseconds.  The graph replays the very same kernels on static buffers, so activations -- and with them
    statistics and masks -- are bit-identical to the eager loop.

    The statistics hooks on the block's linears cannot run inside a graph.  During capture they are replaced by
    recorders that keep every linear's input / output tensor alive (so the graph's memory pool never recycles
    them); after each replay the real hooks are called on those static tensors, exactly as a forward would."""

    def __init__(self, layer, x, cache, subset, autocast, tuple_output):
        from collections import OrderedDict
        self.x = x.clone()
        self.cache = {k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in cache.items()}
        self.records = []
        self.storage = storage_signature(layer)
        modules = list(subset.values())
        saved = [m._forward_hooks for m in modules]

        def recorder(mod, inp, out):
            self.records.append((mod, inp[0], out))
        try:
            for m in modules:
                m._forward_hooks = OrderedDict({0: recorder})
            side = torch.cuda.Stream(device=x.device)
            side.wait_stream(torch.cuda.current_stream(x.device))
            with torch.cuda.stream(side), torch.no_grad(), autocast():
                layer(self.x, **self.cache)                   # warm-up: lazy initialisation must not land in the capture
            torch.cuda.current_stream(x.device).wait_stream(side)
            self.records.clear()
            versions = []

            def recorder(mod, inp, out):                             # noqa: F811  (the capture's recorder also notes versions)
                self.records.append((mod, inp[0], out))
                versions.append((inp[0], inp[0]._version))
            for m in modules:
                m._forward_hooks = OrderedDict({0: recorder})
            def body():
                with torch.no_grad(), autocast():
                    return layer(self.x, **self.cache)
            self.graph, y = capture_graph(body, x.device)
            self.y = y[0] if tuple_output else y
            # the real hooks run AFTER the whole replay, on these static tensors: a block that writes into a linear's
            # input in place after the linear has run would show them other activations than the eager loop does
            if any(t._version != v for t, v in versions):
                raise RuntimeError("the block modifies a hooked linear's input in place after the linear ran")
        finally:
            for m, h in zip(modules, saved):
                m._forward_hooks = h
        graph_stats["captured"] += 1

    def run(self, x, cache):
        self.x.copy_(x)
        for k, v in cache.items():
            if isinstance(v, torch.Tensor):
                self.cache[k].copy_(v)
        self.graph.replay()
        for mod, xin, out in self.records:
            for hook in list(mod._forward_hooks.values()):
                hook(mod, (xin,), out)
        graph_stats["replayed"] += 1
        return self.y.clone()


def walk_blocks(model, inps, outs, caches, module_to_process, n_samples, autocast, prune_block, tuple_output,
                memo_cache=None, pad_ragged=False):
    """The block loop of `_prune`: for every block, `prune_block(i, layer, subset, run)`
    is called with `run()` = one pass of the block over all samples (filling `outs`);
    afterwards the block runs again with whatever weights `prune_block` left, and
    inputs/outputs swap (wanda_pruner.py:287-347).

    Batched replay (default, `VLMC_BATCH_REPLAY=G` sets the group size): up to G samples whose inputs
    and cached kwargs have identical shapes (`plan_groups`) are concatenated along the batch dimension and go through
    the block in one call -- batch-1 forwards of a 2048-wide block leave the matrix cores idle, and the 2 x 128 x 87 of
    them are > 75 % of a FlanT5-XL prune once the statistics and select kernels take 14 ms.  Per-sample statistics are
    kept in the reference's sample order (the hooks see `stacked_samples()`).  `VLMC_BATCH_REPLAY=1` is the reference's
    per-sample loop (:308-311, :343-346), replayed from HIP graphs.  Either way the activations are the GPU's, not
    the reference host's: both modes track the reference's masks to the same near-tie agreement
    (tests/test_pruner_gpu.py::test_replay_modes_agree_with_the_reference_side_by_side)."""
    layers = get_module_recursive(model, module_to_process)
    n_samples = min(n_samples, len(inps))
    state = {"inps": inps, "outs": outs}
    group_max = replay_group_size()

    def run_pass(before_sample=None, outputs=True):
        """`outputs=False`: the caller only wants its hooks on the linears fed (the first pass of every pruner)."""
        with phases.phase("replay"):
            if outputs or group_max == 1:
                _run_pass(before_sample, None, True)
            else:
                with statistics_only(subset, tail_learned) as so:
                    _run_pass(before_sample, so, False)

    def materialise(lst):
        """the per-sample views of a padded pass's output that were not cut when it ran (below)"""
        pend = lazy.pop(id(lst), None)
        for key_, (y_, lens_) in (pend or {}).items():
            slices_ = [y_[t:t + 1, :tj] for t, tj in enumerate(lens_)]
            slices_[0]._vlmc_stack = (y_, key_, slices_)
            for t, j in enumerate(key_):
                lst[j] = slices_[t]

    def _run_pass(before_sample, so, outputs):
        cur_in, cur_out = state["inps"], state["outs"]
        if outputs:
            lazy.pop(id(cur_out), None)                               # (whatever an earlier pass left pending there is overwritten now)
        pend_in = lazy.get(id(cur_in))
        if pend_in is not None and (group_max == 1 or "chunks" not in plan or {tuple(c) for c in plan["chunks"]} != set(pend_in)
                                     or any(plan["pad"].get(k_) is None for k_ in pend_in)):
            materialise(cur_in)                                       # (another route than the padded groups the outputs were left stacked for)
            pend_in = None
        keys = None
        sig = None
        for k_ in [k_ for k_, g_ in graphs.items() if g_ is not False]:     # the first pass's graphs, if the block's
            sig = storage_signature(layer) if sig is None else sig          # tensors are still where they were
            if graphs[k_].storage != sig:
                del graphs[k_]
        if group_max == 1 and graph_replay_enabled() and n_samples and cur_in[0].is_cuda and \
                not getattr(layer, "_vlmc_no_graph", False):
            keys = [_stack_key(cur_in[t], caches[t]) for t in range(n_samples)]
            counts = {}
            for k in keys:
                counts[k] = counts.get(k, 0) + 1
        if keys is not None:                                      # per-sample loop, replayed from HIP graphs
            j = 0
            while j < n_samples:
                if before_sample is not None:
                    before_sample(j)
                bg = None
                if counts[keys[j]] >= GRAPH_MIN_SAMPLES and graphs.get(keys[j]) is not False:
                    bg = graphs.get(keys[j])
                    if bg is None:
                        try:
                            bg = graphs[keys[j]] = BlockGraph(layer, cur_in[j], caches[j], subset, autocast, tuple_output)
                        except Exception as e:      # block not capturable (host sync, data-dependent shapes): eager loop
                            graphs[keys[j]] = False
                            layer._vlmc_no_graph = True         # do not try again in the second pass
                            graph_stats["fallbacks"] += 1
                            print(f"graph replay disabled for this block ({type(e).__name__}: {e})")
                            bg = None
                if bg is not None:
                    cur_out[j] = bg.run(cur_in[j], caches[j])
                else:
                    with torch.no_grad(), autocast():
                        y = layer(cur_in[j], **caches[j])
                    cur_out[j] = y[0] if tuple_output else y
                j += 1
            return
        # blocks of a tower map [.., T, d] to [.., T, d]: the plan of the first pass holds while the shapes do
        # (inputs that are still one stacked tensor per padded group: the shapes are the plan's -- a block maps [.., T, d] to [.., T, d])
        shapes = plan["shapes"] if pend_in is not None else [tuple(cur_in[j].shape) for j in range(n_samples)]
        if plan.get("shapes") != shapes:
            plan["shapes"] = shapes
            padded = plan_padded(cur_in, caches, n_samples, group_max) if (pad_ragged and group_max > 1) else None
            if padded is not None:                                    # ragged samples: padded groups (`pad_ragged`: the caller's hooks take lengths)
                plan["chunks"] = [c for c, _ in padded]
                plan["pad"] = {tuple(c): sp for c, sp in padded}
            else:
                plan["chunks"] = plan_groups(cur_in, caches, n_samples, group_max) if group_max > 1 else [[j] for j in range(n_samples)]
                plan["pad"] = {}
        chunks = plan["chunks"]
        kind = "full" if outputs else "stat"
        for chunk in chunks:
            if before_sample is not None:
                before_sample(chunk[0])
            if so is not None:
                so.new_forward()
            with torch.no_grad(), autocast():
                if len(chunk) == 1:
                    j = chunk[0]
                    try:
                        y = layer(cur_in[j], **caches[j])
                    except _TailStop:
                        continue
                    if so is not None:
                        so.end_forward(True)
                    cur_out[j] = y[0] if tuple_output else y
                elif plan["pad"].get(tuple(chunk)) is not None:
                    # a PADDED group of ragged samples: one forward; the outputs are handed on padded (the next block takes
                    # the same tensor), every sample sees its own rows of it
                    key, spec = tuple(chunk), plan["pad"][tuple(chunk)]
                    prev = getattr(cur_in[chunk[0]], "_vlmc_stack", None) if pend_in is None else None
                    if pend_in is not None:
                        x = pend_in[key][0]                           # the previous block's stacked output, never cut per sample
                    elif prev is not None and prev[1] == key and all(cur_in[j] is prev[2][t] for t, j in enumerate(chunk)):
                        x = prev[0]
                    else:
                        x = _pad_inputs([cur_in[j] for j in chunk], spec["tp"])
                    kw = stacked_kwargs.get(key)
                    if kw is None:
                        kw = stacked_kwargs[key] = _pad_caches([caches[j] for j in chunk], spec)
                    _CTX.stacked, _CTX.stacked_lengths = (len(chunk), 1, key), spec["lengths"]
                    graph_stats["padded_forwards"] = graph_stats.get("padded_forwards", 0) + 1
                    try:
                        with forward.padded_rows(spec.get("rows"), spec["lengths"]):     # linears and attention skip the padding rows
                            y = layer(x, **kw)
                    except _TailStop:
                        continue
                    finally:
                        _CTX.stacked = _CTX.stacked_lengths = None
                    if so is not None:
                        so.end_forward(True)
                    y = y[0] if tuple_output else y
                    # the next block takes the stacked tensor itself; the samples' own views of it (128 x 2 indexing calls: 0.25 ms per
                    # block of a host-paced tower) are cut when somebody wants them -- another route, or the end of the walk
                    lazy.setdefault(id(cur_out), {})[key] = (y, spec["T"])
                else:
                    b0 = cur_in[chunk[0]].shape[0]
                    _CTX.stacked = (len(chunk), b0, tuple(chunk))
                    key = tuple(chunk)
                    # the group's inputs are usually the slices of ONE tensor -- the previous pass's stacked output -- and the
                    # cached kwargs are the same for every block of the tower: neither needs concatenating again
                    prev = getattr(cur_in[chunk[0]], "_vlmc_stack", None)
                    if prev is not None and prev[1] == key and all(cur_in[j] is prev[2][t] for t, j in enumerate(chunk)):
                        x = prev[0]
                    else:
                        x = torch.cat([cur_in[j] for j in chunk], dim=0)
                    kw = stacked_kwargs.get(key)
                    if kw is None:
                        kw = _stack_caches([caches[j] for j in chunk], b0)
                        kw = stacked_kwargs[key] = kw if kw is not None else False
                    if kw is False:                                 # kwargs that cannot be stacked: sample by sample
                        _CTX.stacked = None
                        for t, j in enumerate(chunk):
                            if t and before_sample is not None:     # (chunk[0] was announced above; the per-sample
                                before_sample(j)                    # loop announces every sample, like group_max == 1)
                            if so is not None:
                                so.new_forward()
                            try:
                                y = layer(cur_in[j], **caches[j])
                            except _TailStop:
                                continue
                            if so is not None:
                                so.end_forward(True)
                            cur_out[j] = y[0] if tuple_output else y
                        continue
                    try:
                        y = layer(x, **kw)
                    except _TailStop:
                        continue
                    finally:
                        _CTX.stacked = None
                    if so is not None:
                        so.end_forward(True)
                    y = y[0] if tuple_output else y
                    slices = list(y.split(b0, dim=0))                   # (one call: 128 Python-level slices were 0.1 ms per block pass)
                    slices[0]._vlmc_stack = (y, key, slices)
                    for t, j in enumerate(chunk):
                        cur_out[j] = slices[t]

    graphs, plan, stacked_kwargs, tail_learned, lazy = {}, {}, {}, {}, {}
    sibling_names = []
    for i in range(len(layers)):
        layer = layers[i]
        subset = find_layers(layer)
        graphs.clear()             # per block; the second pass reuses the first one's graphs when storage is unchanged
        # the block's linears run on the batch-invariant MFMA kernel in both passes (vlmc/forward.py): how the samples are
        # grouped -- or sharded over GPUs -- does not reach the statistics
        # linears fed one tensor (q / k / v, wi_0 / wi_1) share a launch: the groups the first block's forward showed are
        # carried over to the following blocks by name, so that their first pass is fused too (vlmc/forward.py)
        for names in sibling_names:
            if all(n in subset for n in names):
                forward.register_siblings([subset[n] for n in names])
        with forward.invariant_linears(subset.values(), roots=(layer,)):
            prune_block(i, layer, subset, run_pass, state)
            run_pass()
        if i == 0:
            by_id = {id(m): n for n, m in subset.items()}
            sibling_names = [tuple(by_id[id(m)] for m in g) for g in forward.sibling_groups(subset.values())
                             if all(id(m) in by_id for m in g)]
        state["inps"], state["outs"] = state["outs"], state["inps"]
    materialise(state["inps"])
    materialise(state["outs"])
    if memo_cache is not None and not tuple_output:
        seed_tower_memo(memo_cache, module_to_process, layers, state["inps"][:n_samples], autocast)
    return model
