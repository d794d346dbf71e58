"""The capture phases: run the model's own forward until block 0 of the tower that is to be pruned (the reference's Catcher,
wanda_pruner.py:213-273 / :583-625) -- per calibration batch as the reference does (`_capture_once`), or with the batches of one shape stacked
into one forward and checked against a sample's own forward bit for bit (`_capture_merged`).  Split out of `calibration.py` in round 6."""
from __future__ import annotations

import contextlib
import os
import threading

import torch
import torch.nn as nn

from vlmc import forward, phases

from lavis.compression.pruners import replay_towers as towers_
from lavis.compression.pruners.replay_state import (  # noqa: F401
    REPLAY_TOKEN_BUDGET,
    _CTX,
    _Stop,
    _keys_for,
    calibration_shard,
    capture_streams,
    get_module_recursive,
    graph_replay_enabled,
    graph_stats,
    later_check_enabled,
    replay_group_size,
    tower_batch_enabled,
    tower_graph_enabled,
    tower_memo_enabled,
    tower_pad_enabled,
    tower_predict_enabled,
)
from lavis.compression.pruners.replay_padding import (  # noqa: F401
    PAD_MASK_KEYS,
)
from lavis.compression.pruners.replay_towers import (  # noqa: F401
    TowerGraph,
    TowerMemo,
    _LaterEqual,
    _wrap_towers,
    with_frozen_towers,
)


def capture_block_inputs(model, dataloader, n_samples, module_to_process, forward_to_cache, lora_model, *, vit,
                         model_prefix=None, count_batches=False, done_towers=None, proxy_cache=None):
    """Run the model until block 0 of `module_to_process` is reached, for the first
    `n_samples` calibration samples; return (inps, outs, caches) like the reference.

    vit=True  -> catcher signature (inp, rel_pos_bias, dense=True)         (:595-608)
    vit=False -> catcher signature (inp, dense=True, **kwargs), caching the
                 family's kwargs (:238-253)
    count_batches=True reproduces the SparseGPT pruners' stop rule (`i >= n_samples` on the
    batch index, sparsegpt_pruner.py:391-393) instead of Wanda's sample count.
    """
    done_towers = with_frozen_towers(model, done_towers, module_to_process)
    with phases.phase("capture " + module_to_process):
        return _capture_block_inputs(model, dataloader, n_samples, module_to_process, forward_to_cache, lora_model, vit=vit,
                                     model_prefix=model_prefix, count_batches=count_batches, done_towers=done_towers,
                                     proxy_cache=proxy_cache)


def _capture_block_inputs(model, dataloader, n_samples, module_to_process, forward_to_cache, lora_model, *, vit,
                          model_prefix, count_batches, done_towers, proxy_cache):
    total, batches = 0, []
    for batch in dataloader:                       # which batches the reference would consume
        if total >= n_samples:
            break
        if count_batches:
            total += 1
        elif vit or "image" in batch:
            total += batch["image"].shape[0]
        else:
            total += len(batch["text_input"])
        batches.append(batch)
    args = (model, batches, module_to_process, forward_to_cache, lora_model)
    kw = dict(vit=vit, model_prefix=model_prefix, done_towers=done_towers, proxy_cache=proxy_cache)
    p0 = next(model.parameters(), None)
    if merged_capture_enabled() and p0 is not None and p0.is_cuda and _CTX.later is None:
        _CTX.later = _LaterEqual()                              # (remembered tower inputs against what the merged forward feeds them)
        # (what the attempt finds in the cache: a memo whose `entries` dict is still the same object afterwards was only replayed
        # from -- `begin("record")` starts a new dict; a memo seeded by the walk, seed_tower_memo, is in "record" mode without ever
        # having been begun: the mode alone does not tell)
        keys_before = {k: (id(v.entries) if isinstance(v, TowerMemo) else None) for k, v in proxy_cache.items()} if proxy_cache is not None else {}
        try:
            try:
                res = _capture_merged(*args, **kw)
            except (RuntimeError, TypeError, IndexError, AssertionError, AttributeError) as e:
                # a model whose forward does not take the stacked batch (it assumes batch 1 somewhere): its own way, per sample --
                # if the trouble is not the merge (out of memory, a broken model) the per-sample route meets it again and raises
                # Out of memory is not a refusal of the stacked batch, and `VLMC_STRICT=1` (CI of the engine itself) wants every such
                # exception raised: a regression inside the merged path must not hide behind its fallback.
                if isinstance(e, torch.cuda.OutOfMemoryError) or os.environ.get("VLMC_STRICT") == "1":
                    raise
                res = None
                graph_stats["merged_capture_errors"] = graph_stats.get("merged_capture_errors", 0) + 1
                import warnings
                warnings.warn(f"vlmc: the stacked calibration forward was declined ({type(e).__name__}: {e}); this capture phase forwards "
                              "one calibration batch at a time, as the reference does", RuntimeWarning)
            bad = res is not None and _CTX.later.failed()
        finally:
            _CTX.later = None
        if res is not None and not bad:
            return res
        graph_stats["merged_capture_declined"] = graph_stats.get("merged_capture_declined", 0) + 1
        # Whatever the declined attempt left behind must not serve the per-sample route that follows: a TowerMemo it
        # created -- or re-recorded: `_wrap_towers` begins a stale memo anew with an empty record -- holds outputs cut out of the merged
        # forward, the very values the comparison has just refused (or never checked); likewise the block-0 arguments / catcher
        # calls it noted for the next phase.  Memos that only REPLAYED during the attempt hold the previous phase's per-sample
        # records and stay, unless a remembered input was refuted (`bad`): then every record goes, as on the per-sample route.
        for key, val in list(proxy_cache.items() if proxy_cache is not None else []):
            if isinstance(val, TowerMemo):
                rewritten = key not in keys_before or keys_before[key] != id(val.entries)
                if bad or rewritten:
                    val._drop()
                    if rewritten:
                        del proxy_cache[key]
            elif isinstance(key, tuple) and key and key[0] in ("calls", "block0") and (bad or key not in keys_before):
                del proxy_cache[key]
    if done_towers and proxy_cache is not None and later_check_enabled() and p0 is not None and p0.is_cuda and _CTX.later is None:
        # what finished towers remember of the previous phase is trusted while the forwards run and verified afterwards
        _CTX.later = _LaterEqual()
        try:
            res = _capture_once(*args, **kw)
            bad = _CTX.later.failed()
        finally:
            _CTX.later = None
        if not bad:
            return res
        # a remembered input was not what this phase fed its tower: forget the records, run the phase again, comparing at once
        graph_stats["later_failed"] = graph_stats.get("later_failed", 0) + 1
        for key, val in list(proxy_cache.items()):
            if isinstance(val, TowerMemo):
                val._drop()
            elif isinstance(key, tuple) and key and key[0] == "block0":
                del proxy_cache[key]                       # ... nor the block-0 arguments remembered for run_predicted
    return _capture_once(*args, **kw)


try:
    MERGED_CAPTURE_MIN = max(2, int(os.environ.get("VLMC_CAPTURE_MERGED_MIN", "24")))
except ValueError:
    MERGED_CAPTURE_MIN = 24


def merged_capture_enabled():
    """Calibration batches of one structure run the model's forward to the next tower as ONE stacked batch
    (`_capture_merged`; `VLMC_CAPTURE_MERGED=0`: one forward per calibration batch, as the reference's loop)."""
    return os.environ.get("VLMC_CAPTURE_MERGED", "1") != "0" and tower_batch_enabled() and replay_group_size() > 1 and \
        graph_replay_enabled() and torch.cuda.is_available()


def _batch_signature(batch):
    if not isinstance(batch, dict):
        return None
    sig = []
    for k in sorted(batch):
        v = batch[k]
        if isinstance(v, torch.Tensor):
            if v.dim() < 1 or v.requires_grad:
                return None
            sig.append((k, "T", tuple(v.shape), v.dtype, v.device))
        elif isinstance(v, (list, tuple)):
            sig.append((k, "L", type(v), len(v), tuple(type(e) for e in v)))
        elif v is None or isinstance(v, (bool, int, float, str)):
            sig.append((k, "V", v))
        else:
            return None
    return tuple(sig)


def _merge_batches(batches):
    out = {}
    for k, v0 in batches[0].items():
        if isinstance(v0, torch.Tensor):
            out[k] = torch.cat([b[k] for b in batches], dim=0)
        elif isinstance(v0, (list, tuple)):
            out[k] = type(v0)(e for b in batches for e in b[k])
        else:
            out[k] = v0
    return out


def all_linears(model, proxy_cache):
    """Every exact-type nn.Linear of the model (one walk per prune): during a capture phase they all run on the batch-invariant
    kernel -- the towers' and the glue between them (`t5_proj`, ..) -- so that what a sample's forward hands the next tower does
    not depend on how many samples share the forward."""
    key = ("all_linears", id(model))
    lin = proxy_cache.get(key) if proxy_cache is not None else None
    if lin is None:
        lin = [m for m in model.modules() if type(m) is nn.Linear]
        if proxy_cache is not None:
            proxy_cache[key] = lin
    return lin


def _capture_merged(model, batches, module_to_process, forward_to_cache, lora_model, *, vit, model_prefix, done_towers, proxy_cache):
    """The capture phase with the calibration batches of one structure STACKED into one forward of the model.

    The reference forwards every calibration batch on its own up to the tower that is to be pruned
    (wanda_pruner.py:213-273: Catcher); 128 batch-1 forwards of the model's Python per phase were half of a prune's wall-clock
    once the towers themselves ran stacked (the host paced them, the GPU idled), and a never-pruned tower in the way -- the
    Q-Former -- cost every sample an aborted forward and a repeated one.  The model's own forward takes batches: the samples
    whose batch dicts have one structure (same tensor shapes, same list lengths) are concatenated and forwarded ONCE; the
    Catcher's one call is cut back into per-sample inputs and kwargs.  What makes a sample's slice carry the bits of its own
    forward: every nn.Linear on the way runs on the batch-invariant kernel (`all_linears`), attention and norms of the finished
    towers on the invariant kernels of vlmc/forward.py, everything else on the way is row-wise.  It is CHECKED, not assumed:
    sample 0 is also forwarded alone (the same route, batch 1); its captured tensors say which kwargs carry the batch
    dimension, and they must equal slice 0 of the merged capture bit for bit -- otherwise, or when a batch does not merge, or
    when the model turns out to pad inside the merged forward (mask kwargs that differ between samples), this returns None
    and the phase runs the reference's way (`_capture_once`).  Finished towers are entered through their proxies: a tower
    whose outputs are remembered from its own walk (the ViT) hands them over stacked, the others run as the model calls them.
    """
    rank, world = calibration_shard()
    if world > 1:
        if len(batches) % world != 0:
            return None
        per = len(batches) // world
        mine = batches[rank * per:(rank + 1) * per]
    else:
        mine = batches
    if len(mine) < MERGED_CAPTURE_MIN:
        return None                                              # (few samples: the forward of one sample alone that the merge is checked against costs what it saves)
    sigs = [_batch_signature(b) for b in mine]
    if any(s_ is None for s_ in sigs):
        return None
    order, groups = [], {}
    for j, s_ in enumerate(sigs):
        if s_ not in groups:
            groups[s_] = []
            order.append(s_)
        groups[s_].append(j)
    layers = get_module_recursive(model, module_to_process)
    keys = None if vit else _keys_for(model_prefix)
    final = proxy_cache is not None and proxy_cache.get(("last_tower",)) == module_to_process
    want_calls = vit and proxy_cache is not None and tower_memo_enabled() and graph_replay_enabled()
    got = []

    class MergedCatcher(nn.Module):
        def __init__(self, module):
            super().__init__()
            self.module = module

        def forward(self, inp, *args, **kwargs):
            got.append((inp, args, dict(kwargs)))
            raise _Stop

    def run(idxs, alone=False):
        """-> the Catcher's call, None (the forward did not reach it), or "later" (postponed at a finished tower)"""
        got.clear()
        batch = mine[idxs[0]] if len(idxs) == 1 else _merge_batches([mine[j] for j in idxs])
        n_def = sum(len(t.deferred) for t in towers)
        if alone:
            _CTX.capture_sample = idxs[0]                       # the per-sample route: finished towers are traced (their wiring, their shapes)
        else:
            _CTX.capture_group = list(idxs)
        try:
            forward_to_cache(model, batch, lora_model)
        except ValueError:                                     # _Stop / _Defer, or the reference's bare ValueError
            pass
        finally:
            _CTX.capture_group = _CTX.capture_sample = None
        if len(got) == 1:
            return got[0]
        return "later" if sum(len(t.deferred) for t in towers) > n_def else None

    def tensors_of(call):
        inp, args, kw = call
        return [("#inp", inp)] + [(f"#{i}", a) for i, a in enumerate(args)] + sorted(kw.items())

    layers[0] = MergedCatcher(layers[0])
    undo = _wrap_towers(model, [t for t in (done_towers or []) if t != module_to_process], proxy_cache, record=not final)
    arrived, calls = [], []
    # how the model calls block 0 of THIS tower, by sample: should the next phase take the per-sample route, its stacked pass
    # through this tower starts from these (TowerGraph.run_predicted)
    first = {} if (proxy_cache is not None and not final and tower_graph_enabled() and tower_predict_enabled()) else None
    try:
        towers = []
        for blocks_, i_, _orig in undo:
            tg_ = blocks_[i_].__dict__.get("_tower")
            if tg_ is not None and not any(tg_[0] is t for t in towers):
                towers.append(tg_[0])
        order.sort(key=lambda s_: -len(groups[s_]))             # the largest group first: its first sample is the one forwarded alone
        scout = groups[order[0]][0]
        if len(groups[order[0]]) < 2:
            return None                                          # nothing to merge
        # ragged batches (more than two shapes): a merged forward per token count would run every finished tower once per count
        # (measured: 532 against 500 ms for the per-sample route, which pads them into one stacked pass).  Instead every group's
        # forward is postponed at a finished tower's first block, the tower runs ONCE, padded, for the samples of all groups
        # (TowerGraph.enter_group -> run_deferred -> _run_padded) and the groups' forwards are repeated.
        defer = len(order) > 2
        if defer and not (os.environ.get("VLMC_CAPTURE_MERGED_RAGGED", "1") == "1" and tower_pad_enabled() and tower_graph_enabled()):
            # (on since round 6: the encoder's phase of the ragged reference-op prune is 55-59 ms instead of 100 in the synchronising
            # phase timers.  In round 5 the whole prune was level with it -- the per-sample Python it removes ran behind the GPU tail of
            # the preceding walk -- but with the linears skipping the padding rows that tail is shorter and the host shows: 479 -> 463 ms,
            # same box, tools/ragged_prof.py)
            return None
        pruned_on_the_way = [t for t in towers if not t.memo_serves and t.path not in towers_.FROZEN_TOWERS]
        if defer and pruned_on_the_way and (os.environ.get("VLMC_CAPTURE_MERGED_PRUNED", "1") == "0" or
                                            not all(t.predicted for t in pruned_on_the_way)):
            # ragged batches and a PRUNED tower on the way whose outputs are not remembered (the decoder's phase: 24 encoder blocks).
            # Round 5 measured 145-173 ms for that phase on this route against 87 per sample: every group's forward ran twice (postponed
            # at the tower, repeated), and handing a group its blocks' outputs stacked the samples' pieces per block and output -- the
            # position bias 24 times.  Since round 6 the tower runs BEFORE the forwards on its remembered block-0 arguments
            # (run_predicted, as on the per-sample route) and a tensor several blocks hand on is cut and stacked once.  Without
            # remembered arguments (a tower whose own phase did not run through this module): the per-sample route.
            return None
        flags = []                                               # device-side verdicts, read once at the end (no wait per forward)
        with torch.no_grad(), forward.invariant_linears(all_linears(model, proxy_cache), roots=[b for t in towers for b in t.mods]):
            everyone = list(range(len(mine)))
            group_of = {j: groups[s_][0] for s_ in order for j in groups[s_]}     # (the samples of a group stay neighbours in a tower's padded stack)
            if defer:
                for t in pruned_on_the_way:
                    t.run_predicted(everyone, group_of)         # (towers whose wiring a previous prune traced: now; else after the scout has traced it)
            _CTX.keep_ready = defer
            one = run([scout], alone=defer)                     # one sample alone: the shapes of a batch-1 call, and the bits to hold the merge to
            for _ in range(len(towers) + 1):                     # (a tower whose wiring an earlier phase traced postpones this forward too)
                if one != "later":
                    break
                for t in towers:
                    if t.deferred:
                        t.run_deferred()
                one = run([scout], alone=True)
            _CTX.keep_ready = False
            if one is None or one == "later" or not isinstance(one[0], torch.Tensor) or one[0].dim() < 2:
                return None
            if defer:
                for t in pruned_on_the_way:
                    t.run_predicted(everyone, group_of)         # (no-op for the samples that have their outputs)
            names1 = tensors_of(one)
            batched = None                                       # name -> the per-sample batch extent of a tensor that carries the batch dimension, else 0
            rows = max(1, one[0].numel() // max(1, one[0].shape[-1]))
            per = max(2, min(replay_group_size(), REPLAY_TOKEN_BUDGET // rows))
            pending = [groups[s_][c0:c0 + per] for s_ in order for c0 in range(0, len(groups[s_]), per)]
            _CTX.group_defer = defer
            sweeps = 0
            while pending:
                sweeps += 1
                if sweeps > len(towers) + 2:
                    return None
                again = []
                for chunk in pending:
                    call = run(chunk)
                    if call == "later":
                        again.append(chunk)
                        continue
                    if call is None or not isinstance(call[0], torch.Tensor):
                        return None
                    g = len(chunk)
                    names = tensors_of(call)
                    if len(names) != len(names1) or [n for n, _ in names] != [n for n, _ in names1]:
                        return None
                    learn = batched is None
                    if learn:
                        if g < 2 or chunk[0] != scout:
                            return None
                        batched = {}
                    pieces = {}
                    for (name, v), (_n1, v1) in zip(names, names1):
                        if isinstance(v, torch.Tensor) != isinstance(v1, torch.Tensor):
                            return None
                        if not isinstance(v, torch.Tensor):
                            if v is not v1 and v != v1:
                                return None                     # a plain argument that depends on the batch
                            continue
                        if v.dtype != v1.dtype or v.dim() != v1.dim():
                            return None
                        if learn:
                            # carries the batch dimension: g times the batch-1 extent in front, the rest as in the batch-1 call
                            if v.dim() >= 1 and v.shape[0] == g * v1.shape[0] and v.shape[1:] == v1.shape[1:]:
                                batched[name] = v1.shape[0]
                            elif v.shape == v1.shape:
                                batched[name] = 0
                            else:
                                return None                     # (e.g. the model padded: another token count than the sample alone)
                        b_ = batched[name]
                        if b_:
                            if v.dim() < 1 or v.shape[0] != g * b_:
                                return None
                            pieces[name] = v.split(b_, dim=0)
                        if learn:                                # the merge against the sample's own forward, bit for bit
                            mine0 = pieces[name][0] if name in pieces else v
                            flags.append((mine0 == v1).all() if mine0.shape == v1.shape else torch.zeros((), dtype=torch.bool, device=v.device))
                        # a mask the model built for padding inside the merged forward: the samples would differ in it
                        if name in PAD_MASK_KEYS and name in pieces and g > 1:
                            flags.append((v == v[:b_].repeat(g, *([1] * (v.dim() - 1)))).all())
                    if not batched.get("#inp"):
                        return None
                    inp, args, kw = call
                    for t, j in enumerate(chunk):
                        pick = lambda name, v: (pieces[name][t] if name in pieces else v)
                        inp_j = pick("#inp", inp)
                        args_j = tuple(pick(f"#{i}", a_) for i, a_ in enumerate(args))
                        kw_j = {k: pick(k, v) for k, v in kw.items()}
                        if want_calls:
                            calls.append((j, TowerMemo._snapshot((inp_j,) + args_j, kw_j)))
                        if first is not None:
                            a_, k_ = (inp_j,) + args_j, dict(kw_j)
                            first[j] = (a_, k_, TowerMemo.context(), [(t_, t_._version) for t_ in TowerGraph._ext(a_, k_)])
                        if vit:
                            rel_pos_bias = args_j[0] if args_j else kw_j.get("rel_pos_bias")
                            dense = args_j[1] if len(args_j) > 1 else kw_j.get("dense", True)
                            cache = {"rel_pos_bias": rel_pos_bias}
                        else:
                            dense = kw_j.pop("dense", True)
                            cache = {k: kw_j[k] for k in keys}
                        if lora_model:
                            cache["dense"] = dense
                        arrived.append((j, inp_j, cache))
                    graph_stats["merged_forwards"] = graph_stats.get("merged_forwards", 0) + 1
                for t in towers:
                    if t.deferred:
                        t.run_deferred()
                pending = again
        if flags and not bool(torch.stack(flags).all()):
            graph_stats["merged_capture_mismatch"] = graph_stats.get("merged_capture_mismatch", 0) + 1
            return None
    except KeyError:
        return None                                             # (a kwarg the reference's key list names is missing: its path)
    finally:
        _CTX.capture_group = _CTX.capture_sample = None
        _CTX.group_defer = _CTX.keep_ready = False
        layers[0] = layers[0].module
        for blocks, i, orig in undo:
            tg = blocks[i].__dict__.get("_tower")
            if tg is not None:
                tg[0].deferred, tg[0].ready, tg[0].live, tg[0].trace, tg[0].btrace = [], {}, None, None, None
            blocks[i].__dict__["_memo"] = None
            blocks[i].__dict__["_tower"] = None
            blocks[i] = orig
    if len(arrived) != len(mine):
        return None
    arrived.sort(key=lambda a: a[0])
    if want_calls:
        calls.sort(key=lambda c: c[0])
        proxy_cache[("calls", module_to_process)] = [c[1] for c in calls]
    if first is not None:
        proxy_cache[("block0", module_to_process)] = first
    for a in arrived:
        if isinstance(a[1], torch.Tensor):
            a[1].requires_grad = False
    return [a[1] for a in arrived], [None] * len(arrived), [a[2] for a in arrived]


def _capture_once(model, batches, module_to_process, forward_to_cache, lora_model, *, vit, model_prefix, done_towers,
                  proxy_cache):
    layers = get_module_recursive(model, module_to_process)
    keys = None if vit else _keys_for(model_prefix)
    arrived = []                                   # (index of the calibration forward, block-0 input, cached kwargs)
    rank, world = calibration_shard()
    # how the model calls block 0, for seed_tower_memo (towers whose blocks all get the same kwargs: the ViT)
    calls = [] if (vit and proxy_cache is not None and tower_memo_enabled() and graph_replay_enabled()
                   and torch.cuda.is_available()) else None

    # how the model calls block 0 of THIS tower, by sample: the next phase's stacked pass through it starts from these
    # (TowerGraph.run_predicted).  References, not copies: the walk replaces `inps[j]`, it never writes into it.
    final = proxy_cache is not None and proxy_cache.get(("last_tower",)) == module_to_process   # (the pruner says so: no phase follows)
    first = {} if (proxy_cache is not None and not final and graph_replay_enabled() and tower_batch_enabled() and tower_graph_enabled()
                   and tower_predict_enabled() and torch.cuda.is_available()) else None

    class Catcher(nn.Module):
        def __init__(self, module):
            super().__init__()
            self.module = module

        def forward(self, inp, *args, **kwargs):
            if first is not None and _CTX.capture_sample is not None and not torch.is_grad_enabled():
                a_, k_ = (inp,) + tuple(args), dict(kwargs)
                first[_CTX.capture_sample] = (a_, k_, TowerMemo.context(), [(t, t._version) for t in TowerGraph._ext(a_, k_)])
            if calls is not None:
                calls.append((_CTX.capture_sample if _CTX.capture_sample is not None else len(calls),
                              TowerMemo._snapshot((inp,) + tuple(args), kwargs)))
            if vit:
                rel_pos_bias = args[0] if args else kwargs.get("rel_pos_bias")
                dense = args[1] if len(args) > 1 else kwargs.get("dense", True)
                cache = {"rel_pos_bias": rel_pos_bias}
            else:
                dense = kwargs.pop("dense", True)
                cache = {k: kwargs[k] for k in keys}
            inp.requires_grad = False
            if lora_model:
                cache["dense"] = dense
            arrived.append((_CTX.capture_sample if _CTX.capture_sample is not None else len(arrived), inp, cache))
            if main_stream is not None:               # produced on a side stream, consumed on the caller's: tell the allocator
                for t in [inp] + list(cache.values()):
                    if isinstance(t, torch.Tensor) and t.is_cuda:
                        t.record_stream(main_stream)
            raise _Stop

    # side streams for the forwards (kept by the pruner from phase to phase): only worth it when finished towers are run
    # through, and only on a GPU model
    sides, main_stream = [], None
    p0 = next(model.parameters(), None)
    if capture_streams() > 1 and done_towers and p0 is not None and p0.is_cuda and graph_replay_enabled():
        main_stream = torch.cuda.current_stream(p0.device)
        holder = proxy_cache if proxy_cache is not None else {}
        sides = holder.get(("streams", p0.device.index))
        if sides is None or len(sides) != capture_streams():
            sides = holder[("streams", p0.device.index)] = [torch.cuda.Stream(device=p0.device) for _ in range(capture_streams())]
    layers[0] = Catcher(layers[0])
    # blocks of towers that were pruned before this one (`done_towers`: their module paths) replay from HIP graphs
    undo = _wrap_towers(model, [t for t in (done_towers or []) if t != module_to_process], proxy_cache, record=not final)
    try:
        if world > 1 and len(batches) % world != 0:
            raise RuntimeError(f"calibration sharding needs the {len(batches)} calibration batches to divide evenly "
                               f"over {world} ranks (set VLMC_SHARD_CALIB=0 to run as replicas)")
        per = len(batches) // world
        mine = batches[rank * per:(rank + 1) * per] if world > 1 else batches
        towers = []
        for blocks_, i_, _orig in undo:
            tg_ = blocks_[i_].__dict__.get("_tower")
            if tg_ is not None and not any(tg_[0] is t for t in towers):
                towers.append(tg_[0])
        # the finished towers' linears run on the batch-invariant kernel whichever way a sample gets through them (alone,
        # from a graph, or stacked with others): the captured inputs do not depend on the route
        # (.. and so do the linears between the towers: the per-sample route and the merged one hand the next tower the same bits)
        with forward.invariant_linears(all_linears(model, proxy_cache) if proxy_cache is not None else [m for t in towers for m in t.linears],
                                       roots=[b for t in towers for b in t.mods]):
            pending, sweeps = list(range(len(mine))), 0
            while pending:
                sweeps += 1
                # towers whose block-0 arguments are remembered from their own capture phase run stacked NOW, on the caller's
                # stream, and every sample below gets through them in its first forward; a tower no forward has been traced
                # through yet (its wiring is unknown) is shown its scouts first
                scouts = set()
                if sweeps == 1:
                    for t in towers:
                        scouts |= t.scouts(pending)
                for t in towers:
                    t.run_predicted(pending)
                if sides:
                    for st in sides:
                        st.wait_stream(main_stream)
                    _CTX.stream_set = tuple([main_stream] + list(sides))
                later = [j for j in pending if j not in scouts] if scouts else []
                for n_, j in enumerate([j for j in pending if j in scouts] if scouts else pending):
                    _CTX.capture_sample = j
                    if sides:
                        _CTX.capture_slot = n_ % len(sides)
                    try:
                        with (torch.cuda.stream(sides[_CTX.capture_slot]) if sides else contextlib.nullcontext()):
                            forward_to_cache(model, mine[j], lora_model)
                    except ValueError:                 # _Stop / _Defer, or the reference's bare ValueError
                        pass
                _CTX.capture_sample = _CTX.capture_slot = None
                if sides:
                    for st in sides:
                        main_stream.wait_stream(st)
                pending = list(later)
                for t in towers:
                    if t.deferred:
                        if main_stream is not None:            # arguments made on the side streams, used on the caller's
                            for rec in t.deferred:
                                for e in TowerGraph._ext(rec["args"], rec["kwargs"]):
                                    e.record_stream(main_stream)
                        pending += t.run_deferred()
                pending = sorted(set(pending))
                if sweeps > 2 * len(towers) + 3 and pending:   # cannot happen with towers in sequence; never loop forever
                    raise RuntimeError("calibration capture: postponed forwards do not get through the finished towers "
                                       "(set VLMC_TOWER_BATCH=0)")
    finally:
        _CTX.capture_slot = _CTX.capture_sample = None
        _CTX.stream_set = ()
        if sides:
            for st in sides:
                main_stream.wait_stream(st)
        layers[0] = layers[0].module
        for blocks, i, orig in undo:
            tg = blocks[i].__dict__.get("_tower")
            if tg is not None:
                tg[0].deferred, tg[0].ready, tg[0].live, tg[0].trace, tg[0].btrace = [], {}, None, None, None
            blocks[i].__dict__["_memo"] = None
            blocks[i].__dict__["_tower"] = None
            blocks[i] = orig
    arrived.sort(key=lambda a: a[0])                   # postponed forwards arrive late; the reference's order is by sample
    inps, caches = [a[1] for a in arrived], [a[2] for a in arrived]
    if calls is not None:
        calls.sort(key=lambda c: c[0])
        calls = [c[1] for c in calls]
    if calls is not None:
        proxy_cache[("calls", module_to_process)] = calls
    if first is not None:
        proxy_cache[("block0", module_to_process)] = first
    return inps, [None] * len(inps), caches
