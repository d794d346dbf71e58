"""Finished towers during a capture phase: what they remember from phase to phase (`TowerMemo`), their stacked / padded passes and the
HIP-graph route (`TowerGraph`), the proxies that stand in for their blocks (`GraphedModule`, `_wrap_towers`), the never-pruned towers in
between (`FROZEN_TOWERS`), and the deferred bit-for-bit comparisons of a phase (`_LaterEqual`).  Split out of `calibration.py` in round 6."""
from __future__ import annotations

import contextlib
import os
import threading

import torch
import torch.nn as nn

from vlmc import forward, phases

from lavis.compression.pruners.replay_state import (  # noqa: F401
    MEMO_MAX_BYTES,
    REPLAY_TOKEN_BUDGET,
    _CTX,
    _bits_equal,
    block_tensors,
    capture_graph,
    find_layers,
    get_module_recursive,
    graph_replay_enabled,
    graph_stats,
    replay_group_size,
    storage_signature,
    tower_batch_enabled,
    tower_graph_enabled,
    tower_memo_enabled,
    tower_pad_enabled,
    tower_predict_enabled,
)
from lavis.compression.pruners.replay_padding import (  # noqa: F401
    _pad_caches,
    _pad_inputs,
    int32_on,
    plan_padded,
    row_map,
)


class _LaterEqual:
    """Bit-for-bit comparisons whose answer is collected at the end of a capture phase instead of one device round trip
    per calibration forward (`torch.equal` waits for the GPU: 2-3 of them per forward were 40 ms of a FlanT5-XL prune).
    `same(r, v)` answers what can be answered on the host (shapes, dtypes, devices), assumes the bits agree and notes the
    pair; `failed()` compares all noted pairs in one stacked `torch.equal` per shape -- and reports a tensor that was
    written to since it was noted as a failure.  Only `capture_block_inputs` installs one: it can run the phase again
    the plain way when the assumption turns out wrong."""

    def __init__(self):
        self.pairs = []

    def same(self, r, v):
        if r.shape != v.shape or r.dtype != v.dtype or r.device != v.device:
            return False
        if r is not v:
            self.pairs.append((r, r._version, v, v._version))
        return True

    def failed(self):
        groups = {}
        for r, rv, v, vv in self.pairs:
            if r._version != rv or v._version != vv:
                return True
            groups.setdefault((tuple(r.shape), r.dtype, r.device), []).append((r, v))
        self.pairs = []
        verdicts = []                                                  # device-side, read back ONCE (a `torch.equal` per shape was 20-30 waits per phase)
        for prs in groups.values():
            nbytes = prs[0][0].numel() * prs[0][0].element_size()
            per = max(1, min(256, (256 << 20) // max(1, nbytes)))      # two stacked copies of at most 256 MB each
            for t in range(0, len(prs), per):
                part = prs[t:t + per]
                verdicts.append((torch.stack([a for a, _ in part]) == torch.stack([b for _, b in part])).all())
        by_dev = {}
        for v in verdicts:
            by_dev.setdefault(v.device, []).append(v)
        return not all(bool(torch.stack(vs).all()) for vs in by_dev.values())


class _HiddenOnly(tuple):
    """What a block of a REMEMBERED tower returns when the tower's blocks return `(hidden_states, more..)` -- a BERT layer's
    `(layer_output, present_key_value)` (Qformer.py:470-474): element 0 is the hidden states; the other outputs were never computed
    (the block did not run), so reading them raises instead of handing the model a stand-in value."""

    def __new__(cls, hidden, n):
        return super().__new__(cls, (hidden,) + (None,) * (n - 1))

    def __getitem__(self, i):
        if isinstance(i, int) and (i == 0 or i == -len(self)):
            return tuple.__getitem__(self, 0)
        raise RuntimeError("tower memo: only the hidden states (element 0) of a remembered block exist; the model reads another of the "
                           "block's outputs -- set VLMC_TOWER_MEMO=0")

    def __iter__(self):
        raise RuntimeError("tower memo: only the hidden states (element 0) of a remembered block exist; the model unpacks the block's "
                           "outputs -- set VLMC_TOWER_MEMO=0")


def _memo_kind(result):
    """How a block hands on its hidden states: None = the tensor itself; (tuple | list, n) = element 0 of a sequence of n outputs
    (n = 1: transformers' `(hidden_states,)`; n > 1: a BERT layer's `(layer_output, present_key_value)`); False = neither."""
    if isinstance(result, torch.Tensor):
        return None
    if type(result) in (tuple, list) and len(result) >= 1 and isinstance(result[0], torch.Tensor):
        return (type(result), len(result))
    return False


class TowerMemo:
    """What a FINISHED tower produced for each calibration forward of one capture phase, for the next phase.

    A three-tower model (ViT -> T5 encoder -> T5 decoder) runs its own forward over the calibration set once per
    tower; the decoder's capture re-runs the ViT on the same images with the same, already pruned weights as the
    encoder's capture did -- 128 x 39 batch-1 block forwards, a tenth of a whole FlanT5-XL prune.  While the
    encoder's inputs are captured the memo records, per forward, the inputs of the tower's first block and the
    output of its last block; in the next phase the first block compares its inputs BIT FOR BIT with the record of
    the same forward and, if they agree, the blocks hand their input through and the last one returns the recorded
    output -- the tensor the blocks would compute again.  Guards: the tower's parameters and buffers must be where
    they were and their absolute values sum (float64, per tensor) to what they summed when the record was made; the blocks are in eval
    mode; every recorded forward called the blocks exactly once each, in order, with a single tensor as output;
    anything else leaves the blocks to run."""

    def __init__(self, fingerprint, n_blocks):
        self.fp, self.n = fingerprint, n_blocks
        self.entries, self.mode, self.ok = {}, "record", True        # entries: index of the calibration forward -> record
        self.cursor, self.hit, self.pending, self.expect, self.bytes = 0, None, None, 0, 0
        self.current = 0
        self.wrap = False            # how the blocks return their hidden states (_memo_kind): None = the tensor itself, (tuple | list, n) =
                                     # element 0 of a sequence of n outputs (BERT-style layers: `layer(...)[0]`); False = not seen yet

    @staticmethod
    def fingerprint(blocks):
        """(addresses, per-tensor float64 sums) of the tower's parameters and buffers, or None if they cannot be taken
        (tensors on several devices, exotic dtypes): then there is no memo."""
        ts = [t for b in blocks for t in (b if isinstance(b, list) else block_tensors(b)[0])]    # (blocks or their tensor lists)
        try:
            # sum |x| in float64 per tensor, one fused launch per dtype (a reduction per tensor was 1 600 launches per prune)
            by = {}
            for n_, t in enumerate(ts):
                by.setdefault((t.dtype, t.device), []).append(n_)
            sums = [None] * len(ts)
            for (dt, _dev), idx in by.items():
                if dt.is_floating_point:
                    vals = torch._foreach_norm([ts[n_].detach() for n_ in idx], 1, dtype=torch.float64)
                else:
                    vals = [torch.sum(ts[n_].detach(), dtype=torch.float64) for n_ in idx]
                for n_, v in zip(idx, vals):
                    sums[n_] = v
            sums = torch.stack(sums) if ts else torch.zeros(0)
        except Exception:
            return None
        return tuple(t.data_ptr() for t in ts), sums

    def matches(self, fp):
        # (inside a capture phase the sums are compared with everything else at the end of the phase, _LaterEqual: asking
        # now would make the host wait for whatever the GPU still has queued from the tower before)
        return fp is not None and self.ok and bool(self.entries) and self.fp[0] == fp[0] and self.fp[1].shape == fp[1].shape \
            and _bits_equal(self.fp[1], fp[1])

    def begin(self, mode):
        self.mode, self.cursor, self.hit, self.pending, self.expect = mode, 0, None, None, 0
        if mode == "record":
            self.entries, self.bytes = {}, 0

    def hand(self, t):
        """hidden states `t` in the form the tower's blocks return them"""
        if not self.wrap:
            return t
        kind, n = self.wrap
        return kind((t,)) if n == 1 else _HiddenOnly(t, n)

    @staticmethod
    def context():
        """What besides its inputs and weights decides a block's output: the autocast state."""
        on = torch.is_autocast_enabled()
        return on, (torch.get_autocast_gpu_dtype() if on else None)

    @staticmethod
    def keep(v):
        """A tensor the memo remembers: the tensor ITSELF with the version it has now (`fresh` refuses it once somebody wrote into it),
        not a copy -- the copies were 1 700 launches and 190 MB of traffic per prune for the Q-Former's arguments alone (every sample's
        image states), and the walk's inputs / a seeded tower's outputs have always been kept this way.  `VLMC_MEMO_COPY=1`: copies."""
        d = v.detach()
        if os.environ.get("VLMC_MEMO_COPY", "0") == "1":
            return d.clone()
        d._vlmc_version = d._version                      # (the alias shares the version counter of what it was detached from)
        return d

    @staticmethod
    def fresh(t):
        return getattr(t, "_vlmc_version", None) in (None, t._version)

    @staticmethod
    def _snapshot(args, kwargs):
        snap = lambda v: TowerMemo.keep(v) if isinstance(v, torch.Tensor) else v
        return [snap(a) for a in args], {k: snap(v) for k, v in kwargs.items()}, TowerMemo.context()

    @staticmethod
    def _same(rec, args, kwargs):
        rargs, rkw, ctx = rec
        if ctx != TowerMemo.context() or len(rargs) != len(args) or sorted(rkw) != sorted(kwargs):
            return False
        for r, v in list(zip(rargs, args)) + [(rkw[k], kwargs[k]) for k in rkw]:
            if isinstance(r, torch.Tensor) != isinstance(v, torch.Tensor):
                return False
            if isinstance(r, torch.Tensor):
                if not TowerMemo.fresh(r) or not _bits_equal(r, v):
                    return False
            elif r is not v and r != v:
                return False
        return True

    def _drop(self):
        self.ok, self.entries, self.hit, self.pending = False, {}, None, None

    def _group_hit(self, group, args, kwargs):
        """A merged calibration forward (samples `group` stacked along the batch): the remembered outputs of those samples, stacked,
        if the stacked input is their remembered inputs (bits compared like every remembered tensor: `_bits_equal`)."""
        ents = [self.entries.get(j) for j in group]
        if not ents or any(e is None for e in ents) or not args or not isinstance(args[0], torch.Tensor):
            return None
        (rargs0, rkw0, ctx0), x = ents[0][0], args[0]
        b = rargs0[0].shape[0]
        if ctx0 != self.context() or len(rargs0) != len(args) or sorted(rkw0) != sorted(kwargs) or x.shape[0] != b * len(group) or \
                x.shape[1:] != rargs0[0].shape[1:]:
            return None
        # the other arguments: plain values as remembered; tensors either carry the batch (a BERT layer's extended masks, the image
        # states of its cross-attention: cut per sample and held against each sample's record) or are one tensor for every sample
        others = []
        for pos, (r, v) in enumerate(list(zip(rargs0[1:], args[1:])) + [(rkw0[k], kwargs[k]) for k in sorted(rkw0)]):
            if isinstance(r, torch.Tensor) != isinstance(v, torch.Tensor):
                return None
            if not isinstance(v, torch.Tensor):
                if r is not v and r != v:
                    return None
                others.append(None)
            elif v.dim() >= 2 and v.shape[0] == b * len(group) and r.shape[0] == b and v.shape[1:] == r.shape[1:]:
                others.append(v.split(b, dim=0))
            elif v.shape == r.shape:
                others.append(v)
            else:
                return None
        parts = x.split(b, dim=0)
        for t, ((rec, _out), part) in enumerate(zip(ents, parts)):
            rargs, rkw, ctx = rec
            if ctx != ctx0 or len(rargs) != len(args) or sorted(rkw) != sorted(rkw0) or not self.fresh(rargs[0]) or not self.fresh(_out) or \
                    not _bits_equal(rargs[0], part):
                return None
            for o, r in zip(others, list(rargs[1:]) + [rkw[k] for k in sorted(rkw0)]):
                if o is None:
                    continue
                if not isinstance(r, torch.Tensor) or not self.fresh(r) or not _bits_equal(r, o[t] if isinstance(o, tuple) else o):
                    return None
        return torch.cat([e[1] for e in ents], dim=0)

    def enter(self, index, args, kwargs):
        """-> (handled, value).  Called by block `index` of the tower before it would run."""
        if not self.ok:
            return False, None
        group = _CTX.capture_group
        if index == 0:
            self.expect, self.hit, self.pending = 0, None, None
            self.hit_fresh = False
            # which calibration forward this is: the capture loop says so (it may run a forward twice); else they are counted
            if group is not None:
                self.current = None
            elif _CTX.capture_sample is not None:
                self.current = _CTX.capture_sample
            else:
                self.current, self.cursor = self.cursor, self.cursor + 1
        if index != self.expect:                                   # blocks skipped or repeated inside one forward
            if self.hit is not None:
                raise RuntimeError("tower memo: the model called the tower's blocks in another order than when the memo "
                                   "was recorded (set VLMC_TOWER_MEMO=0)")
            self._drop()
            return False, None
        self.expect = index + 1
        if group is not None and self.mode == "record":
            # a merged forward: the per-sample calls are cut out of the stacked one, the last block's output likewise (leave)
            if index == 0:
                self.pending = None
                x = args[0] if args and isinstance(args[0], torch.Tensor) else None
                g = len(group)
                if x is not None and x.dim() >= 2 and x.shape[0] % g == 0:
                    b = x.shape[0] // g
                    cut = lambda v: (v.split(b, dim=0) if isinstance(v, torch.Tensor) and v.dim() >= 2 and v.shape[0] == g * b else None)
                    ca, ck = [cut(v) for v in args], {k: cut(v) for k, v in kwargs.items()}
                    self.pending = ("group", list(group), b,
                                    [self._snapshot([c[t] if c is not None else v for c, v in zip(ca, args)],
                                                    {k: (ck[k][t] if ck[k] is not None else v) for k, v in kwargs.items()}) for t in range(g)])
            return False, None
        if self.mode == "record":
            if index == 0:
                self.pending = self._snapshot(args, kwargs) if args and isinstance(args[0], torch.Tensor) else None
                if self.pending is None:
                    self._drop()
            return False, None
        if index == 0 and group is not None:
            self.hit = self._group_hit(group, args, kwargs)
            self.hit_fresh = self.hit is not None
            graph_stats["memo_hits" if self.hit is not None else "memo_misses"] += 1
        elif index == 0:
            ent = self.entries.get(self.current)
            if ent is not None and self.fresh(ent[1]) and self._same(ent[0], args, kwargs):
                self.hit = ent[1]
                graph_stats["memo_hits"] += 1
            else:
                graph_stats["memo_misses"] += 1
        if self.hit is None:
            return False, None
        if index == self.n - 1:
            out, self.hit = (self.hit if self.hit_fresh else self.hit.clone()), None
            return True, self.hand(out)
        return True, self.hand(args[0])

    def leave(self, index, result):
        if _CTX.capture_group is not None:
            if self.ok and self.mode == "record" and index == self.n - 1 and isinstance(self.pending, tuple) and self.pending[0] == "group" \
                    and self.expect == self.n:
                _tag, group, b, snaps = self.pending
                self.pending = None
                kind = _memo_kind(result)
                if kind is False or (self.wrap is not False and self.wrap != kind):
                    return
                self.wrap = kind
                out = result if kind is None else result[0]
                if out.shape[0] != b * len(group):
                    return
                for j, snap, o in zip(group, snaps, out.split(b, dim=0)):
                    self.entries[j] = (snap, self.keep(o))
                    self.bytes += o.numel() * o.element_size() + sum(v.numel() * v.element_size() for v in list(snap[0]) + list(snap[1].values())
                                                                     if isinstance(v, torch.Tensor))
                    graph_stats["memo_recorded"] += 1
                if self.bytes > MEMO_MAX_BYTES:
                    self._drop()
            return
        if self.ok and self.mode == "record":
            # every block must hand its hidden states on the same way: the tensor, or a 1-tuple / 1-list of it
            kind = _memo_kind(result)
            if kind is False or (self.wrap is not False and self.wrap != kind):
                self._drop()
                return
            self.wrap = kind
            if kind is not None:
                result = result[0]
        if self.ok and self.mode == "record" and index == self.n - 1:
            if isinstance(result, torch.Tensor) and self.pending is not None and self.expect == self.n:
                self.entries[self.current] = (self.pending, self.keep(result))
                self.bytes += result.numel() * result.element_size() + sum(
                    v.numel() * v.element_size() for v in list(self.pending[0]) + list(self.pending[1].values())
                    if isinstance(v, torch.Tensor))
                self.pending = None
                graph_stats["memo_recorded"] += 1
                if self.bytes > MEMO_MAX_BYTES:
                    self._drop()
            else:
                self._drop()


def seed_tower_memo(proxy_cache, module_to_process, layers, final_outs, autocast):
    """After the walk over a tower whose blocks all receive the SAME kwargs in the model's own forward (the ViT): the
    second pass of the last block has just produced, per calibration sample, what the pruned tower makes of the inputs
    the catcher saw -- the next capture phase need not run the tower at all.  `proxy_cache[("calls", tower)]` holds the
    catcher's record of how the model called block 0 (capture_block_inputs).  (Under VLMC_BATCH_REPLAY the outputs come
    from the stacked forwards, like everything else downstream of a batched pass.)"""
    calls = proxy_cache.pop(("calls", module_to_process), None) if proxy_cache is not None else None
    if not (calls and tower_memo_enabled() and graph_replay_enabled() and len(layers) >= 2):
        return False
    n = min(len(calls), len(final_outs))
    if n == 0 or not all(isinstance(o, torch.Tensor) and o.is_cuda for o in final_outs[:n]):
        return False
    states = [block_tensors(mod) for mod in layers]
    if any(tr for _, tr in states):
        return False
    with autocast():
        ctx = TowerMemo.context()                    # the walk's forwards ran under this autocast state
    fp = TowerMemo.fingerprint([ts for ts, _ in states])
    if fp is None:
        return False
    memo = TowerMemo(fp, len(layers))
    memo.wrap = None                                 # (the blocks of such a tower return the tensor itself)
    memo.entries = {j: ((c[0], c[1], ctx), final_outs[j].detach()) for j, c in enumerate(calls[:n])}
    proxy_cache[("memo", module_to_process)] = memo
    graph_stats["memo_recorded"] += n
    return True


class _Defer(ValueError):
    """Raised by a finished tower's first block to abort a calibration forward whose tower pass is postponed: the tower
    will run for many samples at once (TowerGraph.run_deferred) and the forward be repeated (a ValueError, like the
    catcher's stop, so that `forward_to_cache` wrappers that swallow it keep working)."""


class TowerGraph:
    """All blocks of a FINISHED tower as ONE HIP graph per calibration forward.

    While the next tower's inputs are captured, the model's own forward walks an already pruned tower block by block; with a
    proxy and a graph per block that is 24 graph launches, 48 buffer copies and ~100 us of Python per block and sample --
    the host, not the GPU, bounded the capture of the T5 decoder's inputs (128 x 24 encoder block forwards).  The first
    forwards through the tower are traced: which argument of block i is which output of an earlier block, which is an
    outside tensor (by object identity), what is a plain value.  If every outside tensor is already an argument of block
    0, the whole chain can run when block 0 is entered: it is captured once per (stream slot, argument signature) and
    replayed from then on; the proxies of blocks 1.. hand out the outputs the graph has already produced, after checking
    that the model passed on exactly the tensors it was given (same objects, unmodified).  Any deviation -- another
    argument, an in-place edit, blocks called out of order -- falls back to the per-block path from that block on and
    switches the tower graph off.  Same kernels on the same values as the per-block graphs: bit-identical.

    Tower batching (the default when the tower qualifies): a batch-1 pass through a 24-block tower is ~700 kernels of a
    few microseconds, 128 times over.  With the wiring known, the forward of sample j is ABORTED at block 0 (`_Defer`), its
    arguments are kept, and once every sample of the sweep has arrived the tower runs ONCE per group of equal-shape
    samples, stacked along the batch dimension (`run_deferred`); the capture loop then repeats those forwards, and this
    time the proxies hand out the per-sample slices.  A tower qualifies when every linear in it runs on the
    batch-invariant kernel (16-bit weights, vlmc/forward.py) and every outside tensor has the hidden states' batch
    dimension: the stacked pass then gives every sample the bits its own pass would."""

    # Traces of a wiring before it is used.  One: every later forward is checked against it call by call (`_serve`: the very
    # tensors that were handed out, unmodified, the same plain values) and leaves the traced path the moment the model does
    # something else, so a second trace buys no safety -- and a traced forward is an eager batch-1 pass through the whole tower
    # (9 ms for the 24 T5 encoder blocks: two of them were 18 of the 26 ms a rank of 8 spends capturing the decoder's inputs).
    # `VLMC_TOWER_TRACES=2`: rounds 2-4's two identical traces.
    try:
        NEED = max(1, int(os.environ.get("VLMC_TOWER_TRACES", "1")))
    except ValueError:
        NEED = 1

    def __init__(self, modules):
        self.mods, self.n = list(modules), len(modules)
        self.plans, self.traces, self.wirings = {}, {}, {}
        self.by_struct = {}                   # argument signature without the tensor extents -> a key whose wiring is known
        self.shapes = {}                      # key -> (token count of the traced sample, per block the shapes of its outputs): what a padded pass is trimmed by
        self.deferred, self.ready = [], {}    # forwards postponed at block 0; their per-block outputs once the tower ran
        # block-0 calls of this tower as its OWN capture phase saw them, by sample (capture_block_inputs): what the model
        # will hand block 0 again in the next phase, if nothing upstream changed -- run_predicted()
        self.predicted, self.memo_serves, self.path = {}, False, None
        self.btrace = None                    # a trace that runs the tower STACKED for a whole group of remembered calls (_begin_batched_trace)
        self.linears = [m for mod in self.mods for m in find_layers(mod).values()]
        self._linears_ok = {}                     # autocast state -> every linear of the tower can run on the invariant kernel
        self.off = False
        self.live = None                      # replay in progress: {"plan", "given": id -> (tensor, version)}
        self.trace = None                     # recording in progress

    # -- helpers ---------------------------------------------------------------------------------------------------
    @staticmethod
    def _flat(out):
        """The leaves of a block's output in order: the output itself, or the entries of a (nested) tuple / list of outputs --
        a BERT layer returns `(hidden_states, (key, value))` (Qformer.py:470-474), a T5 block `(hidden, position_bias, ..)`."""
        if not isinstance(out, (tuple, list)):
            return [out]
        flat = []
        for o in out:
            if isinstance(o, (tuple, list)):
                flat += TowerGraph._flat(o)
            else:
                flat.append(o)
        return flat

    @staticmethod
    def _like(out, flat):
        """`flat` (as many leaves as `_flat(out)` has) in the nesting of `out`."""
        it = iter(flat)

        def build(o):
            if isinstance(o, (tuple, list)):
                return type(o)(build(e) for e in o) if type(o) in (tuple, list) else tuple(build(e) for e in o)
            return next(it)
        return build(out)

    @staticmethod
    def _shape_of(out):
        """the nesting of an output without its leaves (what a traced call is compared by)"""
        if isinstance(out, (tuple, list)):
            return ("L" if isinstance(out, list) else "T",) + tuple(TowerGraph._shape_of(o) for o in out)
        return None

    def _key0(self, args, kwargs, ctx=None):
        """What decides the tower's kernels and the wiring, from block 0's arguments (the stream slot is not part of it)."""
        sig = [TowerMemo.context() if ctx is None else ctx, self.mods[0].training]
        first = {}
        for pos, v in enumerate(list(args) + [kwargs[k] for k in sorted(kwargs)]):
            s_ = GraphedModule._sig(v)
            if s_ is NotImplemented:
                return None
            sig.append(s_)
            if isinstance(v, torch.Tensor):
                sig.append(first.setdefault(id(v), pos))                 # which arguments are one and the same tensor
        return tuple(sig) + tuple(sorted(kwargs))

    @staticmethod
    def _struct(key):
        return tuple(("T", len(e[1])) + e[2:] if isinstance(e, tuple) and len(e) == 4 and e[0] == "T" else e for e in key)

    def _wiring(self, key):
        """The wiring for a block-0 signature: its own, else the one traced for a signature that differs in tensor EXTENTS only
        (ragged calibration samples: 7 sequence lengths were 7 eager batch-1 trace forwards per tower and phase, ~9 ms each).
        A wiring says which argument is which earlier output / outside tensor / plain value, nothing about sizes; a model that
        does pass a size-dependent plain value to a later block is caught where every served forward is checked (`_serve`
        compares each plain value with the traced one and leaves the tower path on a difference).  `VLMC_TOWER_SHARE_WIRING=0`:
        one trace per exact signature, as in rounds 2-4."""
        if key is None:
            return False
        w = self.wirings.get(key)
        if w is None and os.environ.get("VLMC_TOWER_SHARE_WIRING", "1") != "0":
            k2 = self.by_struct.get(self._struct(key))
            if k2 is not None and self.wirings.get(k2):
                w = self.wirings[key] = self.wirings[k2]
                graph_stats["shared_wirings"] = graph_stats.get("shared_wirings", 0) + 1
        return w

    def _wire(self, v, known):
        if isinstance(v, torch.Tensor):
            src = known.get(id(v))
            if src is None or src[0] is not v:
                return None
            return src[1]                                     # ("ext", k) or ("out", block, position)
        if v is None or isinstance(v, (bool, int, float, str)):
            return ("val", v)
        return None

    # -- called by the proxies -----------------------------------------------------------------------------------------
    def enter(self, index, args, kwargs):
        """-> (handled, value)"""
        if self.off or torch.is_grad_enabled():
            self.live = self.trace = None
            return False, None
        if index == 0:
            self.live = self.trace = self.btrace = None
            key = self._key0(args, kwargs)
            wiring = self._wiring(key)
            if os.environ.get("VLMC_DEBUG_TOWERS"):
                print("TOWER", getattr(self, "path", "?"), "sample", _CTX.capture_sample, "key", key is not None, "wiring",
                      None if wiring is None else (False if wiring is False else "known"), "ready", len(self.ready), "predicted", len(self.predicted),
                      "batchable", self._batchable(args, kwargs) if key is not None else None, "memo_serves", self.memo_serves, flush=True)
            if wiring is False or key is None:
                return False, None
            if wiring is None:                                            # not known yet: trace this forward
                known = {}
                ext = []
                for v in list(args) + [kwargs[k] for k in sorted(kwargs)]:
                    if isinstance(v, torch.Tensor) and id(v) not in known:
                        known[id(v)] = (v, ("ext", len(ext)))
                        ext.append(v)
                self.trace = {"key": key, "known": known, "calls": [], "next": 0, "shapes": [],
                              "len": args[0].shape[1] if args and isinstance(args[0], torch.Tensor) and args[0].dim() >= 2 else None}
                self.btrace = self._begin_batched_trace(key, args, kwargs, ext)
                if self.btrace is not None:
                    return self._batched_step(0, args, kwargs)
                return False, None
            if _CTX.capture_sample is not None and tower_batch_enabled():
                # (the merged route forwards its scout alone AND in its group: the record serves both)
                r = self.ready.get(_CTX.capture_sample) if _CTX.keep_ready else self.ready.pop(_CTX.capture_sample, None)
                if r is not None:
                    if r.get("key", key) == key and self._same_inputs(r, args, kwargs):
                        given, k, seen = {}, 0, set()
                        for v in list(args) + [kwargs[kk] for kk in sorted(kwargs)]:
                            if isinstance(v, torch.Tensor) and id(v) not in seen:
                                seen.add(id(v))
                                given[("ext", k)] = (v, v._version)
                                k += 1
                        self.live = {"outs": r["outs"], "calls": wiring, "given": given, "clone": False}
                        return True, self._hand_out(0)
                elif self._batchable(args, kwargs):
                    # (measured, round 5: answering a tower nothing was predicted for -- the frozen Q-Former in the encoder's capture
                    # phase -- per sample from its HIP graph instead of abort + stacked pass + repeat: 334 against 289 ms per prune)
                    self.deferred.append({"j": _CTX.capture_sample, "key": key, "args": args, "kwargs": kwargs,
                                          "ctx": TowerMemo.context()})
                    raise _Defer
            plan = self.plans.get((_CTX.capture_slot, key))
            if plan is None:                                              # one graph (and its buffers) per stream slot
                plan = self.plans[(_CTX.capture_slot, key)] = self._build(wiring, args, kwargs)
            if plan is False:
                return False, None
            return self._replay(plan, args, kwargs)
        if self.live is not None:
            return self._serve(index, args, kwargs)
        if self.btrace is not None and self.trace is not None:
            return self._batched_step(index, args, kwargs)
        return False, None

    # -- merged calibration forwards of ragged batches (calibration._capture_merged) ------------------------------------------------
    def enter_group(self, index, args, kwargs):
        """-> (handled, value) for block `index` called with the STACKED samples `_CTX.capture_group`."""
        grp = _CTX.capture_group
        if self.off or torch.is_grad_enabled():
            self.live = None
            return False, None
        if index > 0:
            return self._serve(index, args, kwargs) if self.live is not None else (False, None)
        self.live = self.trace = self.btrace = None
        g = len(grp)
        if not (args and isinstance(args[0], torch.Tensor) and args[0].dim() >= 2 and args[0].shape[0] % g == 0):
            return False, None
        # the group's arguments as per-sample calls: tensors with g times the per-sample batch extent in front are cut, the rest is shared
        b = args[0].shape[0] // g

        def cut(v):
            if isinstance(v, torch.Tensor) and v.dim() >= 1 and v.shape[0] == g * b and v.dim() >= 2:
                return v.split(b, dim=0)
            return None
        cuts_a = [cut(v) for v in args]
        cuts_k = {k: cut(v) for k, v in kwargs.items()}
        per = [(tuple(c[t] if c is not None else v for c, v in zip(cuts_a, args)),
                {k: (cuts_k[k][t] if cuts_k[k] is not None else v) for k, v in kwargs.items()}) for t in range(g)]
        keys = [self._key0(a_, k_) for a_, k_ in per]
        wiring = self._wiring(keys[0])
        if not wiring or not self._batchable(per[0][0], per[0][1]) or any(self._wiring(k_) is not wiring for k_ in keys[1:]):
            return False, None
        if all(j in self.ready for j in grp):
            recs = [self.ready.pop(j) for j in grp]
            if all(self._same_inputs(r, a_, k_) for r, (a_, k_) in zip(recs, per)):
                outs, stacked = [], {}
                # neighbours of one length in ONE padded pass (its stack is sorted by length): what the blocks before the last hand the
                # group is a VIEW of the pass's outputs -- nobody computes on it (the model passes it to the next block, which is served);
                # the last block's outputs, which the model does compute on, are stacked into tensors of their own as before
                ps, t0 = recs[0].get("pass"), recs[0].get("t")
                near = ps is not None and all(r.get("pass") is ps and r.get("t") == t0 + k_ for k_, r in enumerate(recs)) and \
                    len({ps["lens"][r["t"]] for r in recs}) == 1 and os.environ.get("VLMC_GROUP_VIEWS", "1") != "0"
                n_views = self.n - 1 if near else 0
                whole = len(ps["lens"]) if near else 0
                for i in range(n_views):
                    flat = []
                    for o, c_ in zip(ps["flats"][i], ps["cut"][i]):
                        if c_ is None:
                            flat.append(o)
                            continue
                        hit = stacked.get(id(o))
                        if hit is None or hit[0] is not o:
                            v = o
                            if c_[0] is not None:                        # carries the batch: the group's rows of the stack
                                b0_ = o.shape[0] // whole
                                v = v.narrow(0, t0 * b0_, g * b0_)
                            for d in c_[1]:
                                v = v.narrow(d, 0, ps["lens"][t0])
                            hit = stacked[id(o)] = (o, v)
                        flat.append(hit[1])
                    outs.append(self._like(ps["outs"][i], flat))
                if n_views:
                    graph_stats["group_views"] = graph_stats.get("group_views", 0) + 1
                for i in range(n_views, self.n):
                    firsts = recs[0]["outs"][i]
                    flats = [self._flat(r["outs"][i]) for r in recs]
                    flat = []
                    for pos, o0 in enumerate(flats[0]):
                        if isinstance(o0, torch.Tensor) and g > 1:
                            # (the samples' pieces of one tensor -- the position bias every T5 block hands on -- are stacked once)
                            parts = [fl[pos] for fl in flats]
                            ids = tuple(id(p_) for p_ in parts)
                            hit = stacked.get(ids)
                            if hit is None:
                                hit = stacked[ids] = (parts, torch.cat(parts, dim=0))
                            flat.append(hit[1])
                        else:
                            flat.append(o0)
                    outs.append(self._like(firsts, flat))
                given, k, seen = {}, 0, set()
                for v in list(args) + [kwargs[kk] for kk in sorted(kwargs)]:
                    if isinstance(v, torch.Tensor) and id(v) not in seen:
                        seen.add(id(v))
                        given[("ext", k)] = (v, v._version)
                        k += 1
                self.live = {"outs": outs, "calls": wiring, "given": given, "clone": False}
                return True, self._hand_out(0)
            return False, None
        ctx = TowerMemo.context()
        for j, key_j, (a_, k_) in zip(grp, keys, per):
            self.deferred.append({"j": j, "key": key_j, "args": a_, "kwargs": k_, "ctx": ctx, "group": grp[0]})   # (groups of one length stay together in the padded stack)
        raise _Defer

    # -- tracing the wiring WHILE the tower runs stacked ---------------------------------------------------------------------
    # The forward that traces a tower's wiring used to run the tower eagerly for its one sample (24 T5 blocks: 7-9 ms of host
    # time, the GPU idle) and the stacked pass for everybody came afterwards.  With the block-0 arguments of the other samples
    # remembered (run_predicted), the tracing forward can BE the stacked pass: every block it calls is run once for the whole
    # group -- its arguments resolved through the wiring learnt so far to the stacked outputs of earlier blocks / the stacked
    # remembered arguments -- and the forward is handed its own slices.  When its last block returns, the wiring is known and
    # every sample of the group has its outputs.  Anything the wiring cannot express (an argument that is neither an earlier
    # output nor an outside tensor nor a plain value, a handed-out tensor written to, blocks out of order) ends the batched
    # trace: from that block on the forward runs eagerly, as a plain trace does, and nothing is kept for the others.
    def _begin_batched_trace(self, key, args, kwargs, ext):
        j = _CTX.capture_sample
        if self.NEED != 1 or j is None or not (tower_batch_enabled() and tower_predict_enabled()) or self.memo_serves or \
                os.environ.get("VLMC_TOWER_BATCHED_TRACE", "1") == "0" or j not in self.predicted or not self._batchable(args, kwargs):
            return None
        group = []
        for jj in sorted(self.predicted):
            pa, pk, ctx, versions = self.predicted[jj]
            if jj in self.ready or any(t._version != v for t, v in versions) or ctx != TowerMemo.context():
                continue
            if self._key0(pa, pk, ctx) == key:
                group.append({"j": jj, "key": key, "args": pa, "kwargs": pk, "ctx": ctx})
        pos = next((i for i, r in enumerate(group) if r["j"] == j), None)
        if pos is None or len(group) < 2:
            return None
        x0 = args[0]
        rows = max(1, x0.numel() // max(1, x0.shape[-1]))
        per = max(1, min(replay_group_size(), REPLAY_TOKEN_BUDGET // rows))
        c0 = (pos // per) * per
        chunk = group[c0:c0 + per]
        if len(chunk) < 2 or not self._same_inputs(chunk[pos - c0], args, kwargs):       # (bits: checked at the end of the phase)
            return None
        exts = [self._ext(r["args"], r["kwargs"]) for r in chunk]
        if any(len(e) != len(ext) for e in exts):
            return None
        try:
            stacked = [torch.cat([e[k] for e in exts], dim=0) for k in range(len(ext))]
        except Exception:
            return None
        return {"chunk": chunk, "t": pos - c0, "b0": x0.shape[0], "g": len(chunk), "ext": stacked, "outs": [],
                "ver": {id(v): v._version for v in ext}}

    def _batched_step(self, index, args, kwargs):
        tr, bt = self.trace, self.btrace

        def give_up(forget_wiring):
            if forget_wiring:
                self.wirings[tr["key"]] = False
            self.trace = self.btrace = None
            return False, None
        if index != tr["next"]:
            return give_up(False)
        vals = list(args) + list(kwargs.values())
        if any(isinstance(v, torch.Tensor) and id(v) in bt["ver"] and v._version != bt["ver"][id(v)] for v in vals):
            return give_up(True)                                          # the model wrote into a tensor it was handed
        wires = [self._wire(v, tr["known"]) for v in args]
        kwires = {k: self._wire(v, tr["known"]) for k, v in kwargs.items()}
        if any(w is None for w in wires) or any(w is None for w in kwires.values()):
            return give_up(True)

        def resolve(w):
            if w[0] == "ext":
                return bt["ext"][w[1]]
            if w[0] == "out":
                return self._flat(bt["outs"][w[1]])[w[2]]
            return w[1]
        out = self.mods[index](*[resolve(w) for w in wires], **{k: resolve(w) for k, w in kwires.items()})
        flat = self._flat(out)
        if not all(o is None or isinstance(o, torch.Tensor) for o in flat):
            return give_up(True)
        # The stacked call ran on THIS forward's capture stream; the other samples' slices of its outputs are consumed by
        # forwards on the other capture streams (and on the caller's).  The stream joins of the sweep order that work; what
        # they do not cover is the buffers' lifetime -- freed after the consumer's Python returns, a block goes back to the
        # PRODUCING stream's pool and can be handed out again while the consumer's kernels still read it.
        for o in flat:
            if isinstance(o, torch.Tensor) and o.is_cuda:
                for st in _CTX.stream_set:
                    o.record_stream(st)
        b0, g, t = bt["b0"], bt["g"], bt["t"]
        parts = [(o.split(b0, dim=0) if isinstance(o, torch.Tensor) and o.dim() >= 1 and o.shape[0] == g * b0 else None) for o in flat]
        mine = [(sp[t] if sp is not None else o) for o, sp in zip(flat, parts)]
        for pos, o in enumerate(mine):
            if isinstance(o, torch.Tensor):
                tr["known"][id(o)] = (o, ("out", index, pos))
                bt["ver"][id(o)] = o._version
        tr["calls"].append((tuple(wires), tuple(sorted(kwires.items())), self._shape_of(out), False,
                            tuple(o is None for o in flat)))
        tr["shapes"].append(tuple(tuple(o.shape) if isinstance(o, torch.Tensor) else None for o in mine))
        tr["next"] = index + 1
        bt["outs"].append(out)
        bt.setdefault("parts", []).append(parts)
        if index == self.n - 1:
            self._finish_trace(tr)
            if self.wirings.get(tr["key"]):                               # every other sample of the group has its outputs now
                for t2, rec in enumerate(bt["chunk"]):
                    if t2 == t:
                        continue
                    outs2 = []
                    for o_, pp in zip(bt["outs"], bt["parts"]):
                        fl = [(sp[t2] if sp is not None else o) for o, sp in zip(self._flat(o_), pp)]
                        outs2.append(self._like(o_, fl))
                    self.ready[rec["j"]] = {"outs": outs2, "args": rec["args"], "kwargs": rec["kwargs"], "key": tr["key"]}
                graph_stats["tower_predicted"] = graph_stats.get("tower_predicted", 0) + len(bt["chunk"]) - 1
                graph_stats["tower_batches"] = graph_stats.get("tower_batches", 0) + 1
                graph_stats["batched_traces"] = graph_stats.get("batched_traces", 0) + 1
            self.trace = self.btrace = None
        return True, self._like(out, mine)

    def leave(self, index, args, kwargs, out):
        tr = self.trace
        if tr is None:
            return
        if index != tr["next"]:
            self.trace = None
            return
        tr["next"] = index + 1
        wires = [self._wire(v, tr["known"]) for v in args]
        kwires = {k: self._wire(v, tr["known"]) for k, v in kwargs.items()}
        flat = self._flat(out)
        if any(w is None for w in wires) or any(w is None for w in kwires.values()) or \
                not all(o is None or isinstance(o, torch.Tensor) for o in flat):
            self.wirings[tr["key"]] = False                               # an argument from outside appears after block 0
            self.trace = None
            return
        for pos, o in enumerate(flat):
            if isinstance(o, torch.Tensor):
                tr["known"][id(o)] = (o, ("out", index, pos))
        tr["calls"].append((tuple(wires), tuple(sorted(kwires.items())), self._shape_of(out), False,
                            tuple(o is None for o in flat)))
        tr["shapes"].append(tuple(tuple(o.shape) if isinstance(o, torch.Tensor) else None for o in flat))
        if index == self.n - 1:
            self._finish_trace(tr)
            self.trace = None

    # -- building and running the graph ----------------------------------------------------------------------------------
    def tracing(self):
        return self.trace is not None

    def _finish_trace(self, tr):
        key, calls = tr["key"], tuple(tr["calls"])
        seen = self.traces.setdefault(key, [])
        if seen and seen[-1] != calls:
            self.wirings[key] = False
            return
        seen.append(calls)
        if len(seen) >= self.NEED:
            self.wirings[key] = calls
            self.shapes[key] = (tr.get("len"), tuple(tr.get("shapes", ())))
            self.by_struct.setdefault(self._struct(key), key)

    def _build(self, calls, args, kwargs):
        try:
            ext, seen = [], set()
            for v in list(args) + [kwargs[k] for k in sorted(kwargs)]:
                if isinstance(v, torch.Tensor) and id(v) not in seen:
                    seen.add(id(v))
                    ext.append(v.clone())
            outs = []

            def resolve(w):
                if w[0] == "ext":
                    return ext[w[1]]
                if w[0] == "out":
                    return self._flat(outs[w[1]])[w[2]]
                return w[1]

            def body():
                for i, (wires, kwires, _t, _l, _n) in enumerate(calls):
                    outs.append(self.mods[i](*[resolve(w) for w in wires], **{k: resolve(w) for k, w in kwires}))
                return outs
            graph, _ = capture_graph(body, ext[0].device)
            graph_stats["captured"] += 1
            graph_stats["tower_graphs"] = graph_stats.get("tower_graphs", 0) + 1
            return {"graph": graph, "ext": ext, "outs": outs, "calls": calls}
        except Exception as e:
            graph_stats["fallbacks"] += 1
            print(f"tower graph not built ({type(e).__name__}: {e})")
            return False

    def _hand_out(self, index):
        live = self.live
        out = live["outs"][index]
        flat = [(o.clone() if live["clone"] else o) if isinstance(o, torch.Tensor) else o for o in self._flat(out)]
        for pos, o in enumerate(flat):
            if isinstance(o, torch.Tensor):
                live["given"][("out", index, pos)] = (o, o._version)
        return self._like(out, flat)

    def _replay(self, plan, args, kwargs):
        given = {}
        k = 0
        seen = set()
        for v in list(args) + [kwargs[kk] for kk in sorted(kwargs)]:
            if isinstance(v, torch.Tensor) and id(v) not in seen:
                seen.add(id(v))
                plan["ext"][k].copy_(v)
                given[("ext", k)] = (v, v._version)
                k += 1
        plan["graph"].replay()
        graph_stats["replayed"] += 1
        # (the graph's output buffers are overwritten by the slot's next replay: hand out copies)
        self.live = {"outs": plan["outs"], "calls": plan["calls"], "given": given, "clone": True}
        return True, self._hand_out(0)

    def _serve(self, index, args, kwargs):
        live = self.live
        wires, kwire_items = live["calls"][index][0], live["calls"][index][1]          # kwire_items: sorted (name, wire) pairs
        ok = len(args) == len(wires) and len(kwargs) == len(kwire_items) and all(k in kwargs for k, _ in kwire_items)
        if ok:
            for v, w in list(zip(args, wires)) + [(kwargs[k], w_) for k, w_ in kwire_items]:
                if w[0] == "val":
                    ok = not isinstance(v, torch.Tensor) and (v is w[1] or v == w[1])
                else:
                    g = live["given"].get(w)
                    ok = g is not None and g[0] is v and v._version == g[1]
                if not ok:
                    break
        if not ok:                      # the model did something else with the tower this time: per-block path from here
            self.live = None
            self.off = True
            graph_stats["fallbacks"] += 1
            return False, None
        out = self._hand_out(index)
        if index == self.n - 1:
            self.live = None
        return True, out

    # -- tower batching ----------------------------------------------------------------------------------------------------
    def _positional_names(self, n_args):
        """Names of the parameters that arguments 1 .. n_args - 1 of a block call bind to (argument 0: the hidden states), or
        None when the block's forward does not say (`*args`)."""
        if n_args <= 1:
            return ()
        names = self.__dict__.get("_pos_names")
        if names is None:
            import inspect
            try:
                ps = list(inspect.signature(self.mods[0].forward).parameters.values())
            except (TypeError, ValueError):
                ps = []
            names = []
            for p_ in ps:
                if p_.kind not in (p_.POSITIONAL_ONLY, p_.POSITIONAL_OR_KEYWORD):
                    break
                names.append(p_.name)
            self.__dict__["_pos_names"] = names = tuple(names)
        return names[1:n_args] if len(names) >= n_args else None

    @staticmethod
    def _ext(args, kwargs):
        out, seen = [], set()
        for v in list(args) + [kwargs[k] for k in sorted(kwargs)]:
            if isinstance(v, torch.Tensor) and id(v) not in seen:
                seen.add(id(v))
                out.append(v)
        return out

    def _batchable(self, args, kwargs, ctx=None):
        """Every linear of the tower on the batch-invariant kernel, every outside tensor stackable along the batch."""
        from vlmc import forward as fw
        if not (args and isinstance(args[0], torch.Tensor) and args[0].dim() >= 2 and fw.enabled() and self.linears):
            return False
        ctx = TowerMemo.context() if ctx is None else ctx
        ok = self._linears_ok.get(ctx)                 # (168 linears of a Flan-T5-XL tower: asked once per autocast state)
        if ok is None:
            ok = True
            for m in self.linears:
                w = getattr(m, "weight", None)
                # 16-bit weights (the kernel's K % 8), or fp32 weights outside autocast (the reference's Q-Former: the fp32 kernel)
                if type(m) is not nn.Linear or w is None or (ctx[0] and ctx[1] != w.dtype) or \
                        not ((w.dtype in (torch.float16, torch.bfloat16) and w.shape[1] % 8 == 0) or (w.dtype is torch.float32 and fw.f32_enabled())):
                    ok = False
                    break
            self._linears_ok[ctx] = ok
        if not ok:
            return False
        b0 = args[0].shape[0]
        return all(e.dim() >= 1 and e.shape[0] == b0 for e in self._ext(args, kwargs))

    @staticmethod
    def _same_inputs(r, args, kwargs):
        a, b = TowerGraph._ext(r["args"], r["kwargs"]), TowerGraph._ext(args, kwargs)
        return len(a) == len(b) and all(_bits_equal(x, y) for x, y in zip(a, b))

    def scouts(self, samples):
        """Samples to send through the tower FIRST: per argument signature among the remembered block-0 calls whose wiring is
        not known yet, the NEED forwards that trace it (they would run eagerly in any order: a wiring is only trusted after
        NEED identical traces)."""
        if not self.predicted or self.memo_serves or self.off or not (tower_batch_enabled() and tower_predict_enabled()):
            return set()
        out, count = set(), {}
        shared = os.environ.get("VLMC_TOWER_SHARE_WIRING", "1") != "0" and tower_pad_enabled()
        for j in samples:
            rec = self.predicted.get(j)
            if rec is None:
                continue
            key = self._key0(rec[0], rec[1], rec[2])
            if key is None or self._wiring(key) is not None:
                continue
            # signatures that differ in extents only share a wiring, and their samples can share a PADDED stacked pass
            # (run_deferred): one scout for all of them
            ckey = self._struct(key) if shared else key
            have = count.get(ckey, len(self.traces.get(key, [])))
            if have < self.NEED:
                out.add(j)
                count[ckey] = have + 1
        return out

    def run_predicted(self, samples, group_of=None):
        """The stacked tower pass BEFORE the forwards that will ask for it.

        A finished tower normally learns a sample's block-0 arguments by running the model's forward up to block 0 and
        aborting it there (`_Defer`), and the forward is repeated once the stacked pass has run: two walks of the model's
        Python per sample, the GPU idle during the first (128 samples: 2 x 25 ms of host around 28 ms of GPU in the T5
        decoder's capture phase).  But block 0 of this tower was called with those arguments before -- when the tower's own
        inputs were captured, one phase ago -- and nothing upstream of it has been pruned since.  So: run the stacked pass
        on the REMEMBERED arguments, let every sample's single forward pick its slices up, and check the assumption the way
        every other remembered tensor is checked (`_same_inputs` -> `_bits_equal`: shapes at once, bits at the end of the
        phase; a mismatch reruns the phase without memos or predictions).  Same kernels on the same bits."""
        if not self.predicted or self.memo_serves or self.off or not tower_batch_enabled() or not tower_predict_enabled():
            return 0
        recs = []
        for j in samples:
            rec = self.predicted.get(j)
            if rec is None or j in self.ready:
                continue
            args, kwargs, ctx, versions = rec
            if any(t._version != v for t, v in versions):               # written to since it was captured
                continue
            key = self._key0(args, kwargs, ctx)
            wiring = self._wiring(key) if key is not None else None
            if not wiring or not self._batchable(args, kwargs, ctx):
                continue
            recs.append({"j": j, "key": key, "args": args, "kwargs": kwargs, "ctx": ctx, "group": (group_of or {}).get(j, -1)})
        if recs:
            held, self.deferred = self.deferred, recs
            try:
                self.run_deferred()
            finally:
                self.deferred = held
            graph_stats["tower_predicted"] = graph_stats.get("tower_predicted", 0) + len(recs)
        return len(recs)

    @torch.no_grad()
    def _run_padded(self, groups, fw):
        """Ragged samples of ONE argument structure (signatures that differ in their token count only) through the tower as ONE
        padded stacked pass instead of one pass per length: hidden rows behind a sample's own are zero, its mask columns the
        dtype's minimum (`plan_padded`: what the model's own extended mask holds for padding), and every block output is cut back
        to the sample's length before it is handed out -- along exactly the dimensions in which the padded pass's shapes differ
        from the shapes the TRACED sample produced (its length is never the padded length: the pad is one longer if need be, so a
        dimension that depends on the token count always differs and a constant one never does).  The live rows carry the bits of
        the sample's own pass (linears, norms: row-wise, batch-invariant; attention: vlmc_attn_fwd / vlmc_attn_matmul +
        vlmc_softmax_rows, padding-invariant) -- the argument of walk_blocks' padded groups, tested there and in
        tests/test_replay_invariance_gpu.py.  Returns the groups that are left for the per-signature passes."""
        by_struct = {}
        for key in groups:
            by_struct.setdefault(self._struct(key), []).append(key)
        left = dict(groups)
        for st, keys in by_struct.items():
            base = self.by_struct.get(st)
            calls = self.wirings.get(base) if base is not None else None
            t_s, shapes = self.shapes.get(base, (None, None))
            if len(keys) < 2 or not calls or t_s is None or len(shapes) != len(calls):
                continue
            # by token count, then by sample: the samples of one length -- a group of the merged capture -- are then NEIGHBOURS in the
            # padded stack, and their stacked outputs are views of it (enter_group)
            recs = sorted((r for k in keys for r in groups[k]),
                          key=lambda r: (r["args"][0].shape[1] if isinstance(r["args"][0], torch.Tensor) and r["args"][0].dim() >= 2 else 0,
                                         r.get("group", -1), r["j"]))
            ctx = recs[0]["ctx"]
            # arguments behind the hidden states that the model passes POSITIONALLY (a BERT layer of the Q-Former: Qformer.py:541-550)
            # are padded under the names the block's forward gives them -- the names plan_padded knows masks and states by
            n_pos = len(recs[0]["args"])
            pos_names = self._positional_names(n_pos)
            if pos_names is None or any(len(r["args"]) != n_pos or r["ctx"] != ctx or any(nm in r["kwargs"] for nm in pos_names) for r in recs):
                continue
            named = [dict(r["kwargs"], **dict(zip(pos_names, r["args"][1:]))) for r in recs]
            plan = plan_padded([r["args"][0] for r in recs], named, len(recs), replay_group_size())
            n_ext = len(self._ext(recs[0]["args"], recs[0]["kwargs"]))
            if plan is None or any(spec["sp"] is not None and len(set(spec["S"])) > 1 for _c, spec in plan):
                continue                                                 # (a second RAGGED length -- cross-attention states -- is not handled here)
            done, ok = {}, True
            for chunk, spec in plan:
                crecs = [recs[i] for i in chunk]
                tp = spec["tp"] + 1 if spec["tp"] == t_s else spec["tp"]
                x = _pad_inputs([r["args"][0] for r in crecs], tp)
                padded = _pad_caches([named[i] for i in chunk], dict(spec, tp=tp))
                dev_ = x.device
                kw = {k: padded[k] for k in crecs[0]["kwargs"]}
                ext = self._ext((x,) + tuple(padded[nm] for nm in pos_names), kw)
                if len(ext) != n_ext:
                    ok = False
                    break
                outs = []

                def resolve(w):
                    if w[0] == "ext":
                        return ext[w[1]]
                    if w[0] == "out":
                        return self._flat(outs[w[1]])[w[2]]
                    return w[1]
                with torch.autocast(device_type="cuda", dtype=ctx[1], enabled=ctx[0]) if ctx[0] else contextlib.nullcontext(), \
                        fw.invariant_linears(self.linears, roots=self.mods), \
                        fw.padded_rows({(len(crecs), tp): row_map(spec["T"], tp, dev_)},
                                       {tp: int32_on(spec["T"], dev_)},    # linears and attention skip the padding rows
                                       {(len(crecs), tp): tuple(spec["T"])}):
                    for i, (wires, kwires, _t, _l, _n) in enumerate(calls):
                        outs.append(self.mods[i](*[resolve(w) for w in wires], **{k: resolve(w) for k, w in kwires}))
                g, lens = len(crecs), spec["T"]
                # per block output: which dimensions carry the token count, and the per-sample pieces
                cut, views = [], {}
                for out, sh in zip(outs, shapes):
                    flat = self._flat(out)
                    if len(flat) != len(sh):
                        ok = False
                        break
                    row = []
                    for o, s0 in zip(flat, sh):
                        if o is None or s0 is None:
                            if (o is None) != (s0 is None):
                                ok = False
                            row.append(None)
                            continue
                        if o.dim() != len(s0) or o.dim() < 1 or o.shape[0] not in (g * s0[0], s0[0]):
                            ok = False
                            break
                        dims = [d for d in range(1, o.dim()) if o.shape[d] != s0[d]]
                        if any(s0[d] != t_s or o.shape[d] != tp for d in dims):
                            ok = False
                            break
                        row.append((o.split(s0[0], dim=0) if o.shape[0] == g * s0[0] and g > 1 else None, dims))
                    if not ok:
                        break
                    cut.append(row)
                if not ok:
                    break
                flats = [self._flat(out) for out in outs]
                this_pass = {"outs": outs, "flats": flats, "cut": cut, "lens": lens}
                for t, rec in enumerate(crecs):
                    n_t, mine = lens[t], []
                    for out, fl, row in zip(outs, flats, cut):
                        flat = []
                        for o, c_ in zip(fl, row):
                            if c_ is None:
                                flat.append(o)
                                continue
                            # a tensor that several blocks return (T5: every block hands the position bias on) is cut ONCE per sample,
                            # and the sample is handed the same view each time: enter_group stacks it once per group for the same reason
                            hit = views.get((id(o), t))
                            if hit is not None and hit[0] is o and hit[2] == c_[1]:
                                flat.append(hit[1])
                                continue
                            v = c_[0][t] if c_[0] is not None else o
                            for d in c_[1]:
                                v = v.narrow(d, 0, n_t)
                            views[(id(o), t)] = (o, v, c_[1])
                            flat.append(v)
                        mine.append(self._like(out, flat))
                    done[rec["j"]] = {"outs": mine, "args": rec["args"], "kwargs": rec["kwargs"], "key": rec["key"], "pass": this_pass, "t": t}
                graph_stats["tower_batches"] = graph_stats.get("tower_batches", 0) + 1
                graph_stats["tower_padded_passes"] = graph_stats.get("tower_padded_passes", 0) + 1
            if not ok:
                continue                                                 # (nothing of this structure was handed out: the per-signature passes run)
            for k in keys:
                self.wirings.setdefault(k, calls)
                left.pop(k)
            self.ready.update(done)
        return left

    @torch.no_grad()
    def run_deferred(self):
        """The tower for every postponed forward, stacked per group of equal signature; returns their indices."""
        from vlmc import forward as fw
        todo, self.deferred = self.deferred, []
        groups = {}
        for rec in todo:
            groups.setdefault(rec["key"], []).append(rec)
        if len(groups) > 1 and tower_pad_enabled():
            groups = self._run_padded(groups, fw)
        for key, recs in groups.items():
            calls = self.wirings[key]
            x0 = recs[0]["args"][0]
            rows = max(1, x0.numel() // max(1, x0.shape[-1]))
            per = max(1, min(replay_group_size(), REPLAY_TOKEN_BUDGET // rows))
            b0 = x0.shape[0]
            ctx = recs[0]["ctx"]
            for c0 in range(0, len(recs), per):
                chunk = recs[c0:c0 + per]
                exts = [self._ext(r["args"], r["kwargs"]) for r in chunk]
                ext = [torch.cat([e[k] for e in exts], dim=0) for k in range(len(exts[0]))]
                outs = []

                def resolve(w):
                    if w[0] == "ext":
                        return ext[w[1]]
                    if w[0] == "out":
                        return self._flat(outs[w[1]])[w[2]]
                    return w[1]
                with torch.autocast(device_type="cuda", dtype=ctx[1], enabled=ctx[0]) if ctx[0] else contextlib.nullcontext(), \
                        fw.invariant_linears(self.linears, roots=self.mods):
                    for i, (wires, kwires, _t, _l, _n) in enumerate(calls):
                        outs.append(self.mods[i](*[resolve(w) for w in wires], **{k: resolve(w) for k, w in kwires}))
                g = len(chunk)
                # per-sample views of every block output: one split per tensor (not one Python slice per tensor and sample)
                parts = [[(o.split(b0, dim=0) if o.dim() >= 1 and o.shape[0] == g * b0 else None) if isinstance(o, torch.Tensor) else None
                          for o in self._flat(out)] for out in outs]
                for t, rec in enumerate(chunk):
                    mine = []
                    for out, pp in zip(outs, parts):
                        flat = [(sp[t] if sp is not None else o) for o, sp in zip(self._flat(out), pp)]
                        mine.append(self._like(out, flat))
                    self.ready[rec["j"]] = {"outs": mine, "args": rec["args"], "kwargs": rec["kwargs"], "key": key}
                graph_stats["tower_batches"] = graph_stats.get("tower_batches", 0) + 1
        return [rec["j"] for rec in todo]


class GraphedModule(nn.Module):
    """Stands in for a block of an ALREADY PRUNED tower while the model's own forward runs the calibration batches up
    to the next tower (`capture_block_inputs`): the first call with a given argument signature runs eagerly, the second
    is captured in a HIP graph, later ones replay it -- same kernels, identical activations, about half the wall-clock
    of a batch-1 eager block.  Anything unusual (gradients enabled, arguments that are not tensors / None / plain
    scalars, outputs that are not tensors or flat tuples of them, a failing capture) falls through to the module."""

    def __init__(self, module):
        super().__init__()
        self.__dict__["_wrapped"] = module           # not registered as a sub-module: the model's structure is untouched
        self._seen, self._graphs, self._off = {}, {}, False
        self._storage = storage_signature(module)

    def __getattr__(self, name):
        return getattr(self.__dict__["_wrapped"], name)

    def __call__(self, *args, **kwargs):
        # (a proxy carries no hooks: nn.Module's call machinery in front of `forward` was a third of the cost of handing a
        # remembered output through the 39 + 24 proxies of a calibration forward)
        memo = self.__dict__.get("_memo")
        if memo is not None:
            # the middle blocks of a tower whose output is remembered (TowerMemo.enter's hit path, inlined: 37 of a ViT-g
            # forward's 39 proxy calls): hand the input through
            m, index = memo
            if m.hit is not None and m.ok and 0 < index == m.expect < m.n - 1:
                m.expect = index + 1
                return m.hand(args[0])
        return self.forward(*args, **kwargs)

    @staticmethod
    def _sig(v):
        if isinstance(v, torch.Tensor):
            return ("T", tuple(v.shape), v.dtype, v.device) if v.is_cuda and not v.requires_grad else NotImplemented
        if v is None or isinstance(v, (bool, int, float, str)):
            return ("V", v)
        return NotImplemented

    def forward(self, *args, **kwargs):
        memo = self.__dict__.get("_memo")                       # (TowerMemo, index of this block in its tower) or None
        if memo is None or torch.is_grad_enabled():
            return self._tower_forward(*args, **kwargs)
        handled, value = memo[0].enter(memo[1], args, kwargs)
        if handled:
            return value
        out = self._tower_forward(*args, **kwargs)
        memo[0].leave(memo[1], out)
        return out

    def _tower_forward(self, *args, **kwargs):
        if _CTX.capture_group is not None:
            # a merged calibration forward runs the tower as the model calls it -- unless the batches are ragged: then every group's
            # forward is postponed at the tower's first block, the tower runs ONCE, padded, for the samples of all groups, and the
            # repeated forwards are handed their groups' outputs (TowerGraph.enter_group)
            tg = self.__dict__.get("_tower")
            if tg is not None and _CTX.group_defer:
                handled, value = tg[0].enter_group(tg[1], args, kwargs)
                if handled:
                    return value
            return self.__dict__["_wrapped"](*args, **kwargs)
        tg = self.__dict__.get("_tower")                        # (TowerGraph, index) or None
        if tg is None:
            return self._forward(*args, **kwargs)
        handled, value = tg[0].enter(tg[1], args, kwargs)
        if handled:
            return value
        # (while the tower is traced for its own graph the blocks run eagerly: a graph per block would be captured for nothing)
        out = self.__dict__["_wrapped"](*args, **kwargs) if tg[0].tracing() else self._forward(*args, **kwargs)
        tg[0].leave(tg[1], args, kwargs, out)
        return out

    def _forward(self, *args, **kwargs):
        mod = self.__dict__["_wrapped"]
        if self._off or torch.is_grad_enabled():
            return mod(*args, **kwargs)
        names = sorted(kwargs)
        # besides its arguments, the autocast state and the train / eval flags decide which kernels a block runs
        key = (TowerMemo.context(), mod.training, _CTX.capture_slot) + \
            tuple(self._sig(a) for a in args) + tuple((k, self._sig(kwargs[k])) for k in names)
        if any(x is NotImplemented or (isinstance(x, tuple) and len(x) == 2 and x[1] is NotImplemented) for x in key[3:]) or \
                not any(isinstance(a, torch.Tensor) for a in list(args) + list(kwargs.values())):
            return mod(*args, **kwargs)
        ent = self._graphs.get(key)
        if ent is None:
            n = self._seen[key] = self._seen.get(key, 0) + 1
            if n < 2:
                return mod(*args, **kwargs)          # also the warm-up the capture needs
            try:
                sargs = [a.clone() if isinstance(a, torch.Tensor) else a for a in args]
                skw = {k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in kwargs.items()}
                dev_ = next(a for a in list(sargs) + list(skw.values()) if isinstance(a, torch.Tensor)).device
                graph, out = capture_graph(lambda: mod(*sargs, **skw), dev_)
                flat = out if isinstance(out, (tuple, list)) else (out,)
                if not all(o is None or isinstance(o, torch.Tensor) for o in flat):
                    raise TypeError("block output is not a tensor or a flat tuple of tensors")
                ent = self._graphs[key] = (graph, sargs, skw, out)
                graph_stats["captured"] += 1
            except Exception as e:
                self._off = True
                graph_stats["fallbacks"] += 1
                print(f"graph replay disabled for a block during capture ({type(e).__name__}: {e})")
                return mod(*args, **kwargs)
        graph, sargs, skw, out = ent
        for s_, a in zip(sargs, args):
            if isinstance(a, torch.Tensor):
                s_.copy_(a)
        for k, v in kwargs.items():
            if isinstance(v, torch.Tensor):
                skw[k].copy_(v)
        graph.replay()
        graph_stats["replayed"] += 1
        if isinstance(out, (tuple, list)):
            return type(out)(o.clone() if isinstance(o, torch.Tensor) else o for o in out)
        return out.clone()


def _wrap_towers(model, towers, proxy_cache=None, record=True):
    """Replace the blocks of the given module lists by GraphedModule proxies; returns the undo list.  `proxy_cache`
    (owned by the pruner) keeps the proxies -- and their graphs -- from one capture phase to the next.  `record=False` (the
    last capture phase of a prune): a tower whose outputs are not remembered yet is not recorded either -- nobody would ask."""
    undo = []
    if not (towers and graph_replay_enabled() and torch.cuda.is_available()):
        return undo
    for path in towers:
        try:
            blocks = get_module_recursive(model, path)
        except AttributeError:
            continue
        proxies, originals, states = [], [], []
        for i in range(len(blocks)):
            if isinstance(blocks[i], GraphedModule):
                continue
            ts, training = block_tensors(blocks[i])                # one walk per block and phase (was six)
            if (ts[0] if ts else torch.empty(0)).is_cuda:
                mod = blocks[i]
                sig = storage_signature(mod, ts)
                states.append((ts, training, sig))
                proxy = proxy_cache.get(id(mod)) if proxy_cache is not None else None
                if proxy is None or proxy.__dict__["_wrapped"] is not mod or proxy._storage != sig:
                    proxy = GraphedModule(mod)
                    if proxy_cache is not None:
                        proxy_cache[id(mod)] = proxy
                undo.append((blocks, i, mod))
                blocks[i] = proxy
                proxy.__dict__["_memo"] = None
                proxies.append(proxy)
                originals.append(mod)
        # the whole tower as one graph per calibration forward (TowerGraph), kept with the proxies from phase to phase
        any_training = any(tr for _, tr, _ in states)
        sigs = tuple(sig for _, _, sig in states)
        if tower_graph_enabled() and len(proxies) == len(blocks) >= 2 and not any_training:
            tg = proxy_cache.get(("tower_graph", path)) if proxy_cache is not None else None
            if tg is None or len(tg.mods) != len(originals) or any(a is not b for a, b in zip(tg.mods, originals)) or \
                    tg.storage != sigs:
                tg = TowerGraph(originals)
                tg.storage = sigs
                if proxy_cache is not None:
                    proxy_cache[("tower_graph", path)] = tg
            tg.path, tg.memo_serves = path, False
            tg.predicted = (proxy_cache.get(("block0", path)) or {}) if proxy_cache is not None else {}
            for i, proxy in enumerate(proxies):
                proxy.__dict__["_tower"] = (tg, i)
        else:
            for proxy in proxies:
                proxy.__dict__["_tower"] = None
        # the tower's outputs of this phase are remembered for the next one (TowerMemo)
        # (a block in training mode may draw dropout / drop-path masks: its output is not a function of its inputs)
        if proxy_cache is not None and tower_memo_enabled() and len(proxies) == len(blocks) >= 2 and not any_training:
            memo = proxy_cache.get(("memo", path))
            fp = TowerMemo.fingerprint([ts for ts, _, _ in states]) if (memo is not None or record) else None
            if memo is not None and memo.matches(fp):
                memo.begin("replay")
                tg_ = proxies[0].__dict__.get("_tower")
                if tg_ is not None:
                    tg_[0].memo_serves = True                    # the memo hands out the tower's outputs: nothing to predict
            elif fp is not None and record:
                memo = proxy_cache[("memo", path)] = TowerMemo(fp, len(proxies))
                memo.begin("record")
            else:
                memo = None
            for i, proxy in enumerate(proxies):
                proxy.__dict__["_memo"] = (memo, i) if memo is not None else None
    return undo


# Block lists that lie on the model's forward between two pruned towers and are never pruned themselves: the Q-Former of
# BLIP-2 / InstructBLIP (blip2_t5_instruct.py:146-175: `self.Qformer.bert(...)`, Qformer.py: `bert.encoder.layer`).  Once a tower
# upstream of them is finished they are run through like a finished tower -- memo from phase to phase, stacked over the samples of a
# group -- instead of 12 eager batch-1 layers per calibration forward on the host-bound side of every capture phase.
FROZEN_TOWERS = ("Qformer.bert.encoder.layer",)


def with_frozen_towers(model, done_towers, module_to_process):
    if not done_towers:
        return done_towers
    out = list(done_towers)
    for path in FROZEN_TOWERS:
        if path in out or path == module_to_process:
            continue
        try:
            blocks = get_module_recursive(model, path)
        except AttributeError:
            continue
        if isinstance(blocks, nn.ModuleList) and len(blocks) >= 2:
            out.insert(1 if len(out) >= 1 else 0, path)           # (order is informative only: behind the vision tower)
    return out
