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

T5_KEYS = ["attention_mask", "position_bias", "encoder_attention_mask", "encoder_decoder_position_bias",
           "layer_head_mask", "cross_attn_layer_head_mask", "encoder_hidden_states"]      # wanda_pruner.py:225-228
OPT_KEYS = ["attention_mask", "layer_head_mask"]                                         # :230-232
LLM_KEYS = ["attention_mask", "position_ids"]                                            # :234-236


def get_module_recursive(base, module_to_process):
    for part in [p for p in module_to_process.split(".") if p != ""]:
        base = getattr(base, part)
    return base


def prunable_layer_types():
    from lavis.peft.src.peft.tuners.lora import Linear, LoraLayer, Linear8bitLt
    return [nn.Linear, Linear, LoraLayer, Linear8bitLt]


def find_layers(module, layers=None, name=""):
    """{qualified name: module} for every sub-module whose type is EXACTLY one of `layers`."""
    layers = prunable_layer_types() if layers is None else layers
    if type(module) in layers:
        return {name: module}
    res = {}
    for child_name, child in module.named_children():
        res.update(find_layers(child, layers=layers, name=name + "." + child_name if name != "" else child_name))
    return res


class _Stop(ValueError):
    """Raised by the catcher to abort the model forward (the reference raises ValueError)."""


from vlmc.shard import calibration_shard  # noqa: E402,F401  (one answer for capture, replay and the exchanges)


def release_tower_memory():
    """End of a tower's `_prune` (the reference calls `torch.cuda.empty_cache(); gc.collect()` there, wanda_pruner.py:349-351).
    The calibration activations are ordinary tensors whose memory returns to the caching allocator when they go out of
    scope; handing it back to the driver and sweeping the Python heap cost ~0.1 s per prune of FlanT5-XL for nothing the next
    tower needs, so both are opt-in: `VLMC_RELEASE_MEMORY=1`."""
    if os.environ.get("VLMC_RELEASE_MEMORY", "0") == "1":
        import gc
        if torch.cuda.is_available():
            torch.cuda.empty_cache()
        gc.collect()


def _keys_for(model_prefix):
    if "t5_model" in model_prefix:
        return T5_KEYS
    if "opt_model" in model_prefix:
        return OPT_KEYS
    if "llm_model" in model_prefix:
        return LLM_KEYS
    raise ValueError(f"no calibration cache keys known for model prefix {model_prefix!r}")


def graph_replay_enabled():
    """Graph-captured replay (default on for GPU tensors, `VLMC_GRAPH_REPLAY=0` turns it off)."""
    return os.environ.get("VLMC_GRAPH_REPLAY", "1") != "0"


GRAPH_MIN_SAMPLES = 4         # a capture costs about three eager forwards
graph_stats = {"captured": 0, "replayed": 0, "fallbacks": 0, "memo_recorded": 0, "memo_hits": 0, "memo_misses": 0}
MEMO_MAX_BYTES = 4 << 30


class _PruneContext(threading.local):
    """What a running prune keeps BETWEEN the functions of this module -- per calling thread, so that two prunes driven from two
    threads (each on its own device / stream) do not see each other's state (re-entrant per (device, stream);
    rounds 1-4 kept these in five module globals):
      later           the capture phase's _LaterEqual, or None: compare remembered tower inputs at once
      capture_slot    the capture side stream a calibration forward runs on (picks the graph instance and its static buffers)
      capture_sample  index (within this rank's share) of the calibration forward capture_block_inputs is running
      stacked         (samples, batch per sample, sample indices) of the grouped block forward under way (stacked_samples())
      stacked_lengths {padded token count: int32 device tensor [samples]} of a PADDED group of ragged samples, or None
      capture_side    device -> the side stream graphs are captured on
      stream_set      the caller's stream and the capture side streams of the running capture phase"""

    def __init__(self):
        self.later = None
        self.capture_slot = None
        self.capture_sample = None
        self.stacked = None
        self.stacked_lengths = None
        self.capture_group = None          # merged capture: the samples (indices) of the calibration forward that is running
        self.group_defer = False           # .. and finished towers are left for ONE (padded) stacked pass over all groups (ragged batches)
        self.keep_ready = False            # .. and a tower's outputs for the sample forwarded alone stay for the group it belongs to
        self.capture_side = {}
        self.stream_set = ()


_CTX = _PruneContext()


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
        for prs in groups.values():
            nbytes = prs[0][0].numel() * prs[0][0].element_size()
            per = max(1, min(256, (256 << 20) // max(1, nbytes)))      # two stacked copies of at most 256 MB each
            for t in range(0, len(prs), per):
                part = prs[t:t + per]
                if not torch.equal(torch.stack([a for a, _ in part]), torch.stack([b for _, b in part])):
                    return True
        return False




def _bits_equal(r, v):
    if _CTX.later is not None and r.is_cuda:
        return _CTX.later.same(r, v)
    return r.shape == v.shape and r.dtype == v.dtype and r.device == v.device and bool(torch.equal(r, v))


def tower_memo_enabled():
    """Outputs of a finished tower are remembered from one capture phase to the next (`VLMC_TOWER_MEMO=0`: off)."""
    return os.environ.get("VLMC_TOWER_MEMO", "1") != "0"


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




def capture_graph(fn, device):
    """(graph, fn()) with fn's kernels captured in a HIP graph.  `torch.cuda.graph` synchronises the device, collects garbage
    and empties the allocator cache on entry (~1 ms) -- per block, tower, slot and prune that was 0.1 s of a FlanT5-XL prune;
    the capture itself needs none of it."""
    graph = torch.cuda.CUDAGraph()
    cur = torch.cuda.current_stream(device)
    side = _CTX.capture_side.get(device)
    if side is None:
        side = _CTX.capture_side[device] = torch.cuda.Stream(device=device)
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        graph.capture_begin(capture_error_mode="thread_local")
        try:
            out = fn()
        except BaseException:
            try:
                graph.capture_end()
            except Exception:
                pass
            raise
        graph.capture_end()
    cur.wait_stream(side)
    return graph, out


# Capture phases run the calibration forwards round-robin on a few side streams (capture_streams()); the slot a forward
# runs in picks the graph instance -- and with it the static buffers -- its proxies replay (None: the caller's stream).


def capture_streams():
    """`VLMC_CAPTURE_STREAMS=S` (default 4; 1 = the caller's stream only): while the model's own forward runs the
    calibration batches up to the next tower, batch j goes to side stream j mod S.  A batch-1 forward through an already
    pruned tower is a chain of short kernels that leaves the GPU mostly idle (the ~30 kernels of a Flan-T5-XL block take
    160 us where streaming its 96 MB of weights takes 19); independent samples on S streams fill it.  Same kernels, same
    arguments, same results."""
    try:
        return max(1, int(os.environ.get("VLMC_CAPTURE_STREAMS", "4")))
    except ValueError:
        return 1




_gc_depth = 0


def quiet_gc(fn):
    """Decorator for a pruner's `prune()`: the objects alive when it starts (the model's ~10^5 modules, parameters and hooks, the
    calibration batches) are moved to the collector's permanent generation for the duration (`gc.freeze()`), and back afterwards.
    A prune allocates enough containers to trigger a full (generation-2) collection every second or third call, and a full
    collection walks every tracked object of the process: 55-65 ms on the InstructBLIP-FlanT5-XL stand-in -- the +60 ms outliers
    of every timing series of rounds 2-5 (tools/micro/gc_probe.py: 476 / 545 / 481 / 480 / 535 ms with, 473-476 ms without).
    Young collections keep running, nothing is leaked; skipped if somebody else has frozen objects already (their `unfreeze`
    is theirs to call) and with `VLMC_GC_FREEZE=0`."""
    import functools
    import gc

    @functools.wraps(fn)
    def wrapper(*args, **kwargs):
        global _gc_depth
        mine = _gc_depth == 0 and os.environ.get("VLMC_GC_FREEZE", "1") != "0" and gc.isenabled() and gc.get_freeze_count() == 0
        if mine:
            gc.freeze()
        _gc_depth += 1
        try:
            return fn(*args, **kwargs)
        finally:
            _gc_depth -= 1
            if mine:
                gc.unfreeze()
    return wrapper


class _Defer(ValueError):
    """Raised by a finished tower's first block to abort a calibration forward whose tower pass is postponed: the tower
    will run for many samples at once (TowerGraph.run_deferred) and the forward be repeated (a ValueError, like the
    catcher's stop, so that `forward_to_cache` wrappers that swallow it keep working)."""


def tower_batch_enabled():
    """Finished towers run for all calibration samples of one shape in ONE pass (`VLMC_TOWER_BATCH=0`: per sample)."""
    return os.environ.get("VLMC_TOWER_BATCH", "1") != "0"


def tower_predict_enabled():
    """A finished tower's stacked pass starts from the block-0 arguments remembered from its own capture phase
    (TowerGraph.run_predicted; `VLMC_TOWER_PREDICT=0`: every forward is aborted at block 0 and repeated, as in round 3)."""
    return os.environ.get("VLMC_TOWER_PREDICT", "1") != "0"


def tower_pad_enabled():
    """Ragged samples of one argument structure run a finished tower as ONE padded stacked pass (TowerGraph._run_padded;
    `VLMC_TOWER_PAD=0`: one pass per token count, as before)."""
    return os.environ.get("VLMC_TOWER_PAD", "1") != "0" and pad_ragged_enabled()


def tower_graph_enabled():
    """One HIP graph per finished TOWER and calibration forward (`VLMC_TOWER_GRAPH=0`: one per block)."""
    return os.environ.get("VLMC_TOWER_GRAPH", "1") != "0"


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
                for i in range(self.n):
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
            self.deferred.append({"j": j, "key": key_j, "args": a_, "kwargs": k_, "ctx": ctx})
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

    def run_predicted(self, samples):
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
            recs.append({"j": j, "key": key, "args": args, "kwargs": kwargs, "ctx": ctx})
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
            recs = sorted((r for k in keys for r in groups[k]), key=lambda r: r["j"])
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
                    done[rec["j"]] = {"outs": mine, "args": rec["args"], "kwargs": rec["kwargs"], "key": rec["key"]}
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


def later_check_enabled():
    """`VLMC_LATER_EQUAL=0`: every comparison of a remembered input with the one at hand waits for its answer."""
    return os.environ.get("VLMC_LATER_EQUAL", "1") != "0"


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
        pruned_on_the_way = [t for t in towers if not t.memo_serves and t.path not in FROZEN_TOWERS]
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
            if defer:
                for t in pruned_on_the_way:
                    t.run_predicted(everyone)                   # (towers whose wiring a previous prune traced: now; else after the scout has traced it)
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
                    t.run_predicted(everyone)                   # (no-op for the samples that have their outputs)
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


REPLAY_GROUP_DEFAULT = 128
REPLAY_TOKEN_BUDGET = 1 << 16


def replay_group_size():
    """`VLMC_BATCH_REPLAY=G`: replay up to G calibration samples of equal shape through a block in ONE forward call
    (default 128, i.e. the whole calibration set of the reference's scripts; `VLMC_BATCH_REPLAY=1` is the reference's
    per-sample loop, replayed from HIP graphs)."""
    try:
        return max(1, int(os.environ.get("VLMC_BATCH_REPLAY", str(REPLAY_GROUP_DEFAULT))))
    except ValueError:
        return REPLAY_GROUP_DEFAULT


def plan_groups(cur_in, caches, n_samples, group_max):
    """Which calibration samples go through the block together: samples whose input and cached kwargs have identical
    shapes and dtypes (they need not be neighbours: real calibration text is ragged), at most `group_max` per call and at
    most `VLMC_REPLAY_TOKENS` (default 65536) rows of activations per call.  Returns lists of sample indices, ordered by
    their first member; the statistics keep the reference's per-sample order whatever the grouping."""
    try:
        budget = max(1, int(os.environ.get("VLMC_REPLAY_TOKENS", str(REPLAY_TOKEN_BUDGET))))
    except ValueError:
        budget = REPLAY_TOKEN_BUDGET
    buckets = {}
    for j in range(n_samples):
        buckets.setdefault(_stack_key(cur_in[j], caches[j]), []).append(j)
    chunks = []
    for idx in buckets.values():
        x = cur_in[idx[0]]
        rows = max(1, x.numel() // max(1, x.shape[-1]))
        g = max(1, min(group_max, budget // rows))
        chunks += [idx[t:t + g] for t in range(0, len(idx), g)]
    chunks.sort(key=lambda c: c[0])
    return chunks


# While a stacked forward runs: (number of stacked calibration samples, their common batch size, their indices in the
# calibration set).  The statistics hooks read it to keep the reference's per-sample bookkeeping (one `add_batch` per
# sample, :304-314) in the reference's sample order.


def stacked_samples():
    return _CTX.stacked


def stacked_lengths(padded_tokens):
    """During the forward of a PADDED group of ragged samples: the int32 device tensor [samples] of the token rows that are each
    sample's own, for a hook input whose token dimension has `padded_tokens` rows; None otherwise (nothing is padded)."""
    ln = _CTX.stacked_lengths
    return None if ln is None else ln.get(int(padded_tokens))


# ---- ragged calibration text: ONE padded forward per block instead of one per distinct length ---------------------------------
# Real calibration prompts and answers are ragged (blip2_t5_instruct.py:49-53: up to 128 / 256 tokens); grouping the samples by
# shape makes 7 / 18 groups per encoder / decoder block pass on the bench's ragged set, each a walk of the block's Python.  A
# tower whose blocks are called with additive attention masks (the reference's T5 stack always is: extended masks,
# modeling_t5.py:1060-1115) can take all lengths at once: inputs padded with zero rows, masks padded with the dtype's minimum,
# the cross-attention's states padded with zero rows.  A sample's rows keep their bits because every op of the block is
# row-wise, or a product on the batch-invariant kernels (extra key columns do not touch the real ones; masked probabilities are
# exactly 0 in `attn @ v`), or the softmax -- which runs on `vlmc_softmax_rows` during a replay for exactly this reason.  The
# statistics hooks are told each sample's own token count (stacked_lengths).  `VLMC_PAD_RAGGED=0`: groups of equal shapes only.
PAD_MASK_KEYS = {"attention_mask": "self", "encoder_attention_mask": "cross"}
PAD_STATE_KEYS = {"encoder_hidden_states": "cross"}


def pad_ragged_enabled():
    from vlmc import forward as fw
    return os.environ.get("VLMC_PAD_RAGGED", "1") != "0" and fw.enabled() and fw.attn_matmul_enabled() and fw.softmax_enabled()


def plan_padded(cur_in, caches, n_samples, group_max):
    """[(chunk, spec)] covering samples 0 .. n_samples - 1 with PADDED groups, or None when the samples are not ragged or cannot be
    padded (no mask kwarg to hide the padding behind, tensors this function does not know how to pad, mixed dtypes / widths)."""
    if n_samples < 2 or not pad_ragged_enabled():
        return None
    x0, c0 = cur_in[0], caches[0]
    if x0.dim() != 3 or x0.shape[0] != 1 or not x0.is_cuda:
        return None
    T, S = [], []
    for j in range(n_samples):
        x, c = cur_in[j], caches[j]
        if x.dim() != 3 or x.shape[0] != 1 or x.shape[2] != x0.shape[2] or x.dtype != x0.dtype or sorted(c) != sorted(c0):
            return None
        t, s_len = x.shape[1], None
        for k, v in c.items():
            v0 = c0[k]
            if not isinstance(v, torch.Tensor):
                if isinstance(v0, torch.Tensor) or (v is not v0 and v != v0):
                    return None
                continue
            if not isinstance(v0, torch.Tensor) or v.dtype != v0.dtype or v.dim() != v0.dim():
                return None
            if k in PAD_STATE_KEYS:
                if v.dim() != 3 or v.shape[0] != 1 or v.shape[2] != v0.shape[2]:
                    return None
                s_len = v.shape[1]
            elif k not in PAD_MASK_KEYS:
                return None                                           # a tensor kwarg nobody told us how to pad
        for k, kind in PAD_MASK_KEYS.items():
            v = c.get(k)
            if v is None:
                continue
            keys = t if kind == "self" else s_len
            if not (isinstance(v, torch.Tensor) and v.is_floating_point() and v.dim() == 4 and v.shape[0] == 1 and v.shape[1] == 1
                    and keys is not None and v.shape[3] == keys and v.shape[2] in (1, t)):
                return None
        T.append(t)
        S.append(s_len)
    ragged_t, ragged_s = len(set(T)) > 1, len(set(S)) > 1
    if not (ragged_t or ragged_s):
        return None
    if ragged_t and not isinstance(c0.get("attention_mask"), torch.Tensor):
        return None                                                   # nothing to hide padded keys behind
    if ragged_s and (None in S or not isinstance(c0.get("encoder_attention_mask"), torch.Tensor)):
        return None
    try:
        budget = max(1, int(os.environ.get("VLMC_REPLAY_TOKENS", str(REPLAY_TOKEN_BUDGET))))
    except ValueError:
        budget = REPLAY_TOKEN_BUDGET
    # Which samples share a padded forward: ONE group.  (Buckets of similar length -- up to three, 26-34 % fewer rows -- were measured
    # slower in round 5, 486 / 541 against 478 ms: a T5 block forward is ~45 launches whatever its rows; since round 6 the linears skip
    # the padding rows and the fused attention the padding keys, so the rows buckets would save are hardly computed any more.  Removed.)
    buckets = [list(range(n_samples))]
    out = []
    chunks = []
    for bucket in buckets:
        g = max(2, min(group_max, budget // max(T[j] for j in bucket)))
        chunks += [bucket[c_:c_ + g] for c_ in range(0, len(bucket), g)]
    for chunk in chunks:
        tp = max(T[j] for j in chunk)
        sp = max(S[j] for j in chunk) if S[chunk[0]] is not None else None
        if sp is not None and sp == tp:
            sp += 8                                                   # the hooks tell the two kinds of input apart by their padded length
        dev = x0.device
        lengths = {tp: int32_on([T[j] for j in chunk], dev)}
        rows = {(len(chunk), tp): row_map([T[j] for j in chunk], tp, dev)}
        if sp is not None:
            lengths[sp] = int32_on([S[j] for j in chunk], dev)
            rows[(len(chunk), sp)] = row_map([S[j] for j in chunk], sp, dev)
        out.append((chunk, {"T": [T[j] for j in chunk], "S": [S[j] for j in chunk], "tp": tp, "sp": sp, "lengths": lengths,
                            "rows": rows}))
    return out


def int32_on(values, device):
    """A small host list (token counts, a row map) as an int32 device tensor without draining the GPU (vlmc/forward.py: int32_on)."""
    return forward.int32_on(values, device)


def row_map(lengths, padded, device):
    """(int32 device tensor [len(lengths) * padded], number of real rows) for a [samples, padded, d] stack whose sample t owns
    its first lengths[t] token rows: the flattened indices of the real rows in order, then those of the padding rows
    (vlmc_linear_fwd_rows computes the former and clears the latter; vlmc/forward.py: padded_rows)."""
    import numpy as np
    ln = np.asarray(lengths, dtype=np.int64)
    tok = np.arange(padded, dtype=np.int64)[None, :]
    real = tok < ln[:, None]
    flat = (np.arange(len(ln), dtype=np.int64)[:, None] * padded + tok)
    order = np.concatenate([flat[real], flat[~real]]).astype(np.int32)
    return int32_on(order, device), int(real.sum())


def _pad_inputs(xs, tp):
    """[1, T_j, d] tensors -> [n, tp, d], zero rows behind each sample's own"""
    x = torch.nn.utils.rnn.pad_sequence([x_[0] for x_ in xs], batch_first=True)
    if x.shape[1] < tp:
        x = torch.nn.functional.pad(x, (0, 0, 0, tp - x.shape[1]))
    return x


def _pad_caches(group, spec):
    """The cached kwargs of a padded group as one set: masks padded with the dtype's minimum (keys that do not exist; the rows of
    queries that do not exist are never read), cross-attention states with zero rows, everything else as the first sample has it."""
    n, tp, sp = len(group), spec["tp"], spec["sp"]
    out = {}
    for k, v0 in group[0].items():
        if not isinstance(v0, torch.Tensor):
            out[k] = v0
        elif k in PAD_STATE_KEYS:
            x = torch.nn.utils.rnn.pad_sequence([c[k][0] for c in group], batch_first=True)
            out[k] = torch.nn.functional.pad(x, (0, 0, 0, sp - x.shape[1])) if x.shape[1] < sp else x
        else:
            keys = tp if PAD_MASK_KEYS[k] == "self" else sp
            q = tp if v0.shape[2] != 1 else 1
            m = torch.full((n, 1, q, keys), torch.finfo(v0.dtype).min, dtype=v0.dtype, device=v0.device)
            for t, c in enumerate(group):
                v = c[k]
                m[t, :, :v.shape[2], :v.shape[3]] = v[0]
            out[k] = m
    return out


def _stack_key(x, cache):
    sig = [tuple(x.shape), x.dtype]
    for k in sorted(cache):
        v = cache[k]
        sig.append((k, tuple(v.shape), v.dtype) if isinstance(v, torch.Tensor) else (k, repr(v)))
    return tuple(sig)


def _stack_caches(group, b0):
    """The cached kwargs of a group of samples as ONE set of kwargs for a stacked forward, or None if they cannot be:
    tensors that carry the samples' batch dimension (`shape[0] == b0`: attention masks, encoder states, a per-sample
    position bias) are concatenated along it; a tensor WITHOUT it (a ViT `rel_pos_bias` [heads, N, N], a `layer_head_mask`
    [heads]) is passed once if every sample holds the same object or the same bits -- concatenating it would hand the
    block a wrong shape, or broadcast silently where the sizes happen to line up; anything else sends the group to the
    per-sample path."""
    out = {}
    for k in group[0]:
        v0 = group[0][k]
        if not isinstance(v0, torch.Tensor):
            out[k] = v0
        elif v0.dim() >= 1 and v0.shape[0] == b0 and v0.dim() >= 2:
            out[k] = torch.cat([c[k] for c in group], dim=0)
        elif all(c[k] is v0 for c in group[1:]) or all(_bits_equal(c[k], v0) for c in group[1:]):
            out[k] = v0
        else:
            return None
    return out


def block_tensors(layer):
    """(parameters and buffers of a block in a fixed order, whether any of its modules is in training mode): ONE walk over
    `_modules` / `_parameters` / `_buffers`.  `Module.parameters()` + `.buffers()` + `.modules()` are three generator walks
    with a de-duplication set each; a capture phase asks this of every block of every finished tower, and on one rank's
    share of the calibration set those walks were ~8 ms of a 130 ms prune (profiles/r04_scaling_floor.md).  A tensor shared
    by two modules is listed twice: fine for a signature."""
    ts, training, stack = [], False, [layer]
    while stack:
        m = stack.pop()
        training = training or m.training
        for p_ in m._parameters.values():
            if p_ is not None:
                ts.append(p_)
        for b_ in m._buffers.values():
            if b_ is not None:
                ts.append(b_)
        for c_ in reversed(list(m._modules.values())):
            if c_ is not None:
                stack.append(c_)
    return ts, training


def storage_signature(layer, tensors=None):
    """Addresses of every parameter and buffer of a block: a captured graph stays valid exactly as long as these do
    (Wanda / DSnoT prune in place; SparseGPT and the LoRA masks replace tensors)."""
    return tuple(t.data_ptr() for t in (block_tensors(layer)[0] if tensors is None else tensors))


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

    def _run_pass(before_sample, so, outputs):
        cur_in, cur_out = state["inps"], state["outs"]
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
        shapes = [tuple(cur_in[j].shape) for j in range(n_samples)]
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
                    prev = getattr(cur_in[chunk[0]], "_vlmc_stack", None)
                    if prev is not None and prev[1] == key and all(cur_in[j] is prev[2][t] for t, j in enumerate(chunk)):
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
                    slices = [y[t:t + 1, :tj] for t, tj in enumerate(spec["T"])]
                    slices[0]._vlmc_stack = (y, key, slices)
                    for t, j in enumerate(chunk):
                        cur_out[j] = slices[t]
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

    graphs, plan, stacked_kwargs, tail_learned = {}, {}, {}, {}
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
    if memo_cache is not None and not tuple_output:
        seed_tower_memo(memo_cache, module_to_process, layers, state["inps"][:n_samples], autocast)
    return model
