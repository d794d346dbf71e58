"""What the calibration replay engine's modules share: the per-thread prune context (`_CTX`), the engine's counters (`graph_stats`), its
switches, and the small helpers every part needs -- `get_module_recursive` / `find_layers` (wanda_pruner.py:16-48), the cache-key lists of the
Catcher (:225-236), graph capture on a side stream, `block_tensors` / `storage_signature`.  The engine itself: `calibration.py` (the walk, the
public names), `replay_capture.py` (the capture phases), `replay_towers.py` (finished towers: memo, stacked passes, proxies), `replay_padding.py`
(groups of samples and their padding).  Split out of `calibration.py` in round 6; nothing here is new."""
from __future__ import annotations

import contextlib
import os
import threading

import torch
import torch.nn as nn

from vlmc import forward, phases

from vlmc.shard import calibration_shard  # noqa: E402,F401  (one answer for capture, replay and the exchanges)


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


def _bits_equal(r, v):
    if _CTX.later is not None and r.is_cuda:
        return _CTX.later.same(r, v)
    return r.shape == v.shape and r.dtype == v.dtype and r.device == v.device and bool(torch.equal(r, v))


def tower_memo_enabled():
    """Outputs of a finished tower are remembered from one capture phase to the next (`VLMC_TOWER_MEMO=0`: off)."""
    return os.environ.get("VLMC_TOWER_MEMO", "1") != "0"


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


def later_check_enabled():
    """`VLMC_LATER_EQUAL=0`: every comparison of a remembered input with the one at hand waits for its answer."""
    return os.environ.get("VLMC_LATER_EQUAL", "1") != "0"


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


def pad_ragged_enabled():
    from vlmc import forward as fw
    return os.environ.get("VLMC_PAD_RAGGED", "1") != "0" and fw.enabled() and fw.attn_matmul_enabled() and fw.softmax_enabled()


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
