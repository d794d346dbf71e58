#!/usr/bin/env python
"""bench.py -- end-to-end Wanda 50 % unstructured prune of InstructBLIP-FlanT5-XL on MI355X.

Metric (BASELINE.json / SURVEY.md §8(d)): layers/sec and total prune wall-clock, where wall-clock is entry -> exit of
`load_pruner("blipt5_wanda_pruner", ...).prune()` (what the reference's `@print_time prune` reports,
wanda_pruner.py:947) on configs[1]: all 588 prunable linears (39 ViT-g blocks fp16 with the matrix-wide rule, 24 + 24
Flan-T5-XL blocks bf16 with the per-row rule), 128 batch-1 calibration samples, random-init weights of the true
architecture and synthetic calibration data (`vlmc/synthetic.py`), everything resident in HBM before the timed region.

THE WORKLOAD (since round 6): the stand-in whose blocks follow the reference's model files op for op -- EVA attention as
eva_vit.py:129-168 (q / v bias, explicit `q @ k^T`, softmax, `attn @ v`), T5 attention as modeling_t5.py:520-640
(`torch.matmul` scores, bucketed position bias of block 0, extended masks, fp32 softmax), the Q-Former as Qformer.py:205-246 --
with RAGGED calibration text (prompts of 8..128 tokens, answers of 4..16: blip2_t5_instruct.py:49-53 allows 128 / 256).  Rounds
2-5 quoted a friendlier stand-in (attention as one `F.scaled_dot_product_attention` call, all text of one length); that prune is
kept as the sub-object `config.sdpa_equal_lengths`.

One "step" = ONE whole prune through the drop-in pruner API: capture of each tower's inputs by the model's own
forward, block replay over the calibration samples, activation statistics, score / select / apply of every linear.
Before each step the dense weights are copied back (7.4 GB device-to-device, ~3 ms, inside the timed region).
`value` = 588 x steps / seconds.  `config.subtotals_s` (capture / replay / stat / select) come from ONE extra, untimed
prune with synchronising phase timers.  `roofline` is measured inside the timed steps with HIP events carried by the
kernel launches themselves (vlmc_set_launch_events -> hipExtLaunchKernel) on a rotating subset of the launches.
`config.invariance`: the same prune once more as the reference's one-sample-per-forward loop on this GPU, and how many mask bits
differ from the grouped replay's (0 expected).  `kernel_pass`: statistics + select kernels alone on resident activations.

N GPUs: `python bench.py --gpus N` starts N ranks by itself (torch.distributed.run as a child process, before this
process touches a GPU); under a launcher (WORLD_SIZE set) it is one of the ranks.  The 128 calibration samples are
sharded over the ranks (capture, replay and statistics scale 1/N), one RCCL all-gather of per-sample statistics per
block, select replicated: total work fixed => "scaling": "strong".

    python bench.py [--gpus N] [--steps K] [--warmup W]
"""
import argparse
import contextlib
import io
import json
import os
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "vlm-compression_amd"))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec peak
MFMA_PEAK_TFLOPS = 2500.0      # MI355X_MICROARCH.md: dense bf16 / fp16 MFMA
MFMA_F32_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: fp32 matrix (v_mfma_f32_32x32x2 / 16x16x4)
N_CALIB = 128
RATIO = 0.5
PRUNER = "blipt5_wanda_pruner"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--cpu-seconds", type=float, default=25.0, help="budget of the CPU-baseline leg (0 = skip)")
    ap.add_argument("--kernel-pass", default="auto", choices=["auto", "0", "1"],
                    help="also time the statistics + select kernels alone on resident activations (auto: only at N=1)")
    ap.add_argument("--kernel-steps", type=int, default=10)
    ap.add_argument("--event-stride", type=int, default=5,
                    help="HIP events on every N-th launch of a timed kernel (1 = every launch; a timed launch idles the "
                         "GPU for ~10 us).  Not 3, 4 or 7: a block pass is 3 (statistics pass, its tail skipped) + 4 GEMM launches of four "
                         "shapes, and a stride that divides the period samples ONE shape of the four all through a tower")
    ap.add_argument("--launch-timeout", type=float, default=1500.0, help="seconds before self-launched ranks are killed")
    ap.add_argument("--invariance", default="auto", choices=["auto", "0", "1"],
                    help="also run the headline's prune once as the reference's one-sample-per-forward loop and count the mask bits "
                         "that differ from the grouped replay's (auto: only at N=1)")
    ap.add_argument("--sdpa-leg", default="auto", choices=["auto", "0", "1"],
                    help="sub-object: the rounds 2-5 headline -- the same prune on the stand-in that writes attention as "
                         "F.scaled_dot_product_attention, all calibration text of one length (auto: only at N=1)")
    ap.add_argument("--calib-local", default="auto", choices=["auto", "0", "16", "32", "64"],
                    help="rehearse ONE rank's share of an N-GPU run on this GPU (16 / 32 / 64 samples = one rank of 8 / 4 / 2): "
                         "every kernel a rank runs, the statistics exchange filled in with the rank's own rows "
                         "(VLMC_SIMULATE_WORLD, vlmc/shard.py); reported as config.per_rank_floor, never as the headline "
                         "(auto: all three at N=1, two prunes each)")
    return ap.parse_args()


# ----------------------------------------------------------------------------------------------------------------
# N > 1 without a launcher: start the ranks as a child process, before anything here touches a GPU
# ----------------------------------------------------------------------------------------------------------------
def launch_ranks(args):
    import signal
    import socket
    n_dev = torch.cuda.device_count()               # counting devices does not initialise the GPU
    one_device = os.environ.get("VLMC_BENCH_ONE_DEVICE", "0") == "1"
    if n_dev < args.gpus and not one_device:
        raise SystemExit(f"bench.py --gpus {args.gpus}: only {n_dev} GPU(s) visible "
                         f"(VLMC_BENCH_ONE_DEVICE=1 rehearses the N > 1 path with all ranks on cuda:0 over gloo)")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    child = subprocess.Popen(cmd, env=env, start_new_session=True)
    try:
        rc = child.wait(timeout=args.launch_timeout)
    except subprocess.TimeoutExpired:
        os.killpg(child.pid, signal.SIGKILL)          # exactly the process group started above
        child.wait()
        raise SystemExit(f"bench.py: the {args.gpus} ranks did not finish within {args.launch_timeout:.0f} s "
                         f"(collective hang?) -- killed")
    raise SystemExit(rc)


# ----------------------------------------------------------------------------------------------------------------
# HIP events carried by a kernel's own dispatch
# ----------------------------------------------------------------------------------------------------------------
class HipEvents:
    """HIP events of the runtime this process has loaded (ctypes on libamdhip64), handed to the kernel launch itself
    (vlmc_set_launch_events -> hipExtLaunchKernel): start / stop are the kernel's own begin / end on its stream.
    An hipEventRecord between two kernels (torch.cuda.Event.record) costs ~5 us of idle GPU per record on MI355X."""

    def __init__(self):
        import ctypes
        path = "libamdhip64.so"
        with open("/proc/self/maps") as f:
            for line in f:
                if "libamdhip64" in line:
                    path = line.split()[-1]
                    break
        self.ct, self.hip = ctypes, ctypes.CDLL(path)
        self.hip.hipEventCreate.argtypes = [ctypes.POINTER(ctypes.c_void_p)]
        self.hip.hipEventElapsedTime.argtypes = [ctypes.POINTER(ctypes.c_float), ctypes.c_void_p, ctypes.c_void_p]
        self.hip.hipEventDestroy.argtypes = [ctypes.c_void_p]

    def new(self):
        e = self.ct.c_void_p()
        rc = self.hip.hipEventCreate(self.ct.byref(e))
        if rc != 0 or not e.value:
            raise RuntimeError(f"hipEventCreate failed ({rc})")
        return e

    def elapsed_ms(self, a, b):
        ms = self.ct.c_float()
        rc = self.hip.hipEventElapsedTime(self.ct.byref(ms), a, b)
        if rc != 0:
            raise RuntimeError(f"hipEventElapsedTime failed ({rc})")
        return float(ms.value)

    def free(self, *evs):
        for e in evs:
            self.hip.hipEventDestroy(e)


class LaunchProbe:
    """Times launches of product kernels INSIDE the timed prunes: wraps entry points of `vlmc.ops` so that every
    `stride`-th call that issues exactly ONE kernel launch carries a start / stop event pair.  The wrapper changes
    nothing else about the call; the events are read after the timed region."""

    def __init__(self, stride):
        from vlmc import _lib
        self.hipev = HipEvents()
        self.arm = _lib.load().vlmc_set_launch_events
        self.stride = max(1, stride)
        self.records = {}          # kind -> [(start, stop, algorithmic units, launches)]
        self.calls = {}
        self.active = False
        self._undo = []

    def wrap(self, module, name, kind, units_of, single_launch=lambda *a, **k: True):
        real = getattr(module, name)
        probe = self

        def wrapped(*a, **k):
            if not probe.active or torch.cuda.is_current_stream_capturing() or not single_launch(*a, **k):
                return real(*a, **k)                   # (a launch recorded into a HIP graph cannot carry events)
            n = probe.calls[kind] = probe.calls.get(kind, 0) + 1
            if n % probe.stride:
                return real(*a, **k)
            e0, e1 = probe.hipev.new(), probe.hipev.new()
            probe.arm(e0, e1)
            out = real(*a, **k)
            probe.records.setdefault(kind, []).append((e0, e1, units_of(*a, **k), 1))
            return out
        setattr(module, name, wrapped)
        self._undo.append((module, name, real))

    def restore(self):
        for module, name, real in self._undo:
            setattr(module, name, real)
        self._undo = []

    def summary(self, kind):
        evs = self.records.get(kind, [])
        ms = sum(self.hipev.elapsed_ms(a, b) for a, b, _, _ in evs)
        return ms, sum(float(u) for _, _, u, _ in evs), sum(n for _, _, _, n in evs)


def install_probes(probe):
    """The statistics kernel (one launch per group of calibration samples and block) and the per-row select (one
    mixed launch per T5 block) are single launches behind one `vlmc.ops` call each."""
    from vlmc import ops

    def sq_bytes(xs, outs=None, call_tokens=None):
        if call_tokens is not None and any(ct is not None for ct in call_tokens):     # a padded group of ragged samples: only a sample's own rows are read
            # (the token counts live on the device: summed THERE and read after the timed region -- `int(ct.sum())` here made the host
            # wait for the GPU at every probed launch of a walk, and the capture phase behind it started without the host's lead)
            return sum((ct.sum(dtype=torch.float64) if ct is not None else x.shape[0] * x.shape[1]) * (x.shape[-1] * x.element_size()) + x.shape[0] * x.shape[-1] * 4
                       for x, ct in zip(xs, call_tokens))
        return sum(x.numel() * x.element_size() + x.shape[0] * x.shape[-1] * 4 for x in xs)

    def sel_bytes(ws, sqs, mode, ks=None, n=0, m=0, apply_zero=True, **kw):
        return sum(w.numel() * (w.element_size() + 1 + (w.element_size() if apply_zero else 0)) + 4 * w.shape[1] for w in ws)

    def sel_single(ws, sqs, mode, ks=None, **kw):
        if mode != "row" or ws[0].element_size() != 2 or len(ws) > 12:
            return False
        widths = {w.shape[1] for w in ws}
        return (len(widths) == 1 and len(set(ks)) == 1) or \
            (os.environ.get("VLMC_SELECT_MIXED", "1") != "0" and max(widths) <= 8192 and min(widths) <= 2048 < max(widths)
             and all(i % 8 == 0 for i in widths))

    def gemm_flops(x, weight, bias=None, **kw):
        return 2.0 * (x.numel() // x.shape[-1]) * weight.shape[0] * weight.shape[1]

    def group_flops(x, weights, biases=None, **kw):
        return 2.0 * (x.numel() // x.shape[-1]) * sum(w.shape[0] for w in weights) * weights[0].shape[1]

    def rows_flops(x, weights, biases, rowmap, n_real, **kw):           # a padded group of ragged samples: the REAL rows only
        return 2.0 * int(n_real) * sum(w.shape[0] for w in weights) * weights[0].shape[1]

    probe.wrap(ops, "act_sqnorm_batch", "stat", sq_bytes)
    probe.wrap(ops, "wanda_select_batch", "rows", sel_bytes, sel_single)
    # one entry point, two kernels: 16-bit operands -> gemm_nt_* (2.5 PFLOP/s peak); fp32 (the Q-Former) -> gemm_f32_kernel (157 TFLOP/s peak)
    calls32 = {"n": 0}

    def linear_fwd_kind(*a, **k):
        return len(a) < 2 or a[1].dtype != torch.float32
    real_linear = ops.linear_fwd
    probe.wrap(ops, "linear_fwd", "gemm", gemm_flops, linear_fwd_kind)
    wrapped16 = ops.linear_fwd

    def linear_fwd_both(x, weight, *a, **k):                            # (fp32 calls: their own kind, timed like the others)
        if weight.dtype != torch.float32 or not probe.active or torch.cuda.is_current_stream_capturing():
            return wrapped16(x, weight, *a, **k)
        n = probe.calls["gemm32"] = probe.calls.get("gemm32", 0) + 1
        if n % probe.stride:
            return real_linear(x, weight, *a, **k)
        e0, e1 = probe.hipev.new(), probe.hipev.new()
        probe.arm(e0, e1)
        out = real_linear(x, weight, *a, **k)
        if out is not None:
            probe.records.setdefault("gemm32", []).append((e0, e1, gemm_flops(x, weight), 1))
        return out
    ops.linear_fwd = linear_fwd_both
    probe._undo.append((ops, "linear_fwd", real_linear))
    probe.wrap(ops, "linear_fwd_group", "gemm", group_flops)          # q / k / v, wi_0 / wi_1: one launch (vlmc/forward.py)
    probe.wrap(ops, "linear_fwd_rows", "gemm", rows_flops)            # the same products over a row map (padding rows skipped)

    def fused_flops(q, k, v, *rest, **kw):                              # both products of the chain: 4 B H Tq Tk d
        return 4.0 * q.shape[0] * q.shape[1] * q.shape[2] * k.shape[2] * q.shape[3]
    probe.wrap(ops, "attn_fused", "attnf", fused_flops)               # the whole attention chain of a replayed block in one launch

    def sdpa_flops(q, k, v, *a, **kw):                                  # both products: 4 B H Tq Tk d
        return 4.0 * q.shape[0] * q.shape[1] * q.shape[2] * k.shape[2] * q.shape[3]

    def sdpa_single(q, k, v, *a, **kw):
        return ops.sdpa_plan(q, k, v) is not None
    probe.wrap(ops, "sdpa", "attn", sdpa_flops, sdpa_single)            # F.scaled_dot_product_attention of a replayed block


# ----------------------------------------------------------------------------------------------------------------
# the headline: whole prunes through the drop-in API
# ----------------------------------------------------------------------------------------------------------------
class PruneJob:
    def __init__(self, dev, reference_ops=False, ragged=False):
        from vlmc import synthetic
        self.dev = dev
        self.model = synthetic.InstructBlipT5(reference_ops=reference_ops).to(dev).eval()
        synthetic.randomize_(self.model, 0)
        # the parameters live in ONE buffer per dtype (views of it; 256-byte aligned), so that the weight restore that opens every timed
        # step is three device-to-device copies instead of ~1 500 (one `copyBuffer` launch per tensor: ~5 ms of GPU time and as much host time)
        by_dtype = {}
        for p in self.model.parameters():
            by_dtype.setdefault(p.dtype, []).append(p)
        self.flat, self.dense = [], []
        with torch.no_grad():
            for dt, ps in by_dtype.items():
                sizes = [(p.numel() + 127) // 128 * 128 for p in ps]
                flat = torch.zeros(sum(sizes), dtype=dt, device=dev)
                off = 0
                for p, sz in zip(ps, sizes):
                    view = flat[off:off + p.numel()].view(p.shape)
                    view.copy_(p.data)
                    p.data = view
                    off += sz
                self.flat.append(flat)
                self.dense.append(flat.clone())
        self.batches = synthetic.calibration_batches(N_CALIB, dev, vocab=self.model.t5_model.shared.num_embeddings, ragged=ragged)
        self.n_linears = synthetic.prunable_linears(self.model)
        keep = 1 - RATIO
        self.cfg = dict(t5_prune_spec=f"24-{keep!r}-1.0-1.0", vit_prune_spec=f"39-{keep!r}-1.0-1.0", t5_pruning_method="wanda",
                        vit_pruning_method="wanda", num_samples=N_CALIB, max_sparsity_per_layer=1.01)

    def step(self):
        from lavis.compression import load_pruner
        with torch.no_grad():
            for flat, dense in zip(self.flat, self.dense):          # dense weights back (7.4 GB d2d)
                flat.copy_(dense)
        pruner = load_pruner(PRUNER, self.model, self.batches, cfg=dict(self.cfg))
        with contextlib.redirect_stdout(io.StringIO()):
            pruner.prune()

    def masks(self):
        return {n: m.mask.clone() for n, m in self.model.named_modules()
                if isinstance(m, torch.nn.Linear) and isinstance(getattr(m, "mask", None), torch.Tensor)}

    def pruned_fraction(self):
        zeros = total = 0
        for name, mod in self.model.named_modules():
            if isinstance(mod, torch.nn.Linear) and (".blocks." in name or ".block." in name):
                zeros += int((mod.weight == 0).sum())
                total += mod.weight.numel()
        return zeros / max(1, total)


# ----------------------------------------------------------------------------------------------------------------
# kernel pass (last round's headline, kept as a sub-object): statistics + select on resident activations
# ----------------------------------------------------------------------------------------------------------------
def kernel_pass(dev, steps, warmup, stride):
    from vlmc import _lib, ops, workload as wl
    blocks = wl.flan_t5_xl()
    acts, weights0 = [], []
    gi = 0
    for bi, b in enumerate(blocks):
        per_in = []
        for ii, inp in enumerate(b.inputs):
            g = torch.Generator(device=dev).manual_seed(1_000_003 * bi + 1009 * ii)
            x = torch.empty((N_CALIB, inp.tokens, inp.in_features), dtype=b.dtype, device=dev)
            x.normal_(0.1, 1.0, generator=g)
            per_in.append(x)
        acts.append(per_in)
        wb = []
        for lin in b.linears:
            g = torch.Generator(device=dev).manual_seed(gi)
            w = torch.empty((lin.out_features, lin.in_features), dtype=b.dtype, device=dev)
            w.normal_(0.0, 0.02, generator=g)
            wb.append(w)
            gi += 1
        weights0.append(wb)
    n_sets = steps + warmup
    sets = [weights0] + [[[w.clone() for w in wb] for wb in weights0] for _ in range(n_sets - 1)]
    plans_per_set = []
    state = []
    for b in blocks:
        tot = sum(i.in_features for i in b.inputs)
        flat = torch.empty((N_CALIB, tot), dtype=torch.float32, device=dev)
        ins, off = [], 0
        for inp in b.inputs:
            ins.append({"scaler": torch.zeros(inp.in_features, dtype=torch.float32, device=dev),
                        "sqrt": torch.empty(inp.in_features, dtype=torch.float32, device=dev),
                        "normsq": flat[:, off:off + inp.in_features]})
            off += inp.in_features
        state.append({"ins": ins,
                      "masks": [torch.empty((l.out_features, l.in_features), dtype=torch.bool, device=dev) for l in b.linears],
                      "partials": [torch.empty(ops.select_partials(b.mode, l.out_features, l.in_features), dtype=torch.float64,
                                               device=dev) for l in b.linears]})
    for ws_all in sets:
        plans = []
        for bi, b in enumerate(blocks):
            ins = state[bi]["ins"]
            stat = ops.plan_act_sqnorm_batch(acts[bi], [s["normsq"] for s in ins])
            upd = ops.plan_scaler_update_batch([s["scaler"] for s in ins], 0, [s["normsq"] for s in ins], 1, [s["sqrt"] for s in ins])
            ws, sqs, ks, nbytes, li = [], [], [], 0, 0
            for ii, inp in enumerate(b.inputs):
                for lin in inp.linears:
                    w = ws_all[bi][li]
                    ws.append(w)
                    sqs.append(ins[ii]["sqrt"])
                    ks.append(int(lin.in_features * RATIO) if b.mode == "row" else int(lin.out_features * lin.in_features * RATIO))
                    nbytes += wl.select_bytes(lin, w.element_size(), True)
                    li += 1
            sel = ops.plan_select_batch(ws, sqs, b.mode, ks=ks, apply_zero=True, masks=state[bi]["masks"],
                                        partials=state[bi]["partials"])
            sbytes = sum(wl.stat_bytes(inp, N_CALIB, acts[bi][ii].element_size()) for ii, inp in enumerate(b.inputs))
            plans.append((stat, upd, sel, b.mode, nbytes, sbytes))
        plans_per_set.append(plans)

    hipev = HipEvents()
    arm = _lib.load().vlmc_set_launch_events
    events = {"stat": [], "rows": [], "matrix": []}

    def run(plans, phase=None):
        for bi, (stat, upd, sel, mode, nbytes, sbytes) in enumerate(plans):
            timed = phase is not None and (bi + phase) % stride == 0
            if timed:
                a, b_ = hipev.new(), hipev.new()
                arm(a, b_)
                stat()
                events["stat"].append((a, b_, sbytes))
            else:
                stat()
            upd()
            if timed:
                a, b_ = hipev.new(), hipev.new()
                arm(a, b_)
                sel()
                events["rows" if mode == "row" else "matrix"].append((a, b_, nbytes))
            else:
                sel()

    for i in range(warmup):
        run(plans_per_set[i])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        run(plans_per_set[warmup + i], phase=i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    n_lin = sum(len(b.linears) for b in blocks)

    def roof(kind, kernel, tkey, traffic):
        evs = events[kind]
        ms = sum(hipev.elapsed_ms(a, b_) for a, b_, _ in evs)
        nb = sum(x for _, _, x in evs)
        ach = nb / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        return {"bound": "hbm", "kernel": kernel, "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic.get(tkey), "launches": len(evs),
                "avg_launch_us": round(ms * 1e3 / max(1, len(evs)), 2), "bytes_per_launch": round(nb / max(1, len(evs)))}

    traffic = load_traffic()
    out = {"what": "statistics + running mean + fused score/select/apply of all 588 linears on activations resident in HBM "
                   "(128 samples x 257/64/16 tokens), no block forward -- SURVEY.md §8 rows a3-a8 alone",
           "steps": steps, "ms_per_pass": round(dt / steps * 1e3, 3), "layers_per_s": round(n_lin * steps / dt, 1),
           "kernels": [roof("stat", "vlmc::act_sqnorm_kernel", "act_sqnorm_kernel_bytes_per_launch", traffic),
                       roof("rows", "vlmc::select_rows_mixed_kernel", "select_rows_mixed_kernel_bytes_per_launch", traffic),
                       roof("matrix", "vlmc::matrix_fused_kernel", "matrix_fused_kernel_bytes_per_launch", traffic)]}
    del acts, sets, plans_per_set, state
    torch.cuda.empty_cache()
    return out


# ----------------------------------------------------------------------------------------------------------------
# the grouped replay against the reference's own loop, on this GPU
# ----------------------------------------------------------------------------------------------------------------
def invariance_leg(job):
    """The headline's masks (grouped, padded replay) against ONE more prune run as the reference's loop -- one calibration sample
    per block forward (wanda_pruner.py:308-311, :343-346; VLMC_BATCH_REPLAY=1, VLMC_TOWER_BATCH=0) -- on the same GPU: the
    blocks' linears, attention products, softmax, norms and GELU run on batch- and padding-invariant kernels during a replay
    (vlmc/forward.py), so the agreement is expected to be exactly 1.0."""
    grouped = job.masks()
    keep = {k: os.environ.get(k) for k in ("VLMC_BATCH_REPLAY", "VLMC_TOWER_BATCH")}
    os.environ.update(VLMC_BATCH_REPLAY="1", VLMC_TOWER_BATCH="0")
    try:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        job.step()
        torch.cuda.synchronize()
        per_sample_sec = time.perf_counter() - t0
    finally:
        for k, v in keep.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    single = job.masks()
    total = diff = 0
    worst = (0.0, None)
    for n, m in grouped.items():
        d = int((m != single[n]).sum())
        total += m.numel()
        diff += d
        if d / m.numel() > worst[0]:
            worst = (d / m.numel(), n)
    return {"what": "mask bits of the timed (grouped, padded) prune against one more prune run as the reference's one-sample-per-forward "
                    "loop on this GPU", "per_sample_loop_seconds": round(per_sample_sec, 3),
            "mask_agreement_grouped_vs_per_sample": round(1.0 - diff / max(1, total), 9), "mask_elements": total,
            "mask_elements_differing": diff, "worst_linear": {"name": worst[1], "fraction_differing": round(worst[0], 9)}}


def sdpa_leg(dev, steps, stride):
    """Rounds 2-5's headline, kept as a sub-object: the same whole prune on the stand-in whose blocks write attention as ONE
    `F.scaled_dot_product_attention` call (no position bias, no masks) and whose calibration text is all of one length (32 + 32
    prompt tokens, 16 answer tokens): one group of 128 equal samples per block forward, nothing to pad."""
    from vlmc import ops
    job = PruneJob(dev)
    probe = LaunchProbe(stride)

    def sdpa_flops(q, k, v, *a, **kw):                                  # both products: 4 B H Tq Tk d
        return 4.0 * q.shape[0] * q.shape[1] * q.shape[2] * k.shape[2] * q.shape[3]

    def gemm_flops(x, weight, bias=None, **kw):
        return 2.0 * (x.numel() // x.shape[-1]) * weight.shape[0] * weight.shape[1]

    def group_flops(x, weights, biases=None, **kw):
        return 2.0 * (x.numel() // x.shape[-1]) * sum(w.shape[0] for w in weights) * weights[0].shape[1]
    probe.wrap(ops, "sdpa", "attn", sdpa_flops, lambda q, k, v, *a, **kw: ops.sdpa_plan(q, k, v) is not None)
    probe.wrap(ops, "linear_fwd", "gemm", gemm_flops)
    probe.wrap(ops, "linear_fwd_group", "gemm", group_flops)
    try:
        job.step()
        job.step()
        torch.cuda.synchronize()
        probe.active = True
        t0 = time.perf_counter()
        for _ in range(steps):
            job.step()
        torch.cuda.synchronize()
        sec = (time.perf_counter() - t0) / steps
        probe.active = False
    finally:
        probe.restore()
    out = {"what": "the same whole prune on the friendlier stand-in rounds 2-5 quoted: attention written as F.scaled_dot_product_attention, "
                   "128 calibration samples of equal length (257 image tokens, 32 + 32 text, 16 output tokens)",
           "seconds_per_prune": round(sec, 4), "layers_per_s": round(588 / sec, 1), "steps": steps,
           "pruned_fraction": round(job.pruned_fraction(), 6)}
    for kind, name in (("gemm", "vlmc::gemm_nt_kernel"), ("attn", "vlmc::sdpa_fwd_kernel")):
        ms, flops, n = probe.summary(kind)
        if n:
            tfs = flops / (ms * 1e-3) / 1e12
            calls = probe.calls.get(kind, 0)
            out[kind] = {"kernel": name, "bound": "mfma", "achieved": round(tfs, 1), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(tfs / MFMA_PEAK_TFLOPS, 4), "avg_launch_us": round(ms * 1e3 / n, 2),
                         "launches_per_step": round(calls / steps, 1), "gpu_ms_per_step": round(ms / n * calls / steps, 2)}
    del job
    torch.cuda.empty_cache()
    return out


def capture_info():
    """How the capture phases of the prunes so far forwarded the calibration batches (calibration.graph_stats, whole process)."""
    from lavis.compression.pruners import calibration as cal
    g = cal.graph_stats
    return {"what": "model forwards that carried ALL calibration batches of one shape stacked (merged_forwards; each checked against one "
                    "sample's own forward bit for bit) / capture phases that took the per-sample route instead (declined)",
            "merged_forwards": g.get("merged_forwards", 0), "declined": g.get("merged_capture_declined", 0),
            "mismatches": g.get("merged_capture_mismatch", 0), "errors": g.get("merged_capture_errors", 0)}


def load_traffic():
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        return json.load(open(tpath)) if os.path.exists(tpath) else {}
    except Exception:
        return {}


# ----------------------------------------------------------------------------------------------------------------
# CPU baseline: the oracle's PyTorch-CPU op sequence on the GPU box's host cores
# ----------------------------------------------------------------------------------------------------------------
def cpu_baseline(budget_s):
    """`oracle/wanda_torch.py` (the reference's own op sequence: running-mean (||x||_2)^2 statistics -> |W| * sqrt(s) ->
    stable per-row sort / matrix-wide threshold -> scatter + zero), pinned to the reference's goldens by
    tests/test_oracle_golden.py.  Data are generated OUTSIDE the timers; per linear 1 warm-up + median of >= 5 runs,
    with all host threads and with one.  Checker code timed as a baseline, never shipped."""
    from oracle import wanda_torch as OT
    cases = [  # configs[0] (one FlanT5-XL encoder FFN linear, 8 CPU calibration samples of 64 tokens) + the ViT rule
        ("t5.wi_0 5120x2048 bf16 per-row", 5120, 2048, torch.bfloat16, "row", 8, 64),
        ("t5.wo 2048x5120 bf16 per-row", 2048, 5120, torch.bfloat16, "row", 8, 64),
        ("vit.fc1 6144x1408 fp16 matrix-wide", 6144, 1408, torch.float16, "matrix", 8, 257),
    ]
    data = []
    for ci, (name, out_f, in_f, dt, mode, n, T) in enumerate(cases):
        g = torch.Generator().manual_seed(ci)
        w = (torch.randn((out_f, in_f), generator=g) * 0.02).to(dt)
        xs = [(torch.randn((1, T, in_f), generator=torch.Generator().manual_seed(1 + j)) + 0.1).to(dt) for j in range(n)]
        data.append((name, w, xs, mode))
    all_threads = torch.get_num_threads()
    t_start = time.perf_counter()

    def one(w, xs, mode):
        st = OT.WandaStat(w.shape[1])
        for x in xs:
            st.add_batch(x)
        OT.prune_linear(w.clone(), st.scaler_row, mode, ratio=RATIO, apply_zero=True)

    res = {}
    for threads in (all_threads, 1):
        torch.set_num_threads(threads)
        per = []
        for name, w, xs, mode in data:
            one(w, xs, mode)                                                    # warm-up
            ts = []
            while len(ts) < 5 or (len(ts) < 9 and time.perf_counter() - t_start < budget_s * 0.5):
                t0 = time.perf_counter()
                one(w, xs, mode)
                ts.append(time.perf_counter() - t0)
                if time.perf_counter() - t_start > budget_s * 2:
                    break
            med = statistics.median(ts)
            b_sel = w.numel() * (w.element_size() + 1 + w.element_size()) + 4 * w.shape[1]
            per.append({"linear": name, "ms": round(med * 1e3, 2), "runs": len(ts), "GBps": round(b_sel / med / 1e9, 4)})
        res[threads] = per
    torch.set_num_threads(all_threads)
    tot = sum(p["ms"] for p in res[all_threads]) * 1e-3
    tot1 = sum(p["ms"] for p in res[1]) * 1e-3
    return {"value": round(len(cases) / tot, 3), "unit": "layers/s", "cores": all_threads, "kind": "port",
            "sample": f"configs[0]: Wanda 50 % of t5.wi_0 5120x2048 and t5.wo 2048x5120 (bf16, per-row rule, 8 calibration samples "
                      f"x 64 tokens) + vit.fc1 6144x1408 (fp16, matrix-wide rule, 8 x 257 tokens): statistics + score + "
                      f"select + apply per linear, data generated outside the timers, 1 warm-up + median of >= 5 runs; "
                      f"{time.perf_counter() - t_start:.1f} s of CPU work",
            "ms_per_linear": res[all_threads], "one_thread": {"value": round(len(cases) / tot1, 3), "unit": "layers/s",
                                                              "cores": 1, "ms_per_linear": res[1]},
            "bytes_model": "B_sel = out*in*(2 [read W] + 1 [bool mask] + 2 [zeroed W]) + 4*in (SURVEY.md §8d)"}


# ----------------------------------------------------------------------------------------------------------------
def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        launch_ranks(args)                          # never returns
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback of the product path)")
    # VLMC_BENCH_ONE_DEVICE=1 (rehearsing the N > 1 code path on a 1-GPU box): every rank uses cuda:0 and the
    # collectives run over gloo -- never set by the driver
    one_device = os.environ.get("VLMC_BENCH_ONE_DEVICE", "0") == "1"
    if one_device:
        local_rank = 0
        # ranks sharing a device contend for the CUs the fused matrix-wide select keeps to itself: four-launch form
        os.environ.setdefault("VLMC_MATRIX_FUSED", "0")
    elif world > 1 and torch.cuda.device_count() < world:
        raise SystemExit(f"WORLD_SIZE={world} but only {torch.cuda.device_count()} GPU(s) visible")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    backend = None
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        import datetime
        backend = "gloo" if one_device else "nccl"
        kw = {} if one_device else {"device_id": dev}
        dist.init_process_group(backend, timeout=datetime.timedelta(seconds=600), **kw)
    assert N_CALIB % world == 0

    from vlmc import _lib
    _lib.load()                                   # fail loudly if the HIP library is missing

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- groups per block forward: counted through the planner the walk itself calls (one plan per tower and prune) ----------
    from lavis.compression.pruners import calibration as cal
    groups = []
    real_plan, real_padded = cal.plan_groups, cal.plan_padded

    def counting(*a, **k):
        chunks = real_plan(*a, **k)
        groups.append(len(chunks))
        return chunks

    def counting_padded(*a, **k):
        padded = real_padded(*a, **k)
        if padded is not None:
            groups.append(len(padded))
        return padded
    cal.plan_groups, cal.plan_padded = counting, counting_padded

    job = PruneJob(dev, reference_ops=True, ragged=True)
    probe = LaunchProbe(args.event_stride)
    install_probes(probe)
    for _ in range(args.warmup):
        job.step()
    sync()
    groups.clear()
    probe.active = True
    t0 = time.perf_counter()
    for _ in range(args.steps):
        job.step()
    sync()
    elapsed = time.perf_counter() - t0
    probe.active = False
    probe.restore()
    cal.plan_groups, cal.plan_padded = real_plan, real_padded
    rank_ms = [elapsed / args.steps * 1e3]
    if world > 1:
        every = torch.zeros(world, dtype=torch.float64, device=dev)       # (a sum of one-hot vectors: all_reduce is what both backends carry)
        every[rank] = elapsed
        dist.all_reduce(every, op=dist.ReduceOp.SUM)
        rank_ms = [round(float(e) / args.steps * 1e3, 3) for e in every.tolist()]
        elapsed = max(every.tolist())
    pruned_fraction = job.pruned_fraction()
    per_tower = {}
    if groups and args.steps:
        per = len(groups) // args.steps
        for t_, name in enumerate(("visual_encoder.blocks", "t5_model.encoder.block", "t5_model.decoder.block")[:per]):
            per_tower[name] = groups[t_]

    # ---- sub-totals: one extra, untimed prune with synchronising phase timers ------------------------------------
    from vlmc import phases
    os.environ["VLMC_PHASE_TIMERS"] = "1"
    phases.reset()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    job.step()
    torch.cuda.synchronize()
    phased_total = time.perf_counter() - t1
    os.environ["VLMC_PHASE_TIMERS"] = "0"
    sub = {}
    for k, v in phases.times.items():                       # capture is timed per tower; the headline keys are the four of §8(d)
        sub[k.split()[0]] = round(sub.get(k.split()[0], 0.0) + v, 4)
    sub["capture_by_tower"] = {k.split(None, 1)[1]: round(v, 4) for k, v in phases.times.items() if k.startswith("capture ")}
    sub["other_host"] = round(phased_total - sum(phases.times.values()), 4)
    sub["total_with_phase_syncs"] = round(phased_total, 4)

    # ---- roofline of the dominant product kernel of the step --------------------------------------------------------
    traffic = load_traffic()

    def roof(kind, kernel, tkey, bound="hbm"):
        ms, units, n = probe.summary(kind)
        calls = probe.calls.get(kind, 0)
        timed = (f"HIP events carried by the launch (hipExtLaunchKernel) on every {probe.stride}-th launch of this kernel inside "
                 f"the timed prunes")
        avg_us = ms * 1e3 / max(1, n)
        common = {"kernel": kernel, "launches": n, "avg_launch_us": round(avg_us, 2), "launches_per_step": round(calls / args.steps, 1),
                  "gpu_ms_per_step": round(avg_us * calls / args.steps * 1e-3, 2), "traffic": traffic.get(tkey),
                  "traffic_source": "profiles/traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this workload taken when "
                                    "the profiles were collected (static; NOT measured in this run)", "timed": timed}
        if bound in ("mfma", "mfma32"):
            peak = MFMA_PEAK_TFLOPS if bound == "mfma" else MFMA_F32_PEAK_TFLOPS
            ach = units / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
            return dict(common, bound="mfma", achieved=round(ach, 1), peak=peak, unit="TFLOP/s",
                        frac=round(ach / peak, 4), flops_per_launch=round(units / max(1, n)))
        ach = units / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        return dict(common, bound="hbm", achieved=round(ach, 1), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(ach / HBM_PEAK_GBS, 4),
                    bytes_per_launch=round(units / max(1, n)))

    rows = [roof("gemm", "vlmc::gemm_nt_kernel (vlmc_linear_fwd / _group / _rows: the dense calibration forward of the blocks' linears on "
                         "v_mfma_f32_16x16x32, batch-invariant; algorithmic flops = 2 M N K over the REAL rows of a padded group)",
                 "gemm_nt_bytes_per_launch", "mfma"),
            roof("stat", "vlmc::act_sqnorm_kernel (per-sample squared column norms of every distinct linear input of a block; one "
                         "launch per group of calibration samples)", "act_sqnorm_kernel_bytes_per_launch"),
            roof("rows", "vlmc::select_rows_mixed_kernel (score+select+apply, per-row rule; all linears of a T5 block in one "
                         "launch)", "select_rows_mixed_kernel_bytes_per_launch"),
            roof("gemm32", "vlmc::gemm_f32_kernel (vlmc_linear_fwd, fp32: the linears of the reference's fp32 Q-Former, ln_vision / t5_proj side, on "
                           "v_mfma_f32_16x16x4_f32, batch-invariant; peak = the fp32 matrix rate)", "gemm_f32_bytes_per_launch", "mfma32"),
            roof("attnf", "vlmc::attn_fused_kernel (vlmc_attn_fwd: scores, scaling, position bias / mask addends, fp32 softmax and probs @ v "
                          "of a replayed block's attention in one launch, every intermediate rounded like the tensor op it replaces; "
                          "algorithmic flops = 4 B H Tq Tk d; what paces it is the softmax's VALU work and LDS fragment reads, not the "
                          "matrix cores)", "attn_fused_kernel_bytes_per_launch", "mfma")]
    # the dominant kernel = the product kernel with the most GPU time per step (measured, not assumed)
    rows.sort(key=lambda r: -r["gpu_ms_per_step"])
    roofline = rows[0]
    roofline["other"] = rows[1:]

    invariance = None
    if args.invariance == "1" or (args.invariance == "auto" and world == 1):
        try:
            invariance = invariance_leg(job)
        except Exception as e:                    # never lose the bench line to a side measurement
            invariance = {"error": f"{type(e).__name__}: {e}"}

    # ---- one rank's floor of an N-GPU run, rehearsed on this GPU (--calib-local) -----------------------------------
    floor = None
    locals_ = [] if world > 1 or args.calib_local == "0" else ([16, 32, 64] if args.calib_local == "auto" else [int(args.calib_local)])
    if locals_:
        floor = {"what": "rank 0 of W rehearsed on this GPU: capture + replay + statistics of its 128 / W samples, selects of all 588 "
                         "linears, the running-mean recurrence over all 128 rows; the per-block all-gather is filled in with the rank's "
                         "own rows (VLMC_SIMULATE_WORLD) -- no RCCL, no arrival skew.  projected_speedup = this run's N = 1 seconds / "
                         "the rank's seconds: an upper bound for N = W",
                 "ranks": {}}
        fsteps = args.steps if args.calib_local != "auto" else 2
        for loc in locals_:
            os.environ["VLMC_SIMULATE_WORLD"] = str(N_CALIB // loc)
            try:
                job.step()                                            # warm-up at the rank's shapes (code objects, allocator)
                torch.cuda.synchronize()
                busy0 = time.perf_counter()
                for _ in range(fsteps):
                    job.step()
                torch.cuda.synchronize()
                fsec = (time.perf_counter() - busy0) / fsteps
                floor["ranks"][f"world_{N_CALIB // loc}"] = {
                    "calib_samples_on_this_rank": loc, "seconds_per_prune": round(fsec, 4),
                    "projected_speedup": round(elapsed / args.steps / fsec, 2), "pruned_fraction": round(job.pruned_fraction(), 6)}
            except Exception as e:
                floor["ranks"][f"world_{N_CALIB // loc}"] = {"error": f"{type(e).__name__}: {e}"}
            finally:
                os.environ.pop("VLMC_SIMULATE_WORLD", None)

    job = None
    torch.cuda.empty_cache()
    kp = None
    if args.kernel_pass == "1" or (args.kernel_pass == "auto" and world == 1):
        try:
            kp = kernel_pass(dev, args.kernel_steps, 2, max(1, args.event_stride))
            for k in kp["kernels"]:
                if "matrix_fused" in k["kernel"]:
                    roofline["other"].append(dict(k, timed="kernel_pass sub-run (the fused matrix-wide select's call issues no "
                                                           "other launch there)"))
        except Exception as e:                    # never lose the bench line to the side measurement
            kp = {"error": f"{type(e).__name__}: {e}"}

    sdpa = None
    if args.sdpa_leg == "1" or (args.sdpa_leg == "auto" and world == 1):
        torch.cuda.empty_cache()
        try:
            sdpa = sdpa_leg(dev, max(1, min(args.steps, 3)), args.event_stride)
        except Exception as e:
            sdpa = {"error": f"{type(e).__name__}: {e}"}

    out = None
    if rank == 0:
        sec = elapsed / args.steps
        n_lin = 588
        out = {
            "metric": "layers/sec + total prune wall-clock, InstructBLIP-FlanT5-XL Wanda@50% (588 linears, 128 calib samples)",
            "value": round(n_lin * args.steps / elapsed, 1), "unit": "layers/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(sec * 1e3, 3), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f16/bf16", "data": "synthetic",
            "config": {"workload": "configs[1]: one whole load_pruner('blipt5_wanda_pruner').prune() -- Wanda 50% unstructured, "
                                   "InstructBLIP-FlanT5-XL architecture and shapes (39 ViT-g blocks fp16 matrix-wide rule + 24/24 "
                                   "Flan-T5-XL blocks bf16 per-row rule; between them the 12-layer Q-Former, never pruned), the blocks written "
                                   "with the REFERENCE'S OP SEQUENCE (eva_vit.py:129-168: q / v bias, explicit q @ k^T, softmax, attn @ v; "
                                   "modeling_t5.py:520-640: torch.matmul scores, bucketed position bias of block 0, extended masks, fp32 "
                                   "softmax; Qformer.py:205-246), random init, 128 batch-1 calibration samples with RAGGED text (257 image "
                                   "tokens, 32 queries + prompts of 8..128 tokens, answers of 4..16 tokens)",
                       "linears": n_lin, "blocks": 87, "calib_samples": N_CALIB, "ratio": RATIO,
                       "total_prune_wall_clock_s": round(sec, 4), "blocks_per_s": round(87 / sec, 1),
                       "pruned_fraction": round(pruned_fraction, 6), "subtotals_s": sub,
                       "replay": f"grouped: up to {os.environ.get('VLMC_BATCH_REPLAY', '128')} samples per block forward; ragged samples "
                                 "padded into one group, the linears skip the padding rows (vlmc_linear_fwd_rows)",
                       "groups_per_block_forward": per_tower,
                       "capture": capture_info(), "invariance": invariance,
                       "parallelism": f"calib-dp{world}", "backend": backend, "per_rank_floor": floor,
                       "sdpa_equal_lengths": sdpa,
                       "world_size": dist.get_world_size() if world > 1 else 1,
                       "per_rank_ms_per_step": rank_ms,
                       "exchange": ("one all-gather of the per-sample statistics rows per block (87 per prune); seconds inside them in the "
                                    "phase-timed prune: subtotals_s.exchange (includes the ranks' arrival skew)") if world > 1 else None},
            "roofline": roofline,
            "kernel_pass": kp,
        }
        out["cpu_baseline"] = cpu_baseline(args.cpu_seconds) if world == 1 and args.cpu_seconds > 0 else None
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
