#!/usr/bin/env python
"""bench.py -- Wanda 50 % unstructured prune of InstructBLIP-FlanT5-XL on MI355X.

Metric (BASELINE.json): layers/sec (layer = one pruned nn.Linear) and total prune
wall-clock for configs[1]: all 588 prunable linears (39 ViT-g blocks fp16 with the
matrix-wide rule, 24+24 T5 blocks bf16 with the per-row rule), 128 calibration
samples, synthetic weights N(0, 0.02) and synthetic activations N(0.1, 1) of the
model's shapes (257 / 64 / 16 tokens), all resident in HBM before the timed region.

One "step" = one pass of the hot path over the whole model: per transformer block,
activation statistics of every distinct linear input over the 128 samples
(vlmc_act_sqnorm), the running-mean recurrence + sqrt (vlmc_wanda_scaler_update),
and the fused score + select + apply of every linear (vlmc_wanda_select) -- i.e.
SURVEY.md §8 rows a3-a8 without the block forward (row (f)1: the `end_to_end` object times a
whole drop-in prune, replay engine included, beside the bench line).  Each timed step prunes a
fresh copy of the dense weights.  `roofline` comes from HIP events carried by the kernel launches
themselves (vlmc_set_launch_events -> hipExtLaunchKernel) on the launch stream, inside the timed
region, on every `--event-stride`-th block of a step (DESIGN.md section 6 says why).

N GPUs (`torchrun`, one rank per GPU): the 128 calibration samples are sharded in
contiguous ranges, one RCCL all-gather of per-sample squared norms per block, select
replicated on every rank -> total work fixed => "scaling": "strong".

    python bench.py [--gpus N] [--steps K] [--warmup W]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "vlm-compression_amd"))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec peak
N_CALIB = 128
RATIO = 0.5


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="budget of the CPU-baseline leg (0 = skip)")
    ap.add_argument("--e2e", default="auto", choices=["auto", "0", "1"],
                    help="also time one whole drop-in blipt5_wanda_pruner.prune() on the synthetic InstructBLIP-FlanT5-XL "
                         "(auto: only at N=1)")
    ap.add_argument("--event-stride", type=int, default=4,
                    help="HIP events on every N-th block of a step, rotating with the step (1 = every launch; a timed "
                         "launch idles the GPU for ~10 us)")
    ap.add_argument("--weight-sets", type=int, default=0, help="dense weight copies kept in HBM (0 = steps+warmup, capped by memory)")
    return ap.parse_args()


def build_workload(dev, rank, world, n_sets_wanted):
    from vlmc import workload as wl
    blocks = wl.flan_t5_xl()
    n_local = N_CALIB // world
    # ---- activations: rank r holds samples [r*n_local, (r+1)*n_local) of every distinct input
    acts = []
    for bi, b in enumerate(blocks):
        per_in = []
        for ii, inp in enumerate(b.inputs):
            g = torch.Generator(device=dev)
            g.manual_seed(1_000_003 * bi + 1009 * ii + rank)
            x = torch.empty((n_local, inp.tokens, inp.in_features), dtype=b.dtype, device=dev)
            x.normal_(0.1, 1.0, generator=g)
            per_in.append(x)
        acts.append(per_in)
    # ---- weights: as many dense copies as fit (each timed step prunes a fresh one)
    bytes_per_set = sum(l.out_features * l.in_features * 2 for b in blocks for l in b.linears)
    free, _ = torch.cuda.mem_get_info(dev)
    mask_bytes = bytes_per_set // 2
    fit = int((free - mask_bytes - (8 << 30)) // bytes_per_set)
    n_sets = max(1, min(n_sets_wanted, fit))
    sets = []
    for s in range(n_sets):
        ws, gi = [], 0
        for b in blocks:
            wb = []
            for lin in b.linears:
                if s == 0:
                    g = torch.Generator(device=dev)
                    g.manual_seed(gi)
                    w = torch.empty((lin.out_features, lin.in_features), dtype=b.dtype, device=dev)
                    w.normal_(0.0, 0.02, generator=g)
                else:
                    w = sets[0][len(ws)][len(wb)].clone()
                wb.append(w)
                gi += 1
            ws.append(wb)
        sets.append(ws)
    return blocks, acts, sets, n_local


def build_plans(blocks, acts, weights, n_local, world, dev, state):
    """Pre-bind every launch of one step for one weight set: per block ONE batched statistics launch
    (all distinct linear inputs), ONE batched running-mean launch, and one batched select call (the
    library issues one launch per distinct (in_features, k) for the per-row rule, one launch sequence
    for the matrix-wide rule)."""
    from vlmc import ops, workload as wl
    steps = []
    for bi, b in enumerate(blocks):
        ins = state["blocks"][bi]
        stat = ops.plan_act_sqnorm_batch(acts[bi], [s["normsq_local"] for s in ins])
        upd = ops.plan_scaler_update_batch([s["scaler"] for s in ins], 0, [s["normsq_all"] for s in ins], 1,
                                           [s["sqrt"] for s in ins])
        ws, sqs, ks, nbytes, li = [], [], [], 0, 0
        for ii, inp in enumerate(b.inputs):
            for lin in inp.linears:
                w = weights[bi][li]
                ws.append(w)
                sqs.append(ins[ii]["sqrt"])
                ks.append(int(lin.in_features * RATIO) if b.mode == "row" else int(lin.out_features * lin.in_features * RATIO))
                nbytes += wl.select_bytes(lin, w.element_size(), True)
                li += 1
        sel = ops.plan_select_batch(ws, sqs, b.mode, ks=ks, apply_zero=True, masks=state["masks"][bi],
                                    partials=state["partials"][bi])
        n_launch = 0
        if b.mode == "row":
            # csrc/wanda_select.hip: 16-bit rows of <= 2048 and of 2049..8192 columns in one call share ONE mixed launch
            widths = {w.shape[1] for w in ws}
            mixed = (os.environ.get("VLMC_SELECT_MIXED", "1") != "0" and len(ws) <= 12 and ws[0].element_size() == 2
                     and max(widths) <= 8192 and min(widths) <= 2048 < max(widths) and all(i % 8 == 0 for i in widths))
            n_launch = 1 if mixed else len({(w.shape[1], k) for w, k in zip(ws, ks)})
        sbytes = sum(wl.stat_bytes(inp, n_local, acts[bi][ii].element_size()) for ii, inp in enumerate(b.inputs))
        steps.append((stat, upd, sel, b.mode == "row", nbytes, n_launch, sbytes))
    return steps


def alloc_state(blocks, n_local, world, dev):
    from vlmc import ops
    st = {"blocks": [], "masks": [], "partials": [], "flat": []}
    for b in blocks:
        ins = []
        tot = sum(i.in_features for i in b.inputs)
        # one flat [samples, sum(in)] buffer per block: the statistics kernel writes column slices of it, the
        # multi-GPU exchange is a single all-gather of it, the running-mean kernel reads column slices of it
        flat_local = torch.empty((n_local, tot), dtype=torch.float32, device=dev)
        flat_all = torch.empty((N_CALIB, tot), dtype=torch.float32, device=dev) if world > 1 else flat_local
        off = 0
        for inp in b.inputs:
            ins.append({"scaler": torch.zeros(inp.in_features, dtype=torch.float32, device=dev),
                        "sqrt": torch.empty(inp.in_features, dtype=torch.float32, device=dev),
                        "normsq_local": flat_local[:, off:off + inp.in_features],
                        "normsq_all": flat_all[:, off:off + inp.in_features]})
            off += inp.in_features
        st["blocks"].append(ins)
        st["flat"].append((flat_local, flat_all))
        st["masks"].append([torch.empty((l.out_features, l.in_features), dtype=torch.bool, device=dev) for l in b.linears])
        st["partials"].append([torch.empty(ops.select_partials(b.mode, l.out_features, l.in_features), dtype=torch.float64,
                                           device=dev) for l in b.linears])
    return st


class HipEvents:
    """HIP events of the runtime this process has loaded (ctypes on libamdhip64), handed to the kernel launch itself
    (vlmc_set_launch_events -> hipExtLaunchKernel): start / stop are the kernel's own begin / end on its stream.
    An hipEventRecord between two kernels (torch.cuda.Event.record) costs ~5 us of idle GPU per record on MI355X --
    270 records per step were 11 % of the step; this form costs nothing."""

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


def run_step(plans, state, world, events=None, hipev=None, set_events=None, phase=0, stride=1):
    """events = {"stat": [...], "rows": [...]} collects (start, stop, algorithmic bytes, launches): HIP events carried by
    the statistics launch and by the per-row select launch of a block, on the launch stream.  A launch that carries
    events is preceded and followed by ~5 us of idle GPU (its completion signal is waited for), so only every
    `stride`-th block of a step is timed, rotating with the step index: every launch is covered once per `stride` steps."""
    for bi, (stat, upd, sel, is_row, nbytes, n_launch, sbytes) in enumerate(plans):
        timed = events is not None and (bi + phase) % stride == 0
        if timed:
            a, b = hipev.new(), hipev.new()
            set_events(a, b)
            stat()
            events["stat"].append((a, b, sbytes, 1))
        else:
            stat()
        if world > 1:          # ONE all-gather per block: [n_local, sum(in)] -> [128, sum(in)] in sample order
            flat_local, flat_all = state["flat"][bi]
            dist.all_gather_into_tensor(flat_all, flat_local)
        upd()
        if timed and is_row and n_launch == 1:
            a, b = hipev.new(), hipev.new()
            set_events(a, b)
            sel()
            events["rows"].append((a, b, nbytes, n_launch))
        else:
            sel()


def cpu_baseline(blocks, budget_s):
    """Time the oracle (the reference's PyTorch-CPU op sequence, oracle/wanda_torch.py) on a
    bounded sample of the same workload: whole blocks taken round-robin from the three towers
    until the time budget is used.  Checker code timed as a baseline, never shipped."""
    from oracle import wanda_torch as OT
    order = [0, 39, 63]                      # first ViT block, first encoder block, first decoder block
    order += [1, 40, 64, 2, 41, 65]
    t0 = time.perf_counter()
    done, names = 0, []
    for bi in order:
        b = blocks[bi]
        g = torch.Generator().manual_seed(bi)
        for inp in b.inputs:
            st = OT.WandaStat(inp.in_features)
            for j in range(N_CALIB):
                x = (torch.randn((1, inp.tokens, inp.in_features), generator=g) + 0.1).to(b.dtype)
                st.add_batch(x)
            for lin in inp.linears:
                w = (torch.randn((lin.out_features, lin.in_features), generator=g) * 0.02).to(b.dtype)
                OT.prune_linear(w, st.scaler_row, b.mode, ratio=RATIO, apply_zero=True)
                done += 1
        names.append(b.name)
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    return {"value": done / dt, "unit": "layers/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{done} linears of {len(names)} blocks ({', '.join(names)}), {N_CALIB} calib samples each, "
                      f"synthetic data generation included, {dt:.1f} s"}


def end_to_end(dev, world):
    """Whole-prune wall-clock through the drop-in pruner API (capture of the towers' inputs by the model's own forward,
    block replay, statistics, score/select/apply of all 588 linears) on a random-init model of the true shapes.
    Reported beside the timed hot path, never part of `value`."""
    from vlmc import synthetic
    out = {"model": "synthetic InstructBLIP-FlanT5-XL shapes (39 ViT-g fp16 + 24/24 Flan-T5-XL bf16 blocks, Q-Former replaced "
                    "by its 32 query tokens), random init", "pruner": "blipt5_wanda_pruner", "calib_samples": N_CALIB,
           "calib_sharded_over": world}
    model = None
    for key, env in (("seconds", {}), ("seconds_batched32", {"VLMC_BATCH_REPLAY": "32"})):
        saved = {k: os.environ.get(k) for k in ("VLMC_BATCH_REPLAY", "VLMC_GRAPH_REPLAY")}
        os.environ.update(env)
        try:
            dt, model, info = synthetic.time_prune(dev, "blipt5_wanda_pruner", n_samples=N_CALIB, ratio=RATIO, model=model)
        finally:
            for k, v in saved.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        out[key] = round(dt, 3)
        out["layers_per_s" + key[len("seconds"):]] = round(info["linears"] / dt, 1)
        out["pruned_fraction"] = round(info["pruned_fraction"], 6)
    out["replay"] = {"seconds": "per-sample block forwards from HIP graphs (bit-identical to the reference's loop)",
                     "seconds_batched32": "VLMC_BATCH_REPLAY=32 (32 samples per block forward; masks agree up to near-ties)"}
    return out


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback of the product path)")
    # VLMC_BENCH_ONE_DEVICE=1 (testing the N > 1 code path on a 1-GPU box): every rank uses cuda:0 and the
    # collective runs over gloo -- never set by the driver
    one_device = os.environ.get("VLMC_BENCH_ONE_DEVICE", "0") == "1"
    if one_device:
        local_rank = 0
        # several processes on one device contend for the CUs the fused matrix-wide select keeps to itself (its barriers
        # would time out and the exact fallback run): the rehearsal uses the four-launch form
        os.environ.setdefault("VLMC_MATRIX_FUSED", "0")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if one_device:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    assert N_CALIB % world == 0

    from vlmc import _lib
    _lib.load()                                   # fail loudly if the HIP library is missing

    want_sets = args.weight_sets or (args.steps + args.warmup)
    blocks, acts, sets, n_local = build_workload(dev, rank, world, want_sets)
    state = alloc_state(blocks, n_local, world, dev)
    plans = [build_plans(blocks, acts, w, n_local, world, dev, state) for w in sets]
    n_lin = sum(len(b.linears) for b in blocks)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        run_step(plans[i % len(plans)], state, world)
    sync()
    events = {"stat": [], "rows": []}
    hipev = HipEvents()
    set_events = _lib.load().vlmc_set_launch_events
    t0 = time.perf_counter()
    for i in range(args.steps):
        run_step(plans[(args.warmup + i) % len(plans)], state, world, events, hipev, set_events, phase=i,
                 stride=max(1, args.event_stride))
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- roofline of the dominant kernel (largest share of the step's GPU time): the activation
    # ---- statistics kernel; the per-row score+select kernel is reported beside it ----------------
    traffic = {}
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath))
        except Exception:
            traffic = {}

    def roof(kind, kernel, tkey):
        evs = events[kind]
        tot_ms = sum(hipev.elapsed_ms(a, b) for a, b, _, _ in evs)
        tot_bytes = sum(nb for _, _, nb, _ in evs)
        n_launches = sum(nl for _, _, _, nl in evs)
        ach = tot_bytes / (tot_ms * 1e-3) / 1e9 if tot_ms > 0 else 0.0
        return {"bound": "hbm", "kernel": kernel, "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic.get(tkey), "launches": n_launches,
                "avg_launch_us": round(tot_ms * 1e3 / max(1, n_launches), 2),
                "timed": f"HIP events carried by the launch (hipExtLaunchKernel) on every {max(1, args.event_stride)}-th block "
                         f"of a step, rotating with the step index",
                "bytes_per_launch": round(tot_bytes / max(1, n_launches))}

    roofline = roof("stat", "vlmc::act_sqnorm_kernel (per-sample squared column norms of every distinct linear input "
                            "of a block, one launch per block)", "act_sqnorm_kernel_bytes_per_launch")
    roofline["other"] = [roof("rows", "vlmc::select_rows_mixed_kernel (score+select+apply, per-row rule; all linears of a T5 "
                                      "block in one launch)", "select_rows_mixed_kernel_bytes_per_launch")]

    e2e, n_sets = None, len(sets)
    if args.e2e == "1" or (args.e2e == "auto" and world == 1):
        n_sets = len(sets)
        del plans, sets, acts, state
        torch.cuda.empty_cache()
        try:
            e2e = end_to_end(dev, world)
        except Exception as e:                    # never lose the bench line to the side measurement
            e2e = {"error": f"{type(e).__name__}: {e}"}

    out = None
    if rank == 0:
        out = {
            "metric": "layers/sec, Wanda@50% unstructured, InstructBLIP-FlanT5-XL (588 linears, 128 calib samples)",
            "value": round(n_lin * args.steps / elapsed, 1), "unit": "layers/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "configs[1]: Wanda 50% unstructured, full InstructBLIP-FlanT5-XL shapes "
                                   "(39 ViT-g blocks fp16 matrix-wide rule + 24/24 T5 blocks bf16 per-row rule), "
                                   "128 calib samples, tokens 257/64/16; statistics + score/select/apply of every linear",
                       "linears": n_lin, "blocks": len(blocks), "calib_samples": N_CALIB, "ratio": RATIO,
                       "weight_sets": n_sets, "total_prune_wall_clock_s": round(elapsed / args.steps, 5),
                       "parallelism": f"calib-dp{world}"},
            "roofline": roofline,
            "end_to_end": e2e,
        }
        if world == 1 and args.cpu_seconds > 0:
            out["cpu_baseline"] = cpu_baseline(blocks, args.cpu_seconds)
        else:
            out["cpu_baseline"] = None
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
