"""GPU end-to-end: the drop-in `blipt5_*_pruner`s running on the HIP kernels (Wanda first).

(a) every statistics launch and every per-linear select that the pruner issues is
    re-checked bit-for-bit against the CPU oracle on the very tensors it saw (the GPU
    forward of the toy blocks differs from the CPU forward in the last bits, so the
    comparison uses the captured GPU activations);
(b) the final masks agree with the REFERENCE's whole-pruner golden run (CPU forward) up to
    the few near-tie flips that those last-bit activation differences allow."""
import numpy as np
import pytest
import torch

import pruner_helpers as H
from oracle import wanda as OW

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("replay", ["grouped", "per_sample"])
@pytest.mark.parametrize("name", list(H.VARIANTS))
def test_pruner_on_gpu_every_launch_matches_oracle(name, replay, monkeypatch):
    from vlmc import ops, wanda
    monkeypatch.setenv("VLMC_BATCH_REPLAY", "128" if replay == "grouped" else "1")
    real_sq, real_prune = ops.act_sqnorm_batch, wanda.prune_block
    counts = {"sqnorm_launches": 0, "sqnorm_inputs": 0, "select": 0, "blocks": 0}

    def checked_sqnorm(xs, outs=None):
        rows = real_sq(xs, outs)
        for x, r in zip(xs, rows):
            got = r.cpu().numpy()
            for c in range(x.shape[0]):
                want = OW.act_sqnorm(x[c].cpu())
                assert np.array_equal(got[c].view(np.uint32), want.view(np.uint32))
        counts["sqnorm_launches"] += 1
        counts["sqnorm_inputs"] += len(xs)
        return rows

    def checked_block(weights, stats, mode, *, ratios=None, n=0, m=0, apply_zero=True, partials=None):
        W0 = [w.detach().clone().cpu() for w in weights]
        masks = real_prune(weights, stats, mode, ratios=ratios, n=n, m=m, apply_zero=apply_zero, partials=partials)
        for w0, w, st, mk, ratio in zip(W0, weights, stats, masks, ratios):
            want = OW.prune_linear(w0, st.scaler_row.cpu().numpy(), mode, ratio=ratio, n=n, m=m, apply_zero=apply_zero)
            assert np.array_equal(mk.cpu().numpy(), want["mask"]), f"{mode} mask differs from oracle"
            assert torch.equal(w.detach().cpu(), want["weight"])
            counts["select"] += 1
        counts["blocks"] += 1
        return masks

    monkeypatch.setattr(ops, "act_sqnorm_batch", checked_sqnorm)
    monkeypatch.setattr(wanda, "prune_block", checked_block)
    pruned, _ = H.run_pruner(name, "cuda:0")
    assert counts["select"] == 2 * 4 + 2 * 7 + 2 * 11 and counts["blocks"] == 6
    # shared inputs are reduced once: per sample 4 (ViT) / 4 (enc) / 7 (dec) distinct tensors, not 4 / 7 / 11,
    # and all distinct inputs of one sample's block forward go in one launch
    # (grouped replay, the default: the 6 equal-shape samples of a block pass are ONE forward and ONE statistics launch
    # that still yields one row per sample)
    per_pass = 1 if replay == "grouped" else 6
    assert counts["sqnorm_inputs"] == per_pass * (2 * 4 + 2 * 4 + 2 * 7)
    assert counts["sqnorm_launches"] == per_pass * 6
    st = H.compare_with_golden(name, pruned, exact=False, min_mask_agreement=1.0)
    print(name, st)
    for mn, mod in pruned.named_modules():
        if hasattr(mod, "mask") and hasattr(mod, "weight"):
            assert mod.mask.is_cuda and mod.mask.dtype == torch.bool
            if not H.VARIANTS[name]["lora"]:
                assert bool((mod.weight.data[~mod.mask] == 0).all())


def test_pruner_is_deterministic_on_gpu():
    a, _ = H.run_pruner("fp32_r50", "cuda:0")
    b, _ = H.run_pruner("fp32_r50", "cuda:0")
    for (ka, va), (kb, vb) in zip(a.state_dict().items(), b.state_dict().items()):
        assert ka == kb and torch.equal(va, vb), ka


@pytest.mark.parametrize("name", ["fp32_u50", "fp32_2_4"])
def test_sparsegpt_pruner_on_gpu_tracks_reference_run(name):
    """Whole blipt5_sparsegpt_pruner on the GPU (library Cholesky/GEMMs + fused sweep kernel) vs the
    reference's CPU run: same zero pattern up to near-ties, weights close, 2:4 structure exact."""
    import golden_io
    import test_pruner_host_logic as T
    E = golden_io.load("sparsegpt_e2e")
    pruned, _ = T._run_sparsegpt_pruner(name, "cuda:0")
    got = pruned.state_dict()
    tot = agree = 0
    num = den = 0.0
    for key in [k for k in E if k.startswith(f"{name}/sd/")]:
        k = key[len(name) + 4:]
        ref = E[key]
        g = got[k].cpu()
        if ref.dim() != 2 or ".block" not in k or "shared" in k:
            continue
        same = (g == 0) == (ref == 0)
        tot += same.numel()
        agree += int(same.sum())
        clean = same.all(dim=1)
        num += float((g[clean] - ref[clean]).pow(2).sum())
        den += float(ref[clean].pow(2).sum())
        if name == "fp32_2_4" and ("blocks." in k or "block." in k) and g.shape[1] % 4 == 0 and (g == 0).any():
            assert bool(((g == 0).view(g.shape[0], -1, 4).sum(-1) >= 2).all()), k
    assert tot > 0 and agree / tot >= 0.97, agree / tot
    assert (num / den) ** 0.5 < 2e-2


def test_sparsegpt_sweeps_on_side_streams_change_nothing(monkeypatch):
    """n:m mode: the column sweeps of a block's independent linears run on streams of their own (sparsegpt_pruner.py:430-446
    prunes them one after the other): the same kernels on the same data, so the pruned model equals the one-after-the-other
    run bit for bit -- weights and importance scores."""
    import test_pruner_host_logic as T
    monkeypatch.setenv("VLMC_SGPT_SWEEP_STREAMS", "1")
    a, _ = T._run_sparsegpt_pruner("fp32_2_4", "cuda:0")
    monkeypatch.setenv("VLMC_SGPT_SWEEP_STREAMS", "4")
    b, _ = T._run_sparsegpt_pruner("fp32_2_4", "cuda:0")
    sa, sb = a.state_dict(), b.state_dict()
    assert sa.keys() == sb.keys()
    for k in sa:
        assert torch.equal(sa[k], sb[k]), k
    ia = {n: m.weight.importance_score for n, m in a.named_modules() if hasattr(getattr(m, "weight", None), "importance_score")}
    ib = {n: m.weight.importance_score for n, m in b.named_modules() if hasattr(getattr(m, "weight", None), "importance_score")}
    assert ia and ia == ib


@pytest.mark.parametrize("name", list(H.DSNOT_VARIANTS))
def test_dsnot_pruner_on_gpu_every_linear_matches_oracle(name, monkeypatch):
    """Whole blipt5_dsnot_pruner on the GPU: every per-linear refinement is re-checked bit-for-bit against
    the CPU oracle fed with the statistics the GPU produced; the final masks track the reference's run."""
    from types import SimpleNamespace
    from oracle import dsnot as OD
    from vlmc import dsnot
    real = dsnot.prune_linear
    counts = {"linears": 0, "moments": 0}
    real_moments, real_moments_calls = dsnot.act_moments, dsnot.act_moments_calls

    def counted(x):
        counts["moments"] += 1
        return real_moments(x)

    def counted_calls(x, calls):
        counts["moments"] += 1
        return real_moments_calls(x, calls)

    def checked(weight, stat, ratio, **kw):
        W0 = weight.detach().clone().cpu()
        ostat = SimpleNamespace(scaler_row=stat.scaler_row.cpu(), sum_metric_row=stat.sum_row.cpu(),
                                var=stat.var_row.cpu().reshape(-1, 1))
        keep = real(weight, stat, ratio, **kw)
        okw = {k: kw[k] for k in ("initial_method", "max_cycle_time", "update_threshold", "pow_of_var_regrowing")}
        if kw["prune_n"]:
            pruned = OD.prune_nm(W0, ostat, kw["prune_n"], kw["prune_m"], **okw)
        else:
            pruned = OD.prune_unstructured(W0, ostat, ratio, without_DSnoT=kw["without_DSnoT"],
                                           without_same_sign=kw["without_same_sign"], **okw)
        assert torch.equal(keep.cpu(), ~pruned), "refined mask differs from the oracle"
        want_w = W0.clone()
        if kw["apply_zero"]:
            want_w[pruned] = 0
        assert torch.equal(weight.detach().cpu(), want_w)
        counts["linears"] += 1
        return keep

    monkeypatch.setattr(dsnot, "prune_linear", checked)
    monkeypatch.setattr(dsnot, "act_moments", counted)
    monkeypatch.setattr(dsnot, "act_moments_calls", counted_calls)
    pruned, _ = H.run_dsnot_pruner(name, "cuda:0")
    assert counts["linears"] == 2 * 4 + 2 * 7 + 2 * 11
    assert counts["moments"] == 2 * 4 + 2 * 4 + 2 * 7              # one launch per distinct input tensor, all 6 samples in it
    st = H.compare_with_golden(name, pruned, exact=False, min_mask_agreement=1.0, which="dsnot_e2e")
    print(name, st)


@pytest.mark.parametrize("method", ["wanda", "dsnot"])
def test_batched_replay_on_gpu_tracks_reference_run(method, monkeypatch):
    """VLMC_BATCH_REPLAY=4: 4 samples per block forward, per-sample statistics kept; the final masks
    track the reference's per-sample golden run (activations differ in the last bits only)."""
    monkeypatch.setenv("VLMC_BATCH_REPLAY", "4")
    if method == "wanda":
        from vlmc import ops
        real_sq = ops.act_sqnorm_batch
        seen = {"calls": 0}

        def counting(xs, outs=None):
            seen["calls"] += sum(x.shape[0] for x in xs)
            return real_sq(xs, outs)
        monkeypatch.setattr(ops, "act_sqnorm_batch", counting)
        pruned, _ = H.run_pruner("fp32_r50", "cuda:0")
        assert seen["calls"] == 6 * (2 * 4 + 2 * 4 + 2 * 7)          # still one statistics row per sample and distinct input
        st = H.compare_with_golden("fp32_r50", pruned, exact=False, min_mask_agreement=1.0)
    else:
        pruned, _ = H.run_dsnot_pruner("fp32_r50", "cuda:0")
        st = H.compare_with_golden("fp32_r50", pruned, exact=False, min_mask_agreement=1.0, which="dsnot_e2e")
    print(method, st)


def test_replay_modes_agree_with_the_reference_side_by_side(monkeypatch):
    """The reference's whole-pruner golden runs (CPU forward) against (a) the per-sample loop replayed from HIP graphs
    and (b) the grouped replay that is the default: neither can be bit-identical to a CPU forward, both are held to the
    same bar, and the achieved agreement of each is RECORDED (stdout and gpurun_out/replay_agreement.json) next to the
    agreement of the two modes with each other."""
    import json
    import os
    rows = []

    def masks_of(model):
        return {n: m.mask.clone() for n, m in model.named_modules() if hasattr(m, "mask") and torch.is_tensor(m.mask)}

    cases = [("wanda", n, H.run_pruner, "wanda_e2e") for n in H.VARIANTS] + \
            [("dsnot", n, H.run_dsnot_pruner, "dsnot_e2e") for n in H.DSNOT_VARIANTS]
    for method, name, run, which in cases:
        got = {}
        for mode, group in (("per_sample_graph", "1"), ("grouped", "128")):
            monkeypatch.setenv("VLMC_BATCH_REPLAY", group)
            pruned, _ = run(name, "cuda:0")
            st = H.compare_with_golden(name, pruned, exact=False, min_mask_agreement=1.0, which=which)
            got[mode] = (1 - st["mask_diff"] / st["mask_elems"], masks_of(pruned))
        a, b = got["per_sample_graph"][1], got["grouped"][1]
        tot = sum(m.numel() for m in a.values())
        between = 1 - sum(int((a[k] != b[k]).sum()) for k in a) / tot
        rows.append({"method": method, "variant": name, "mask_elements": tot,
                     "agreement_per_sample_graph_vs_reference": got["per_sample_graph"][0],
                     "agreement_grouped_vs_reference": got["grouped"][0], "agreement_between_modes": between})
        print(rows[-1])
        assert between >= 0.99
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "replay_agreement.json"), "w") as f:
        json.dump(rows, f, indent=1)


@pytest.mark.parametrize("method", ["wanda", "wanda_lora_mixed", "dsnot", "sparsegpt"])
def test_graph_captured_replay_is_bit_identical_to_the_eager_loop(method, monkeypatch):
    """The block forwards of the calibration replay run from HIP graphs by default (same kernels, static buffers,
    statistics hooks fired on the captured tensors after each replay): every weight and mask equals the eager loop's."""
    from lavis.compression.pruners import calibration as cal

    def run():
        if method == "wanda":
            return H.run_pruner("fp32_r50", "cuda:0")[0]
        if method == "wanda_lora_mixed":
            return H.run_pruner("fp32_r40_lora", "cuda:0")[0], H.run_pruner("mixed_2_4", "cuda:0")[0]
        if method == "dsnot":
            return H.run_dsnot_pruner("fp32_r50", "cuda:0")[0]
        import test_pruner_host_logic as T
        return T._run_sparsegpt_pruner("fp32_u50", "cuda:0")[0]

    def states(res):
        models = res if isinstance(res, tuple) else (res,)
        out = []
        for m in models:
            sd = dict(m.state_dict())
            for n, mod in m.named_modules():
                if hasattr(mod, "mask") and torch.is_tensor(mod.mask):
                    sd[n + ".mask*"] = mod.mask
            out.append(sd)
        return out

    monkeypatch.setenv("VLMC_BATCH_REPLAY", "1")                           # the per-sample loop is what the graphs replay
    monkeypatch.setenv("VLMC_GRAPH_REPLAY", "0")
    before = dict(cal.graph_stats)
    eager = states(run())
    assert cal.graph_stats == before                                     # nothing captured when switched off
    monkeypatch.setenv("VLMC_GRAPH_REPLAY", "1")
    graphed = states(run())
    assert cal.graph_stats["captured"] > before["captured"] and cal.graph_stats["replayed"] > before["replayed"]
    assert cal.graph_stats["fallbacks"] == before["fallbacks"]
    for a, b in zip(eager, graphed):
        assert a.keys() == b.keys()
        for k in a:
            assert torch.equal(a[k], b[k]), k


def test_tower_memo_hands_over_recorded_outputs_only_when_inputs_and_weights_are_unchanged(monkeypatch):
    """calibration.TowerMemo: phase 1 records (first block's inputs, last block's output) per forward; phase 2 returns
    the record when the inputs are bit-equal, runs the blocks otherwise; changed weights start a new record."""
    from lavis.compression.pruners import calibration as cal
    monkeypatch.setenv("VLMC_TOWER_MEMO", "1")
    monkeypatch.setenv("VLMC_GRAPH_REPLAY", "1")

    class Tower(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.blocks = torch.nn.ModuleList([torch.nn.Sequential(torch.nn.Linear(32, 32), torch.nn.GELU()) for _ in range(3)])

        def forward(self, x, scale=1.0):
            for blk in self.blocks:
                x = blk(x)
            return x * scale

    torch.manual_seed(0)
    model = Tower().to("cuda:0").eval()
    xs = [torch.randn(2, 5, 32, device="cuda:0") for _ in range(5)]
    with torch.no_grad():
        want = [model(x) for x in xs]
        cache = {}

        def phase(inputs):
            undo = cal._wrap_towers(model, ["blocks"], cache)
            try:
                return [model(x) for x in inputs]
            finally:
                for blocks, i, orig in undo:
                    blocks[i].__dict__["_memo"] = None
                    blocks[i] = orig

        s0 = dict(cal.graph_stats)
        got = phase(xs)                                                    # records
        assert all(torch.equal(a, b) for a, b in zip(got, want))
        assert cal.graph_stats["memo_recorded"] == s0["memo_recorded"] + 5 and cal.graph_stats["memo_hits"] == s0["memo_hits"]
        other = torch.randn(2, 5, 32, device="cuda:0")
        got = phase(xs[:3] + [other, xs[4]])                               # replays 4, runs the changed forward
        assert all(torch.equal(a, b) for a, b in zip(got, want[:3] + [model(other), want[4]]))
        assert cal.graph_stats["memo_hits"] == s0["memo_hits"] + 4 and cal.graph_stats["memo_misses"] == s0["memo_misses"] + 1
        model.blocks[1][0].weight.mul_(1.5)                                # same addresses, other values
        want2 = [model(x) for x in xs]
        got = phase(xs)                                                    # fingerprint differs: a new record, no hits
        assert all(torch.equal(a, b) for a, b in zip(got, want2))
        assert cal.graph_stats["memo_hits"] == s0["memo_hits"] + 4
        got = phase(xs)
        assert all(torch.equal(a, b) for a, b in zip(got, want2)) and cal.graph_stats["memo_hits"] == s0["memo_hits"] + 9


def test_tower_memo_in_a_whole_prune_is_bit_identical(monkeypatch):
    """Decoder capture of the toy InstructBLIP with the ViT's outputs taken from the encoder capture's record."""
    monkeypatch.setenv("VLMC_LINEAR_F32", "0")      # (the toy is fp32: this test is about the graph / memo route of towers that are NOT stacked)
    from lavis.compression.pruners import calibration as cal

    def state(model):
        sd = dict(model.state_dict())
        for n, mod in model.named_modules():
            if hasattr(mod, "mask") and torch.is_tensor(mod.mask):
                sd[n + ".mask*"] = mod.mask
        return sd

    monkeypatch.setenv("VLMC_GRAPH_REPLAY", "1")
    monkeypatch.setenv("VLMC_TOWER_MEMO", "0")
    s0 = dict(cal.graph_stats)
    off = state(H.run_pruner("fp32_r50", "cuda:0")[0])
    assert cal.graph_stats["memo_hits"] == s0["memo_hits"] and cal.graph_stats["memo_recorded"] == s0["memo_recorded"]
    monkeypatch.setenv("VLMC_TOWER_MEMO", "1")
    on = state(H.run_pruner("fp32_r50", "cuda:0")[0])
    assert cal.graph_stats["memo_hits"] > s0["memo_hits"] and cal.graph_stats["memo_misses"] == s0["memo_misses"]
    assert off.keys() == on.keys() and all(torch.equal(off[k], on[k]) for k in off)


def test_capture_phase_runs_again_when_a_remembered_input_was_not_the_one_fed(monkeypatch):
    """The comparisons of remembered inputs are answered at the end of a capture phase (`_LaterEqual`); when one of them
    fails -- here the record of forward 0 is tampered with -- the records are dropped and the phase runs again comparing at
    once: same pruned model as without any memo, and `later_failed` counts the event."""
    monkeypatch.setenv("VLMC_LINEAR_F32", "0")      # (the toy is fp32: this test is about the graph / memo route of towers that are NOT stacked)
    from lavis.compression.pruners import calibration as cal

    def state(model):
        sd = dict(model.state_dict())
        for n, mod in model.named_modules():
            if hasattr(mod, "mask") and torch.is_tensor(mod.mask):
                sd[n + ".mask*"] = mod.mask
        return sd

    monkeypatch.setenv("VLMC_LATER_EQUAL", "1")
    monkeypatch.setenv("VLMC_TOWER_MEMO", "0")
    want = state(H.run_pruner("fp32_r50", "cuda:0")[0])
    monkeypatch.setenv("VLMC_TOWER_MEMO", "1")
    real_begin = cal.TowerMemo.begin
    tampered = []

    def begin(self, mode):
        real_begin(self, mode)
        if mode == "replay" and self.entries and not tampered:
            first = self.entries[min(self.entries)]
            first[0][0][0] = first[0][0][0] + 1         # the remembered block-0 input of forward 0 is another tensor than the one fed
            tampered.append(self)
    monkeypatch.setattr(cal.TowerMemo, "begin", begin)
    before = cal.graph_stats.get("later_failed", 0)
    got = state(H.run_pruner("fp32_r50", "cuda:0")[0])
    assert tampered and cal.graph_stats.get("later_failed", 0) == before + 1
    assert want.keys() == got.keys() and all(torch.equal(want[k], got[k]) for k in want)
    # and with the answers taken at once there is nothing to run again: plain misses for that forward (once per sweep)
    tampered.clear()
    monkeypatch.setenv("VLMC_LATER_EQUAL", "0")
    m0 = cal.graph_stats["memo_misses"]
    got = state(H.run_pruner("fp32_r50", "cuda:0")[0])
    assert tampered and cal.graph_stats.get("later_failed", 0) == before + 1 and cal.graph_stats["memo_misses"] > m0
    assert all(torch.equal(want[k], got[k]) for k in want)
    # the memo keeps the tensors themselves, not copies (TowerMemo.keep): one that was written into since is refused at once
    # (not tampered with here through a whole prune: the toy's block-0 input IS the calibration image, and writing into the record
    # would write into the data)
    t = torch.randn(4, 8, device="cuda:0")
    kept = cal.TowerMemo.keep(t[1:3])
    rec = cal.TowerMemo._snapshot([t], {"m": t[0]})
    assert kept.data_ptr() == t[1:3].data_ptr() and cal.TowerMemo.fresh(kept) and cal.TowerMemo._same(rec, [t.clone()], {"m": t[0].clone()})
    t[0].add_(1)                                        # (a view shares its base's version counter)
    assert not cal.TowerMemo.fresh(kept) and not cal.TowerMemo._same(rec, [t.clone()], {"m": t[0].clone()})
    monkeypatch.setenv("VLMC_MEMO_COPY", "1")
    kept = cal.TowerMemo.keep(t)
    t.add_(1)
    assert cal.TowerMemo.fresh(kept) and kept.data_ptr() != t.data_ptr()


def test_graphed_module_proxy_replays_identically_and_falls_back(monkeypatch):
    """calibration.GraphedModule (stand-in for already pruned blocks during capture): eager first call, captured second,
    replayed afterwards, one graph per argument signature; gradients / odd arguments go straight to the module."""
    monkeypatch.setenv("VLMC_LINEAR_F32", "0")      # (the toy is fp32: with its linears on the fp32 kernel the finished tower is stacked, not graphed)
    import toy_models
    from lavis.compression.pruners import calibration as cal
    torch.manual_seed(0)
    blk = toy_models.ToyT5Block(32, 64, is_decoder=True).to("cuda:0").eval()
    proxy = cal.GraphedModule(blk)
    assert proxy.is_decoder is True and proxy.ln0 is blk.ln0              # attribute access reaches the block
    before = dict(cal.graph_stats)
    xs = [torch.randn(1, 5, 32, device="cuda:0") for _ in range(5)]
    enc = [torch.randn(1, 7, 32, device="cuda:0") for _ in range(5)]
    with torch.no_grad():
        for x, e in zip(xs, enc):
            got = proxy(x, attention_mask=None, encoder_hidden_states=e, dense=False)
            want = blk(x, attention_mask=None, encoder_hidden_states=e, dense=False)
            assert isinstance(got, tuple) and torch.equal(got[0], want[0])
        assert cal.graph_stats["captured"] == before["captured"] + 1 and cal.graph_stats["replayed"] == before["replayed"] + 4
        y = proxy(torch.randn(2, 5, 32, device="cuda:0"), attention_mask=None, encoder_hidden_states=torch.randn(2, 7, 32, device="cuda:0"))
        assert y[0].shape == (2, 5, 32) and cal.graph_stats["captured"] == before["captured"] + 1       # new signature: eager first
        odd = proxy(xs[0], attention_mask=None, encoder_hidden_states=enc[0], dense=False, unused_list=[1, 2])   # not graphable
        assert torch.equal(odd[0], blk(xs[0], encoder_hidden_states=enc[0])[0])
    x = xs[0].clone().requires_grad_()
    out = proxy(x, encoder_hidden_states=enc[0])[0]                         # gradients enabled: the module itself
    out.sum().backward()
    assert x.grad is not None
    # capture_block_inputs puts the proxies in and takes them out again
    model = toy_models.init_toy(toy_models.ToyBlipT5(), seed=7).eval().to("cuda:0")
    batches = [{k: t.to("cuda:0") for k, t in b.items()} for b in toy_models.make_batches(6, seed=11)]
    vit_blocks = list(model.visual_encoder.blocks)
    with torch.no_grad():
        inps, _, _ = cal.capture_block_inputs(model, batches, 6, "t5_model.encoder.block", lambda m, b, _l=False: m(b), False,
                                              vit=False, model_prefix="t5_model", done_towers=["visual_encoder.blocks"])
        ref, _, _ = cal.capture_block_inputs(model, batches, 6, "t5_model.encoder.block", lambda m, b, _l=False: m(b), False,
                                             vit=False, model_prefix="t5_model")
    assert all(a is b for a, b in zip(model.visual_encoder.blocks, vit_blocks))
    assert len(inps) == 6 and all(torch.equal(a, b) for a, b in zip(inps, ref))


def test_blocks_that_cannot_be_captured_fall_back_to_the_eager_loop(monkeypatch):
    """A block whose forward synchronises with the host (HF T5's fp16 `torch.isinf(h).any()` clamp does) cannot be
    captured: the replay engine and the capture proxies must notice, run it eagerly and produce the same result."""
    import toy_models
    from lavis.compression.pruners import calibration as cal
    orig = toy_models.ToyT5Block.forward

    def syncing_forward(self, hidden_states, **kw):
        if bool(torch.isinf(hidden_states).any()):            # host sync: illegal while a stream is capturing
            hidden_states = torch.clamp(hidden_states, -1e4, 1e4)
        return orig(self, hidden_states, **kw)
    monkeypatch.setenv("VLMC_BATCH_REPLAY", "1")
    monkeypatch.setenv("VLMC_GRAPH_REPLAY", "0")
    monkeypatch.setattr(toy_models.ToyT5Block, "forward", syncing_forward)
    eager, _ = H.run_pruner("fp32_r50", "cuda:0")
    monkeypatch.setenv("VLMC_GRAPH_REPLAY", "1")
    before = dict(cal.graph_stats)
    graphed, _ = H.run_pruner("fp32_r50", "cuda:0")
    assert cal.graph_stats["fallbacks"] > before["fallbacks"]                 # the T5 blocks were refused ...
    assert cal.graph_stats["captured"] > before["captured"]                   # ... the ViT blocks were not
    for (ka, va), (kb, vb) in zip(eager.state_dict().items(), graphed.state_dict().items()):
        assert ka == kb and torch.equal(va, vb), ka


def _capture_decoder_inputs(model, batches, **env):
    """Inputs of the T5 decoder's first block as captured while the (finished) ViT and encoder towers are run through."""
    from lavis.compression.pruners import calibration as cal
    cache = {}
    with torch.no_grad():
        inps, _, caches = cal.capture_block_inputs(model, batches, len(batches), "t5_model.decoder.block",
                                                   lambda m, b, _l=False: m(b), False, vit=False, model_prefix="t5_model",
                                                   done_towers=["visual_encoder.blocks", "t5_model.encoder.block"], proxy_cache=cache)
    torch.cuda.synchronize()
    return inps, caches, cache


@pytest.mark.parametrize("streams", ["1", "3"])
def test_tower_graph_gives_the_eager_forward_bit_for_bit(streams, monkeypatch):
    """calibration.TowerGraph: the finished towers run as ONE graph per calibration forward (traced twice, then captured per
    stream slot); the captured decoder inputs equal those of the plain eager forward, on one stream and on three."""
    monkeypatch.setenv("VLMC_LINEAR_F32", "0")      # (the toy is fp32: this test is about the graph / memo route of towers that are NOT stacked)
    import toy_models
    from lavis.compression.pruners import calibration as cal
    model = toy_models.init_toy(toy_models.ToyBlipT5(), seed=7).eval().to("cuda:0")
    batches = [{k: t.to("cuda:0") for k, t in b.items()} for b in toy_models.make_batches(12, seed=11)]
    monkeypatch.setenv("VLMC_GRAPH_REPLAY", "0")
    want, want_c, _ = _capture_decoder_inputs(model, batches)
    monkeypatch.setenv("VLMC_GRAPH_REPLAY", "1")
    monkeypatch.setenv("VLMC_TOWER_MEMO", "0")
    monkeypatch.setenv("VLMC_CAPTURE_STREAMS", streams)
    before = dict(cal.graph_stats)
    got, got_c, cache = _capture_decoder_inputs(model, batches)
    assert cal.graph_stats.get("tower_graphs", 0) - before.get("tower_graphs", 0) == 2 * int(streams)     # two towers x slots
    assert cal.graph_stats["fallbacks"] == before["fallbacks"]
    assert len(got) == len(want) == 12
    for a, b, ca, cb in zip(got, want, got_c, want_c):
        assert torch.equal(a, b)
        assert torch.equal(ca["encoder_hidden_states"], cb["encoder_hidden_states"])
    # per-block graphs instead (`VLMC_TOWER_GRAPH=0`): the same again
    monkeypatch.setenv("VLMC_TOWER_GRAPH", "0")
    got2, _, _ = _capture_decoder_inputs(model, batches)
    assert all(torch.equal(a, b) for a, b in zip(got2, want))


def test_tower_graph_falls_back_when_the_model_passes_on_something_else(monkeypatch):
    from lavis.compression.pruners import calibration as cal
    monkeypatch.setenv("VLMC_GRAPH_REPLAY", "1")

    class Tower(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.blocks = torch.nn.ModuleList([torch.nn.Sequential(torch.nn.Linear(32, 32), torch.nn.GELU()) for _ in range(3)])
            self.tweak, self.tweak2 = -1, -1

        def forward(self, x, j):
            for i, blk in enumerate(self.blocks):
                if i == 1 and j == self.tweak:
                    x = x * 2.0                              # a new tensor, not what block 0 returned
                if i == 2 and j == self.tweak2:
                    x.mul_(0.5)                              # what block 1 returned, edited in place
                x = blk(x)
            return x

    torch.manual_seed(0)
    model = Tower().to("cuda:0").eval()
    xs = [torch.randn(2, 5, 32, device="cuda:0") for _ in range(8)]
    with torch.no_grad():
        for tweak, tweak2 in ((5, -1), (-1, 6)):
            model.tweak, model.tweak2 = tweak, tweak2
            want = [model(x, j) for j, x in enumerate(xs)]
            undo = cal._wrap_towers(model, ["blocks"], {})
            before = dict(cal.graph_stats)
            try:
                got = [model(x, j) for j, x in enumerate(xs)]
            finally:
                for blocks, i, orig in undo:
                    blocks[i].__dict__["_memo"] = blocks[i].__dict__["_tower"] = None
                    blocks[i] = orig
            assert cal.graph_stats.get("tower_graphs", 0) == before.get("tower_graphs", 0) + 1
            assert cal.graph_stats["fallbacks"] == before["fallbacks"] + 1      # one sample left the traced path
            for a, b in zip(got, want):
                assert torch.equal(a, b)


@pytest.mark.parametrize("ragged", [False, True])
def test_tower_batching_gives_the_per_sample_forward_bit_for_bit(ragged, monkeypatch):
    """Finished 16-bit towers (their linears on the batch-invariant kernel) run ONCE for all postponed calibration forwards
    of a shape (`TowerGraph.run_deferred`); the forwards are repeated and served slices.  The captured decoder inputs equal
    those of the sample-by-sample route bit for bit, in sample order, also with ragged text (three shapes, interleaved)."""
    monkeypatch.setenv("VLMC_CAPTURE_MERGED", "0")          # (this is about the per-sample route: one model forward per calibration batch)
    import toy_models
    from lavis.compression.pruners import calibration as cal
    monkeypatch.setattr(toy_models.ToyAttention, "use_sdpa", True)        # attention per sample and head by construction
    model = toy_models.init_toy(toy_models.ToyBlipT5(vit_dtype=torch.float16, t5_dtype=torch.bfloat16), seed=7).eval().to("cuda:0")
    lens = [5, 7, 5, 5, 7, 3, 5, 7, 5, 5, 7, 5]
    batches = []
    for j, n in enumerate(lens):
        b = toy_models.make_batches(1, txt_len=n if ragged else 5, out_len=(2 + n % 3) if ragged else 4, seed=100 + j)[0]
        batches.append({k: t.to("cuda:0") for k, t in b.items()})
    monkeypatch.setenv("VLMC_LINEAR_FWD", "1")              # (tower batching needs the invariant kernel)
    monkeypatch.setenv("VLMC_TOWER_MEMO", "0")
    monkeypatch.setenv("VLMC_TOWER_BATCH", "0")
    monkeypatch.setenv("VLMC_TOWER_GRAPH", "0")
    want, want_c, _ = _capture_decoder_inputs(model, batches)
    monkeypatch.setenv("VLMC_TOWER_BATCH", "1")
    monkeypatch.setenv("VLMC_TOWER_GRAPH", "1")
    before = dict(cal.graph_stats)
    got, got_c, _ = _capture_decoder_inputs(model, batches)
    assert cal.graph_stats.get("tower_batches", 0) > before.get("tower_batches", 0)
    assert cal.graph_stats["fallbacks"] == before["fallbacks"]
    assert len(got) == len(want) == len(lens)
    for j, (a, b, ca, cb) in enumerate(zip(got, want, got_c, want_c)):
        assert a.shape == b.shape and torch.equal(a, b), j
        assert torch.equal(ca["encoder_hidden_states"], cb["encoder_hidden_states"]), j


def test_finished_tower_that_hands_a_block_output_on_as_a_keyword(monkeypatch):
    """transformers' T5Stack: block 0 computes the relative position bias and RETURNS it, the stack hands it to every later
    block as `position_bias=`.  The capture machinery must wire that (an output of block 0 is a keyword argument of blocks
    1..n) in all three ways a finished tower is traversed: eagerly, as one graph per sample, stacked for all samples."""
    monkeypatch.setenv("VLMC_CAPTURE_MERGED", "0")          # (this is about the per-sample route: one model forward per calibration batch)
    import torch.nn.functional as F
    from lavis.compression.pruners import calibration as cal

    class Block(torch.nn.Module):
        def __init__(self, first, heads=4, dim=64):
            super().__init__()
            self.q, self.k, self.v, self.o = (torch.nn.Linear(dim, dim, bias=False) for _ in range(4))
            self.heads = heads
            self.rel = torch.nn.Parameter(torch.randn(heads, 16, 16) * 0.5) if first else None

        def forward(self, x, position_bias=None, rel_pos_bias=None):
            B, T, D = x.shape
            if position_bias is None:
                position_bias = self.rel[:, :T, :T].to(x.dtype).unsqueeze(0).expand(B, -1, -1, -1)
            sh = lambda t: t.reshape(B, T, self.heads, -1).transpose(1, 2)
            y = F.scaled_dot_product_attention(sh(self.q(x)), sh(self.k(x)), sh(self.v(x)), attn_mask=position_bias)
            return x + self.o(y.transpose(1, 2).reshape(B, T, D)), position_bias

    class Model(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.enc = torch.nn.Module()
            self.enc.block = torch.nn.ModuleList([Block(i == 0) for i in range(4)])
            self.dec = torch.nn.Module()
            self.dec.block = torch.nn.ModuleList([Block(True) for _ in range(2)])

        def forward(self, samples):
            x, pb = samples["image"], None
            for blk in self.enc.block:
                x, pb = blk(x, position_bias=pb)
            for blk in self.dec.block:
                x = blk(x, None)[0]
            return x

    torch.manual_seed(3)
    model = Model().to("cuda:0").half().eval()
    lens = [12, 12, 9, 12, 9, 12, 12, 9, 12, 12]
    batches = [{"image": (torch.randn(1, n, 64, device="cuda:0") * 0.5).half()} for n in lens]

    def capture(**env):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        with torch.no_grad():
            inps, _, _ = cal.capture_block_inputs(model, batches, len(batches), "dec.block", lambda m, b, _l=False: m(b), False,
                                                  vit=True, done_towers=["enc.block"], proxy_cache={})
        torch.cuda.synchronize()
        return inps

    want = capture(VLMC_GRAPH_REPLAY="0")
    before = dict(cal.graph_stats)
    stacked = capture(VLMC_GRAPH_REPLAY="1", VLMC_TOWER_BATCH="1", VLMC_TOWER_GRAPH="1", VLMC_LINEAR_FWD="1", VLMC_TOWER_MEMO="0")
    assert cal.graph_stats.get("tower_batches", 0) > before.get("tower_batches", 0), "the tower was not run stacked"
    before = dict(cal.graph_stats)
    graphed = capture(VLMC_TOWER_BATCH="0")
    assert cal.graph_stats.get("tower_graphs", 0) > before.get("tower_graphs", 0), "the tower was not run from a graph"
    assert cal.graph_stats["fallbacks"] == before["fallbacks"]
    assert len(want) == len(stacked) == len(graphed) == len(lens)
    for a, b, c in zip(want, stacked, graphed):
        assert torch.equal(a, b) and torch.equal(a, c)


# ---- finished towers: the stacked pass from the block-0 arguments remembered from the tower's own capture phase ----------
@pytest.mark.parametrize("ragged", [False, True])
def test_predicted_tower_pass_gives_the_aborted_and_repeated_forwards_result(ragged, monkeypatch):
    """Decoder capture of the 16-bit toy InstructBLIP: the T5 encoder (just pruned) runs stacked for all samples from the
    block-0 arguments its own capture phase saw (`TowerGraph.run_predicted`), every sample then needs ONE forward instead of
    an aborted one plus a repeated one.  Same pruned model, masks and importance scores, bit for bit."""
    monkeypatch.setenv("VLMC_CAPTURE_MERGED", "0")          # (this is about the per-sample route: one model forward per calibration batch)
    import toy_models
    from lavis.compression.pruners import calibration as cal
    monkeypatch.setattr(toy_models.ToyAttention, "use_sdpa", True)
    monkeypatch.setenv("VLMC_LINEAR_FWD", "1")
    monkeypatch.setenv("VLMC_TOWER_PREDICT", "0")
    b0 = dict(cal.graph_stats)
    want = H.run_16bit_toy("wanda", "cuda:0", n_samples=8, ragged=ragged)
    assert cal.graph_stats.get("tower_predicted", 0) == b0.get("tower_predicted", 0)
    monkeypatch.setenv("VLMC_TOWER_PREDICT", "1")
    b1 = dict(cal.graph_stats)
    got = H.run_16bit_toy("wanda", "cuda:0", n_samples=8, ragged=ragged)
    assert cal.graph_stats.get("tower_predicted", 0) > b1.get("tower_predicted", 0), "no tower ran from remembered arguments"
    assert cal.graph_stats.get("batched_traces", 0) > b1.get("batched_traces", 0), "no wiring was traced while the tower ran stacked"
    assert cal.graph_stats.get("later_failed", 0) == b1.get("later_failed", 0)
    assert cal.graph_stats["fallbacks"] == b1["fallbacks"]
    assert want.keys() == got.keys()
    for k in want:
        assert torch.equal(want[k], got[k]), k
    # the forward that traces the wiring running the tower for its one sample (rounds 2-4's trace): the same again
    monkeypatch.setenv("VLMC_TOWER_BATCHED_TRACE", "0")
    b2 = dict(cal.graph_stats)
    eager = H.run_16bit_toy("wanda", "cuda:0", n_samples=8, ragged=ragged)
    assert cal.graph_stats.get("batched_traces", 0) == b2.get("batched_traces", 0)
    for k in want:
        assert torch.equal(want[k], eager[k]), k


@pytest.mark.parametrize("batched_trace", ["1", "0"])
def test_a_wrong_prediction_is_noticed_and_the_phase_runs_again(batched_trace, monkeypatch):
    """The remembered block-0 arguments of one sample are tampered with between the phases: the stacked pass ran on the wrong
    input, the end-of-phase comparison says so (`later_failed`), the phase is repeated without memos or predictions, and the
    pruned model is the one the plain route gives.  Both ways the stacked pass can start: from the forward that traces the wiring
    (`_begin_batched_trace`) and, with an eager trace, from `run_predicted`."""
    monkeypatch.setenv("VLMC_CAPTURE_MERGED", "0")          # (this is about the per-sample route: one model forward per calibration batch)
    import toy_models
    from lavis.compression.pruners import calibration as cal
    monkeypatch.setattr(toy_models.ToyAttention, "use_sdpa", True)
    monkeypatch.setenv("VLMC_LINEAR_FWD", "1")
    monkeypatch.setenv("VLMC_LATER_EQUAL", "1")
    monkeypatch.setenv("VLMC_TOWER_BATCHED_TRACE", batched_trace)
    monkeypatch.setenv("VLMC_TOWER_PREDICT", "0")
    want = H.run_16bit_toy("wanda", "cuda:0", n_samples=8)
    monkeypatch.setenv("VLMC_TOWER_PREDICT", "1")
    tampered = []

    def tamper(tg):
        j = max(tg.predicted)
        args, kwargs, ctx, versions = tg.predicted[j]
        fake = args[0].clone().add_(1)
        tg.predicted[j] = ((fake,) + tuple(args[1:]), kwargs, ctx, [(fake, fake._version)] + versions[1:])
        tampered.append(j)
    real_run, real_begin = cal.TowerGraph.run_predicted, cal.TowerGraph._begin_batched_trace

    def run_predicted(self, samples, *more):
        if batched_trace == "0" and self.predicted and not self.memo_serves and not tampered and any(w for w in self.wirings.values()):
            tamper(self)
        return real_run(self, samples, *more)

    def begin(self, key, args, kwargs, ext):
        if batched_trace == "1" and self.predicted and not self.memo_serves and not tampered:
            tamper(self)
        return real_begin(self, key, args, kwargs, ext)
    monkeypatch.setattr(cal.TowerGraph, "run_predicted", run_predicted)
    monkeypatch.setattr(cal.TowerGraph, "_begin_batched_trace", begin)
    before = cal.graph_stats.get("later_failed", 0)
    got = H.run_16bit_toy("wanda", "cuda:0", n_samples=8)
    assert tampered and cal.graph_stats.get("later_failed", 0) == before + 1
    assert want.keys() == got.keys()
    for k in want:
        assert torch.equal(want[k], got[k]), k
