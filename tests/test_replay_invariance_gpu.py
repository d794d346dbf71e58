"""The grouped calibration replay does not change a bit of the statistics: with the block's linears on the
batch-invariant MFMA kernel (vlmc/forward.py, csrc/gemm_nt.hip), forwarding the calibration samples one by one (the
reference's loop, wanda_pruner.py:308-311), in groups, or all at once gives identical activations, masks and weights.

(The reference's contract is "one sample per forward"; what makes the grouped default safe is this equality, not a
tolerance.)"""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _blocks():
    from vlmc import synthetic as S
    torch.manual_seed(0)
    vit = S.ViTBlock(1408, 6144, 16).to(DEV).half().eval()
    enc = S.T5Block(2048, 5120, 32, 64, False).to(DEV).bfloat16().eval()
    dec = S.T5Block(2048, 5120, 32, 64, True).to(DEV).bfloat16().eval()
    for blk in (vit, enc, dec):
        S.randomize_(blk, seed=3)
    return vit, enc, dec


@pytest.mark.parametrize("which", ["vit", "enc", "dec"])
def test_block_forward_is_batch_invariant_at_model_width(which, monkeypatch):
    """One transformer block at InstructBLIP-FlanT5-XL's dimensions: 6 samples per forward == 6 forwards of one sample,
    bit for bit, with the linears on vlmc_linear_fwd (attention, norms and activations are per-sample by nature)."""
    from vlmc import forward
    from lavis.compression.pruners import calibration as cal
    vit, enc, dec = _blocks()
    g = torch.Generator(device=DEV).manual_seed(1)
    n = 6
    if which == "vit":
        blk, xs, kw = vit, [(torch.randn(1, 257, 1408, generator=g, device=DEV) * 0.5).half() for _ in range(n)], [{} for _ in range(n)]
        call = lambda x, k: blk(x, None)
    else:
        blk = enc if which == "enc" else dec
        T = 64 if which == "enc" else 16
        xs = [(torch.randn(1, T, 2048, generator=g, device=DEV) * 0.5).bfloat16() for _ in range(n)]
        kw = [dict(encoder_hidden_states=(torch.randn(1, 64, 2048, generator=g, device=DEV) * 0.5).bfloat16()) if which == "dec" else {}
              for _ in range(n)]
        call = lambda x, k: blk(x, **k)[0]
    subset = cal.find_layers(blk)
    before = dict(forward.stats)
    with torch.no_grad(), forward.invariant_linears(subset.values()):
        one = [call(x, k) for x, k in zip(xs, kw)]
        stacked_kw = {k: torch.cat([c[k] for c in kw]) for k in kw[0]}
        allg = call(torch.cat(xs), stacked_kw)
        three = [call(torch.cat(xs[j:j + 3]), {k: torch.cat([c[k] for c in kw[j:j + 3]]) for k in kw[0]}) for j in (0, 3)]
    assert forward.stats["kernel"] - before["kernel"] == len(subset) * (n + 1 + 2) and forward.stats["library"] == before["library"]
    assert torch.equal(allg, torch.cat(one)), f"{which}: grouped forward differs from the per-sample forwards"
    assert torch.equal(torch.cat(three), allg)
    assert not any("forward" in m.__dict__ for m in subset.values())                 # patches are gone
    # sibling linears (q / k / v, wi_0 / wi_1, cross-attention k / v) were learned from the first forward and shared launches
    # in the later ones; with the grouping switched off the block computes the same bits
    grouped = forward.stats["grouped_launches"] - before["grouped_launches"]
    groups = forward.sibling_groups(subset.values())
    if which == "vit":
        assert grouped == 0 and not groups                                          # one fused qkv linear already
    else:
        want = [("q", "k", "v"), ("wi_0", "wi_1")] + ([("k", "v")] if which == "dec" else [])
        names = {id(m): n.split(".")[-1] for n, m in subset.items()}
        assert sorted(tuple(names[id(m)] for m in g) for g in groups) == sorted(want)
        assert grouped == len(want) * (n + 2)                                       # every forward after the first
        assert forward.stats["stash_dropped"] == before["stash_dropped"]
    monkeypatch.setenv("VLMC_LINEAR_GROUP", "0")
    with torch.no_grad(), forward.invariant_linears(subset.values()):
        assert torch.equal(call(torch.cat(xs), stacked_kw), allg)
    assert forward.stats["grouped_launches"] - before["grouped_launches"] == grouped


def _run_16bit_toy(method, group, monkeypatch, ragged=False, sdpa=True):
    """(Attention through SDPA: per sample and head by construction.  The toy's default, an explicit batched `torch.matmul`
    like the reference's model files, is at the GEMM library's discretion -- see the last test.)"""
    import pruner_helpers as H
    import toy_models
    monkeypatch.setattr(toy_models.ToyAttention, "use_sdpa", sdpa)
    monkeypatch.setenv("VLMC_BATCH_REPLAY", str(group))
    monkeypatch.setenv("VLMC_TOWER_BATCH", "0" if group == 1 else "1")     # group 1 = the reference's sample-by-sample route
    return H.run_16bit_toy(method, DEV, ragged=ragged)


@pytest.mark.parametrize("ragged", [False, True])
@pytest.mark.parametrize("method", ["wanda", "dsnot"])
def test_whole_prune_is_identical_for_every_grouping(method, ragged, monkeypatch):
    """fp16 ViT + bf16 T5 toy InstructBLIP: per-sample loop (HIP graphs), groups of 3, and one group give the same masks,
    weights and importance scores bit for bit -- also with ragged text, where groups are formed from non-neighbours."""
    from vlmc import forward
    before = forward.stats["kernel"]
    ref = _run_16bit_toy(method, 1, monkeypatch, ragged=ragged)
    assert forward.stats["kernel"] > before, "the invariant kernel did not run"
    for group in (3, 128):
        got = _run_16bit_toy(method, group, monkeypatch, ragged=ragged)
        assert got.keys() == ref.keys()
        for k in ref:
            assert torch.equal(got[k], ref[k]), (group, k)
    assert sum(1 for k in ref if k.endswith(".mask*")) == 2 * 4 + 2 * 7 + 2 * 11


def test_library_linears_are_not_promised_bit_for_bit(monkeypatch):
    """`VLMC_LINEAR_FWD=0` leaves the block's linears (and the attention products) to the GEMM library: the grouped replay
    still works and tracks the sample-by-sample run up to near-ties, but equality of bits is the library's to give."""
    from vlmc import forward
    before = dict(forward.stats)
    monkeypatch.setenv("VLMC_LINEAR_FWD", "0")
    a = _run_16bit_toy("wanda", 1, monkeypatch, sdpa=False)
    b = _run_16bit_toy("wanda", 128, monkeypatch, sdpa=False)
    assert forward.stats["kernel"] == before["kernel"] and forward.stats["attn_kernel"] == before["attn_kernel"]
    tot = diff = 0
    for k in a:
        if k.endswith(".mask*"):
            tot += a[k].numel()
            diff += int((a[k] != b[k]).sum())
    assert tot and diff / tot < 0.01


@pytest.mark.parametrize("ragged", [False, True])
@pytest.mark.parametrize("method", ["wanda", "dsnot"])
def test_attention_written_as_batched_matmuls_is_identical_for_every_grouping(method, ragged, monkeypatch):
    """The reference's T5 and EVA-ViT write attention as batched `torch.matmul`s (modeling_t5.py:590,638; eva_vit.py:147,164).
    During the replay those products run on `vlmc_attn_matmul` (vlmc/forward.py: invariant_matmuls), batch-invariant like
    the linears: per-sample loop, groups of 3 and one group give the same masks, weights and importance scores, bit for
    bit (round 3 left them to the library: 0.997-0.99997 agreement, run to run)."""
    from vlmc import forward
    before = dict(forward.stats)
    ref = _run_16bit_toy(method, 1, monkeypatch, ragged=ragged, sdpa="matmul16")
    assert forward.stats["attn_kernel"] + forward.stats["attn_fused"] > before["attn_kernel"] + before["attn_fused"], \
        "the attention products did not run on the invariant kernels"
    assert forward.stats["attn_library"] == before["attn_library"]
    for group in (3, 128):
        got = _run_16bit_toy(method, group, monkeypatch, ragged=ragged, sdpa="matmul16")
        assert got.keys() == ref.keys()
        for k in ref:
            assert torch.equal(got[k], ref[k]), (group, k)
    again = _run_16bit_toy(method, 128, monkeypatch, ragged=ragged, sdpa="matmul16")       # and run to run
    for k in ref:
        assert torch.equal(again[k], ref[k]), k


def test_a_sibling_group_whose_products_are_wasted_is_forgotten():
    """once q / k / v are a known group, a call pattern in which q is called alone would keep computing k's and v's
    products for nothing.  After three forwards whose stash was dropped the group is forgotten (and learned again when the
    calls show it again)."""
    from vlmc import forward
    torch.manual_seed(0)
    q, k, v = (torch.nn.Linear(64, 64, bias=False).to(DEV).half() for _ in range(3))
    x = torch.randn(5, 64, device=DEV).half()
    mods = [q, k, v]
    with torch.no_grad():
        with forward.invariant_linears(mods):
            want = [m(x) for m in mods]                      # learned here
        assert forward.sibling_groups(mods) == [(q, k, v)]
        before = dict(forward.stats)
        for _ in range(3):                                   # q alone: one grouped launch each, two products dropped
            with forward.invariant_linears(mods):
                assert torch.equal(q(x), want[0])
        assert forward.stats["grouped_launches"] - before["grouped_launches"] == 3
        assert forward.stats["stash_dropped"] - before["stash_dropped"] == 6
        assert forward.sibling_groups(mods) == []            # forgotten
        with forward.invariant_linears(mods):
            assert torch.equal(q(x), want[0])                # a single launch now
        assert forward.stats["grouped_launches"] - before["grouped_launches"] == 3
        with forward.invariant_linears(mods):
            got = [m(x) for m in mods]                       # the calls show the group again
        assert forward.sibling_groups(mods) == [(q, k, v)] and all(torch.equal(a, b) for a, b in zip(got, want))


@pytest.mark.parametrize("lora_model", [False, True])
def test_ragged_samples_padded_into_one_group_keep_their_bits(lora_model, monkeypatch):
    """Ragged calibration text on a stand-in whose T5 blocks follow the reference's op sequence (extended masks, explicit matmuls,
    fp32 softmax): the default route pads all lengths into ONE forward per block pass (zero rows, masks at the dtype's minimum,
    per-sample token counts for the statistics, the softmax on `vlmc_softmax_rows`); masks, weights and importance scores of a
    whole Wanda prune equal the groups-of-equal-shapes route (`VLMC_PAD_RAGGED=0`) and the per-sample loop bit for bit."""
    from vlmc import forward, synthetic
    from lavis.compression.pruners import calibration
    dev = torch.device(DEV)

    def run(env):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        torch.manual_seed(0)
        model = synthetic.InstructBlipT5(vit_dim=64, vit_hidden=128, vit_heads=4, vit_depth=2, d_model=64, d_ff=128, heads=4, d_kv=16,
                                         enc_depth=3, dec_depth=3, vocab=100, query_tokens=4, reference_ops=True, qformer_dim=64, qformer_heads=4,
                                         qformer_hidden=128, qformer_depth=2, qformer_vocab=50).to(dev).eval()
        batches = synthetic.calibration_batches(16, dev, vit_tokens=9, vit_dim=64, vocab=100, ragged=True)
        before = (calibration.graph_stats.get("padded_forwards", 0), forward.stats["softmax_kernel"] + forward.stats["attn_fused"],
                  calibration.graph_stats.get("tower_padded_passes", 0))
        synthetic.time_prune(dev, n_samples=16, model=model, batches=batches, **({"t5_prune_spec": "3-0.5-1.0-1.0"}))
        out = {n: (m.weight.detach().clone(), m.mask.clone() if hasattr(m, "mask") else None) for n, m in model.named_modules()
               if isinstance(m, torch.nn.Linear) and (".block." in n or ".blocks." in n)}
        seen["tower_padded"] = calibration.graph_stats.get("tower_padded_passes", 0) - before[2]
        return out, calibration.graph_stats.get("padded_forwards", 0) - before[0], \
            forward.stats["softmax_kernel"] + forward.stats["attn_fused"] - before[1]       # (the softmax kernel, alone or inside the fused attention)

    base = {"VLMC_BATCH_REPLAY": "128", "VLMC_TOWER_BATCH": "1", "VLMC_PAD_RAGGED": "1", "VLMC_TOWER_PAD": "1", "VLMC_CAPTURE_MERGED": "0"}
    seen = {}
    before_m = calibration.graph_stats.get("merged_forwards", 0)
    monkeypatch.setattr(calibration.replay_capture, "MERGED_CAPTURE_MIN", 2)
    declined0, slices0 = calibration.graph_stats.get("merged_capture_declined", 0), forward.stats["kernel_slices"]
    views0 = calibration.graph_stats.get("group_views", 0)
    merged, n_m, _ = run({**base, "VLMC_CAPTURE_MERGED": "1", "VLMC_CAPTURE_MERGED_RAGGED": "1"})      # merged forwards per shape, towers deferred (opt-in)
    # (round 6, second half) no phase declined -- the decoder's phase runs merged although the pruned encoder lies on the way -- and the
    # fp32 Q-Former ran as a padded pass whose token slices (its query / text feed-forward halves) were read in place
    assert calibration.graph_stats.get("merged_capture_declined", 0) == declined0 and forward.stats["kernel_slices"] > slices0, \
        (calibration.graph_stats, forward.stats)
    assert calibration.graph_stats.get("group_views", 0) > views0              # .. whose groups were handed views of the padded pass
    variants = {}
    for env_, name_ in (({"VLMC_CAPTURE_MERGED_PRUNED": "0"}, "decoder's capture per sample"), ({"VLMC_ROW_SLICES": "0"}, "slices with their padding rows"),
                        ({"VLMC_MEMO_COPY": "1"}, "memo copies"), ({"VLMC_REPLAY_TOKENS": "40"}, "several padded chunks per block pass"),
                        ({"VLMC_GROUP_VIEWS": "0"}, "groups handed copies, not views of the padded pass")):
        d_ = calibration.graph_stats.get("merged_capture_declined", 0)
        variants[name_], _, _ = run({**base, "VLMC_CAPTURE_MERGED": "1", "VLMC_CAPTURE_MERGED_RAGGED": "1", **env_})
        if "VLMC_CAPTURE_MERGED_PRUNED" in env_:
            assert calibration.graph_stats.get("merged_capture_declined", 0) == d_ + 1           # the decoder's phase, and only it
        for k_ in env_:
            monkeypatch.delenv(k_)
    monkeypatch.delenv("VLMC_CAPTURE_MERGED_RAGGED")
    # (ragged batches: the groups' merged forwards are postponed at the finished towers, which run once, padded, for all samples)
    assert calibration.graph_stats.get("merged_forwards", 0) > before_m and n_m == 2 * 2 * 3, calibration.graph_stats
    rows0, lens0 = forward.stats["kernel_rows"], forward.stats["attn_fused_lens"]
    padded, n_padded, n_softmax = run(base)
    # (round 6) inside the padded groups the linears ran over the row map and the fused attention took the samples' lengths ...
    assert forward.stats["kernel_rows"] > rows0 and forward.stats["attn_fused_lens"] > lens0, forward.stats
    rows1 = forward.stats["kernel_rows"]
    every_row, n_er, _ = run({**base, "VLMC_ROW_MAP": "0"})                  # ... and with every padding row computed, as in round 5: same bits
    assert forward.stats["kernel_rows"] == rows1 and n_er == n_padded
    monkeypatch.delenv("VLMC_ROW_MAP")
    # .. and the finished encoder tower ran ONE padded stacked pass for the samples behind the scout's group while the decoder's
    # inputs were captured (TowerGraph._run_padded)
    assert seen["tower_padded"] >= 1, seen
    towers_per_length, n_tp, _ = run({**base, "VLMC_TOWER_PAD": "0"})
    assert seen["tower_padded"] == 0 and n_tp == n_padded
    shaped, n_shaped, _ = run({**base, "VLMC_PAD_RAGGED": "0"})
    single, _, _ = run({**base, "VLMC_BATCH_REPLAY": "1", "VLMC_TOWER_BATCH": "0"})
    assert n_padded == 2 * 2 * 3 and n_shaped == 0 and n_softmax > 0          # encoder + decoder tower, two passes, three blocks: ONE forward each
    assert padded.keys() == shaped.keys() == single.keys() and len(padded) == 2 * 4 + 3 * 7 + 3 * 11
    for k in padded:
        for other, name in ((shaped, "groups of equal shapes"), (single, "per-sample loop"), (towers_per_length, "towers per token count"),
                            (merged, "merged capture forwards"), (every_row, "padding rows computed")) + tuple((v_, n_) for n_, v_ in variants.items()):
            assert torch.equal(padded[k][0], other[k][0]), (k, name)
            assert (padded[k][1] is None and other[k][1] is None) or torch.equal(padded[k][1], other[k][1]), (k, name)


@pytest.mark.parametrize("n_shapes", [1, 2])
def test_merged_capture_forwards_give_the_per_sample_routes_bits(n_shapes, monkeypatch):
    """calibration._capture_merged: the calibration batches of one shape go through the model's own forward as ONE stacked batch per
    capture phase (the Q-Former in the way is simply run, nothing is aborted and repeated); a whole prune -- masks, weights,
    importance scores of all three towers -- equals the per-sample route's (one model forward per calibration batch) and the
    reference's one-sample-per-forward loop, bit for bit; with two text lengths the batches merge per length."""
    from vlmc import forward, synthetic
    from lavis.compression.pruners import calibration
    dev = torch.device(DEV)

    def run(env):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        torch.manual_seed(0)
        model = synthetic.InstructBlipT5(vit_dim=64, vit_hidden=128, vit_heads=4, vit_depth=2, d_model=64, d_ff=128, heads=4, d_kv=16,
                                         enc_depth=3, dec_depth=3, vocab=100, query_tokens=4, reference_ops=True, qformer_dim=64, qformer_heads=4,
                                         qformer_hidden=128, qformer_depth=2, qformer_vocab=50).to(dev).eval()
        batches = synthetic.calibration_batches(16, dev, vit_tokens=9, vit_dim=64, vocab=100, ragged=False)
        if n_shapes == 2:                                                  # every third sample with a shorter prompt
            for b in batches[::3]:
                b["text_input"] = b["text_input"][:, :5].contiguous()
        m0, d0 = calibration.graph_stats.get("merged_forwards", 0), calibration.graph_stats.get("merged_capture_declined", 0)
        synthetic.time_prune(dev, n_samples=16, model=model, batches=batches, **({"t5_prune_spec": "3-0.5-1.0-1.0"}))
        out = {n: (m.weight.detach().clone(), m.mask.clone() if hasattr(m, "mask") else None, getattr(m.weight, "importance_score", None))
               for n, m in model.named_modules() if isinstance(m, torch.nn.Linear) and (".block." in n or ".blocks." in n)}
        return out, calibration.graph_stats.get("merged_forwards", 0) - m0, calibration.graph_stats.get("merged_capture_declined", 0) - d0

    monkeypatch.setattr(calibration.replay_capture, "MERGED_CAPTURE_MIN", 2)             # (16 samples here; the default asks for 24)
    merged, n_merged, declined = run({"VLMC_CAPTURE_MERGED": "1", "VLMC_BATCH_REPLAY": "128", "VLMC_TOWER_BATCH": "1"})
    assert n_merged == 3 * n_shapes and declined == 0                      # one stacked model forward per tower, capture phase and shape
    per_sample, n0, _ = run({"VLMC_CAPTURE_MERGED": "0"})
    loop, _, _ = run({"VLMC_CAPTURE_MERGED": "0", "VLMC_BATCH_REPLAY": "1", "VLMC_TOWER_BATCH": "0"})
    assert n0 == 0 and merged.keys() == per_sample.keys() == loop.keys() and len(merged) == 2 * 4 + 3 * 7 + 3 * 11
    for k in merged:
        for other, name in ((per_sample, "per-sample capture"), (loop, "the reference's loop")):
            assert torch.equal(merged[k][0], other[k][0]), (k, name)
            assert (merged[k][1] is None and other[k][1] is None) or torch.equal(merged[k][1], other[k][1]), (k, name)
            assert merged[k][2] == other[k][2], (k, name)


def test_a_model_that_cannot_take_the_merged_batch_is_forwarded_per_sample(monkeypatch):
    """A forward that assumes batch 1 somewhere (here: it raises on the stacked batch) sends the capture phase down the per-sample
    route -- the prune's result is the per-sample route's, the refusal is counted."""
    from vlmc import synthetic
    from lavis.compression.pruners import calibration
    dev = torch.device(DEV)
    monkeypatch.setattr(calibration.replay_capture, "MERGED_CAPTURE_MIN", 2)

    def run(picky):
        torch.manual_seed(0)
        model = synthetic.InstructBlipT5(vit_dim=64, vit_hidden=128, vit_heads=4, vit_depth=2, d_model=64, d_ff=128, heads=4, d_kv=16,
                                         enc_depth=2, dec_depth=2, vocab=100, query_tokens=4, qformer_dim=64, qformer_heads=4,
                                         qformer_hidden=128, qformer_depth=2, qformer_vocab=50).to(dev).eval()
        if picky:
            plain = model.forward

            def forward(samples, *a, **k):
                if samples["image"].shape[0] != 1:
                    raise RuntimeError("this model forwards one sample at a time")
                return plain(samples, *a, **k)
            model.forward = forward
        batches = synthetic.calibration_batches(8, dev, vit_tokens=9, vit_dim=64, vocab=100)
        e0 = calibration.graph_stats.get("merged_capture_errors", 0)
        synthetic.time_prune(dev, n_samples=8, model=model, batches=batches)
        return {n: m.weight.detach().clone() for n, m in model.named_modules() if isinstance(m, torch.nn.Linear)}, \
            calibration.graph_stats.get("merged_capture_errors", 0) - e0
    a, errs = run(True)
    b, errs0 = run(False)
    assert errs == 3 and errs0 == 0 and a.keys() == b.keys()          # one refusal per capture phase
    for k in a:
        assert torch.equal(a[k], b[k]), k
