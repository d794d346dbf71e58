"""The synthetic InstructBLIP-FlanT5-XL used for end-to-end timing has exactly the prunable linears of the workload
table (names, shapes, dtypes); a small instance goes through the drop-in pruner on the GPU."""
import pytest
import torch
import torch.nn as nn


def test_synthetic_model_matches_the_workload_table():
    from vlmc import synthetic, workload
    with torch.device("meta"):
        model = synthetic.InstructBlipT5()
    mods = dict(model.named_modules())
    n = total = 0
    for b in workload.flan_t5_xl():
        for lin in b.linears:
            m = mods[f"{b.name}.{lin.name}"]
            assert type(m) is nn.Linear and tuple(m.weight.shape) == (lin.out_features, lin.in_features)
            assert m.weight.dtype == b.dtype
            n += 1
            total += lin.out_features * lin.in_features
    assert n == synthetic.prunable_linears(model) == 588
    # the Q-Former between the towers (BASELINE.json configs[1]: "ViT-g + QFormer + T5"): 12 BERT-base layers, cross-attention to
    # the 1408-wide image tokens in every second one, none of its linears among the prunable ones
    layers = model.Qformer.bert.encoder.layer
    assert len(layers) == 12 and [l.has_cross_attention for l in layers] == [i % 2 == 0 for i in range(12)]
    assert tuple(layers[0].crossattention.self.key.weight.shape) == (768, 1408) and tuple(layers[1].intermediate_query.dense.weight.shape) == (3072, 768)
    assert tuple(model.t5_proj.weight.shape) == (2048, 768)
    assert sum(p.numel() for k, p in model.named_parameters() if p.dim() == 2 and (".blocks." in k or ".block." in k)) == total


@pytest.mark.gpu
def test_small_synthetic_model_through_the_wanda_pruner():
    from vlmc import synthetic
    dev = torch.device("cuda:0")
    model = synthetic.InstructBlipT5(vit_dim=64, vit_hidden=128, vit_heads=4, vit_depth=2, d_model=64, d_ff=128, heads=4, d_kv=16,
                                     enc_depth=2, dec_depth=2, vocab=100, query_tokens=4).to(dev).eval()
    batches = synthetic.calibration_batches(8, dev, vit_tokens=9, vit_dim=64, text_len=5, out_len=3, vocab=100)
    dt, model, info = synthetic.time_prune(dev, n_samples=8, model=model, batches=batches)
    assert info["linears"] == 2 * 4 + 2 * 7 + 2 * 11 and dt > 0
    assert abs(info["pruned_fraction"] - 0.5) < 0.01
    for name, mod in model.named_modules():
        if isinstance(mod, nn.Linear) and ".block." in name:
            assert bool(((mod.weight == 0).sum(dim=1) == mod.weight.shape[1] // 2).all()), name     # per-row rule on the T5 side
        if isinstance(mod, nn.Linear) and name.startswith("Qformer."):
            assert float((mod.weight == 0).float().mean()) < 1e-3 and not hasattr(mod, "mask"), name      # the Q-Former is never pruned


@pytest.mark.gpu
@pytest.mark.parametrize("ragged", [False, True])
def test_q_former_run_as_a_finished_tower_gives_the_per_sample_forwards_result(ragged, monkeypatch):
    """The Q-Former's layers are run through like a finished tower in the language model's capture phases (stacked over the
    samples of a group in the encoder's phase, handed out from the memo in the decoder's): masks and weights of a whole prune
    equal, bit for bit, the route in which every calibration sample goes through them alone (`replay_per_sample`: same
    batch-invariant kernels, one sample per forward).  Against the route that leaves the Q-Former to torch's eager ops (library
    GEMMs: other summation orders) the masks agree to near-ties."""
    monkeypatch.setenv("VLMC_CAPTURE_MERGED", "0")          # (this is about the per-sample route: one model forward per calibration batch)
    from vlmc import synthetic
    from lavis.compression.pruners import calibration
    dev = torch.device("cuda:0")

    def run(frozen, per_sample):
        monkeypatch.setattr(calibration.replay_towers, "FROZEN_TOWERS", frozen)
        monkeypatch.setenv("VLMC_BATCH_REPLAY", "1" if per_sample else "128")
        monkeypatch.setenv("VLMC_TOWER_BATCH", "0" if per_sample else "1")
        torch.manual_seed(0)                                           # (biases and norm weights keep torch's default, RNG-drawn init)
        model = synthetic.InstructBlipT5(vit_dim=64, vit_hidden=128, vit_heads=4, vit_depth=2, d_model=64, d_ff=128, heads=4, d_kv=16,
                                         enc_depth=2, dec_depth=2, vocab=100, query_tokens=4, qformer_dim=64, qformer_heads=4, qformer_hidden=128,
                                         qformer_depth=4, qformer_vocab=50).to(dev).eval()
        batches = synthetic.calibration_batches(12, dev, vit_tokens=9, vit_dim=64, text_len=5, out_len=3, vocab=100, ragged=ragged)
        before = dict(calibration.graph_stats)
        synthetic.time_prune(dev, n_samples=12, model=model, batches=batches)
        delta = {k: v - before.get(k, 0) for k, v in calibration.graph_stats.items() if isinstance(v, (int, float))}
        return {n: (m.weight.detach().clone(), m.mask.clone()) for n, m in model.named_modules() if isinstance(m, nn.Linear) and hasattr(m, "mask")}, delta

    frozen = calibration.FROZEN_TOWERS
    got, stats = run(frozen, False)
    ref, _ = run(frozen, True)
    eager, _ = run((), False)
    assert stats.get("tower_batches", 0) >= 2 and stats.get("memo_recorded", 0) >= 24            # stacked, and remembered for the decoder's phase
    assert got.keys() == ref.keys() == eager.keys() and len(got) == 2 * 4 + 2 * 7 + 2 * 11
    for k in got:
        assert torch.equal(got[k][0], ref[k][0]) and torch.equal(got[k][1], ref[k][1]), k
    agree = sum(int((got[k][1] == eager[k][1]).sum()) for k in got) / sum(got[k][1].numel() for k in got)
    assert agree > 0.99, agree


def test_synthetic_vicuna_has_the_llama_linears():
    from vlmc import synthetic
    with torch.device("meta"):
        model = synthetic.InstructBlipVicuna()
    assert synthetic.prunable_linears(model) == 39 * 4 + 32 * 7
    layer = model.llm_model.model.layers[0]
    shapes = {n: tuple(m.weight.shape) for n, m in layer.named_modules() if isinstance(m, nn.Linear)}
    assert shapes == {"self_attn.q_proj": (4096, 4096), "self_attn.k_proj": (4096, 4096), "self_attn.v_proj": (4096, 4096),
                      "self_attn.o_proj": (4096, 4096), "mlp.gate_proj": (11008, 4096), "mlp.up_proj": (11008, 4096),
                      "mlp.down_proj": (4096, 11008)}
    assert all(p.dtype == torch.float16 for p in model.parameters())


@pytest.mark.gpu
def test_small_synthetic_vicuna_through_the_dsnot_pruner():
    from vlmc import synthetic
    dev = torch.device("cuda:0")
    model = synthetic.InstructBlipVicuna(vit_dim=64, vit_hidden=128, vit_heads=4, vit_depth=2, dim=64, d_ff=176, heads=4, depth=2,
                                         vocab=100, query_tokens=4).to(dev).eval()
    batches = synthetic.calibration_batches(8, dev, vit_tokens=9, vit_dim=64, text_len=5, out_len=3, vocab=100)
    dt, model, info = synthetic.time_prune(dev, "blipt5_dsnot_pruner", n_samples=8, model=model, batches=batches,
                                           t5_model_prefix="llm_model", max_cycle_time=8)
    assert info["linears"] == 2 * 4 + 2 * 7
    assert abs(info["pruned_fraction"] - 0.5) < 0.01


@pytest.mark.gpu
@pytest.mark.parametrize("ragged", [False, True])
def test_q_former_with_the_references_call_contract_takes_the_stacked_routes(ragged, monkeypatch):
    """ The reference-op stand-in calls its Q-Former as the reference does -- fp32 weights outside autocast, layers called
    POSITIONALLY with tensor extended masks, `(layer_output, (key, value))` coming back (Qformer.py:470-474, :541-550, :795-801) -- and the
    engine still runs it like a finished tower: stacked over the samples (fp32 linears, attention products and GELU on the
    batch-invariant fp32 kernels), remembered from the encoder's phase to the decoder's (nested-tuple outputs handed on as
    `(hidden_states, ..)`), merged capture forwards checked bit for bit.  Masks and weights of the whole prune equal the route in
    which every calibration sample goes through everything alone."""
    from vlmc import forward, synthetic
    from lavis.compression.pruners import calibration
    dev = torch.device("cuda:0")
    monkeypatch.setattr(calibration.replay_capture, "MERGED_CAPTURE_MIN", 2)

    def run(per_sample):
        monkeypatch.setenv("VLMC_BATCH_REPLAY", "1" if per_sample else "128")
        monkeypatch.setenv("VLMC_TOWER_BATCH", "0" if per_sample else "1")
        torch.manual_seed(0)
        model = synthetic.InstructBlipT5(vit_dim=64, vit_hidden=128, vit_heads=4, vit_depth=2, d_model=64, d_ff=128, heads=4, d_kv=16,
                                         enc_depth=2, dec_depth=2, vocab=100, query_tokens=4, qformer_dim=64, qformer_heads=4, qformer_hidden=128,
                                         qformer_depth=4, qformer_vocab=50, reference_ops=True).to(dev).eval()
        layer = model.Qformer.bert.encoder.layer[0]
        assert layer.attention.self.query.weight.dtype == torch.float32 and model.t5_proj.weight.dtype == torch.float32
        out = layer(torch.zeros(1, 6, 64, device=dev), torch.zeros(1, 1, 1, 6, device=dev), None, torch.zeros(1, 9, 64, device=dev),
                    torch.zeros(1, 1, 1, 9, device=dev), None, False, 4)
        assert isinstance(out, tuple) and len(out) == 2 and isinstance(out[1], tuple) and len(out[1]) == 2 and out[1][0].shape == (1, 4, 6, 16)
        batches = synthetic.calibration_batches(12, dev, vit_tokens=9, vit_dim=64, text_len=5, out_len=3, vocab=100, ragged=ragged)
        before = dict(calibration.graph_stats)
        lib0 = forward.stats["library"]
        synthetic.time_prune(dev, n_samples=12, model=model, batches=batches)
        delta = {k: v - before.get(k, 0) for k, v in calibration.graph_stats.items() if isinstance(v, (int, float))}
        delta["library_linears"] = forward.stats["library"] - lib0
        return {n: (m.weight.detach().clone(), m.mask.clone()) for n, m in model.named_modules() if isinstance(m, nn.Linear) and hasattr(m, "mask")}, delta

    got, stats = run(False)
    ref, _ = run(True)
    print("graph_stats of the stacked route:", {k: v for k, v in stats.items() if v})
    # the fp32 Q-Former ran stacked -- inside the merged capture forwards (equal lengths), or as a finished tower per group of equal
    # shapes / one padded pass (ragged) -- and was remembered for the decoder's phase; nothing fell back, every linear on an invariant kernel
    assert stats.get("merged_forwards", 0) >= 2 or stats.get("tower_batches", 0) >= 2, stats
    assert stats.get("memo_recorded", 0) >= 12 and stats.get("memo_hits", 0) >= 1, stats
    assert stats.get("merged_capture_mismatch", 0) == 0 and stats.get("merged_capture_errors", 0) == 0 and stats.get("fallbacks", 0) == 0, stats
    assert stats["library_linears"] == 0, stats
    assert got.keys() == ref.keys() and len(got) == 2 * 4 + 2 * 7 + 2 * 11
    for k in got:
        assert torch.equal(got[k][0], ref[k][0]) and torch.equal(got[k][1], ref[k][1]), k
