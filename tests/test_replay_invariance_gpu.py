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
def test_block_forward_is_batch_invariant_at_model_width(which):
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


def _run_16bit_toy(method, group, monkeypatch, n_samples=8, ragged=False):
    import toy_models
    from lavis.compression import load_pruner
    monkeypatch.setenv("VLMC_BATCH_REPLAY", str(group))
    model = toy_models.init_toy(toy_models.ToyBlipT5(vit_dtype=torch.float16, t5_dtype=torch.bfloat16), seed=7).eval().to(DEV)
    lens = [5, 7, 5, 5, 7, 3, 5, 7]
    batches = []
    for j in range(n_samples):
        b = toy_models.make_batches(1, txt_len=lens[j] if ragged else 5, out_len=(2 + lens[j] % 3) if ragged else 4, seed=100 + j)[0]
        batches.append({k: t.to(DEV) for k, t in b.items()})
    spec = "2-0.5-1.0-1.0"
    cfg = dict(t5_prune_spec=spec, vit_prune_spec=spec, t5_pruning_method=method, vit_pruning_method=method,
               num_samples=n_samples, max_sparsity_per_layer=1.01)
    if method == "dsnot":
        cfg["max_cycle_time"] = 8
    pruned, _ = load_pruner(f"blipt5_{method}_pruner", model, batches, cfg=cfg).prune()
    sd = {k: v.clone() for k, v in pruned.state_dict().items()}
    for n, m in pruned.named_modules():
        if hasattr(m, "mask") and torch.is_tensor(m.mask):
            sd[n + ".mask*"] = m.mask.clone()
        if hasattr(m, "weight") and hasattr(m.weight, "importance_score"):
            sd[n + ".importance*"] = torch.tensor(m.weight.importance_score, dtype=torch.float64)
    return sd


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


def test_library_gemms_are_not_batch_invariant_or_at_least_not_promised(monkeypatch):
    """`VLMC_LINEAR_FWD=0` leaves the block's GEMMs to the library: the replay still works (masks agree up to near-ties),
    the kernel counter stays put."""
    from vlmc import forward
    monkeypatch.setenv("VLMC_LINEAR_FWD", "0")
    before = dict(forward.stats)
    a = _run_16bit_toy("wanda", 1, monkeypatch)
    b = _run_16bit_toy("wanda", 128, monkeypatch)
    assert forward.stats["kernel"] == before["kernel"]
    tot = diff = 0
    for k in a:
        if k.endswith(".mask*"):
            tot += a[k].numel()
            diff += int((a[k] != b[k]).sum())
    assert tot and diff / tot < 0.01
