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
