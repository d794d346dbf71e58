"""CPU: the drop-in pruners' host logic (capture, block walk, tower order, inps/outs swap,
hook de-duplication, lora_model semantics, importance read-back) against whole-pruner golden
runs of the REFERENCE on the toy InstructBLIP (tests/golden/wanda_e2e.npz).

The numeric kernels are GPU-only, so `vlmc.ops` is monkeypatched with oracle-backed
stand-ins (tests/oracle_ops.py) -- a test fixture, not a product fallback.  On CPU the toy
model's forward is the same PyTorch code the reference ran, so everything must match the
golden bit for bit."""
import pytest
import torch

import oracle_ops
import pruner_helpers as H


@pytest.mark.parametrize("name", list(H.VARIANTS))
def test_blipt5_wanda_pruner_matches_reference_run(name, monkeypatch):
    oracle_ops.install(monkeypatch)
    pruned, sd = H.run_pruner(name, "cpu")
    assert sd is None                                  # granularity none -> (model, None)
    st = H.compare_with_golden(name, pruned, exact=True, min_mask_agreement=1.0)
    assert st["masks"] == 2 * 4 + 2 * 7 + 2 * 11       # every prunable linear of the toy got a mask


def test_lora_model_keeps_weights_dense_and_sets_mask_buffers(monkeypatch):
    oracle_ops.install(monkeypatch)
    model, _, _ = H.build("fp32_r40_lora")
    before = {k: v.clone() for k, v in model.state_dict().items() if k.endswith("weight")}
    pruned, _ = H.run_pruner("fp32_r40_lora")
    for k, v in pruned.state_dict().items():
        if k.endswith(".weight") and "lora_" not in k and k in before:
            assert torch.equal(v, before[k]), k        # wanda_pruner.py:340-341: no zeroing under lora_model
    masks = [v for k, v in pruned.state_dict().items() if k.endswith(".mask")]
    assert masks and all(m.dtype == torch.bool for m in masks)
    sparsity = 1 - sum(m.sum().item() for m in masks) / sum(m.numel() for m in masks)
    assert 0.37 < sparsity <= 0.40     # int(in*0.4)/in per row, e.g. 12/32


def test_registry_and_load_pruner_contract():
    from lavis.common.registry import registry
    from lavis.compression import load_pruner
    for n in ("t5_wanda_pruner", "vit_wanda_pruner", "blipt5_wanda_pruner"):
        assert registry.get_pruner_class(n) is not None
    with pytest.raises(SystemExit):                    # reference: TypeError -> message + exit(1)
        load_pruner("no_such_pruner", None, None, cfg={})


def test_pruner_has_no_cpu_fallback():
    """Without the test stand-ins a CPU model must raise, not silently compute on the host."""
    with pytest.raises(RuntimeError, match="GPU only"):
        H.run_pruner("fp32_r50", "cpu")
