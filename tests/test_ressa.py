"""RESSA retraining step: the drop-in `ImageTextRetrainTask._train_inner_loop` against
the REFERENCE task's run on the toy InstructBLIP (tests/golden/ressa.npz, made by tests/golden/make_golden.py).

CPU: the SparseLoRA kernels are GPU-only, so `vlmc.sparse_lora.linear` (and the Wanda ops) are monkeypatched with
the oracle's PyTorch expressions -- what is pinned, bit for bit, is the host sequence (dense no-grad pass, sparse
pass, KL mix, gradient accumulation, optimizer cadence, lr schedule calls).
GPU: the same loop on the HIP kernels (`vlmc_lora_effective_weight`, `vlmc_lora_grad`) tracks the reference's
losses and final adapter weights within fp32 tolerance."""
import pytest
import torch

import golden_io
import oracle_ops
import pruner_helpers as H

G = golden_io.load("ressa")


def _install_lora_oracle(monkeypatch):
    from oracle import sparse_lora as OL
    from vlmc import sparse_lora
    monkeypatch.setattr(sparse_lora, "linear",
                        lambda x, w, A, B, mask, bias, scaling, sparse: OL.forward(x, w, A, B, mask, bias, scaling, sparse))


def test_ressa_inner_loop_matches_reference_run_on_cpu(monkeypatch):
    oracle_ops.install(monkeypatch)
    _install_lora_oracle(monkeypatch)
    model, task, stats = H.run_ressa("cpu")
    assert torch.equal(torch.tensor(task.loss_history, dtype=torch.float64), G["losses"])
    assert stats["loss"] == str(G["stats_loss"])
    sd = model.state_dict()
    n = 0
    for key in [k for k in G if k.startswith("sd/")]:
        assert torch.equal(sd[key[3:]], G[key]), key
        n += 1
    assert n > 150
    # only the adapters moved; pruned positions of the frozen weights are still dense (lora_model=True)
    assert any(k.endswith("lora_B.weight") for k in sd)


def test_task_registry_and_step_contract():
    from lavis.common.registry import registry
    from lavis.tasks.image_text_retrain import ImageTextRetrainTask
    assert registry.get_task_class("image_text_retrain") is ImageTextRetrainTask
    t = ImageTextRetrainTask()
    assert t.kl_weight == 0.01 and t.T == 2.0                     # image_text_retrain.py:25-26


@pytest.mark.gpu
def test_ressa_inner_loop_on_gpu_tracks_reference_run():
    model, task, _ = H.run_ressa("cuda:0")
    got = torch.tensor(task.loss_history, dtype=torch.float64)
    assert torch.allclose(got, G["losses"], rtol=2e-3, atol=1e-6), (got, G["losses"])
    sd = model.state_dict()
    worst = 0.0
    for key in [k for k in G if k.startswith("sd/") and "lora_" in k]:
        ref, g = G[key], sd[key[3:]].cpu()
        worst = max(worst, float((g - ref).abs().max()))
    # AdamW normalises each step to ~lr: adapters moved by ~5 * 1e-2; agreement well inside that
    assert worst < 2e-3, worst
    for key in [k for k in G if k.startswith("sd/") and k.endswith(".mask")]:
        assert (sd[key[3:]].cpu() == G[key]).float().mean() > 0.99
