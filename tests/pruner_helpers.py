"""Shared by the CPU host-logic test and the GPU end-to-end test of the drop-in pruners."""
import torch
import torch.nn as nn

import golden_io
import toy_models

E2E = {}

VARIANTS = {
    "fp32_r50": dict(vit_dtype=torch.float32, t5_dtype=torch.float32, ratio=0.5, n=0, m=0, lora=False),
    "mixed_2_4": dict(vit_dtype=torch.float32, t5_dtype=torch.bfloat16, ratio=0.5, n=2, m=4, lora=False),
    "fp32_r40_lora": dict(vit_dtype=torch.float32, t5_dtype=torch.float32, ratio=0.4, n=0, m=0, lora=True),
}


DSNOT_VARIANTS = {
    "fp32_r50": dict(vit_dtype=torch.float32, t5_dtype=torch.float32, ratio=0.5, n=0, m=0, lora=False,
                     kw=dict(max_cycle_time=20)),
    "mixed_2_4": dict(vit_dtype=torch.float32, t5_dtype=torch.bfloat16, ratio=0.5, n=2, m=4, lora=False,
                      kw=dict(max_cycle_time=4)),
    "fp32_r40_lora_mag": dict(vit_dtype=torch.float32, t5_dtype=torch.float32, ratio=0.4, n=0, m=0, lora=True,
                              kw=dict(max_cycle_time=16, initial_method="magnitude", update_threshold=0.02)),
}


def golden(which="wanda_e2e"):
    if which not in E2E:
        E2E[which] = golden_io.load(which)
    return E2E[which]


def wrap_lora(model, r=4, alpha=16):
    """Same replacement the golden generator did with the reference's Linear (weights shared,
    A/B from the same seeded stream)."""
    from lavis.peft.src.peft.tuners.lora import Linear
    g = torch.Generator().manual_seed(4242)
    for parent in list(model.modules()):
        for cname, child in list(parent.named_children()):
            if type(child) is nn.Linear and cname != "t5_proj":
                new = Linear(child.in_features, child.out_features, r=r, lora_alpha=alpha, bias=child.bias is not None)
                new.weight = child.weight
                if child.bias is not None:
                    new.bias = child.bias
                new.mask = torch.ones_like(child.weight.data).bool()
                with torch.no_grad():
                    new.lora_A.weight.copy_(torch.randn(new.lora_A.weight.shape, generator=g) * 0.05)
                    new.lora_B.weight.copy_(torch.randn(new.lora_B.weight.shape, generator=g) * 0.05)
                new.to(child.weight.dtype)
                setattr(parent, cname, new)
    return model


def build(name, device="cpu", variants=None, n_samples=6):
    v = (variants or VARIANTS)[name]
    model = toy_models.init_toy(toy_models.ToyBlipT5(vit_dtype=v["vit_dtype"], t5_dtype=v["t5_dtype"]), seed=7)
    if v["lora"]:
        wrap_lora(model)
    model.eval().to(device)
    batches = [{k: t.to(device) for k, t in b.items()} for b in toy_models.make_batches(n_samples, seed=11)]
    return model, batches, v


def run_pruner(name, device="cpu", n_samples=6):
    from lavis.compression import load_pruner
    model, batches, v = build(name, device, n_samples=n_samples)
    spec = "2-%r-1.0-1.0" % (1 - v["ratio"])
    cfg = dict(t5_prune_spec=spec, vit_prune_spec=spec, t5_pruning_method="wanda", vit_pruning_method="wanda",
               num_samples=n_samples, prune_n=v["n"], prune_m=v["m"], max_sparsity_per_layer=1.01)
    pruner = load_pruner("blipt5_wanda_pruner", model, batches, cfg=cfg)
    pruned, sd = pruner.prune(lora_model=True) if v["lora"] else pruner.prune()
    return pruned, sd


def run_dsnot_pruner(name, device="cpu", n_samples=6):
    from lavis.compression import load_pruner
    model, batches, v = build(name, device, DSNOT_VARIANTS, n_samples=n_samples)
    spec = "2-%r-1.0-1.0" % (1 - v["ratio"])
    cfg = dict(t5_prune_spec=spec, vit_prune_spec=spec, t5_pruning_method="dsnot", vit_pruning_method="dsnot",
               num_samples=n_samples, prune_n=v["n"], prune_m=v["m"], max_sparsity_per_layer=1.01, **v["kw"])
    pruner = load_pruner("blipt5_dsnot_pruner", model, batches, cfg=cfg)
    pruned, sd = pruner.prune(lora_model=True) if v["lora"] else pruner.prune()
    return pruned, sd


def compare_with_golden(name, pruned, exact=True, min_mask_agreement=1.0, weight_rtol=0.0, which="wanda_e2e"):
    G = golden(which)
    stats = {"masks": 0, "mask_elems": 0, "mask_diff": 0}
    sd = pruned.state_dict()
    for key in [k for k in G if k.startswith(f"{name}/sd/")]:
        k = key[len(name) + 4:]
        got, ref = sd[k].cpu(), G[key]
        if k.endswith("mask"):
            continue
        if exact:
            assert torch.equal(got, ref), f"{name}: state tensor {k} differs"
    for mn, mod in pruned.named_modules():
        gk = f"{name}/mask/{mn}"
        sk = f"{name}/sd/{mn}.mask"
        ref = G.get(gk, G.get(sk) if hasattr(mod, "lora_A") else None)
        if ref is None:
            continue
        got = mod.mask.cpu()
        stats["masks"] += 1
        stats["mask_elems"] += ref.numel()
        stats["mask_diff"] += int((got != ref).sum())
        ik = f"{name}/imp/{mn}"
        if ik in G:
            ref_imp = float(G[ik])
            assert abs(mod.weight.importance_score - ref_imp) <= 2e-3 * abs(ref_imp) + 1e-12, (mn, mod.weight.importance_score, ref_imp)
    assert stats["masks"] > 0
    agree = 1 - stats["mask_diff"] / stats["mask_elems"]
    assert agree >= min_mask_agreement, f"{name}: mask agreement {agree:.5f} < {min_mask_agreement}"
    return stats


# ---- RESSA retraining --------------------------------------------------------------
def run_ressa(device="cpu", iters=5):
    """Toy InstructBLIP + SparseLoRA (sparse=True) -> Wanda masks under lora_model -> the drop-in task's
    `_train_inner_loop`; the same script the golden generator ran with the reference's modules."""
    from lavis.compression import load_pruner
    from lavis.peft.src.peft.tuners.lora import mark_only_lora_as_trainable
    from lavis.tasks.image_text_retrain import ImageTextRetrainTask
    model = toy_models.init_toy(toy_models.ToyBlipT5(), seed=7)
    wrap_lora(model)
    for m in model.modules():
        if hasattr(m, "lora_A"):
            m.merge_weights = False
            m.sparse = True
    model.eval().to(device)
    batches = [{k: t.to(device) for k, t in b.items()} for b in toy_models.make_batches(6, seed=11)]
    spec = "2-0.5-1.0-1.0"
    cfg = dict(t5_prune_spec=spec, vit_prune_spec=spec, t5_pruning_method="wanda", vit_pruning_method="wanda",
               num_samples=6, max_sparsity_per_layer=1.01)
    model, _ = load_pruner("blipt5_wanda_pruner", model, batches, cfg=cfg).prune(lora_model=True)
    mark_only_lora_as_trainable(model)
    opt = torch.optim.AdamW([p for p in model.parameters() if p.requires_grad], lr=1e-2, weight_decay=0.05)

    class Sched:
        def step(self, cur_epoch, cur_step):
            for g in opt.param_groups:
                g["lr"] = 1e-2 * (0.9 ** cur_step)

    task = ImageTextRetrainTask()
    task.kl_weight = 0.1
    stats = task._train_inner_loop(epoch=0, iters_per_epoch=iters, model=model, data_loader=iter(batches), optimizer=opt,
                                   lr_scheduler=Sched(), scaler=None, log_freq=1, cuda_enabled=False, accum_grad_iters=2)
    return model, task, stats


# ---- 16-bit toy (fp16 ViT + bf16 T5): the dtypes of the real model, on which the batch-invariant MFMA forward engages --------
def run_16bit_toy(method, device, n_samples=8, ragged=False):
    """Whole `blipt5_<method>_pruner.prune()` on the toy InstructBLIP in the real model's dtypes; returns every state
    tensor, mask and importance score.  Ragged: three text lengths, interleaved."""
    from lavis.compression import load_pruner
    model = toy_models.init_toy(toy_models.ToyBlipT5(vit_dtype=torch.float16, t5_dtype=torch.bfloat16), seed=7).eval().to(device)
    lens = [5, 7, 5, 5, 7, 3, 5, 7]
    batches = []
    for j in range(n_samples):
        ln = lens[j % len(lens)]
        b = toy_models.make_batches(1, txt_len=ln if ragged else 5, out_len=(2 + ln % 3) if ragged else 4, seed=100 + j)[0]
        batches.append({k: t.to(device) for k, t in b.items()})
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
