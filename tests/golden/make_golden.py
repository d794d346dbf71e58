#!/usr/bin/env python
"""Generate the golden fixtures in tests/golden/*.npz by running the REFERENCE
(Shwai-He/VLM-Compression, mounted read-only at /root/reference) on small seeded
inputs.  Runs only in the build container; the GPU box and the test-suite never
import the reference -- they read the committed .npz files.

Import recipe: SURVEY.md Appendix C (namespace stub for `lavis`, stubs for the
missing `lavis.datasets.data_utils`, `omegaconf`, `bitsandbytes`).  No reference
source is copied; only inputs and outputs are stored.

    python tests/golden/make_golden.py            # all groups
    python tests/golden/make_golden.py wanda      # one group
"""
import copy
import importlib.machinery
import os
import sys
import types
from functools import partial

import numpy as np
import torch
import torch.nn as nn

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.dont_write_bytecode = True          # never leave __pycache__ inside /root/reference


def import_reference():
    if not os.path.isdir(REF):
        raise SystemExit("reference not mounted; golden fixtures can only be regenerated in the build container")

    def stub(name, **a):
        m = types.ModuleType(name)
        m.__spec__ = importlib.machinery.ModuleSpec(name, None)
        m.__dict__.update(a)
        sys.modules[name] = m
        return m

    stub("lavis").__path__ = [REF + "/lavis"]
    stub("lavis.datasets").__path__ = []
    stub("lavis.datasets.data_utils", prepare_sample=lambda s, cuda_enabled=True: s)
    stub("omegaconf", OmegaConf=type("OmegaConf", (), {}))
    b = stub("bitsandbytes")
    b.nn = stub("bitsandbytes.nn", Linear8bitLt=type("Linear8bitLt", (nn.Linear,), {}))
    import lavis.compression  # noqa: F401  (registers the pruners)
    torch.cuda.synchronize = lambda *a, **k: None      # sparsegpt_pruner.py:212 on a CPU-only box
    torch.cuda.empty_cache = lambda *a, **k: None


import golden_io  # noqa: E402
import toy_models  # noqa: E402


# --------------------------------------------------------------------------- #
# single-linear towers: the block input IS the linear input
# --------------------------------------------------------------------------- #
class OneLinearT5Block(nn.Module):
    def __init__(self, W):
        super().__init__()
        self.fc = nn.Linear(W.shape[1], W.shape[0], bias=False)
        self.fc.weight.data = W.clone()

    def forward(self, x, dense=False, **kw):
        return (self.fc(x),)


class OneLinearViTBlock(nn.Module):
    def __init__(self, W):
        super().__init__()
        self.fc = nn.Linear(W.shape[1], W.shape[0], bias=False)
        self.fc.weight.data = W.clone()

    def forward(self, x, rel_pos_bias=None, dense=False):
        return self.fc(x)


class OneLinearModel(nn.Module):
    def __init__(self, W, tower):
        super().__init__()
        import contextlib
        self._ctx = contextlib.nullcontext
        if tower == "t5":
            self.t5_model = nn.Module()
            self.t5_model.config = types.SimpleNamespace(use_cache=True)
            self.t5_model.encoder = nn.Module()
            self.t5_model.encoder.block = nn.ModuleList([OneLinearT5Block(W)])
        else:
            self.visual_encoder = nn.Module()
            self.visual_encoder.blocks = nn.ModuleList([OneLinearViTBlock(W)])
        self.tower = tower

    def maybe_autocast(self, dtype=None):
        return self._ctx()

    def forward(self, samples):
        if self.tower == "t5":
            kw = dict(attention_mask=None, position_bias=None, encoder_hidden_states=None,
                      encoder_attention_mask=None, encoder_decoder_position_bias=None, layer_head_mask=None,
                      cross_attn_layer_head_mask=None)
            return self.t5_model.encoder.block[0](samples["x"], **kw)
        return self.visual_encoder.blocks[0](samples["x"], None)


def ref_prune_one_linear(W, xs, tower, ratio=0.5, n=0, m=0):
    """Run the reference Wanda loop body on one linear with hook inputs `xs`."""
    from lavis.compression.pruners import wanda_pruner as R
    model = OneLinearModel(W, tower)
    loader = [{"x": x, "image": x, "text_input": [0] * x.shape[0]} for x in xs]
    pr = R.BLIPT5LayerWandaPruner(model=model, data_loader=loader, t5_prune_spec="1-%r-1.0-1.0" % (1 - ratio),
                                  vit_prune_spec="1-%r-1.0-1.0" % (1 - ratio), t5_pruning_method="wanda",
                                  vit_pruning_method="wanda", num_samples=len(xs), prune_n=n, prune_m=m,
                                  max_sparsity_per_layer=1.01)
    sd = pr.get_sparsity(ratio, sparsity_ratio_granularity=None)
    cls = R.T5LayerWandaPruner if tower == "t5" else R.VITLayerWandaPruner
    pr.prepare_calibration_input_encoder = partial(cls.prepare_calibration_input_encoder, pr)
    if tower == "t5":
        cls._prune(pr, model, loader, model_prefix="t5_model", module_to_process="t5_model.encoder.block",
                   n_samples=len(xs), sparsity_ratio=sd, lora_model=False)
        fc = model.t5_model.encoder.block[0].fc
    else:
        cls._prune(pr, model, loader, model_prefix="visual_encoder", module_to_process="visual_encoder.blocks",
                   n_samples=len(xs), sparsity_ratio=sd, lora_model=False)
        fc = model.visual_encoder.blocks[0].fc
    return fc.mask.clone(), fc.weight.data.clone(), float(fc.weight.importance_score)


def craft_weight(out_f, in_f, dtype, seed, ties=True):
    g = torch.Generator().manual_seed(seed)
    W = (torch.randn(out_f, in_f, generator=g) * 0.02).to(dtype)
    if ties:
        W[0, :] = W[0, 0]                         # whole row of equal |w| (ties decided by scale)
        W[1, ::3] = 0                             # exact zeros -> zero scores, index-ordered ties
        W[2, :] = 0                               # all-zero row
        W[3, 1::2] = -W[3, 0::2]                  # +/- pairs
        W[5:9, 4:12] = W[5:9, 4:5]                # equal blocks
    return W


def make_xs(n, T, in_f, dtype, seed, mean=0.1, equal_cols=True):
    g = torch.Generator().manual_seed(seed)
    xs = []
    for _ in range(n):
        x = (torch.randn(1, T, in_f, generator=g) + mean).to(dtype)
        if equal_cols:
            x[..., 8:16] = x[..., 8:9]            # identical channels -> identical scales -> ties across columns
        xs.append(x)
    return xs


# --------------------------------------------------------------------------- #
def gen_wanda():
    from lavis.compression.pruners import wanda_pruner as R
    out = {}
    # ---- G1: WrappedGPT.add_batch recurrence -----------------------------
    cases = [("bf16", torch.bfloat16, 16, 128, 8, 1), ("fp16", torch.float16, 5, 64, 6, 1),
             ("fp32", torch.float32, 16, 64, 8, 1), ("bf16_b2", torch.bfloat16, 7, 64, 4, 2),
             ("fp32_b3", torch.float32, 9, 32, 3, 3)]
    for name, dt, T, in_f, n, b in cases:
        g = torch.Generator().manual_seed(100 + len(out))
        layer = nn.Linear(in_f, 4, bias=False)
        w = R.WrappedGPT(layer)
        states = []
        for j in range(n):
            x = ((torch.randn(b, T, in_f, generator=g) * (1 + j % 3)) + 0.1).to(dt)
            out[f"g1/{name}/x{j}"] = x
            w.add_batch(x, None)
            states.append(w.scaler_row.clone())
        out[f"g1/{name}/states"] = torch.stack(states)
        out[f"g1/{name}/n"] = n
        out[f"g1/{name}/b"] = b
    # ---- G2: per-row select (T5/LLM rule) --------------------------------
    k = 0
    for name, dt, ratio in [("bf16_r50", torch.bfloat16, 0.5), ("bf16_r30", torch.bfloat16, 0.3),
                            ("fp16_r50", torch.float16, 0.5), ("fp32_r37", torch.float32, 0.37)]:
        W = craft_weight(48, 128, dt, seed=200 + k)
        xs = make_xs(8, 16, 128, dt, seed=300 + k)
        mask, Wn, imp = ref_prune_one_linear(W, xs, "t5", ratio=ratio)
        out.update({f"g2/{name}/W": W, f"g2/{name}/xs": torch.cat(xs), f"g2/{name}/ratio": ratio,
                    f"g2/{name}/mask": mask, f"g2/{name}/Wn": Wn, f"g2/{name}/imp": imp})
        k += 1
    # ---- G3: matrix-wide select (ViT rule) -------------------------------
    for name, dt, ratio in [("fp16_r50", torch.float16, 0.5), ("bf16_r25", torch.bfloat16, 0.25),
                            ("fp32_r60", torch.float32, 0.6)]:
        W = craft_weight(40, 96, dt, seed=400 + k)
        xs = make_xs(6, 9, 96, dt, seed=500 + k)
        mask, Wn, imp = ref_prune_one_linear(W, xs, "vit", ratio=ratio)
        out.update({f"g3/{name}/W": W, f"g3/{name}/xs": torch.cat(xs), f"g3/{name}/ratio": ratio,
                    f"g3/{name}/mask": mask, f"g3/{name}/Wn": Wn, f"g3/{name}/imp": imp})
        k += 1
    # ---- G4: n:m (tie-free weights; torch.topk tie order is implementation-defined)
    for name, dt, n, m, tower in [("bf16_2_4_t5", torch.bfloat16, 2, 4, "t5"), ("fp16_4_8_vit", torch.float16, 4, 8, "vit"),
                                  ("fp32_2_4_vit", torch.float32, 2, 4, "vit"), ("bf16_4_8_t5", torch.bfloat16, 4, 8, "t5")]:
        W = craft_weight(32, 64, dt, seed=600 + k, ties=False)
        xs = make_xs(5, 11, 64, dt, seed=700 + k, equal_cols=False)
        mask, Wn, imp = ref_prune_one_linear(W, xs, tower, n=n, m=m)
        out.update({f"g4/{name}/W": W, f"g4/{name}/xs": torch.cat(xs), f"g4/{name}/n": n, f"g4/{name}/m": m,
                    f"g4/{name}/mask": mask, f"g4/{name}/Wn": Wn, f"g4/{name}/imp": imp})
        k += 1
    golden_io.save("wanda_unit", out)
    print("wanda_unit.npz:", len(out), "arrays")


def _wrap_lora_reference(model, r=4, alpha=16):
    """Swap every prunable nn.Linear for the reference SparseLoRA Linear sharing the
    weight, like LoraModel._replace_module (lora.py:190-208)."""
    from lavis.peft.src.peft.tuners.lora import Linear as RefLoraLinear
    g = torch.Generator().manual_seed(4242)
    for parent in list(model.modules()):
        for cname, child in list(parent.named_children()):
            if type(child) is nn.Linear and cname != "t5_proj":
                new = RefLoraLinear(child.in_features, child.out_features, r=r, lora_alpha=alpha,
                                    bias=child.bias is not None)
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


def gen_wanda_e2e():
    """G8: whole-pruner runs on the toy InstructBLIP (ViT -> encoder -> decoder)."""
    from lavis.compression.pruners import wanda_pruner as R
    variants = {
        "fp32_r50": dict(vit_dtype=torch.float32, t5_dtype=torch.float32, ratio=0.5, n=0, m=0, lora=False),
        "mixed_2_4": dict(vit_dtype=torch.float32, t5_dtype=torch.bfloat16, ratio=0.5, n=2, m=4, lora=False),
        "fp32_r40_lora": dict(vit_dtype=torch.float32, t5_dtype=torch.float32, ratio=0.4, n=0, m=0, lora=True),
    }
    out = {}
    for name, v in variants.items():
        model = toy_models.init_toy(toy_models.ToyBlipT5(vit_dtype=v["vit_dtype"], t5_dtype=v["t5_dtype"]), seed=7)
        if v["lora"]:
            _wrap_lora_reference(model)
        model.eval()
        batches = toy_models.make_batches(6, seed=11)
        spec = "2-%r-1.0-1.0" % (1 - v["ratio"])
        pr = R.BLIPT5LayerWandaPruner(model=model, data_loader=batches, t5_prune_spec=spec, vit_prune_spec=spec,
                                      t5_pruning_method="wanda", vit_pruning_method="wanda", num_samples=6,
                                      prune_n=v["n"], prune_m=v["m"], max_sparsity_per_layer=1.01)
        pruned, _ = pr.prune(lora_model=True) if v["lora"] else pr.prune()
        for k_, t in pruned.state_dict().items():
            out[f"{name}/sd/{k_}"] = t
        for mn, mod in pruned.named_modules():
            if hasattr(mod, "mask") and "mask" not in dict(mod.named_buffers(recurse=False)):
                out[f"{name}/mask/{mn}"] = mod.mask
            if hasattr(mod, "weight") and hasattr(mod.weight, "importance_score"):
                out[f"{name}/imp/{mn}"] = float(mod.weight.importance_score)
        out[f"{name}/ratio"] = v["ratio"]
    golden_io.save("wanda_e2e", out)
    print("wanda_e2e.npz:", len(out), "arrays")


def gen_sparse_lora():
    """G7: the reference's SparseLoRA Linear -- forward (dense / sparse=True / sparse=False), the
    weight it hands to F.linear (captured by wrapping F.linear), grads of x, A, B, and merge()."""
    import contextlib
    from lavis.peft.src.peft.tuners import lora as RL
    out = {}
    in_f, out_f, r, alpha = 96, 80, 4, 16
    g = torch.Generator().manual_seed(77)
    base_W = torch.randn(out_f, in_f, generator=g) * 0.05
    base_b = torch.randn(out_f, generator=g) * 0.01
    A0 = torch.randn(r, in_f, generator=g) * 0.1
    B0 = torch.randn(out_f, r, generator=g) * 0.1
    M0 = torch.rand(out_f, in_f, generator=g) > 0.5
    X0 = torch.randn(2, 5, in_f, generator=g)
    GY = torch.randn(2, 5, out_f, generator=g)
    out.update({"W": base_W, "b": base_b, "A": A0, "B": B0, "M": M0, "X": X0, "GY": GY, "r": r, "alpha": alpha})
    cases = {"fp32": (torch.float32, None), "bf16": (torch.bfloat16, None), "bf16_autocast": (torch.bfloat16, torch.bfloat16),
             "fp32_autocast_bf16": (torch.float32, torch.bfloat16)}
    for cname, (wd, ac) in cases.items():
        for sparse in (True, False):
            lin = RL.Linear(in_f, out_f, r=r, lora_alpha=alpha, bias=True)
            with torch.no_grad():
                lin.weight.copy_(base_W); lin.bias.copy_(base_b)
                lin.lora_A.weight.copy_(A0); lin.lora_B.weight.copy_(B0)
            lin.weight.data = lin.weight.data.to(wd)
            lin.bias.data = lin.bias.data.to(wd)
            lin.mask = M0.clone()
            lin.sparse = sparse
            captured = {}
            real_linear = RL.F.linear

            def spy(x, w, bias=None):
                captured["w"] = w.detach().clone()
                return real_linear(x, w, bias)
            RL.F.linear = spy
            ctx = torch.autocast("cpu", dtype=ac) if ac is not None else contextlib.nullcontext()
            x = X0.to(wd if ac is None else torch.float32).clone().requires_grad_(True)
            with ctx:
                y = lin(x)
            RL.F.linear = real_linear
            y.backward(GY.to(y.dtype))
            key = f"{cname}/sparse{int(sparse)}"
            out.update({f"{key}/y": y.detach(), f"{key}/weff": captured["w"], f"{key}/gx": x.grad,
                        f"{key}/gA": lin.lora_A.weight.grad, f"{key}/gB": lin.lora_B.weight.grad})
            if ac is None:
                with torch.no_grad():
                    yd = lin(X0.to(wd), dense=True)
                out[f"{key}/y_dense"] = yd
                lin.merge()
                out[f"{key}/merged"] = lin.weight.data.clone()
                out[f"{key}/A_after_merge_is_reinit"] = int(not torch.equal(lin.lora_A.weight.data, A0))
                out[f"{key}/B_after_merge_is_zero"] = int(bool((lin.lora_B.weight.data == 0).all()))
        lin = RL.Linear(in_f, out_f, r=r, lora_alpha=alpha, bias=True)
        out["state_dict_keys"] = np.array(sorted(lin.state_dict().keys()))
    golden_io.save("sparse_lora", out)
    print("sparse_lora.npz:", len(out), "arrays")


def gen_sparsegpt():
    """G5: the reference's SparseGPT class -- Hessian after add_batch, fasterprune() for
    unstructured 0.5 / 0.3 and 2:4 / 4:8 on [32,256] (two 128-column blocks), a dead input
    channel, and a rank-deficient Hessian that exercises the damping loop."""
    from lavis.compression.pruners import sparsegpt_pruner as RS
    out = {}
    # 6 samples x 48 tokens = 288 > 256 columns: positive definite Hessians, except `rankdef`
    cases = [("fp32_u50", torch.float32, 0.5, 0, 0, 48, False), ("bf16_u30", torch.bfloat16, 0.3, 0, 0, 48, False),
             ("fp32_2_4", torch.float32, 0.5, 2, 4, 48, False), ("fp16_4_8", torch.float16, 0.5, 4, 8, 48, False),
             ("fp32_dead", torch.float32, 0.5, 0, 0, 48, True), ("fp32_rankdef", torch.float32, 0.5, 0, 0, 3, False)]
    for i, (name, dt, sparsity, n, m, T, dead) in enumerate(cases):
        g = torch.Generator().manual_seed(900 + i)
        lin = nn.Linear(256, 32, bias=False)
        W = (torch.randn(32, 256, generator=g) * 0.05).to(dt)
        lin.weight.data = W.clone()
        sg = RS.SparseGPT(lin)
        xs = []
        for j in range(6):
            # activations are fp16-representable so that the fixture stores them in 2 bytes
            x = ((torch.randn(1, T, 256, generator=g) + 0.1) * (1 + 0.2 * j)).half().to(dt)
            if dead:
                x[..., 17] = 0
                x[..., 200] = 0
            xs.append(x if dt != torch.float32 else x.half())
            sg.add_batch(x, None)
        if name in ("fp32_u50", "fp32_rankdef"):          # the others are re-derived (and verified) by the oracle
            out[f"{name}/H"] = sg.H.clone()
        out[f"{name}/W"] = W
        out[f"{name}/xs"] = torch.cat(xs)
        sg.fasterprune(sparsity, prune_n=n, prune_m=m, percdamp=0.01, blocksize=128)
        out[f"{name}/Wn"] = lin.weight.data.clone()
        out[f"{name}/imp"] = float(lin.weight.importance_score)
        out[f"{name}/sparsity"] = sparsity
        out[f"{name}/n"] = n
        out[f"{name}/m"] = m
    golden_io.save("sparsegpt", out)
    print("sparsegpt.npz:", len(out), "arrays")


def gen_sparsegpt_e2e():
    """Whole-pruner run of the reference's blipt5_sparsegpt_pruner on the toy InstructBLIP."""
    from lavis.compression.pruners import sparsegpt_pruner as RS
    out = {}
    for name, v in {"fp32_u50": dict(ratio=0.5, n=0, m=0), "fp32_2_4": dict(ratio=0.5, n=2, m=4)}.items():
        model = toy_models.init_toy(toy_models.ToyBlipT5(), seed=7).eval()
        batches = toy_models.make_batches(6, seed=11)
        spec = "2-%r-1.0-1.0" % (1 - v["ratio"])
        pr = RS.BLIPT5LayerSparseGPTPruner(model=model, data_loader=batches, t5_prune_spec=spec, vit_prune_spec=spec,
                                           t5_pruning_method="sparsegpt", vit_pruning_method="sparsegpt", num_samples=6,
                                           prune_n=v["n"], prune_m=v["m"], max_sparsity_per_layer=1.01)
        pruned, _ = pr.prune()
        for k_, t in pruned.state_dict().items():
            out[f"{name}/sd/{k_}"] = t
        for mn, mod in pruned.named_modules():
            if hasattr(mod, "weight") and hasattr(mod.weight, "importance_score"):
                out[f"{name}/imp/{mn}"] = float(mod.weight.importance_score)
    golden_io.save("sparsegpt_e2e", out)
    print("sparsegpt_e2e.npz:", len(out), "arrays")


def ref_dsnot_one_linear(W, xs, tower, ratio=0.5, n=0, m=0, **dsnot_kw):
    """Run the reference DSnoT loop body on one linear with hook inputs `xs`."""
    from lavis.compression.pruners import dsnot_pruner as R
    model = OneLinearModel(W, tower)
    loader = [{"x": x, "image": x, "text_input": [0] * x.shape[0]} for x in xs]
    pr = R.BLIPT5LayerDSnoTPruner(model=model, data_loader=loader, t5_prune_spec="1-%r-1.0-1.0" % (1 - ratio),
                                  vit_prune_spec="1-%r-1.0-1.0" % (1 - ratio), t5_pruning_method="dsnot",
                                  vit_pruning_method="dsnot", num_samples=len(xs), prune_n=n, prune_m=m,
                                  max_sparsity_per_layer=1.01, **dsnot_kw)
    sd = pr.get_sparsity(ratio, ratio, sparsity_ratio_granularity=None)
    cls = R.T5LayerDSnoTPruner if tower == "t5" else R.VITLayerDSnoTPruner
    pr.prepare_calibration_input_encoder = partial(cls.prepare_calibration_input_encoder, pr)
    if tower == "t5":
        cls._prune(pr, model, loader, "cpu", model_prefix="t5_model", module_to_process="t5_model.encoder.block",
                   n_samples=len(xs), sparsity_ratio=sd, lora_model=False)
        fc = model.t5_model.encoder.block[0].fc
    else:
        cls._prune(pr, model, loader, "cpu", model_prefix="visual_encoder", module_to_process="visual_encoder.blocks",
                   n_samples=len(xs), sparsity_ratio=sd, lora_model=False)
        fc = model.visual_encoder.blocks[0].fc
    return fc.mask.clone(), fc.weight.data.clone()


def gen_dsnot():
    from lavis.compression.pruners import dsnot_pruner as R
    out = {}
    # ---- statistics (DSnoT WrappedGPT.add_batch) ------------------------------------------
    for name, dt, T, in_f, ncalls, b in [("bf16", torch.bfloat16, 16, 128, 6, 1), ("fp16_b2", torch.float16, 7, 64, 4, 2),
                                         ("fp32", torch.float32, 12, 64, 5, 1)]:
        g = torch.Generator().manual_seed(40 + len(out))
        w = R.WrappedGPT(nn.Linear(in_f, 4, bias=False))
        for j in range(ncalls):
            x = ((torch.randn(b, T + j, in_f, generator=g) * (1 + j % 3)) + 0.2).to(dt)
            out[f"stat/{name}/x{j}"] = x
            w.add_batch(x, None)
            out[f"stat/{name}/scaler{j}"] = w.scaler_row.clone()
            out[f"stat/{name}/sum{j}"] = w.sum_metric_row.clone()
            out[f"stat/{name}/var{j}"] = w.var.flatten().clone()
        out[f"stat/{name}/n"] = ncalls
    # ---- per-linear pruning ----------------------------------------------------------------
    cases = [
        ("t5_bf16_r50", "t5", torch.bfloat16, 40, 256, 0.5, 0, 0, {}),
        ("t5_fp32_r30_mag", "t5", torch.float32, 24, 256, 0.3, 0, 0, dict(initial_method="magnitude")),
        ("t5_bf16_plain", "t5", torch.bfloat16, 24, 256, 0.5, 0, 0, dict(without_DSnoT=True)),
        ("t5_fp16_walk", "t5", torch.float16, 32, 64, 0.5, 0, 0, dict(max_cycle_time=24)),
        ("t5_fp32_samesign", "t5", torch.float32, 24, 256, 0.4, 0, 0, dict(without_same_sign=False, update_threshold=0.02)),
        ("vit_fp16_r50", "vit", torch.float16, 32, 256, 0.5, 0, 0, {}),
        ("t5_bf16_2_4", "t5", torch.bfloat16, 32, 256, 0.5, 2, 4, {}),
        ("vit_fp32_4_8", "vit", torch.float32, 24, 256, 0.5, 4, 8, dict(max_cycle_time=40)),
    ]
    for i, (name, tower, dt, out_f, in_f, ratio, n, m, kw) in enumerate(cases):
        W = craft_weight(out_f, in_f, dt, seed=1200 + i, ties=(n == 0))
        xs = make_xs(6, 12, in_f, dt, seed=1300 + i, mean=0.3, equal_cols=(n == 0))
        mask, Wn = ref_dsnot_one_linear(W, xs, tower, ratio=ratio, n=n, m=m, **kw)
        out.update({f"{name}/W": W, f"{name}/xs": torch.cat(xs), f"{name}/mask": mask, f"{name}/Wn": Wn,
                    f"{name}/ratio": ratio, f"{name}/n": n, f"{name}/m": m, f"{name}/tower": np.array(tower)})
        for k_, v_ in kw.items():
            out[f"{name}/kw/{k_}"] = np.array(v_)
    golden_io.save("dsnot", out)
    print("dsnot.npz:", len(out), "arrays")


def gen_dsnot_e2e():
    """Whole-pruner run of the reference's blipt5_dsnot_pruner on the toy InstructBLIP."""
    from lavis.compression.pruners import dsnot_pruner as RD
    variants = {
        "fp32_r50": dict(t5_dtype=torch.float32, ratio=0.5, n=0, m=0, lora=False, kw=dict(max_cycle_time=20)),
        "mixed_2_4": dict(t5_dtype=torch.bfloat16, ratio=0.5, n=2, m=4, lora=False, kw=dict(max_cycle_time=4)),
        "fp32_r40_lora_mag": dict(t5_dtype=torch.float32, ratio=0.4, n=0, m=0, lora=True,
                                  kw=dict(max_cycle_time=16, initial_method="magnitude", update_threshold=0.02)),
    }
    out = {}
    for name, v in variants.items():
        model = toy_models.init_toy(toy_models.ToyBlipT5(vit_dtype=torch.float32, t5_dtype=v["t5_dtype"]), seed=7)
        if v["lora"]:
            _wrap_lora_reference(model)
        model.eval()
        batches = toy_models.make_batches(6, seed=11)
        spec = "2-%r-1.0-1.0" % (1 - v["ratio"])
        pr = RD.BLIPT5LayerDSnoTPruner(model=model, data_loader=batches, t5_prune_spec=spec, vit_prune_spec=spec,
                                       t5_pruning_method="dsnot", vit_pruning_method="dsnot", num_samples=6,
                                       prune_n=v["n"], prune_m=v["m"], max_sparsity_per_layer=1.01, **v["kw"])
        pruned, _ = pr.prune(lora_model=True) if v["lora"] else pr.prune()
        for k_, t in pruned.state_dict().items():
            out[f"{name}/sd/{k_}"] = t
        for mn, mod in pruned.named_modules():
            if hasattr(mod, "mask") and "mask" not in dict(mod.named_buffers(recurse=False)):
                out[f"{name}/mask/{mn}"] = mod.mask
        out[f"{name}/ratio"] = v["ratio"]
    golden_io.save("dsnot_e2e", out)
    print("dsnot_e2e.npz:", len(out), "arrays")


def gen_ressa():
    """RESSA retraining (image_text_retrain.py:95-203) on the toy InstructBLIP: Wanda masks under lora_model,
    then the reference task's `_train_inner_loop` for a few AdamW steps on CPU, fp32, no AMP."""
    from lavis.compression.pruners import wanda_pruner as R

    def stub(name, **a):
        m = types.ModuleType(name)
        m.__spec__ = importlib.machinery.ModuleSpec(name, None)
        m.__dict__.update(a)
        sys.modules[name] = m
        return m
    # import the task module without lavis/tasks/__init__.py (which pulls in every task and dataset) and without timm
    stub("timm"); stub("timm.models"); stub("timm.models.hub")
    stub("lavis.tasks").__path__ = [REF + "/lavis/tasks"]
    from lavis.tasks.image_text_retrain import ImageTextRetrainTask
    from lavis.peft.src.peft.tuners.lora import mark_only_lora_as_trainable
    out = {}
    model = toy_models.init_toy(toy_models.ToyBlipT5(), seed=7)
    _wrap_lora_reference(model)
    for m in model.modules():
        if hasattr(m, "lora_A"):
            m.merge_weights = False                      # LoraConfig default (lora.py:68-71)
            m.sparse = True                              # train.py:607-609 with --sparse
    model.eval()
    batches = toy_models.make_batches(6, seed=11)
    spec = "2-0.5-1.0-1.0"
    pr = R.BLIPT5LayerWandaPruner(model=model, data_loader=batches, t5_prune_spec=spec, vit_prune_spec=spec,
                                  t5_pruning_method="wanda", vit_pruning_method="wanda", num_samples=6,
                                  max_sparsity_per_layer=1.01)
    model, _ = pr.prune(lora_model=True)
    mark_only_lora_as_trainable(model)
    params = [p for p in model.parameters() if p.requires_grad]
    opt = torch.optim.AdamW(params, lr=1e-2, weight_decay=0.05)

    class Sched:
        def step(self, cur_epoch, cur_step):
            for gparam in opt.param_groups:
                gparam["lr"] = 1e-2 * (0.9 ** cur_step)

    task = ImageTextRetrainTask()
    task.kl_weight = 0.1
    losses = []
    orig_backward = torch.Tensor.backward

    def spy(self, *a, **k):
        losses.append(float(self.detach()))
        return orig_backward(self, *a, **k)
    torch.Tensor.backward = spy
    try:
        stats = task._train_inner_loop(epoch=0, iters_per_epoch=5, model=model, data_loader=iter(batches), optimizer=opt,
                                       lr_scheduler=Sched(), scaler=None, log_freq=1, cuda_enabled=False,
                                       accum_grad_iters=2)
    finally:
        torch.Tensor.backward = orig_backward
    out["losses"] = torch.tensor(losses, dtype=torch.float64)
    out["stats_loss"] = np.array(stats["loss"])
    for k_, t in model.state_dict().items():
        out[f"sd/{k_}"] = t
    golden_io.save("ressa", out)
    print("ressa.npz:", len(out), "arrays; losses", losses)


def gen_ecoflap():
    """ECoFLaP first stage (LayerSparsity, layer_single_base_pruner.py:111-728) + the Wanda / DSnoT prune it steers,
    on the toy InstructBLIP: the sparsity dict and the final masks of the reference."""
    import numpy.random as npr
    from lavis.compression.pruners import dsnot_pruner as RD
    from lavis.compression.pruners import wanda_pruner as R
    variants = {
        "wanda_block_aobd_sum": dict(cls="wanda", gran="block", score="aobd_sum", kw={}),
        "wanda_layer_obd_avg": dict(cls="wanda", gran="layer", score="obd_avg", kw={}),
        "wanda_model_gradient_sum": dict(cls="wanda", gran="model", score="gradient_sum", kw={}),
        "wanda_block_olmezo": dict(cls="wanda", gran="block", score="olmezo-gradient_sum", kw=dict(num_noise=2)),
        "wanda_block_olmezo_aobd": dict(cls="wanda", gran="block", score="olmezo-aobd_sum", kw=dict(num_noise=2)),
        "wanda_layer_lmezo_obd": dict(cls="wanda", gran="layer", score="lmezo-obd_sum", kw={}),
        "dsnot_block_per_model": dict(cls="dsnot", gran="block", score="aobd_sum", kw=dict(prune_per_model=True, max_cycle_time=4)),
    }
    out = {}
    for name, v in variants.items():
        torch.manual_seed(0)
        npr.seed(1234)
        model = toy_models.init_toy(toy_models.ToyBlipT5(), seed=7).eval()
        batches = toy_models.make_batches(6, seed=11)
        spec = "2-0.5-1.0-1.0"
        cls = R.BLIPT5LayerWandaPruner if v["cls"] == "wanda" else RD.BLIPT5LayerDSnoTPruner
        pr = cls(model=model, data_loader=batches, t5_prune_spec=spec, vit_prune_spec=spec, t5_pruning_method=v["cls"],
                 vit_pruning_method=v["cls"], num_samples=6, max_sparsity_per_layer=0.8, score_method=v["score"],
                 sparsity_ratio_granularity=v["gran"], num_data_first_stage=4, **v["kw"])
        pruned, sd = pr.prune()
        keys = sorted(sd.keys())
        out[f"{name}/keys"] = np.array(keys)
        out[f"{name}/sparsity"] = torch.tensor([float(sd[k]) for k in keys], dtype=torch.float64)
        for mn, mod in pruned.named_modules():
            if hasattr(mod, "mask") and "mask" not in dict(mod.named_buffers(recurse=False)):
                out[f"{name}/mask/{mn}"] = mod.mask
    golden_io.save("ecoflap", out)
    print("ecoflap.npz:", len(out), "arrays")


def gen_global():
    """Global pruners (global_pruner.py:49-383) on the toy InstructBLIP: final weights of the reference's
    blipt5_{mag,rand,aobd,mezo}_pruner for the three scope layouts, iterative pruning, and `get_mask` with a
    per-layer cap on explicit scores."""
    import numpy.random as npr
    from lavis.compression.pruners import global_pruner as RG
    variants = {
        "mag_global": dict(cls=RG.BLIPT5MagPruner, kw=dict(is_global=True)),
        "mag_per_model_it2": dict(cls=RG.BLIPT5MagPruner, kw=dict(is_global=True, prune_per_model=True, iteration=2)),
        "mag_layerwise_mixed": dict(cls=RG.BLIPT5MagPruner, kw=dict(is_global=False), t5_dtype=torch.bfloat16),
        "rand_global": dict(cls=RG.BLIPT5RandPruner, kw=dict(is_global=True)),
        "aobd_global": dict(cls=RG.BLIPT5AOBDPruner, kw=dict(is_global=True)),
        "aobd_layerwise_it2": dict(cls=RG.BLIPT5AOBDPruner, kw=dict(is_global=False, iteration=2)),
        "mezo_global": dict(cls=RG.BLIPT5AMeZoPruner, kw=dict(is_global=True, num_noise=2)),
    }
    out = {}
    for name, v in variants.items():
        torch.manual_seed(0)
        npr.seed(1234)
        model = toy_models.init_toy(toy_models.ToyBlipT5(t5_dtype=v.get("t5_dtype", torch.float32)), seed=7).eval()
        batches = toy_models.make_batches(6, seed=11)
        spec = "2-0.6-1.0-1.0"
        pr = v["cls"](model=model, data_loader=batches, t5_prune_spec=spec, vit_prune_spec=spec, num_samples=4, **v["kw"])
        pruned, _ = pr.prune()
        for k_, p_ in pruned.named_parameters():
            if p_.dim() == 2 and ".block" in k_:
                out[f"{name}/{k_}"] = p_.data.clone()
    # get_mask / get_layerwise_mask on explicit scores, with the per-layer cap active
    g = torch.Generator().manual_seed(3)
    scores = {f"layer{i}": torch.randn(8 + 4 * i, 16, generator=g) * (1 + i) for i in range(4)}
    scores["layer1"] = (scores["layer1"] * 2).round() / 2                     # ties
    pr = RG.BLIPT5MagPruner(model=toy_models.ToyBlipT5(), data_loader=[])
    for k_, t in scores.items():
        out[f"get_mask/scores/{k_}"] = t.clone()
    for k_, t in pr.get_mask({a: b.clone() for a, b in scores.items()}, 0.5, 0.6).items():
        out[f"get_mask/capped/{k_}"] = t
    for k_, t in pr.get_mask({a: b.clone() for a, b in scores.items()}, 0.3, 1.0).items():
        out[f"get_mask/uncapped/{k_}"] = t
    for k_, t in pr.get_layerwise_mask({a: b.clone() for a, b in scores.items()}, 0.45).items():
        out[f"get_mask/layerwise/{k_}"] = t
    golden_io.save("global", out)
    print("global.npz:", len(out), "arrays")


def gen_vicuna_e2e():
    """The `llm_model...model.layers` branch (wanda_pruner.py:233-236,1031-1038; dsnot_pruner.py likewise) on a toy
    InstructBLIP-Vicuna: BLIPT5 Wanda (50 % per row, 2:4 on an fp16 language model) and DSnoT pruners with
    `t5_model_prefix="llm_model"` -> final weights and masks."""
    from lavis.compression.pruners import dsnot_pruner as RD
    from lavis.compression.pruners import wanda_pruner as R
    variants = {
        "wanda_r50": dict(cls=R.BLIPT5LayerWandaPruner, method="wanda", llm_dtype=torch.float32, kw={}),
        "wanda_2_4_bf16": dict(cls=R.BLIPT5LayerWandaPruner, method="wanda", llm_dtype=torch.bfloat16, kw=dict(prune_n=2, prune_m=4)),
        "dsnot_r50": dict(cls=RD.BLIPT5LayerDSnoTPruner, method="dsnot", llm_dtype=torch.float32, kw=dict(max_cycle_time=12)),
    }
    out = {}
    for name, v in variants.items():
        torch.manual_seed(0)
        model = toy_models.init_toy(toy_models.ToyBlipVicuna(llm_dtype=v["llm_dtype"]), seed=5).eval()
        batches = toy_models.make_batches(6, seed=13)
        spec = "2-0.5-1.0-1.0"
        pr = v["cls"](model=model, data_loader=batches, t5_prune_spec=spec, vit_prune_spec=spec, t5_pruning_method=v["method"],
                      vit_pruning_method=v["method"], num_samples=6, t5_model_prefix="llm_model", max_sparsity_per_layer=1.01,
                      **v["kw"])
        pruned, _ = pr.prune()
        for k_, t in pruned.state_dict().items():
            out[f"{name}/sd/{k_}"] = t
        for mn, mod in pruned.named_modules():
            if hasattr(mod, "mask") and torch.is_tensor(mod.mask):
                out[f"{name}/mask/{mn}"] = mod.mask
    golden_io.save("vicuna_e2e", out)
    print("vicuna_e2e.npz:", len(out), "arrays")


def gen_nm_ties():
    """n:m selection with TIES inside the m-groups (wanda_pruner.py:326-329: `torch.topk(tmp, n, dim=1, largest=False)`).  Which
    of several equal scores torch.topk returns is implementation-defined: the reference's CPU answer (this container's torch;
    its CUDA kernel differs again) is RECORDED here, so that the repo's own policy -- lowest column first, the stable order of
    the per-row rule -- is held against it explicitly instead of being avoided (tests/test_nm_ties.py)."""
    out = {}
    k = 0
    for name, dt, n, m, tower in [("bf16_2_4_t5", torch.bfloat16, 2, 4, "t5"), ("fp16_4_8_vit", torch.float16, 4, 8, "vit"),
                                  ("fp32_2_4_vit", torch.float32, 2, 4, "vit"), ("bf16_4_8_t5", torch.bfloat16, 4, 8, "t5")]:
        W = craft_weight(32, 64, dt, seed=900 + k, ties=True)
        W[9, :] = 0.01                                 # a whole row of one value
        W[10, 0::4] = W[10, 1::4]                       # pairs inside every 4-group
        xs = make_xs(5, 11, 64, dt, seed=950 + k, equal_cols=True)
        for x in xs:
            x[..., 32:64] = x[..., 32:33]              # 32 identical channels: every group there ties on the activation side
        mask, Wn, imp = ref_prune_one_linear(W, xs, tower, n=n, m=m)
        out.update({f"{name}/W": W, f"{name}/xs": torch.cat(xs), f"{name}/n": n, f"{name}/m": m,
                    f"{name}/mask": mask, f"{name}/Wn": Wn, f"{name}/imp": imp})
        k += 1
    # DSnoT n:m (dsnot_pruner.py:407-552): a walk long enough to come back to m-groups both of whose kept entries were
    # already swapped out (their metrics sit at rowmax + 1): `torch.topk(pruning_block, 1, largest=False)` (:517-519) then
    # picks one of two EQUAL values
    for name, tower, dt, out_f in [("dsnot_t5_fp32_2_4", "t5", torch.float32, 32), ("dsnot_vit_bf16_2_4", "vit", torch.bfloat16, 24)]:
        W = craft_weight(out_f, 64, dt, seed=990 + k, ties=False)
        xs = make_xs(6, 12, 64, dt, seed=995 + k, mean=0.3, equal_cols=False)
        kw = dict(max_cycle_time=19, update_threshold=0.0)
        mask, Wn = ref_dsnot_one_linear(W, xs, tower, ratio=0.5, n=2, m=4, **kw)
        out.update({f"{name}/W": W, f"{name}/xs": torch.cat(xs), f"{name}/mask": mask, f"{name}/Wn": Wn, f"{name}/n": 2, f"{name}/m": 4,
                    f"{name}/tower": np.array(tower)})
        for k_, v_ in kw.items():
            out[f"{name}/kw/{k_}"] = np.array(v_)
        k += 1
    golden_io.save("nm_ties", out)
    print("nm_ties.npz:", len(out), "arrays")


def _main_statements(path, keep):
    """The top-level statements of `main()` in one of the reference's drivers for which `keep(node, source_segment)` holds,
    compiled from the file where it lies (nothing of it is stored): the save block of train.py and the reload blocks of
    evaluate_new.py are straight-line code over `args`, `model`, `job_id`, ... and can be driven on a toy model."""
    import ast
    src = open(path).read()
    tree = ast.parse(src)
    main = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "main")
    picked = [n for n in main.body if keep(n, ast.get_source_segment(src, n) or "")]
    assert picked, path
    return compile(ast.Module(body=picked, type_ignores=[]), path, "exec")


def gen_formats():
    """On-disk formats: the artefacts the reference's OWN save block writes (train.py:677-714) for a toy
    InstructBLIP whose towers went through the reference's `get_peft_model` (train.py:413-486), and what the reference's OWN
    reload blocks (evaluate_new.py:226-276) make of that checkpoint -- key lists, dtypes, shapes, yaml texts, reloaded tensors.
    Also `return_reorder_indice` on its docstring example and more inputs (dsnot_pruner.py:1881-1925)."""
    import tempfile
    import time
    import types as _types
    from lavis.peft.src.peft.mapping import get_peft_model
    from lavis.peft.src.peft.tuners.lora import LoraConfig
    out = {}
    model = toy_models.init_toy(toy_models.ToyBlipT5(), seed=3)
    model.t5_model.config.to_dict = lambda: {"model_type": "t5"}       # what get_peft_model asks a tower for (mapping.py:195-196)
    model.visual_encoder.model_type = "vit"
    model.t5_model = get_peft_model(model.t5_model, LoraConfig(r=4, lora_alpha=16, target_modules=[".q", ".k", ".v", ".o", ".wi_0", "wi_1", "wo"],
                                                               lora_dropout=0.0, bias="none", task_type="CAUSAL_LM"))
    model.visual_encoder = get_peft_model(model.visual_encoder, LoraConfig(r=2, lora_alpha=16, target_modules=[".qkv", ".proj", ".fc1", ".fc2"],
                                                                           lora_dropout=0.0, bias="none", task_type="ViT"))
    g = torch.Generator().manual_seed(11)
    k = 0
    for name, mod in model.named_modules():
        if hasattr(mod, "lora_A") and hasattr(mod, "mask"):
            mod.mask = torch.rand(mod.weight.shape, generator=g) > 0.5          # what prune(lora_model=True) leaves (wanda_pruner.py:339)
            mod.weight.importance_score = float(k) * 0.25 + 0.125               # (:320)
            with torch.no_grad():
                mod.lora_B.weight.copy_(torch.randn(mod.lora_B.weight.shape, generator=g) * 0.01)
            k += 1
    sparsity_dict = {"t5_model.base_model.model.encoder.block.0.layer.0.SelfAttention.q.weight": 0.5, "visual_encoder.blocks.1.mlp.fc2.weight": 0.25}
    save_block = _main_statements(REF + "/train.py", lambda n, seg: seg.startswith("if args.save_pruned_model"))
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:
        os.chdir(tmp)
        try:
            import yaml
            env = {"args": _types.SimpleNamespace(save_pruned_model=True, pruning_method="blipt5_wanda_pruner"), "model": model, "job_id": "job42",
                   "sparsity_dict": sparsity_dict, "start": time.time() - 1.5, "os": os, "torch": torch, "time": time}
            saved_cuda = torch.cuda.max_memory_allocated
            torch.cuda.max_memory_allocated = lambda *a, **k: 3 * 1024 ** 3        # (a CPU-only box)
            exec(save_block, env)
            torch.cuda.max_memory_allocated = saved_cuda
            listing = sorted(os.path.join(d, f)[2:] for d, _, fs in os.walk(".") for f in fs)
            out["files"] = np.array(listing)
            ckpt = "pruned_checkpoint/V+L/blipt5_wanda_pruner/job42.pth"
            state = torch.load(ckpt)
            out["ckpt/keys"] = np.array(list(state.keys()))
            out["ckpt/dtypes"] = np.array([str(v.dtype) for v in state.values()])
            out["ckpt/shapes"] = np.array([",".join(map(str, v.shape)) for v in state.values()])
            out["sparsity_yaml"] = np.array(open("sparsity_dict/job42.yaml").read())
            stats = yaml.safe_load(open("training_statistics/job42.yaml"))
            out["stats/keys"] = np.array(sorted(stats))
            out["stats/memory"] = float(stats["memory"])
            scores = torch.load("importance_scores/job42.pth")
            out["scores/keys"] = np.array(list(scores.keys()))
            out["scores/values"] = np.array([float(v) for v in scores.values()])
            # ---- the reference's reload of that checkpoint into a fresh, unwrapped model --------------------------------
            fresh = toy_models.init_toy(toy_models.ToyBlipT5(), seed=99)
            eva = _types.ModuleType("lavis.models.eva_vit")
            eva.interpolate_pos_embed = lambda m, sd: None                     # (timm-based file; positions do not change here)
            sys.modules.setdefault("lavis.models", _types.ModuleType("lavis.models"))
            sys.modules["lavis.models.eva_vit"] = eva
            reload_blocks = _main_statements(REF + "/evaluate_new.py", lambda n, seg: seg.startswith("if args.t5_pruned_checkpoint is not None")
                                             or seg.startswith("if args.vit_pruned_checkpoint is not None"))
            exec(reload_blocks, {"args": _types.SimpleNamespace(t5_pruned_checkpoint=ckpt, vit_pruned_checkpoint=ckpt), "model": fresh, "torch": torch,
                                 "print": lambda *a, **k: None})
            for key, v in fresh.state_dict().items():
                out["reloaded/" + key] = v
            for key, v in state.items():
                out["ckpt/tensor/" + key] = v
        finally:
            os.chdir(cwd)
    # ---- return_reorder_indice ----------------------------------------------------------------------------------------------
    from lavis.compression.pruners.dsnot_pruner import return_reorder_indice
    doc = torch.tensor([[-2.0, 1.0, -3.0, 4.0, 0.5], [1.0, -1.0, 2.0, -2.0, 3.0]])
    cases = {"doc_like": doc, "zeros_and_ties": torch.tensor([[0.0, -1.0, 0.0, 2.0, -1.0, 2.0], [3.0, 0.0, 0.0, 0.0, -4.0, 1.0]]),
             "all_negative": -torch.rand(3, 7, generator=g) - 0.1, "all_positive": torch.rand(3, 7, generator=g) + 0.1,
             "random": torch.randn(8, 33, generator=g)}
    cases["docstring"] = torch.tensor([[1., -2., 3.], [-2., 2., -4.], [5., 6., -7.], [-6., -7., -4.]])      # the example of :1883-1892
    for name, t in cases.items():
        out[f"reorder/{name}/in"] = t
        out[f"reorder/{name}/out"] = return_reorder_indice(t.clone())
    golden_io.save("formats", out)
    print("formats.npz:", len(out), "arrays")


GROUPS = {"formats": gen_formats, "nm_ties": gen_nm_ties, "wanda": gen_wanda, "wanda_e2e": gen_wanda_e2e, "sparse_lora": gen_sparse_lora, "sparsegpt": gen_sparsegpt,
          "sparsegpt_e2e": gen_sparsegpt_e2e, "dsnot": gen_dsnot,
          "dsnot_e2e": gen_dsnot_e2e, "ressa": gen_ressa, "ecoflap": gen_ecoflap, "global": gen_global, "vicuna_e2e": gen_vicuna_e2e}

if __name__ == "__main__":
    import_reference()
    torch.set_num_threads(1)
    todo = sys.argv[1:] or list(GROUPS)
    for gname in todo:
        GROUPS[gname]()
