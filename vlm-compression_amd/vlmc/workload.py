"""Shapes of the BASELINE.json workloads.

`flan_t5_xl()` lists, block by block, the distinct linear inputs and the linears
that consume them for InstructBLIP-FlanT5-XL: 39 EVA ViT-g blocks (fp16, matrix-wide
Wanda rule), 24 T5 encoder and 24 T5 decoder blocks (bf16, per-row rule) -- 588
prunable linears, 3.70 G weights.  Token counts are the synthetic-replay sizes of
BASELINE.md config 2 (257 / 64 / 16).
"""
from __future__ import annotations

from dataclasses import dataclass, field

import torch


@dataclass
class LinearSpec:
    name: str
    out_features: int
    in_features: int


@dataclass
class InputSpec:
    """One distinct activation tensor [samples, tokens, in] feeding one or more linears."""
    name: str
    tokens: int
    in_features: int
    linears: list = field(default_factory=list)


@dataclass
class BlockSpec:
    name: str
    tower: str            # "vit" | "t5_enc" | "t5_dec" | "llm"
    dtype: torch.dtype
    mode: str             # Wanda unstructured rule for this tower: "matrix" (ViT) or "row" (LLM)
    inputs: list = field(default_factory=list)

    @property
    def linears(self):
        return [l for i in self.inputs for l in i.linears]


def vit_g_block(i, tokens=257, dim=1408, hidden=6144):
    b = BlockSpec(f"visual_encoder.blocks.{i}", "vit", torch.float16, "matrix")
    b.inputs = [
        InputSpec("norm1", tokens, dim, [LinearSpec("attn.qkv", 3 * dim, dim)]),
        InputSpec("attn_out", tokens, dim, [LinearSpec("attn.proj", dim, dim)]),
        InputSpec("norm2", tokens, dim, [LinearSpec("mlp.fc1", hidden, dim)]),
        InputSpec("fc1_act", tokens, hidden, [LinearSpec("mlp.fc2", dim, hidden)]),
    ]
    return b


def t5_block(i, decoder, tokens, enc_tokens=64, d_model=2048, d_ff=5120, dtype=torch.bfloat16):
    side = "decoder" if decoder else "encoder"
    b = BlockSpec(f"t5_model.{side}.block.{i}", "t5_dec" if decoder else "t5_enc", dtype, "row")
    sa = "layer.0.SelfAttention."
    b.inputs = [
        InputSpec("self_norm", tokens, d_model, [LinearSpec(sa + n, d_model, d_model) for n in "qkv"]),
        InputSpec("self_ctx", tokens, d_model, [LinearSpec(sa + "o", d_model, d_model)]),
    ]
    ff = "layer.1.DenseReluDense."
    if decoder:
        ca = "layer.1.EncDecAttention."
        ff = "layer.2.DenseReluDense."
        b.inputs += [
            InputSpec("cross_norm", tokens, d_model, [LinearSpec(ca + "q", d_model, d_model)]),
            InputSpec("encoder_states", enc_tokens, d_model, [LinearSpec(ca + n, d_model, d_model) for n in "kv"]),
            InputSpec("cross_ctx", tokens, d_model, [LinearSpec(ca + "o", d_model, d_model)]),
        ]
    b.inputs += [
        InputSpec("ff_norm", tokens, d_model, [LinearSpec(ff + "wi_0", d_ff, d_model), LinearSpec(ff + "wi_1", d_ff, d_model)]),
        InputSpec("ff_act", tokens, d_ff, [LinearSpec(ff + "wo", d_model, d_ff)]),
    ]
    return b


def flan_t5_xl(vit_tokens=257, enc_tokens=64, dec_tokens=16):
    """InstructBLIP-FlanT5-XL, in the order the reference prunes it: ViT, encoder, decoder
    (wanda_pruner.py:969-1031)."""
    blocks = [vit_g_block(i, vit_tokens) for i in range(39)]
    blocks += [t5_block(i, False, enc_tokens) for i in range(24)]
    blocks += [t5_block(i, True, dec_tokens, enc_tokens) for i in range(24)]
    return blocks


def count(blocks):
    lin = sum(len(b.linears) for b in blocks)
    weights = sum(l.out_features * l.in_features for b in blocks for l in b.linears)
    act = sum(i.tokens * i.in_features for b in blocks for i in b.inputs)
    return {"blocks": len(blocks), "linears": lin, "weights": weights, "act_elems_per_sample": act}


def select_bytes(lin: LinearSpec, elem_size=2, apply_zero=True):
    """Algorithmic HBM bytes of one select launch:
    read W + write bool mask + write zeroed W back + read sqrt(scaler_row)."""
    z = 1 if apply_zero else 0
    return lin.out_features * lin.in_features * (elem_size + 1 + z * elem_size) + 4 * lin.in_features


def stat_bytes(inp: InputSpec, samples, elem_size=2):
    """Algorithmic HBM bytes of the statistics pass over one distinct input tensor."""
    return samples * inp.tokens * inp.in_features * elem_size + samples * inp.in_features * 4
