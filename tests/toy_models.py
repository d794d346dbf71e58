"""Toy InstructBLIP-shaped model that honours the block call contract of the
reference's pruners:

* `model.visual_encoder.blocks[i](x, rel_pos_bias, dense=...) -> tensor`
  (eva_vit.Block contract, lavis/models/eva_vit.py:193-208)
* `model.t5_model.{encoder,decoder}.block[i](hidden, attention_mask=..., position_bias=...,
  encoder_hidden_states=..., ..., dense=...) -> (hidden,)`
  (T5Block contract, lavis/models/blip2_models/modeling_t5.py:748-763)
* `model.maybe_autocast(dtype=None)`, `model.t5_model.config.use_cache`
* `model(samples, vit_dense=..., llm_dense=...)` (blip2_t5_instruct.py:136-221)

Used by tests/golden/make_golden.py (driving the *reference* pruners) and by the
parity tests (driving this repo's drop-in pruners), so both see the same model.
All linears are bias-free or biased `nn.Linear`; when wrapped by SparseLoRA the
blocks forward the `dense` flag to them like the reference model files do.
"""
from __future__ import annotations

import contextlib
import types

import torch
import torch.nn as nn
import torch.nn.functional as F


def _lin(mod, x, dense):
    """Call a linear the way the reference model code does: plain nn.Linear gets
    `lin(x)`; a SparseLoRA Linear gets `lin(x, dense=dense)` (eva_vit.py:56,138)."""
    if type(mod) is nn.Linear:
        return mod(x)
    return mod(x, dense=dense)


class ToyAttention(nn.Module):
    # False (the goldens were generated this way): explicit fp32 matmul + softmax, like the reference's T5 / EVA attention.
    # True: F.scaled_dot_product_attention in the input dtype -- per (sample, head) by construction, which a batched
    # `torch.matmul` is not (the GEMM library picks its kernel by batch count); the batch-invariance tests set it.
    # "matmul16": batched matmuls in the activation dtype, the reference's own op sequence -- on the GPU these run on
    # vlmc_attn_matmul during the replay (vlmc/forward.py: invariant_matmuls).
    use_sdpa = False

    def __init__(self, dim, heads, fused_qkv, cross=False):
        super().__init__()
        self.heads, self.fused, self.cross = heads, fused_qkv, cross
        if fused_qkv:
            self.qkv = nn.Linear(dim, 3 * dim, bias=False)
            self.proj = nn.Linear(dim, dim, bias=True)
        else:
            self.q = nn.Linear(dim, dim, bias=False)
            self.k = nn.Linear(dim, dim, bias=False)
            self.v = nn.Linear(dim, dim, bias=False)
            self.o = nn.Linear(dim, dim, bias=False)

    def forward(self, x, kv=None, dense=False):
        B, T, D = x.shape
        h = self.heads
        if self.fused:
            q, k, v = _lin(self.qkv, x, dense).reshape(B, T, 3, D).unbind(2)
        else:
            src = x if kv is None else kv
            q, k, v = _lin(self.q, x, dense), _lin(self.k, src, dense), _lin(self.v, src, dense)
        S = k.shape[1]
        if self.use_sdpa is True:
            q, k, v = (t.reshape(B, -1, h, D // h).transpose(1, 2) for t in (q, k, v))
            y = F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(B, T, D)
            return _lin(self.proj if self.fused else self.o, y, dense)
        if self.use_sdpa == "matmul16":
            # the reference's op sequence in the activation dtype (modeling_t5.py:590-638: `torch.matmul` scores, fp32 softmax
            # cast back, `torch.matmul(attn, v)`; eva_vit.py:144-164 with `q * scale` first): batched 16-bit matmuls
            q, k, v = (t.reshape(B, -1, h, D // h).transpose(1, 2) for t in (q, k, v))
            scores = torch.matmul(q * (D // h) ** -0.5, k.transpose(3, 2))
            a = torch.softmax(scores.float(), dim=-1).type_as(scores)
            y = (a @ v).transpose(1, 2).reshape(B, T, D)
            return _lin(self.proj if self.fused else self.o, y, dense)
        q = q.reshape(B, T, h, D // h).transpose(1, 2).float()
        k = k.reshape(B, S, h, D // h).transpose(1, 2).float()
        v = v.reshape(B, S, h, D // h).transpose(1, 2).float()
        a = torch.softmax(q @ k.transpose(-1, -2) / (D // h) ** 0.5, dim=-1)
        y = (a @ v).transpose(1, 2).reshape(B, T, D).to(x.dtype)
        return _lin(self.proj if self.fused else self.o, y, dense)


class ToyViTBlock(nn.Module):
    """4 prunable linears: attn.qkv, attn.proj, mlp.fc1, mlp.fc2 (eva_vit.py:474-487)."""

    def __init__(self, dim, hidden, heads=2):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim)
        self.attn = ToyAttention(dim, heads, fused_qkv=True)
        self.norm2 = nn.LayerNorm(dim)
        self.mlp = nn.Module()
        self.mlp.fc1 = nn.Linear(dim, hidden)
        self.mlp.fc2 = nn.Linear(hidden, dim)

    def forward(self, x, rel_pos_bias=None, dense=False):
        x = x + self.attn(self.norm1(x), dense=dense)
        h = F.gelu(_lin(self.mlp.fc1, self.norm2(x), dense))
        return x + _lin(self.mlp.fc2, h, dense)


class ToyRMSNorm(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(dim))

    def forward(self, x):
        v = x.float().pow(2).mean(-1, keepdim=True)
        return (x.float() * torch.rsqrt(v + 1e-6)).to(x.dtype) * self.weight


class ToyT5Block(nn.Module):
    """Encoder block: 7 linears (q,k,v,o,wi_0,wi_1,wo); decoder block: 11
    (adds cross q,k,v,o) -- modeling_t5.py:320-343,748-763.  Returns a tuple."""

    def __init__(self, dim, d_ff, heads=2, is_decoder=False):
        super().__init__()
        self.is_decoder = is_decoder
        self.ln0 = ToyRMSNorm(dim)
        self.SelfAttention = ToyAttention(dim, heads, fused_qkv=False)
        if is_decoder:
            self.ln1 = ToyRMSNorm(dim)
            self.EncDecAttention = ToyAttention(dim, heads, fused_qkv=False, cross=True)
        self.ln2 = ToyRMSNorm(dim)
        self.DenseReluDense = nn.Module()
        self.DenseReluDense.wi_0 = nn.Linear(dim, d_ff, bias=False)
        self.DenseReluDense.wi_1 = nn.Linear(dim, d_ff, bias=False)
        self.DenseReluDense.wo = nn.Linear(d_ff, dim, bias=False)

    def forward(self, hidden_states, attention_mask=None, position_bias=None, encoder_hidden_states=None,
                encoder_attention_mask=None, encoder_decoder_position_bias=None, layer_head_mask=None,
                cross_attn_layer_head_mask=None, dense=False, **unused):
        x = hidden_states
        x = x + self.SelfAttention(self.ln0(x), dense=dense)
        if self.is_decoder:
            x = x + self.EncDecAttention(self.ln1(x), kv=encoder_hidden_states, dense=dense)
        h = self.ln2(x)
        ff = self.DenseReluDense
        h = F.gelu(_lin(ff.wi_0, h, dense)) * _lin(ff.wi_1, h, dense)
        x = x + _lin(ff.wo, h, dense)
        return (x,)


class ToyBlipT5(nn.Module):
    def __init__(self, vit_dim=32, vit_hidden=64, vit_depth=2, t5_dim=32, t5_ff=64, enc_depth=2, dec_depth=2,
                 vocab=50, vit_dtype=torch.float32, t5_dtype=torch.float32):
        super().__init__()
        self.visual_encoder = nn.Module()
        self.visual_encoder.blocks = nn.ModuleList([ToyViTBlock(vit_dim, vit_hidden) for _ in range(vit_depth)])
        self.visual_encoder.to(vit_dtype)
        self.t5_proj = nn.Linear(vit_dim, t5_dim)
        t5 = nn.Module()
        t5.config = types.SimpleNamespace(use_cache=True, d_model=t5_dim)
        t5.shared = nn.Embedding(vocab, t5_dim)
        t5.encoder = nn.Module()
        t5.encoder.block = nn.ModuleList([ToyT5Block(t5_dim, t5_ff) for _ in range(enc_depth)])
        t5.decoder = nn.Module()
        t5.decoder.block = nn.ModuleList([ToyT5Block(t5_dim, t5_ff, is_decoder=True) for _ in range(dec_depth)])
        self.t5_model = t5
        self.t5_proj.to(t5_dtype)
        self.t5_model.to(t5_dtype)
        self.vit_dtype, self.t5_dtype = vit_dtype, t5_dtype

    def maybe_autocast(self, dtype=None):
        return contextlib.nullcontext()

    def forward(self, samples, vit_dense=False, llm_dense=False):
        x = samples["image"].to(self.vit_dtype)
        for blk in self.visual_encoder.blocks:
            x = blk(x, None, dense=vit_dense)
        t5 = self.t5_model
        img = self.t5_proj(x.to(self.t5_dtype))
        txt = t5.shared(samples["text_input"])
        h = torch.cat([img, txt], dim=1)
        kw = dict(attention_mask=None, position_bias=None, encoder_hidden_states=None, encoder_attention_mask=None,
                  encoder_decoder_position_bias=None, layer_head_mask=None, cross_attn_layer_head_mask=None)
        for blk in t5.encoder.block:
            h = blk(h, dense=llm_dense, **kw)[0]
        d = t5.shared(samples["text_output"])
        kw["encoder_hidden_states"] = h
        for blk in t5.decoder.block:
            d = blk(d, dense=llm_dense, **kw)[0]
        logits = d.float() @ t5.shared.weight.float().t()            # tied lm head, [B, out_len, vocab]
        return {"loss": d.float().pow(2).mean(), "logits": logits}


def make_batches(n, vit_tokens=9, vit_dim=32, txt_len=5, out_len=4, vocab=50, seed=0, image_mean=0.1):
    g = torch.Generator().manual_seed(seed)
    out = []
    for _ in range(n):
        out.append({
            "image": torch.randn(1, vit_tokens, vit_dim, generator=g) + image_mean,
            "text_input": torch.randint(0, vocab, (1, txt_len), generator=g),
            "text_output": torch.randint(0, vocab, (1, out_len), generator=g),
        })
    return out


def init_toy(model, seed=0, std=0.05):
    """Deterministic weights (independent of torch's default init RNG stream)."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if p.dim() >= 2:
                p.copy_((torch.randn(p.shape, generator=g) * std).to(p.dtype))
            elif n.endswith("bias"):
                p.copy_((torch.randn(p.shape, generator=g) * 0.01).to(p.dtype))
    return model


# ---- InstructBLIP-Vicuna shaped toy (BASELINE.json configs 3-4: the `llm_model...model.layers` branch) -------------------
class ToyLlamaAttention(nn.Module):
    def __init__(self, dim, heads):
        super().__init__()
        self.heads = heads
        self.q_proj = nn.Linear(dim, dim, bias=False)
        self.k_proj = nn.Linear(dim, dim, bias=False)
        self.v_proj = nn.Linear(dim, dim, bias=False)
        self.o_proj = nn.Linear(dim, dim, bias=False)

    def forward(self, x, position_ids=None, dense=False):
        B, T, D = x.shape
        h = self.heads
        q, k, v = _lin(self.q_proj, x, dense), _lin(self.k_proj, x, dense), _lin(self.v_proj, x, dense)
        if position_ids is not None:                     # a stand-in for the rotary embedding: the kwarg must arrive
            rot = torch.cos(position_ids.float() * 0.1)[..., None].to(x.dtype)
            q, k = q * rot, k * rot
        q = q.reshape(B, T, h, D // h).transpose(1, 2).float()
        k = k.reshape(B, T, h, D // h).transpose(1, 2).float()
        v = v.reshape(B, T, h, D // h).transpose(1, 2).float()
        causal = torch.full((T, T), float("-inf"), device=x.device).triu(1)
        a = torch.softmax(q @ k.transpose(-1, -2) / (D // h) ** 0.5 + causal, dim=-1)
        return _lin(self.o_proj, (a @ v).transpose(1, 2).reshape(B, T, D).to(x.dtype), dense)


class ToyLlamaLayer(nn.Module):
    """7 linears: self_attn.{q,k,v,o}_proj, mlp.{gate,up,down}_proj (modeling_llama.py:160,204-206,253)."""

    def __init__(self, dim, d_ff, heads=2):
        super().__init__()
        self.input_layernorm = ToyRMSNorm(dim)
        self.self_attn = ToyLlamaAttention(dim, heads)
        self.post_attention_layernorm = ToyRMSNorm(dim)
        self.mlp = nn.Module()
        self.mlp.gate_proj = nn.Linear(dim, d_ff, bias=False)
        self.mlp.up_proj = nn.Linear(dim, d_ff, bias=False)
        self.mlp.down_proj = nn.Linear(d_ff, dim, bias=False)

    def forward(self, hidden_states, attention_mask=None, position_ids=None, dense=False, **unused):
        x = hidden_states
        x = x + self.self_attn(self.input_layernorm(x), position_ids=position_ids, dense=dense)
        h = self.post_attention_layernorm(x)
        h = F.silu(_lin(self.mlp.gate_proj, h, dense)) * _lin(self.mlp.up_proj, h, dense)
        return (x + _lin(self.mlp.down_proj, h, dense),)


class ToyBlipVicuna(nn.Module):
    def __init__(self, vit_dim=32, vit_hidden=64, vit_depth=2, dim=32, d_ff=88, depth=3, vocab=50, vit_dtype=torch.float32,
                 llm_dtype=torch.float32):
        super().__init__()
        self.visual_encoder = nn.Module()
        self.visual_encoder.blocks = nn.ModuleList([ToyViTBlock(vit_dim, vit_hidden) for _ in range(vit_depth)])
        self.visual_encoder.to(vit_dtype)
        self.llm_proj = nn.Linear(vit_dim, dim)
        llm = nn.Module()
        llm.config = types.SimpleNamespace(use_cache=True, hidden_size=dim)
        llm.model = nn.Module()
        llm.model.embed_tokens = nn.Embedding(vocab, dim)
        llm.model.layers = nn.ModuleList([ToyLlamaLayer(dim, d_ff) for _ in range(depth)])
        llm.model.norm = ToyRMSNorm(dim)
        self.llm_model = llm
        self.llm_proj.to(llm_dtype)
        self.llm_model.to(llm_dtype)
        self.vit_dtype, self.llm_dtype = vit_dtype, llm_dtype

    def maybe_autocast(self, dtype=None):
        return contextlib.nullcontext()

    def forward(self, samples, vit_dense=False, llm_dense=False):
        x = samples["image"].to(self.vit_dtype)
        for blk in self.visual_encoder.blocks:
            x = blk(x, None, dense=vit_dense)
        m = self.llm_model.model
        img = self.llm_proj(x.to(self.llm_dtype))
        h = torch.cat([img, m.embed_tokens(samples["text_input"]), m.embed_tokens(samples["text_output"])], dim=1)
        pos = torch.arange(h.shape[1], device=h.device)[None].expand(h.shape[0], -1)
        for layer in m.layers:
            h = layer(h, attention_mask=None, position_ids=pos, dense=llm_dense)[0]
        h = m.norm(h)
        logits = h.float() @ m.embed_tokens.weight.float().t()
        return {"loss": h.float().pow(2).mean(), "logits": logits}
