"""Random-init model with the architecture and shapes of InstructBLIP-FlanT5-XL, for end-to-end timing of the
drop-in pruners when no checkpoint can be downloaded (bench.py `end_to_end`, tools/).

What the pruners need from the reference's model classes is kept: the module names
(`visual_encoder.blocks[i].{attn.qkv, attn.proj, mlp.fc1, mlp.fc2}`, `t5_model.{encoder,decoder}.block[i].layer[..]`
`.{SelfAttention,EncDecAttention}.{q,k,v,o}`, `.DenseReluDense.{wi_0,wi_1,wo}`), the block call contracts
(eva_vit.Block: `blk(x, rel_pos_bias, dense=)`; T5Block: `blk(hidden, attention_mask=..., ..., dense=) -> (hidden,)`),
`maybe_autocast`, `t5_model.config.use_cache` and `model(samples)["loss"]` (blip2_t5_instruct.py:136-221).
Shapes: SURVEY.md Appendix A (ViT-g 39 x 1408/6144, 16 heads, 257 tokens, fp16; Flan-T5-XL 24 + 24 x 2048/5120,
32 heads of 64, bf16); the Q-Former (never pruned: 12 BERT-base layers over 32 queries + the instruction, cross-attention to
the image tokens in every second layer, `QFormer` below) between them since round 5 -- BASELINE.json configs[1] names it.
"""
from __future__ import annotations

import contextlib
import types

import torch
import torch.nn as nn
import torch.nn.functional as F


def _lin(mod, x, dense):
    return mod(x) if type(mod) is nn.Linear else mod(x, dense=dense)


class Attention(nn.Module):
    def __init__(self, dim, heads, d_kv, fused_qkv):
        super().__init__()
        self.heads, self.d_kv, self.fused = heads, d_kv, fused_qkv
        inner = heads * d_kv
        if fused_qkv:
            self.qkv = nn.Linear(dim, 3 * inner, bias=False)
            self.proj = nn.Linear(inner, dim)
        else:
            self.q = nn.Linear(dim, inner, bias=False)
            self.k = nn.Linear(dim, inner, bias=False)
            self.v = nn.Linear(dim, inner, bias=False)
            self.o = nn.Linear(inner, dim, bias=False)

    def forward(self, x, kv=None, dense=False):
        B, T, _ = x.shape
        if self.fused:
            q, k, v = _lin(self.qkv, x, dense).reshape(B, T, 3, -1).unbind(2)
        else:
            src = x if kv is None else kv
            q, k, v = _lin(self.q, x, dense), _lin(self.k, src, dense), _lin(self.v, src, dense)
        S = k.shape[1]
        q = q.reshape(B, T, self.heads, self.d_kv).transpose(1, 2)
        k = k.reshape(B, S, self.heads, self.d_kv).transpose(1, 2)
        v = v.reshape(B, S, self.heads, self.d_kv).transpose(1, 2)
        y = F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(B, T, -1)
        return _lin(self.proj if self.fused else self.o, y, dense)


class EvaAttentionOps(nn.Module):
    """EVA-ViT attention written with the reference's own op sequence (eva_vit.py:129-168): qkv linear without bias plus the
    concatenated (q_bias, 0, v_bias), `q * scale`, an explicit batched `q @ k^T`, optional `rel_pos_bias`, softmax in the
    activation dtype, `attn @ v`, proj.  The two products are batched `torch.matmul`s: the GEMM library's, not per sample."""

    def __init__(self, dim, heads):
        super().__init__()
        self.heads, self.scale = heads, (dim // heads) ** -0.5
        self.qkv = nn.Linear(dim, 3 * dim, bias=False)
        self.q_bias = nn.Parameter(torch.zeros(dim))
        self.v_bias = nn.Parameter(torch.zeros(dim))
        self.proj = nn.Linear(dim, dim)

    def forward(self, x, rel_pos_bias=None, dense=False):
        B, N, C = x.shape
        qkv_bias = torch.cat((self.q_bias, torch.zeros_like(self.v_bias, requires_grad=False), self.v_bias))
        qkv = _lin(self.qkv, x, dense) + qkv_bias
        qkv = qkv.reshape(B, N, 3, self.heads, -1).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0], qkv[1], qkv[2]
        q = q * self.scale
        attn = q @ k.transpose(-2, -1)
        if rel_pos_bias is not None:
            attn = attn + rel_pos_bias
        attn = attn.softmax(dim=-1)
        x = (attn @ v).transpose(1, 2).reshape(B, N, -1)
        return _lin(self.proj, x, dense)


class T5AttentionOps(nn.Module):
    """T5 attention with the reference's op sequence (modeling_t5.py:520-640): unscaled `torch.matmul(q, k^T)`, the position
    bias -- the bucketed relative bias in the tower's first block, zeros in the others when none is handed over (the
    pruners replay every block with block 0's kwargs: position_bias=None) -- plus the extended mask, softmax in fp32 cast
    back, `torch.matmul(attn, v)`, o.  Returns (output, position_bias) like the reference returns it in its tuple."""

    def __init__(self, dim, heads, d_kv, has_relative_attention_bias=False, is_decoder=False, num_buckets=32, max_distance=128):
        super().__init__()
        self.heads, self.d_kv, self.is_decoder = heads, d_kv, is_decoder
        self.num_buckets, self.max_distance = num_buckets, max_distance
        inner = heads * d_kv
        self.q = nn.Linear(dim, inner, bias=False)
        self.k = nn.Linear(dim, inner, bias=False)
        self.v = nn.Linear(dim, inner, bias=False)
        self.o = nn.Linear(inner, dim, bias=False)
        self.has_relative_attention_bias = has_relative_attention_bias
        if has_relative_attention_bias:
            self.relative_attention_bias = nn.Embedding(num_buckets, heads)

    def compute_bias(self, q_len, k_len, device):
        """Bucketed relative position bias [1, heads, q_len, k_len] (modeling_t5.py:411-482: half of the buckets exact, half
        logarithmic up to max_distance; bidirectional in the encoder)."""
        rel = torch.arange(k_len, device=device)[None, :] - torch.arange(q_len, device=device)[:, None]
        nb = self.num_buckets
        buckets = torch.zeros_like(rel)
        if not self.is_decoder:
            nb //= 2
            buckets = buckets + (rel > 0).long() * nb
            rel = rel.abs()
        else:
            rel = -torch.min(rel, torch.zeros_like(rel))
        exact = nb // 2
        import math
        large = exact + (torch.log(rel.float() / exact) / math.log(self.max_distance / exact) * (nb - exact)).long()
        large = torch.min(large, torch.full_like(large, nb - 1))
        buckets = buckets + torch.where(rel < exact, rel, large)
        return self.relative_attention_bias(buckets).permute(2, 0, 1).unsqueeze(0)

    def forward(self, x, mask=None, kv=None, position_bias=None, dense=False):
        B, T, _ = x.shape
        src = x if kv is None else kv

        def shape(t):
            return t.view(B, -1, self.heads, self.d_kv).transpose(1, 2)
        q, k, v = shape(_lin(self.q, x, dense)), shape(_lin(self.k, src, dense)), shape(_lin(self.v, src, dense))
        scores = torch.matmul(q, k.transpose(3, 2))
        if position_bias is None:
            if not self.has_relative_attention_bias:
                position_bias = torch.zeros((1, self.heads, T, k.shape[2]), device=scores.device, dtype=scores.dtype)
            else:
                position_bias = self.compute_bias(T, k.shape[2], scores.device)
            if mask is not None:
                position_bias = position_bias + mask
        scores += position_bias
        attn = F.softmax(scores.float(), dim=-1).type_as(scores)
        y = torch.matmul(attn, v).transpose(1, 2).contiguous().view(B, -1, self.heads * self.d_kv)
        return _lin(self.o, y, dense), position_bias


class ViTBlock(nn.Module):
    def __init__(self, dim, hidden, heads, reference_ops=False):
        super().__init__()
        self.reference_ops = reference_ops
        self.norm1 = nn.LayerNorm(dim)
        self.attn = EvaAttentionOps(dim, heads) if reference_ops else Attention(dim, heads, dim // heads, fused_qkv=True)
        self.norm2 = nn.LayerNorm(dim)
        self.mlp = nn.Module()
        self.mlp.fc1 = nn.Linear(dim, hidden)
        self.mlp.fc2 = nn.Linear(hidden, dim)

    def forward(self, x, rel_pos_bias=None, dense=False):
        if self.reference_ops:                                # eva_vit.py:216-221 without layer scale / drop path
            x = x + self.attn(self.norm1(x), rel_pos_bias=rel_pos_bias, dense=dense)
        else:
            x = x + self.attn(self.norm1(x), dense=dense)
        return x + _lin(self.mlp.fc2, F.gelu(_lin(self.mlp.fc1, self.norm2(x), dense)), dense)


class RMSNorm(nn.Module):
    def __init__(self, dim, eps=1e-6):     # (transformers' T5LayerNorm: the same attribute names, the same op sequence)
        super().__init__()
        self.weight = nn.Parameter(torch.ones(dim))
        self.variance_epsilon = eps

    def forward(self, x):
        v = x.to(torch.float32).pow(2).mean(-1, keepdim=True)
        h = x * torch.rsqrt(v + self.variance_epsilon)
        if self.weight.dtype in (torch.float16, torch.bfloat16):
            h = h.to(self.weight.dtype)
        return self.weight * h


class _T5Sub(nn.Module):
    pass


class T5Block(nn.Module):
    """`layer[0].SelfAttention`, (`layer[1].EncDecAttention`,) `layer[-1].DenseReluDense` with T5 v1.1's gated GELU."""

    def __init__(self, dim, d_ff, heads, d_kv, is_decoder, reference_ops=False, has_relative_attention_bias=False):
        super().__init__()
        self.is_decoder, self.reference_ops = is_decoder, reference_ops
        subs = []
        sa = _T5Sub()
        sa.layer_norm = RMSNorm(dim)
        sa.SelfAttention = T5AttentionOps(dim, heads, d_kv, has_relative_attention_bias, is_decoder) if reference_ops else \
            Attention(dim, heads, d_kv, fused_qkv=False)
        subs.append(sa)
        if is_decoder:
            ca = _T5Sub()
            ca.layer_norm = RMSNorm(dim)
            ca.EncDecAttention = T5AttentionOps(dim, heads, d_kv, False, is_decoder) if reference_ops else \
                Attention(dim, heads, d_kv, fused_qkv=False)
            subs.append(ca)
        ff = _T5Sub()
        ff.layer_norm = RMSNorm(dim)
        ff.DenseReluDense = nn.Module()
        ff.DenseReluDense.wi_0 = nn.Linear(dim, d_ff, bias=False)
        ff.DenseReluDense.wi_1 = nn.Linear(dim, d_ff, bias=False)
        ff.DenseReluDense.wo = nn.Linear(d_ff, dim, bias=False)
        subs.append(ff)
        self.layer = nn.ModuleList(subs)

    def forward(self, hidden_states, attention_mask=None, position_bias=None, encoder_hidden_states=None,
                encoder_attention_mask=None, encoder_decoder_position_bias=None, layer_head_mask=None,
                cross_attn_layer_head_mask=None, dense=False, **unused):
        x = hidden_states
        sa = self.layer[0]
        biases = ()
        if self.reference_ops:
            y, position_bias = sa.SelfAttention(sa.layer_norm(x), mask=attention_mask, position_bias=position_bias, dense=dense)
            x = x + y
            biases = (position_bias,)
        else:
            x = x + sa.SelfAttention(sa.layer_norm(x), dense=dense)
        if self.is_decoder:
            ca = self.layer[1]
            if self.reference_ops:
                y, encoder_decoder_position_bias = ca.EncDecAttention(ca.layer_norm(x), mask=encoder_attention_mask, kv=encoder_hidden_states,
                                                                      position_bias=encoder_decoder_position_bias, dense=dense)
                x = x + y
                biases = biases + (encoder_decoder_position_bias,)
            else:
                x = x + ca.EncDecAttention(ca.layer_norm(x), kv=encoder_hidden_states, dense=dense)
        ff = self.layer[-1]
        h = ff.layer_norm(x)
        d = ff.DenseReluDense
        x = x + _lin(d.wo, F.gelu(_lin(d.wi_0, h, dense)) * _lin(d.wi_1, h, dense), dense)
        return (x,) + biases                       # (hidden, position bias[, cross-attention position bias]): use_cache is off


# ---- the Q-Former between the vision tower and the language model (never pruned) ----------------------------------------------
# lavis/models/blip2_models/Qformer.py (BertLayer :378-470, BertSelfAttention :111-260, BertEncoder :490-580, BertModel :713-803 /
# :866-930) as blip2_t5_instruct.py:143-175 drives it: 12 BERT-base layers (hidden 768, 12 heads, FFN 3072) over [32 learned queries |
# instruction text]; every second layer the queries cross-attend to the image tokens (encoder width 1408); queries and text have their
# own feed-forward halves; the first 32 rows of the last layer go to `t5_proj`.  The call CONTRACT is the reference's:
#   * a layer is called POSITIONALLY -- `layer(hidden_states, attention_mask, head_mask, encoder_hidden_states,
#     encoder_attention_mask, past_key_value, output_attentions, query_length)` (:541-550) -- with TENSOR extended masks,
#     `(1 - mask[:, None, None, :]) * -10000.0` in the module's dtype (:795-801), head_mask / past_key_value None;
#   * a layer returns `(layer_output, present_key_value)` with present_key_value = the self-attention's `(key_layer, value_layer)`
#     (:470-474, :200): a nested tuple, of which the encoder reads element 0 (use_cache is off for a non-decoder);
#   * the reference keeps the Q-Former, `ln_vision` and `t5_proj` in fp32 and calls them OUTSIDE `maybe_autocast`
#     (blip2_t5_instruct.py:144 closes the autocast block before `self.Qformer.bert(...)`; under it `ln_vision` returns fp32):
#     `qformer_dtype=torch.float32` (the default of the reference-op stand-in) does the same; the fp16 form is kept for the
#     rounds 2-5 stand-in (`reference_ops=False`).
# Module names are the reference's (`.attention.self.{query,key,value}`, `.attention.output.dense`, `.crossattention...`,
# `.intermediate[_query].dense`, `.output[_query].dense`: the targets of `qformer_lora_target_modules`).
class _QfSelfAttention(nn.Module):
    def __init__(self, dim, heads, kv_dim, reference_ops):
        super().__init__()
        self.heads, self.reference_ops = heads, reference_ops
        self.query, self.key, self.value = nn.Linear(dim, dim), nn.Linear(kv_dim, dim), nn.Linear(kv_dim, dim)

    def transpose_for_scores(self, x):
        return x.view(*x.shape[:-1], self.heads, -1).permute(0, 2, 1, 3)

    def forward(self, hidden_states, attention_mask=None, head_mask=None, encoder_hidden_states=None, encoder_attention_mask=None,
                past_key_value=None, output_attentions=False):
        cross = encoder_hidden_states is not None
        src = encoder_hidden_states if cross else hidden_states
        if cross:
            attention_mask = encoder_attention_mask
        k = self.transpose_for_scores(self.key(src))
        v = self.transpose_for_scores(self.value(src))
        q = self.transpose_for_scores(self.query(hidden_states))
        B, T, D = hidden_states.shape
        if self.reference_ops or attention_mask is not None:              # Qformer.py:205-246: scores, / sqrt(d), + mask, softmax, probs @ v
            scores = torch.matmul(q, k.transpose(-1, -2))
            scores = scores / (D // self.heads) ** 0.5
            if attention_mask is not None:
                scores = scores + attention_mask
            y = torch.matmul(nn.functional.softmax(scores, dim=-1), v)
        else:
            y = F.scaled_dot_product_attention(q, k, v)
        return y.permute(0, 2, 1, 3).reshape(B, T, D), (k, v)


class _QfOutput(nn.Module):                         # BertSelfOutput / BertOutput: dense, (dropout,) LayerNorm(dense + residual)
    def __init__(self, in_dim, dim, eps=1e-12):
        super().__init__()
        self.dense = nn.Linear(in_dim, dim)
        self.LayerNorm = nn.LayerNorm(dim, eps=eps)

    def forward(self, x, residual):
        return self.LayerNorm(self.dense(x) + residual)


class _QfAttention(nn.Module):
    def __init__(self, dim, heads, kv_dim, reference_ops):
        super().__init__()
        self.self = _QfSelfAttention(dim, heads, kv_dim, reference_ops)
        self.output = _QfOutput(dim, dim)

    def forward(self, hidden_states, attention_mask=None, head_mask=None, encoder_hidden_states=None, encoder_attention_mask=None,
                past_key_value=None, output_attentions=False):
        ctx, present = self.self(hidden_states, attention_mask, head_mask, encoder_hidden_states, encoder_attention_mask, past_key_value,
                                 output_attentions)
        return self.output(ctx, hidden_states), present


class _QfIntermediate(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.dense = nn.Linear(dim, hidden)

    def forward(self, x):
        return F.gelu(self.dense(x))


class QFormerLayer(nn.Module):
    def __init__(self, dim, heads, hidden, encoder_width, cross, reference_ops=False):
        super().__init__()
        self.attention = _QfAttention(dim, heads, dim, reference_ops)
        self.has_cross_attention = cross
        if cross:
            self.crossattention = _QfAttention(dim, heads, encoder_width, reference_ops)
        self.intermediate, self.output = _QfIntermediate(dim, hidden), _QfOutput(hidden, dim)
        self.intermediate_query, self.output_query = _QfIntermediate(dim, hidden), _QfOutput(hidden, dim)

    def forward(self, hidden_states, attention_mask=None, head_mask=None, encoder_hidden_states=None, encoder_attention_mask=None,
                past_key_value=None, output_attentions=False, query_length=0):
        a, present_key_value = self.attention(hidden_states, attention_mask, head_mask, output_attentions=output_attentions,
                                              past_key_value=past_key_value[:2] if past_key_value is not None else None)
        if query_length > 0:
            qa = a[:, :query_length, :]
            if self.has_cross_attention:
                assert encoder_hidden_states is not None, "encoder_hidden_states must be given for cross-attention layers"
                qa, _ = self.crossattention(qa, attention_mask, head_mask, encoder_hidden_states, encoder_attention_mask,
                                            output_attentions=output_attentions)
            out = self.output_query(self.intermediate_query(qa), qa)
            if a.shape[1] > query_length:
                ta = a[:, query_length:, :]
                out = torch.cat([out, self.output(self.intermediate(ta), ta)], dim=1)
        else:
            out = self.output(self.intermediate(a), a)
        return (out,) + (present_key_value,)                 # (layer_output, present_key_value): Qformer.py:470-474


class QFormer(nn.Module):
    """`Qformer.bert(input_ids, attention_mask=, query_embeds=, encoder_hidden_states=, encoder_attention_mask=)` of the reference,
    reduced to what the forward of blip2_t5_instruct.py:146-175 uses: `.bert.embeddings`, the extended masks of BertModel.forward
    (:713-803, :889-921), `.bert.encoder.layer[i]` called as BertEncoder.forward calls them (:541-550)."""

    def __init__(self, dim=768, heads=12, hidden=3072, depth=12, encoder_width=1408, vocab=30523, cross_attention_freq=2, reference_ops=False):
        super().__init__()
        self.bert = nn.Module()
        self.bert.embeddings = nn.Module()
        self.bert.embeddings.word_embeddings = nn.Embedding(vocab, dim)
        self.bert.embeddings.position_embeddings = nn.Embedding(512, dim)
        self.bert.embeddings.LayerNorm = nn.LayerNorm(dim, eps=1e-12)
        self.bert.encoder = nn.Module()
        self.bert.encoder.layer = nn.ModuleList([QFormerLayer(dim, heads, hidden, encoder_width, i % cross_attention_freq == 0, reference_ops)
                                                 for i in range(depth)])
        self.vocab = vocab

    def forward(self, input_ids, attention_mask=None, query_embeds=None, encoder_hidden_states=None, encoder_attention_mask=None):
        emb = self.bert.embeddings
        B, Q = query_embeds.shape[:2]
        if input_ids is not None:
            pos = torch.arange(input_ids.shape[1], device=input_ids.device)[None]
            x = torch.cat([query_embeds, emb.word_embeddings(input_ids) + emb.position_embeddings(pos)], dim=1)
        else:
            x = query_embeds
        x = emb.LayerNorm(x)
        dtype = x.dtype
        ext = enc_ext = None
        if attention_mask is not None:                                    # get_extended_attention_mask (:795-801)
            ext = (1.0 - attention_mask[:, None, None, :].to(dtype)) * -10000.0
        if encoder_attention_mask is not None:                            # invert_attention_mask of the image tokens (:906-916)
            enc_ext = (1.0 - encoder_attention_mask[:, None, None, :].to(dtype)) * -10000.0
        for layer in self.bert.encoder.layer:                             # BertEncoder.forward :541-550: positional, use_cache off
            layer_outputs = layer(x, ext, None, encoder_hidden_states, enc_ext, None, False, Q)
            x = layer_outputs[0]
        return x


class InstructBlipT5(nn.Module):
    def __init__(self, vit_dim=1408, vit_hidden=6144, vit_heads=16, vit_depth=39, d_model=2048, d_ff=5120, heads=32, d_kv=64,
                 enc_depth=24, dec_depth=24, vocab=32128, query_tokens=32, vit_dtype=torch.float16, t5_dtype=torch.bfloat16,
                 reference_ops=False, qformer=True, qformer_dim=768, qformer_heads=12, qformer_hidden=3072, qformer_depth=12,
                 qformer_vocab=30523, qformer_dtype=None):
        """reference_ops=True: the blocks' attention follows the reference's model files op for op -- EVA attention as
        eva_vit.py:129-168 (q / v bias, explicit `q @ k^T`, softmax, `attn @ v`), T5 attention as modeling_t5.py:520-640
        (`torch.matmul` scores, bucketed position bias in block 0 handed on by the stack, fp32 softmax, extended masks) --
        instead of `F.scaled_dot_product_attention`.  The products of attention are then batched library GEMMs."""
        super().__init__()
        self.reference_ops = reference_ops
        self.visual_encoder = nn.Module()
        self.visual_encoder.blocks = nn.ModuleList([ViTBlock(vit_dim, vit_hidden, vit_heads, reference_ops) for _ in range(vit_depth)])
        self.visual_encoder.to(vit_dtype)
        # qformer=False: rounds 1-4's stand-in (the first 32 image tokens projected straight to the language model's width)
        # the reference keeps ln_vision, the Q-Former, its query tokens and t5_proj in fp32 and runs them outside autocast
        # (blip2_t5_instruct.py:76-95, :143-175); the friendlier stand-in of rounds 2-5 held them in the towers' 16-bit dtypes
        qdt = qformer_dtype if qformer_dtype is not None else (torch.float32 if reference_ops else vit_dtype)
        self.qformer_dtype = qdt
        if qformer:
            self.ln_vision = nn.LayerNorm(vit_dim).to(qdt)
            self.Qformer = QFormer(qformer_dim, qformer_heads, qformer_hidden, qformer_depth, vit_dim, qformer_vocab, 2, reference_ops).to(qdt)
            self.query_embeds = nn.Parameter(torch.zeros(1, query_tokens, qformer_dim, dtype=qdt))
            self.t5_proj = nn.Linear(qformer_dim, d_model).to(torch.float32 if qdt == torch.float32 else t5_dtype)
        else:
            self.Qformer = None
            self.t5_proj = nn.Linear(vit_dim, d_model).to(t5_dtype)
        t5 = nn.Module()
        t5.config = types.SimpleNamespace(use_cache=True, d_model=d_model)
        t5.shared = nn.Embedding(vocab, d_model)
        t5.encoder = nn.Module()
        t5.encoder.block = nn.ModuleList([T5Block(d_model, d_ff, heads, d_kv, False, reference_ops, i == 0) for i in range(enc_depth)])
        t5.decoder = nn.Module()
        t5.decoder.block = nn.ModuleList([T5Block(d_model, d_ff, heads, d_kv, True, reference_ops, i == 0) for i in range(dec_depth)])
        self.t5_model = t5.to(t5_dtype)
        self.query_tokens, self.vit_dtype, self.t5_dtype = query_tokens, vit_dtype, t5_dtype

    def maybe_autocast(self, dtype=None):
        return contextlib.nullcontext()

    def encode_image(self, image):
        x = image.to(self.vit_dtype)
        for blk in self.visual_encoder.blocks:
            x = blk(x, None)
        return x

    def forward(self, samples, vit_dense=False, llm_dense=False):
        x = samples["image"].to(self.vit_dtype)
        for blk in self.visual_encoder.blocks:
            x = blk(x, None, dense=vit_dense)
        t5 = self.t5_model
        if self.Qformer is not None:
            # blip2_t5_instruct.py:144-175: ln_vision, [queries | instruction text] through the Q-Former, its first 32 rows to t5_proj
            # (the Q-Former has its own tokenizer: the synthetic prompt's ids are folded into its vocabulary)
            # (under the reference's fp16 autocast `ln_vision` computes and returns fp32: the same values as the fp32 module on x.float())
            img_embeds = self.ln_vision(x.to(self.qformer_dtype))
            ids = samples["text_input"] % self.Qformer.vocab
            B = x.shape[0]
            if self.reference_ops:
                # blip2_t5_instruct.py:145-165: attention masks of ones (batch 1: the tokenizer's `padding="longest"` pads nothing),
                # queries | text; the image tokens' mask
                image_atts = torch.ones(img_embeds.shape[:-1], dtype=torch.long, device=x.device)
                atts = torch.ones((B, self.query_tokens + ids.shape[1]), dtype=torch.long, device=x.device)
                qout = self.Qformer(ids, attention_mask=atts, query_embeds=self.query_embeds.expand(B, -1, -1),
                                    encoder_hidden_states=img_embeds, encoder_attention_mask=image_atts)
            else:
                qout = self.Qformer(ids, query_embeds=self.query_embeds.expand(B, -1, -1), encoder_hidden_states=img_embeds)
            # (the reference's bf16 autocast `torch.cat` would carry the fp32 projection on as fp32 hidden states; the stand-in's T5
            # stack is bf16 throughout: the projection is rounded to it here)
            img = self.t5_proj(qout[:, :self.query_tokens].to(self.t5_proj.weight.dtype)).to(self.t5_dtype)
        else:
            img = self.t5_proj(x[:, :self.query_tokens].to(self.t5_dtype))      # stands in for the Q-Former's 32 queries
        h = torch.cat([img, t5.shared(samples["text_input"])], dim=1)
        kw = dict(attention_mask=None, position_bias=None, encoder_hidden_states=None, encoder_attention_mask=None,
                  encoder_decoder_position_bias=None, layer_head_mask=None, cross_attn_layer_head_mask=None)
        d = t5.shared(samples["text_output"])
        if self.reference_ops:
            # T5Stack.forward (modeling_t5.py:1060-1260): extended masks (0 where attended, dtype-min where not; all text is
            # real at batch 1), the biases of block 0 handed to the following blocks
            B, S, T = h.shape[0], h.shape[1], d.shape[1]
            kw["attention_mask"] = torch.zeros((B, 1, 1, S), dtype=h.dtype, device=h.device)
            for blk in t5.encoder.block:
                out = blk(h, dense=llm_dense, **kw)
                h, kw["position_bias"] = out[0], out[1]
            causal = torch.ones((T, T), dtype=torch.bool, device=d.device).tril()
            kw.update(attention_mask=torch.where(causal, 0.0, torch.finfo(d.dtype).min).to(d.dtype)[None, None].expand(B, 1, T, T).contiguous(),
                      position_bias=None, encoder_hidden_states=h,
                      encoder_attention_mask=torch.zeros((B, 1, 1, S), dtype=h.dtype, device=h.device))
            for blk in t5.decoder.block:
                out = blk(d, dense=llm_dense, **kw)
                d, kw["position_bias"], kw["encoder_decoder_position_bias"] = out[0], out[1], out[2]
        else:
            for blk in t5.encoder.block:
                h = blk(h, dense=llm_dense, **kw)[0]
            kw["encoder_hidden_states"] = h
            for blk in t5.decoder.block:
                d = blk(d, dense=llm_dense, **kw)[0]
        logits = d @ t5.shared.weight.t()
        return {"loss": logits.float().logsumexp(-1).mean(), "logits": logits}


# ---- InstructBLIP-Vicuna-7B shapes (BASELINE.json configs 3-4) -------------------------------------------------------------
class LlamaAttention(nn.Module):
    def __init__(self, dim, heads):
        super().__init__()
        self.heads = heads
        self.q_proj = nn.Linear(dim, dim, bias=False)
        self.k_proj = nn.Linear(dim, dim, bias=False)
        self.v_proj = nn.Linear(dim, dim, bias=False)
        self.o_proj = nn.Linear(dim, dim, bias=False)

    def forward(self, x, dense=False):
        B, T, D = x.shape
        q, k, v = (_lin(m, x, dense).reshape(B, T, self.heads, D // self.heads).transpose(1, 2)
                   for m in (self.q_proj, self.k_proj, self.v_proj))
        y = F.scaled_dot_product_attention(q, k, v, is_causal=True).transpose(1, 2).reshape(B, T, D)
        return _lin(self.o_proj, y, dense)


class LlamaLayer(nn.Module):
    def __init__(self, dim, d_ff, heads):
        super().__init__()
        self.input_layernorm = RMSNorm(dim)
        self.self_attn = LlamaAttention(dim, heads)
        self.post_attention_layernorm = RMSNorm(dim)
        self.mlp = nn.Module()
        self.mlp.gate_proj = nn.Linear(dim, d_ff, bias=False)
        self.mlp.up_proj = nn.Linear(dim, d_ff, bias=False)
        self.mlp.down_proj = nn.Linear(d_ff, dim, bias=False)

    def forward(self, hidden_states, attention_mask=None, position_ids=None, dense=False, **unused):
        x = hidden_states
        x = x + self.self_attn(self.input_layernorm(x), dense=dense)
        h = self.post_attention_layernorm(x)
        return (x + _lin(self.mlp.down_proj, F.silu(_lin(self.mlp.gate_proj, h, dense)) * _lin(self.mlp.up_proj, h, dense), dense),)


class InstructBlipVicuna(nn.Module):
    """ViT-g (fp16) + 32 LLaMA layers 4096 / 11008 (fp16): `llm_model.model.layers[i].{self_attn.*_proj, mlp.*_proj}`."""

    def __init__(self, vit_dim=1408, vit_hidden=6144, vit_heads=16, vit_depth=39, dim=4096, d_ff=11008, heads=32, depth=32,
                 vocab=32000, query_tokens=32, dtype=torch.float16):
        super().__init__()
        self.visual_encoder = nn.Module()
        self.visual_encoder.blocks = nn.ModuleList([ViTBlock(vit_dim, vit_hidden, vit_heads) for _ in range(vit_depth)])
        self.llm_proj = nn.Linear(vit_dim, dim)
        llm = nn.Module()
        llm.config = types.SimpleNamespace(use_cache=True, hidden_size=dim)
        llm.model = nn.Module()
        llm.model.embed_tokens = nn.Embedding(vocab, dim)
        llm.model.layers = nn.ModuleList([LlamaLayer(dim, d_ff, heads) for _ in range(depth)])
        llm.model.norm = RMSNorm(dim)
        self.llm_model = llm
        self.to(dtype)
        self.query_tokens, self.vit_dtype, self.llm_dtype = query_tokens, dtype, dtype

    def maybe_autocast(self, dtype=None):
        return contextlib.nullcontext()

    def forward(self, samples, vit_dense=False, llm_dense=False):
        x = samples["image"].to(self.vit_dtype)
        for blk in self.visual_encoder.blocks:
            x = blk(x, None, dense=vit_dense)
        m = self.llm_model.model
        h = torch.cat([self.llm_proj(x[:, :self.query_tokens]), m.embed_tokens(samples["text_input"]),
                       m.embed_tokens(samples["text_output"])], dim=1)
        pos = torch.arange(h.shape[1], device=h.device)[None].expand(h.shape[0], -1)
        for layer in m.layers:
            h = layer(h, attention_mask=None, position_ids=pos, dense=llm_dense)[0]
        logits = m.norm(h) @ m.embed_tokens.weight.t()
        return {"loss": logits.float().logsumexp(-1).mean(), "logits": logits}


def randomize_(model, seed=0, std=0.02):
    """Seeded in-place N(0, std) weights (every rank builds the same model)."""
    g = torch.Generator(device=next(model.parameters()).device).manual_seed(seed)
    with torch.no_grad():
        for p in model.parameters():
            if p.dim() >= 2:
                p.copy_(torch.randn(p.shape, generator=g, device=p.device, dtype=torch.float32) * std)
    return model


RAGGED_TEXT = (8, 16, 24, 32, 32, 48, 64, 128)        # prompt lengths of a ragged calibration set, cycled (mean 44)
RAGGED_OUT = (4, 8, 16, 16)


def calibration_batches(n, device, vit_tokens=257, vit_dim=1408, text_len=32, out_len=16, vocab=32128, seed=1, ragged=False):
    """`ragged`: prompt lengths 8..128 and output lengths 4..16, interleaved -- real calibration text is not of one length
    (blip2_t5_instruct.py:49-53: max_txt_len 128, max_output_txt_len 256), so the replay forms several groups per block."""
    g = torch.Generator(device=device).manual_seed(seed)
    out = []
    for j in range(n):
        tl = RAGGED_TEXT[j % len(RAGGED_TEXT)] if ragged else text_len
        ol = RAGGED_OUT[(j // 3) % len(RAGGED_OUT)] if ragged else out_len
        out.append({"image": (torch.randn(1, vit_tokens, vit_dim, generator=g, device=device) * 0.5).half(),
                    "text_input": torch.randint(0, vocab, (1, tl), generator=g, device=device),
                    "text_output": torch.randint(0, vocab, (1, ol), generator=g, device=device)})
    return out


def prunable_linears(model):
    n = 0
    for name, mod in model.named_modules():
        if isinstance(mod, nn.Linear) and (".blocks." in name or ".block." in name or ".layers." in name):
            n += 1
    return n


def time_prune(device, pruner_name="blipt5_wanda_pruner", n_samples=128, ratio=0.5, model=None, seed=0, quiet=True, batches=None,
               **cfg):
    """Wall-clock of one whole `load_pruner(...).prune()` on the synthetic InstructBLIP-FlanT5-XL (capture of the three
    towers' inputs, block replay, statistics, score/select/apply of all 588 linears).  Returns (seconds, model, info)."""
    import contextlib
    import io
    import time

    from lavis.compression import load_pruner
    vicuna = cfg.get("t5_model_prefix") == "llm_model"
    if model is None:
        model = (InstructBlipVicuna() if vicuna else InstructBlipT5()).to(device).eval()
    randomize_(model, seed)
    if batches is None:
        emb = model.llm_model.model.embed_tokens if vicuna else model.t5_model.shared
        batches = calibration_batches(n_samples, device, vocab=emb.num_embeddings)       # token ids must fit the embedding
    keep = 1 - ratio
    method = pruner_name.split("_")[1]
    full = dict(t5_prune_spec=f"{32 if vicuna else 24}-{keep!r}-1.0-1.0", vit_prune_spec=f"39-{keep!r}-1.0-1.0", t5_pruning_method=method,
                vit_pruning_method=method, num_samples=n_samples, max_sparsity_per_layer=1.01)
    full.update(cfg)
    pruner = load_pruner(pruner_name, model, batches, cfg=full)
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    with (contextlib.redirect_stdout(io.StringIO()) if quiet else contextlib.nullcontext()):
        pruner.prune()
    torch.cuda.synchronize(device)
    dt = time.perf_counter() - t0
    zeros = total = 0
    for name, mod in model.named_modules():
        if isinstance(mod, nn.Linear) and (".blocks." in name or ".block." in name or ".layers." in name):
            zeros += int((mod.weight == 0).sum())
            total += mod.weight.numel()
    return dt, model, {"linears": prunable_linears(model), "pruned_fraction": zeros / max(1, total), "weights": total}
