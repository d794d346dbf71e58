"""GPU: `vlmc_attn_fwd` (csrc/attn_fused_kernel.hpp) -- the attention chain of the reference's model files (eva_vit.py:145-164,
modeling_t5.py:588-640, Qformer.py:205-246) in one launch -- against the SAME chain op by op on this library's unfused kernels
(`vlmc_attn_matmul`, torch's elementwise ops, `vlmc_softmax_rows`): bit for bit, that is the contract that lets the replay engine
swap one for the other under model code it does not own (vlmc/forward.py: LazyScores).  Also: the chain against fp64 (tolerance),
batch / padding invariance, the softmax kernel's dtypes, the entry point's refusals, and the stand-in attention modules with the
lazy route on and off."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _ops():
    from vlmc import ops
    return ops


def _unfused(q, k, v, mul=None, div=None, adds=(), f32_softmax=False):
    """the chain as separate tensor ops on the unfused kernels"""
    ops = _ops()
    s = ops.attn_matmul(q, k.transpose(-1, -2))
    if div is not None:
        s = s / div
    if mul is not None:
        s = s * mul
    for t in adds:
        s = s + t
    if f32_softmax:
        p = ops.softmax_rows(s.float()).type_as(s)
    else:
        p = ops.softmax_rows(s)
    return ops.attn_matmul(p, v)


def _bits(t):
    return t.contiguous().view(torch.int16)


def _heads(B, T, H, d, dtype, g, scale=1.0):
    """[B, H, T, d] as a view of a [B, T, H d] linear output"""
    return (torch.randn(B, T, H * d, generator=g, device=DEV) * scale).to(dtype).view(B, T, H, d).transpose(1, 2)


def _gen(seed):
    return torch.Generator(device=DEV).manual_seed(seed)


CASES = [
    # name, dtype, B, H, Tq, Tk, d, multiplier, addend shapes, fp32 detour
    ("t5 encoder block", torch.bfloat16, 3, 32, 96, 96, 64, None, [(1, 32, 96, 96), (3, 1, 1, 96)], True),
    ("t5 decoder self", torch.bfloat16, 4, 32, 16, 16, 64, None, [(4, 32, 16, 16)], True),
    ("t5 cross attention", torch.bfloat16, 4, 32, 16, 130, 64, None, [(4, 32, 16, 130)], True),
    ("eva vit-g", torch.float16, 2, 16, 257, 257, 88, None, [], False),
    ("eva with rel_pos_bias", torch.float16, 2, 16, 50, 50, 88, None, [(16, 50, 50)], False),
    ("qformer self", torch.float16, 3, 12, 45, 45, 64, 1.0 / 8.0, [(3, 1, 1, 45)], False),
    ("qformer cross", torch.float16, 3, 12, 32, 257, 64, 1.0 / 8.0, [], False),
    ("llama-like head_dim 128", torch.float16, 2, 8, 70, 70, 128, 1.0 / math.sqrt(128), [(2, 1, 70, 70)], True),
    ("one query, one key tile", torch.bfloat16, 2, 4, 1, 5, 64, None, [], True),
    ("33 keys", torch.float16, 2, 4, 17, 33, 32, 0.5, [(33,)], False),
    ("512 keys", torch.bfloat16, 1, 4, 40, 512, 64, None, [(1, 1, 1, 512)], True),
    ("352 keys at head_dim 96", torch.float16, 1, 3, 20, 352, 96, None, [], False),
    ("288 keys at head_dim 128", torch.bfloat16, 1, 3, 20, 288, 128, None, [], False),
]


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_fused_chain_has_the_bits_of_the_unfused_sequence(case):
    name, dtype, B, H, Tq, Tk, d, mul, add_shapes, f32 = case
    ops = _ops()
    g = _gen(len(name))
    q, k, v = _heads(B, Tq, H, d, dtype, g, 0.6), _heads(B, Tk, H, d, dtype, g, 0.6), _heads(B, Tk, H, d, dtype, g)
    adds = []
    for i, sh in enumerate(add_shapes):
        t = (torch.randn(sh, generator=g, device=DEV) * 2).to(dtype)
        if i == len(add_shapes) - 1 and sh[-1] > 8:                       # the last addend doubles as a mask: the tail keys are off
            t[..., -(sh[-1] // 5):] = torch.finfo(dtype).min
        adds.append(t)
    import numpy as np
    m32 = None if mul is None else float(np.float32(mul))
    got = ops.attn_fused(q, k, v, m32, adds)
    want = _unfused(q, k, v, mul=m32, adds=adds, f32_softmax=f32)
    assert got.shape == want.shape == (B, H, Tq, d)
    assert got.transpose(1, 2).is_contiguous()                             # [B, Tq, H, d]: the model's reshape is a view
    assert torch.equal(_bits(got), _bits(want)), f"{name}: {int((_bits(got) != _bits(want)).sum())} of {got.numel()} entries differ"
    # .. and both are what the chain is in exact arithmetic, up to the roundings it prescribes
    s = q.double() @ k.double().transpose(-1, -2)
    if m32 is not None:
        s = s * m32
    for t in adds:
        s = s + t.double()
    ref = torch.softmax(s, -1) @ v.double()
    tol = 3e-2 if dtype == torch.bfloat16 else 4e-3
    assert (got.double() - ref).abs().max() <= tol * max(1.0, float(ref.abs().max()))


def test_permuted_position_bias_is_read_through_its_strides():
    """modeling_t5.py:475-482: `values.permute([2, 0, 1]).unsqueeze(0)` -- the bias of a tower's first block is a [Tq, Tk, H] table
    lookup seen as [1, H, Tq, Tk] (key stride H), and `position_bias + mask` keeps that layout"""
    ops = _ops()
    g = _gen(21)
    B, H, T, d, dtype = 3, 32, 77, 64, torch.bfloat16
    q, k, v = _heads(B, T, H, d, dtype, g, 0.6), _heads(B, T, H, d, dtype, g, 0.6), _heads(B, T, H, d, dtype, g)
    table = torch.randn(T, T, H, generator=g, device=DEV).to(dtype).permute(2, 0, 1).unsqueeze(0)
    mask = torch.zeros(B, 1, 1, T, dtype=dtype, device=DEV)
    mask[1, ..., 60:] = torch.finfo(dtype).min
    bias = table + mask
    assert bias.stride(-1) != 1                                           # (TensorIterator keeps the permuted layout)
    for adds in ([table], [bias], [table, mask]):
        got = ops.attn_fused(q, k, v, None, adds)
        assert torch.equal(_bits(got), _bits(_unfused(q, k, v, adds=adds, f32_softmax=True)))
        assert torch.equal(_bits(got), _bits(ops.attn_fused(q, k, v, None, [t.contiguous() for t in adds])))


def test_division_by_a_python_number_is_torchs_multiplication_by_the_fp32_reciprocal():
    """`scores / math.sqrt(d)` (Qformer.py:244, modeling_llama.py): what the lazy route hands the kernel for it"""
    import numpy as np
    g = _gen(5)
    for dtype in (torch.float16, torch.bfloat16):
        s = (torch.randn(4096, 257, generator=g, device=DEV) * 30).to(dtype)
        for c in (8.0, math.sqrt(88), math.sqrt(128), 3, 11.313708498984761):
            m = float(np.float32(1.0) / np.float32(c))
            assert torch.equal(s / c, (s.float() * torch.tensor(m, dtype=torch.float32, device=DEV)).to(dtype)), (dtype, c)


def test_a_sample_has_the_same_bits_alone_in_a_group_and_padded():
    ops = _ops()
    g = _gen(11)
    dtype, H, d = torch.bfloat16, 32, 64
    lens = [37, 128, 5, 96]
    T = max(lens)
    qs = [_heads(1, n, H, d, dtype, g, 0.6) for n in lens]
    ks = [_heads(1, n, H, d, dtype, g, 0.6) for n in lens]
    vs = [_heads(1, n, H, d, dtype, g) for n in lens]
    bias = (torch.randn(1, H, T, T, generator=g, device=DEV)).to(dtype)
    alone = [ops.attn_fused(q, k, v, None, [bias[:, :, :n, :n].contiguous()]) for q, k, v, n in zip(qs, ks, vs, lens)]
    # padded into one group: zero rows, the mask at the dtype's minimum on the keys of the padding (calibration.py: padded groups)
    def pad(ts):
        out = torch.zeros(len(ts), H, T, d, dtype=dtype, device=DEV)
        for i, t in enumerate(ts):
            out[i, :, :t.shape[2]] = t[0]
        return out
    mask = torch.zeros(len(lens), 1, 1, T, dtype=dtype, device=DEV)
    for i, n in enumerate(lens):
        mask[i, ..., n:] = torch.finfo(dtype).min
    full = bias + mask                                                    # T5 adds the mask into the position bias
    grouped = ops.attn_fused(pad(qs), pad(ks), pad(vs), None, [full])
    for i, n in enumerate(lens):
        assert torch.equal(_bits(grouped[i, :, :n]), _bits(alone[i][0])), f"sample {i} ({n} tokens) changed bits inside the padded group"
    # and against the unfused sequence on the padded group
    assert torch.equal(_bits(grouped), _bits(_unfused(pad(qs), pad(ks), pad(vs), adds=[full], f32_softmax=True)))


@pytest.mark.parametrize("dtype,H,d,lens_q,lens_k,nadd", [
    (torch.bfloat16, 32, 64, [40, 160, 48, 96, 33], None, 1),                 # T5 encoder self-attention, ragged (keys = queries)
    (torch.bfloat16, 32, 64, [4, 16, 8, 16, 1], [40, 160, 48, 96, 33], 2),    # T5 cross-attention: ragged answers over ragged prompts
    (torch.float16, 12, 64, [40, 64, 33], None, 1),                           # Q-Former self-attention with text
    (torch.float16, 16, 88, [257, 100, 17], None, 1),                         # head_dim 88, three K-steps along d, the 8-wave kernel
])
def test_lengths_skip_the_padding_and_keep_the_live_bits(dtype, H, d, lens_q, lens_k, nadd):
    """vlmc_attn_fwd_lens: with per-sample key / query counts the masked keys are neither staged nor multiplied and the padding
    queries come out as zeros; every live row has the bits of the launch without lengths (and therefore of the sample alone)."""
    ops = _ops()
    g = _gen(17 + d)
    lens_k = lens_q if lens_k is None else lens_k
    B, Tq, Tk = len(lens_q), max(lens_q), max(lens_k)
    q = _heads(B, Tq, H, d, dtype, g, 0.6)
    k = _heads(B, Tk, H, d, dtype, g, 0.6)
    v = _heads(B, Tk, H, d, dtype, g)
    for i in range(B):                                                       # padding rows hold zeros, as the row-mapped linears leave them
        q[i, :, lens_q[i]:] = 0
        k[i, :, lens_k[i]:] = 0
        v[i, :, lens_k[i]:] = 0
    mask = torch.zeros(B, 1, 1, Tk, dtype=dtype, device=DEV)
    for i, n in enumerate(lens_k):
        mask[i, ..., n:] = torch.finfo(dtype).min
    bias = torch.randn(1, H, Tq, Tk, generator=g, device=DEV).to(dtype)
    adds = [bias + mask] if nadd == 1 else [bias, mask]
    ql = torch.tensor(lens_q, dtype=torch.int32, device=DEV)
    kl = torch.tensor(lens_k, dtype=torch.int32, device=DEV)
    full = ops.attn_fused(q, k, v, None, adds)
    for kw in (dict(q_len=ql, k_len=kl), dict(k_len=kl), dict(q_len=ql)):
        cut = ops.attn_fused(q, k, v, None, adds, **kw)
        for i in range(B):
            nq = lens_q[i]
            assert torch.equal(_bits(cut[i, :, :nq]), _bits(full[i, :, :nq])), (kw.keys(), i)
            if "q_len" in kw:
                assert bool((_bits(cut[i, :, nq:]) == 0).all()), "a padding query is not +0"
            else:
                assert torch.equal(_bits(cut[i, :, nq:]), _bits(full[i, :, nq:]))
    with pytest.raises(ValueError):
        ops.attn_fused(q, k, v, None, [], k_len=kl)                          # nothing masks the skipped keys
    with pytest.raises(TypeError):
        ops.attn_fused(q, k, v, None, adds, k_len=kl.long())


def test_operands_are_read_in_place_through_their_strides():
    ops = _ops()
    g = _gen(3)
    B, N, H, d = 2, 50, 16, 88
    qkv = torch.randn(B, N, 3 * H * d, generator=g, device=DEV).half().reshape(B, N, 3, H, d).permute(2, 0, 3, 1, 4)   # eva_vit.py:141
    q, k, v = qkv[0], qkv[1], qkv[2]
    got = ops.attn_fused(q, k, v)
    want = ops.attn_fused(q.contiguous(), k.contiguous(), v.contiguous())
    assert torch.equal(_bits(got), _bits(want)) and torch.equal(_bits(got), _bits(_unfused(q, k, v)))
    # rows that are not 16-byte aligned: staged through registers instead of LDS-DMA, same bits
    base = torch.randn(B * H * N * d + 4, generator=g, device=DEV).half()
    k2 = base[1:1 + B * H * N * d].view(B, H, N, d)
    v2 = base[3:3 + B * H * N * d].view(B, H, N, d)
    assert torch.equal(_bits(ops.attn_fused(q, k2, v2)), _bits(ops.attn_fused(q, k2.clone(), v2.clone())))


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16])
def test_softmax_rows_dtypes_and_padding_invariance(dtype):
    ops = _ops()
    g = _gen(7)
    for n in (1, 5, 16, 17, 257, 1000, 5000):
        x = (torch.randn(37, n, generator=g, device=DEV) * 4).to(dtype)
        y = ops.softmax_rows(x)
        assert y.dtype == dtype and y.shape == x.shape
        ref = torch.softmax(x.double(), -1)
        tol = {torch.float32: 2e-6, torch.float16: 1e-3, torch.bfloat16: 8e-3}[dtype]
        assert (y.double() - ref).abs().max() <= tol
        if dtype != torch.float32:
            y32 = ops.softmax_rows(x, torch.float32)                      # softmax(x, -1, dtype=torch.float32)
            assert y32.dtype == torch.float32 and torch.equal(y32.to(dtype), y)
            assert torch.equal(y32, ops.softmax_rows(x.float()))
        # masked entries behind the row leave its bits alone
        pad = torch.full((37, n + 83), torch.finfo(dtype).min, dtype=dtype, device=DEV)
        pad[:, :n] = x
        yp = ops.softmax_rows(pad)
        assert torch.equal(yp[:, :n], y) and not yp[:, n:].any()
    # strided rows
    x = (torch.randn(8, 4, 64, generator=g, device=DEV)).to(dtype)
    assert torch.equal(ops.softmax_rows(x[:, ::2]), ops.softmax_rows(x[:, ::2].contiguous()))


def test_entry_point_refuses_what_it_does_not_compute():
    ops = _ops()
    g = _gen(1)
    q, k, v = _heads(1, 8, 2, 64, torch.float16, g), _heads(1, 8, 2, 64, torch.float16, g), _heads(1, 8, 2, 64, torch.float16, g)
    assert ops.attn_fused_plan(q, k, v) == (1, 2, 8, 8, 64)
    assert ops.attn_fused_plan(q.float(), k.float(), v.float()) is None                                       # fp32
    assert ops.attn_fused_plan(q, k.bfloat16(), v) is None                                                    # mixed dtypes
    assert ops.attn_fused_plan(q[..., :60], k[..., :60], v[..., :60]) is None                                 # head_dim % 8
    assert ops.attn_fused_plan(q, k, v[:, :, :7]) is None                                                     # v has another length
    big = _heads(1, 513, 2, 64, torch.float16, g)
    assert ops.attn_fused_plan(q, big, big) is None                                                           # more keys than LDS holds
    assert ops.attn_fused_plan(q, k, v, [torch.zeros(8, device=DEV)]) is None                                 # an fp32 addend
    assert ops.attn_fused_plan(q, k, v, [torch.zeros(3, 8, 8, device=DEV).half()]) is None                    # not broadcastable
    assert ops.attn_fused_plan(q, k, v, [torch.zeros(8, 16, device=DEV).half()[:, ::2]]) is not None          # strided along the keys: read in place
    with pytest.raises(TypeError):
        ops.attn_fused(q.float(), k.float(), v.float())
    from vlmc import _lib
    import ctypes
    st = (ctypes.c_int64 * 3)(*q.stride()[:3])
    out = torch.empty(1, 8, 2, 64, dtype=torch.float16, device=DEV)
    rc = _lib.load().vlmc_attn_fwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), 1, 1, 2, 8, 600, 64, st, st, st, 0, 0.0,
                                   None, None, None, None, None)
    assert rc != 0 and b"keys per head" in _lib.load().vlmc_last_error()


@pytest.mark.parametrize("kind", ["t5", "t5 first block", "eva", "qformer"])
def test_stand_in_attention_modules_give_the_same_bits_on_the_lazy_route(kind, monkeypatch):
    """vlmc/synthetic.py's reference-op attention modules (the op sequences of the reference's model files) under the replay's
    patches: lazy + fused against `VLMC_ATTN_FUSED=0` (every op on its own kernel)."""
    from vlmc import forward, synthetic
    torch.manual_seed(0)
    if kind.startswith("t5"):
        mod = synthetic.T5AttentionOps(256, 4, 64, has_relative_attention_bias=kind.endswith("first block")).to(DEV, torch.bfloat16)
        x = torch.randn(5, 40, 256, device=DEV).bfloat16()
        mask = torch.zeros(5, 1, 1, 40, device=DEV, dtype=torch.bfloat16)
        mask[2:, ..., 29:] = torch.finfo(torch.bfloat16).min
        call = lambda: mod(x, mask=mask)[0]
    elif kind == "eva":
        mod = synthetic.EvaAttentionOps(352, 4).to(DEV, torch.float16)
        with torch.no_grad():
            mod.q_bias.normal_(0, 0.1), mod.v_bias.normal_(0, 0.1)
        x = torch.randn(3, 257, 352, device=DEV).half()
        call = lambda: mod(x)
    else:
        mod = synthetic._QfSelfAttention(256, 4, 256, True).to(DEV, torch.float16)
        x = torch.randn(3, 45, 256, device=DEV).half()
        mask = torch.zeros(3, 1, 1, 45, device=DEV, dtype=torch.float16)
        mask[1, ..., 40:] = torch.finfo(torch.float16).min
        call = lambda: mod(x, mask, None, None, None, None, False)[0]          # (Qformer.py:176-185: positional, returns (context, (key, value)))
    linears = [m for m in mod.modules() if isinstance(m, torch.nn.Linear)]
    forward._FUSED_OK.clear()
    with torch.no_grad(), forward.invariant_linears(linears, roots=[mod]):
        n0, c0 = forward.stats["attn_fused"], forward.stats["attn_fused_checks"]
        a = call()
        b = call()
        assert forward.stats["attn_fused"] == n0 + 2 and forward.stats["attn_fused_checks"] == c0 + 1
    monkeypatch.setenv("VLMC_ATTN_FUSED", "0")
    with torch.no_grad(), forward.invariant_linears(linears, roots=[mod]):
        n0 = forward.stats["attn_fused"]
        c = call()
        assert forward.stats["attn_fused"] == n0
    assert torch.equal(_bits(a), _bits(b)) and torch.equal(_bits(a), _bits(c))
    assert a.isfinite().all()
