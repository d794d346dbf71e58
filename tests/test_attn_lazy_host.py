"""CPU: the lazy attention chain of `vlmc/forward.py` (LazyScores) with the kernels replaced by torch stand-ins -- what is recorded,
what reaches the fused entry point, and that anything the chain does not cover is computed from the literal record (the model's
code sees the values it would have seen)."""
import functools
import math

import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from vlmc import forward, ops


_MATMUL, _SOFTMAX = torch.matmul, torch.softmax                         # the originals (the stand-ins run under the patches)


def _mm(a, b, *_a, **_k):
    return _MATMUL(a.float(), b.float()).to(a.dtype)


def _softmax_rows(x, out_dtype=None):
    return _SOFTMAX(x.float(), -1).to(x.dtype if out_dtype is None else out_dtype)


class _Fused:
    """stands in for ops.attn_fused: the chain, op by op, from the arguments the fused kernel would get"""
    def __init__(self):
        self.calls = []

    def __call__(self, q, k, v, mul=None, adds=(), _plan=None):
        self.calls.append({"mul": mul, "adds": list(adds), "q": q, "k": k, "v": v})
        s = _mm(q, k.transpose(-1, -2))
        if mul is not None:
            s = (s.float() * torch.tensor(mul, dtype=torch.float32)).to(q.dtype)
        for t in adds:
            s = (s.float() + t.float()).to(q.dtype)
        p = _SOFTMAX(s.float(), -1).to(q.dtype)
        out = _mm(p, v)
        return out.transpose(1, 2).contiguous().transpose(1, 2)            # [B, H, Tq, d] as a view of [B, Tq, H, d]


@pytest.fixture
def fake_kernels(monkeypatch):
    # (one thread: the stand-ins are compared bit for bit with a second evaluation of the same CPU products, and a threaded fp32 matmul
    # on a loaded host has been seen to split its reduction differently from one call to the next)
    n_threads = torch.get_num_threads()
    torch.set_num_threads(1)
    fused = _Fused()
    monkeypatch.setattr(ops, "attn_matmul", lambda a, b, _plan=None, _try=False: _mm(a, b))
    monkeypatch.setattr(ops, "attn_fused_plan", functools.partial(ops.attn_fused_plan, _cuda_only=False))
    monkeypatch.setattr(ops, "attn_fused", fused)
    monkeypatch.setattr(ops, "softmax_rows", _softmax_rows)
    monkeypatch.setattr(ops, "_need_gpu", lambda *a: None)
    monkeypatch.setattr(ops, "_attn_max_keys", {8: 512, 16: 512})
    monkeypatch.setattr(forward, "_FUSED_OK", {})
    monkeypatch.setattr(torch.cuda, "is_current_stream_capturing", lambda: False)
    yield fused
    torch.set_num_threads(n_threads)


def _qkv(B=2, H=3, Tq=5, Tk=7, d=8, dtype=torch.bfloat16, seed=0):
    g = torch.Generator().manual_seed(seed)
    q = torch.randn(B, Tq, H, d, generator=g).to(dtype).transpose(1, 2)
    k = torch.randn(B, Tk, H, d, generator=g).to(dtype).transpose(1, 2)
    v = torch.randn(B, Tk, H, d, generator=g).to(dtype).transpose(1, 2)
    return q, k, v


def _t5(q, k, v, bias):                                                   # modeling_t5.py:588-640
    scores = torch.matmul(q, k.transpose(3, 2))
    scores += bias
    attn = F.softmax(scores.float(), dim=-1).type_as(scores)
    attn = F.dropout(attn, p=0.1, training=False)
    return torch.matmul(attn, v)


def _eva(q, k, v, scale):                                                 # eva_vit.py:145-164
    q = q * scale
    attn = q @ k.transpose(-2, -1)
    attn = attn.softmax(dim=-1)
    attn = nn.Dropout(0.0)(attn)
    return attn @ v


def _qformer(q, k, v, mask):                                              # Qformer.py:205-264
    s = torch.matmul(q, k.transpose(-1, -2))
    s = s / math.sqrt(q.shape[-1])
    s = s + mask
    p = nn.Softmax(dim=-1)(s)
    p = nn.Dropout(0.1).eval()(p)
    return torch.matmul(p, v)


def _run(fn, *a):
    with torch.no_grad(), forward.invariant_matmuls():
        return fn(*a)


def test_t5_chain_reaches_the_fused_entry_point_with_its_addend(fake_kernels, monkeypatch):
    q, k, v = _qkv()
    bias = torch.randn(1, 3, 5, 7).bfloat16()
    want = _t5(q, k, v, bias)                                             # plain torch
    before = dict(forward.stats)
    got = _run(_t5, q, k, v, bias)
    assert forward.stats["attn_fused_checks"] == before["attn_fused_checks"] + 1        # first time: compared with the unfused sequence
    assert forward.stats["attn_fused"] == before["attn_fused"] + 1
    c = fake_kernels.calls[-1]
    assert c["mul"] is None and len(c["adds"]) == 1 and c["adds"][0] is bias and c["q"] is q and c["v"] is v
    assert c["k"].data_ptr() == k.data_ptr() and c["k"].shape == k.shape
    assert torch.allclose(got.float(), want.float(), atol=2e-2)
    got2 = _run(_t5, q, k, v, bias)                                       # verified signature: no second check, no unfused pass
    assert forward.stats["attn_fused_checks"] == before["attn_fused_checks"] + 1
    assert forward.stats["attn_chain_unfused"] == before["attn_chain_unfused"] + 1
    assert torch.equal(got, got2)


def test_eva_and_qformer_chains(fake_kernels):
    q, k, v = _qkv(dtype=torch.float16, d=16)
    got = _run(_eva, q, k, v, 0.25)
    c = fake_kernels.calls[-1]
    assert c["mul"] is None and not c["adds"]                             # (q * scale happens before the product: a plain tensor op)
    assert torch.allclose(got.float(), _eva(q, k, v, 0.25).float(), atol=2e-2)
    mask = torch.zeros(2, 1, 1, 7, dtype=torch.float16)
    mask[1, ..., 5:] = torch.finfo(torch.float16).min
    got = _run(_qformer, q, k, v, mask)
    c = fake_kernels.calls[-1]
    import numpy as np
    assert c["mul"] == float(np.float32(1) / np.float32(4.0)) and len(c["adds"]) == 1 and c["adds"][0] is mask
    assert torch.allclose(got.float(), _qformer(q, k, v, mask).float(), atol=2e-2)
    assert got[1].isfinite().all()


def test_anything_else_realizes_the_tensor(fake_kernels):
    q, k, v = _qkv()
    n_fused = len(fake_kernels.calls)

    def odd(q, k, v):
        s = torch.matmul(q, k.transpose(-1, -2))
        assert type(s) is forward.LazyScores and s.shape == (2, 3, 5, 7) and s.dtype == torch.bfloat16 and s.dim() == 4
        s = torch.tanh(s)                                                 # not an op of the chain
        assert type(s) is torch.Tensor
        return torch.matmul(torch.softmax(s, -1), v)
    got = _run(odd, q, k, v)
    s = torch.tanh(_mm(q, k.transpose(-1, -2)))
    assert torch.equal(got, _mm(_softmax_rows(s), v))
    assert len(fake_kernels.calls) == n_fused

    def three_adds(q, k, v):
        s = torch.matmul(q, k.transpose(-1, -2))
        b = torch.ones(7).bfloat16()
        s = s + b
        s = s + b
        s = s + b                                                         # a third addend: realized here
        assert type(s) is torch.Tensor
        return s
    got = _run(three_adds, q, k, v)
    s = _mm(q, k.transpose(-1, -2))
    for _ in range(3):
        s = s + torch.ones(7).bfloat16()
    assert torch.equal(got, s)

    def reads_the_scores(q, k, v):
        s = torch.matmul(q, k.transpose(-1, -2))
        p = s.softmax(-1)
        out = p @ v                                                       # fused
        return out, p.sum(-1), s.amax()                                   # .. and both lazies are asked for afterwards
    out, psum, smax = _run(reads_the_scores, q, k, v)
    assert torch.allclose(psum.float(), torch.ones_like(psum).float(), atol=2e-2)
    assert smax == _mm(q, k.transpose(-1, -2)).amax()


def test_in_place_and_out_of_place_addends_keep_python_semantics(fake_kernels):
    q, k, v = _qkv()
    b = torch.randn(2, 1, 1, 7).bfloat16()

    def f(q, k, v):
        s = torch.matmul(q, k.transpose(-1, -2))
        t = s + b                                                         # a new tensor
        s += b
        s += b                                                            # s has two addends now, t one
        return torch.matmul(F.softmax(t, -1), v), torch.matmul(F.softmax(s, -1), v)
    o1, o2 = _run(f, q, k, v)
    s0 = _mm(q, k.transpose(-1, -2))
    t = s0 + b
    s = (s0 + b) + b
    assert torch.equal(o1, _mm(_softmax_rows(t), v)) and torch.equal(o2, _mm(_softmax_rows(s), v))
    assert [len(c["adds"]) for c in fake_kernels.calls[-2:]] == [1, 2]


def test_a_lazy_tensor_that_escapes_is_computed_when_the_patches_go(fake_kernels):
    q, k, v = _qkv()
    with torch.no_grad(), forward.invariant_matmuls():
        s = torch.matmul(q, k.transpose(-1, -2))
        assert s._real is None
    assert s._real is not None and torch.equal(s + 0, _mm(q, k.transpose(-1, -2)))
    assert "softmax" not in torch.Tensor.__dict__                         # patches gone, the tensor still works


def test_a_failed_comparison_switches_the_signature_off(fake_kernels, monkeypatch):
    q, k, v = _qkv()
    bias = torch.randn(1, 3, 5, 7).bfloat16()
    monkeypatch.setattr(ops, "attn_fused", lambda *a, **kw: _Fused()(*a, **kw) + 1)
    with pytest.warns(RuntimeWarning, match="does not reproduce"):
        got = _run(_t5, q, k, v, bias)
    s = (_mm(q, k.transpose(-1, -2)).float() + bias.float()).to(q.dtype)
    assert torch.equal(got, _mm(_softmax_rows(s), v))                     # the unfused result was returned
    n = forward.stats["attn_fused"]
    _run(_t5, q, k, v, bias)
    assert forward.stats["attn_fused"] == n                               # and the chain stays unfused


def test_with_gradients_or_switched_off_nothing_is_lazy(fake_kernels, monkeypatch):
    q, k, v = _qkv()
    with forward.invariant_matmuls():
        s = torch.matmul(q.clone().requires_grad_(), k.transpose(-1, -2))
        assert type(s) is torch.Tensor
    monkeypatch.setenv("VLMC_ATTN_FUSED", "0")
    with torch.no_grad(), forward.invariant_matmuls():
        assert type(torch.matmul(q, k.transpose(-1, -2))) is torch.Tensor
