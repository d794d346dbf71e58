"""The compiled host path (csrc/fastpath/fast_bind.cpp -> vlmc/_fast) and the ctypes route call the same C ABI: same bits, same
refusals."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _both(monkeypatch, fn):
    from vlmc import ops
    assert ops._fast is not None, "vlmc/_fast is not built (python -c 'import __graft_entry__ as g; g.build()')"
    fast = fn()
    monkeypatch.setattr(ops, "_fast", None)
    slow = fn()
    monkeypatch.undo()
    return fast, slow


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_linear_routes_agree(dtype, monkeypatch):
    from vlmc import ops
    g = torch.Generator(device=DEV).manual_seed(0)
    x = torch.randn(3, 37, 256, generator=g, device=DEV).to(dtype)
    ws = [torch.randn(n, 256, generator=g, device=DEV).to(dtype) for n in (64, 200, 96)]
    bs = [torch.randn(64, generator=g, device=DEV).to(dtype), None, torch.randn(96, generator=g, device=DEV).to(dtype)]
    for w, b in zip(ws, bs):
        a, c = _both(monkeypatch, lambda: ops.linear_fwd(x, w, b))
        assert a.shape == (3, 37, w.shape[0]) and torch.equal(a, c)
    a, c = _both(monkeypatch, lambda: ops.linear_fwd_group(x, ws, bs))
    assert len(a) == 3 and all(torch.equal(p, q) for p, q in zip(a, c))
    assert all(torch.equal(p, ops.linear_fwd(x, w, b)) for p, w, b in zip(a, ws, bs))
    # a strided input (column slice of a wider buffer) and an expanded one
    wide = torch.randn(37, 512, generator=g, device=DEV).to(dtype)
    a, c = _both(monkeypatch, lambda: ops.linear_fwd(wide[:, 256:], ws[0], bs[0]))
    assert torch.equal(a, c) and torch.equal(a, ops.linear_fwd(wide[:, 256:].contiguous(), ws[0], bs[0]))
    row = torch.randn(1, 256, generator=g, device=DEV).to(dtype).expand(5, 256)
    a, c = _both(monkeypatch, lambda: ops.linear_fwd(row, ws[1]))
    assert torch.equal(a, c)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16, torch.float32])
def test_row_mapped_linear_routes_agree(dtype, monkeypatch):
    """vlmc_linear_fwd_rows through vlmc/_fast and through ctypes: same outputs (16-bit group launch; fp32 one launch per member), and
    the calls the kernel does not take fall through to the ctypes route's checks."""
    from vlmc import ops
    g = torch.Generator(device=DEV).manual_seed(3)
    lengths, tp, K = [9, 3, 12, 1, 7], 12, 256
    x = (torch.randn(len(lengths), tp, K, generator=g, device=DEV) * 0.5).to(dtype)
    real = [j * tp + t for j, n in enumerate(lengths) for t in range(n)]
    pad = [j * tp + t for j, n in enumerate(lengths) for t in range(n, tp)]
    rowmap = torch.tensor(real + pad, dtype=torch.int32, device=DEV)
    ws = [(torch.randn(n, K, generator=g, device=DEV) * 0.05).to(dtype) for n in (64, 200)]
    bs = [(torch.randn(64, generator=g, device=DEV) * 0.1).to(dtype), None]
    a, c = _both(monkeypatch, lambda: ops.linear_fwd_rows(x, ws, bs, rowmap, len(real)))
    assert len(a) == 2 and all(p.shape == (len(lengths), tp, w.shape[0]) and torch.equal(p, q) for p, q, w in zip(a, c, ws))
    with pytest.raises(ValueError):
        ops.linear_fwd_rows(x, ws, bs, rowmap[:-1].contiguous(), len(real))          # (the compiled path answers None, the ctypes route says why)


def test_attention_products_routes_agree(monkeypatch):
    from vlmc import ops
    g = torch.Generator(device=DEV).manual_seed(1)
    qkv = (torch.randn(2, 33, 3 * 4 * 88, generator=g, device=DEV) * 0.5).half().reshape(2, 33, 3, 4, 88).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    a, c = _both(monkeypatch, lambda: ops.attn_matmul(q, k.transpose(-2, -1)))
    assert torch.equal(a, c)
    p = a.softmax(-1)
    a, c = _both(monkeypatch, lambda: ops.attn_matmul(p, v))
    assert torch.equal(a, c)
    a, c = _both(monkeypatch, lambda: ops.attn_matmul(p[0], v[0]))                     # 3-D
    assert torch.equal(a, c) and a.shape == (4, 33, 88)
    a, c = _both(monkeypatch, lambda: ops.attn_matmul(p[:1], v))                       # broadcast batch
    assert torch.equal(a, c) and a.shape == (2, 4, 33, 88)


def test_both_routes_refuse_the_same_calls(monkeypatch):
    from vlmc import ops
    x = torch.randn(4, 64, device=DEV)
    w = torch.randn(8, 64, device=DEV)
    for fast in (True, False):
        if not fast:
            monkeypatch.setattr(ops, "_fast", None)
        assert ops.linear_fwd(x, w).dtype == torch.float32               # fp32: taken since round 6 (the fp32 kernel, both routes)
        with pytest.raises(TypeError):
            ops.linear_fwd(x.double(), w.double())                       # fp64
        with pytest.raises(TypeError):
            ops.linear_fwd(x.half(), w)                                  # mixed dtypes
        assert ops.linear_fwd(x.half(), w, _try=True) is None
        with pytest.raises(TypeError):
            ops.linear_fwd_group(x.half(), [w.half(), w])                # mixed dtypes
        with pytest.raises(ValueError):
            ops.linear_fwd_group(x.half(), [w.half()] * 5)               # more than 4 members
        with pytest.raises(TypeError):
            ops.attn_matmul(x.half()[None], w.half().t()[None].float())
        with pytest.raises(RuntimeError):
            ops.linear_fwd(x.cpu().half(), w.cpu().half())               # no CPU fallback on either route


def test_sdpa_and_rms_norm_routes_agree(monkeypatch):
    from vlmc import ops
    g = torch.Generator(device=DEV).manual_seed(3)
    qkv = torch.randn(3, 70, 3 * 8 * 64, generator=g, device=DEV).to(torch.bfloat16)
    q, k, v = (t.reshape(3, 70, 8, 64).transpose(1, 2) for t in qkv.reshape(3, 70, 3, 8 * 64).unbind(2))
    a, c = _both(monkeypatch, lambda: ops.sdpa(q, k, v))
    assert a.shape == (3, 8, 70, 64) and torch.equal(a, c) and a.stride() == c.stride()
    a, c = _both(monkeypatch, lambda: ops.sdpa(q[:, :, :16], k, v, scale=0.2))
    assert torch.equal(a, c)
    a, c = _both(monkeypatch, lambda: ops.sdpa(q, k, v, causal=True))
    assert torch.equal(a, c) and not torch.equal(a, ops.sdpa(q, k, v))
    big = torch.randn(1, 2, 8, 300, 64, generator=g, device=DEV).to(torch.float16)       # more keys than a head may have
    a, c = _both(monkeypatch, lambda: ops.sdpa(big[0, :, :, :8], big[0], big[0], _try=True))
    assert a is None and c is None
    for fn in (lambda: ops.sdpa(q.float(), k.float(), v.float()), lambda: ops.sdpa(q, k, v, scale=-1.0)):
        for route in (True, False):
            if not route:
                monkeypatch.setattr(ops, "_fast", None)
            with pytest.raises(TypeError):
                fn()
            monkeypatch.undo()
    x = torch.randn(4, 9, 520, generator=g, device=DEV).to(torch.float16)
    w = (torch.randn(520, generator=g, device=DEV) * 0.2 + 1).to(torch.float16)
    a, c = _both(monkeypatch, lambda: ops.rms_norm(x, w, 1e-6, 0))
    assert a.shape == x.shape and torch.equal(a, c)
    a, c = _both(monkeypatch, lambda: ops.rms_norm(x[:, :, :], w, 1e-5, 1))
    assert torch.equal(a, c)
    for route in (True, False):
        if not route:
            monkeypatch.setattr(ops, "_fast", None)
        with pytest.raises(TypeError):
            ops.rms_norm(x, w.float(), 1e-6, 0)
        monkeypatch.undo()
