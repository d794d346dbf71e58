"""`vlmc_attn_matmul` (csrc/attn_matmul.hip): the batched products of attention -- `q @ k.transpose(-2, -1)`, `attn @ v`
(eva_vit.py:147,164; modeling_t5.py:590,638) -- held against fp64, against exact integer arithmetic, and against ITSELF over
every way of grouping the samples: a product's bits do not depend on the batch it is computed in."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _tol(dtype, a, b):
    """One rounding of the 16-bit output + fp32 accumulation error (tests/test_gemm_gpu.py's bar)."""
    eps = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -10
    K = a.shape[-1]
    mag = torch.matmul(a.double().abs(), b.double().abs())
    return eps, 4e-7 * (K ** 0.5) * mag


def _check(a, b):
    from vlmc import ops
    got = ops.attn_matmul(a, b)
    want = torch.matmul(a.double(), b.double())
    assert got.shape == want.shape and got.dtype == a.dtype and got.is_contiguous()
    eps, acc = _tol(a.dtype, a, b)
    err = (got.double() - want).abs()
    assert bool((err <= eps * want.abs() + acc + 1e-30).all()), float((err - eps * want.abs() - acc).max())
    return got


def _eva(B, N, H, d, g, dtype=torch.float16):
    """q, k, v as eva_vit.py:136-147 slices them out of the fused qkv product"""
    qkv = (torch.randn(B, N, 3 * H * d, generator=g, device=DEV) * 0.5).to(dtype)
    qkv = qkv.reshape(B, N, 3, H, d).permute(2, 0, 3, 1, 4)
    return qkv[0], qkv[1], qkv[2]


def _t5(B, T, S, H, d, g, dtype=torch.bfloat16):
    def shape(t, L):
        return t.view(B, L, H, d).transpose(1, 2)                      # modeling_t5.py:539-541
    q = shape((torch.randn(B, T, H * d, generator=g, device=DEV) * 0.5).to(dtype), T)
    k = shape((torch.randn(B, S, H * d, generator=g, device=DEV) * 0.5).to(dtype), S)
    v = shape((torch.randn(B, S, H * d, generator=g, device=DEV) * 0.5).to(dtype), S)
    return q, k, v


@pytest.mark.parametrize("tr", ["1", "0"])
def test_products_of_the_reference_attention_against_fp64(tr, monkeypatch):
    monkeypatch.setenv("VLMC_ATTN_TR", tr)
    g = torch.Generator(device=DEV).manual_seed(0)
    # EVA ViT-g: 16 heads of 88, 257 tokens (rows of 257 scores are not 16-byte aligned)
    q, k, v = _eva(3, 257, 16, 88, g)
    attn = _check(q * 0.1, k.transpose(-2, -1))
    p = attn.softmax(dim=-1)
    _check(p, v)
    # Flan-T5-XL: 32 heads of 64; encoder self-attention, decoder self- and cross-attention, ragged lengths
    for T, S in ((64, 64), (40, 40), (16, 16), (4, 4), (16, 160), (7, 93), (1, 33)):
        q, k, v = _t5(2, T, S, 32, 64, g)
        sc = _check(q, k.transpose(3, 2))
        _check(torch.softmax(sc.float(), dim=-1).type_as(sc), v)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("shape", [(1, 1, 1, 1, 1), (2, 3, 5, 7, 9), (1, 2, 64, 64, 64), (2, 1, 65, 63, 33), (1, 3, 130, 70, 100),
                                   (2, 2, 17, 129, 8), (1, 1, 200, 9, 257), (3, 2, 31, 31, 95), (1, 2, 8, 24, 321)])
def test_shapes_tails_and_both_layouts(shape, dtype):
    """[b0, b1, M, K] x [b0, b1, K, N] for odd sizes: B contiguous along n (NN), along k (NT, a transposed view), both
    transposing routes of the NN operand, 3-D operands and a broadcast batch dimension."""
    from vlmc import ops
    b0, b1, M, N, K = shape
    g = torch.Generator(device=DEV).manual_seed(sum(shape))
    a = torch.randn(b0, b1, M, K, generator=g, device=DEV).to(dtype)
    bn = torch.randn(b0, b1, K, N, generator=g, device=DEV).to(dtype)
    bt = bn.transpose(-1, -2).contiguous().transpose(-1, -2)                 # the same matrix, contiguous along k
    nn = _check(a, bn)
    nt = _check(a, bt)
    assert torch.equal(nn, nt), "the two layouts of B give different bits"
    os.environ["VLMC_ATTN_TR"] = "0"
    try:
        assert torch.equal(ops.attn_matmul(a, bn), nn), "transposing read and transposing write disagree"
    finally:
        del os.environ["VLMC_ATTN_TR"]
    assert torch.equal(ops.attn_matmul(a[0], bn[0]), nn[0])                  # 3-D
    if b0 > 1:
        assert torch.equal(ops.attn_matmul(a, bn[:1]), torch.stack([ops.attn_matmul(a[i], bn[0]) for i in range(b0)]))   # broadcast


def test_integer_data_is_exact():
    from vlmc import ops
    g = torch.Generator(device=DEV).manual_seed(5)
    for dtype in (torch.float16, torch.bfloat16):
        a = torch.randint(-2, 3, (2, 3, 70, 257), generator=g, device=DEV).to(dtype)
        b = torch.randint(-2, 3, (2, 3, 257, 40), generator=g, device=DEV).to(dtype)
        want = torch.matmul(a.double(), b.double())
        assert float(want.abs().max()) <= 256                           # exactly representable in bf16 as well
        assert torch.equal(ops.attn_matmul(a, b).double(), want)
        assert torch.equal(ops.attn_matmul(a, b.transpose(-1, -2).contiguous().transpose(-1, -2)).double(), want)


def test_a_product_has_the_same_bits_in_any_batch():
    """The property the grouped calibration replay rests on: sample j's scores and context are the same bits whether the
    launch holds 1, 3 or 12 samples, a subset of the heads, or a subset of the query rows."""
    from vlmc import ops
    g = torch.Generator(device=DEV).manual_seed(9)
    for (q, k, v) in (_eva(12, 257, 16, 88, g), _t5(12, 48, 72, 32, 64, g)):
        sc = ops.attn_matmul(q, k.transpose(-2, -1))
        p = torch.softmax(sc.float(), dim=-1).to(sc.dtype)
        ctx = ops.attn_matmul(p, v)
        for j in range(12):
            assert torch.equal(ops.attn_matmul(q[j:j + 1], k[j:j + 1].transpose(-2, -1)), sc[j:j + 1])
            assert torch.equal(ops.attn_matmul(p[j:j + 1], v[j:j + 1]), ctx[j:j + 1])
        for j in range(0, 12, 3):
            assert torch.equal(ops.attn_matmul(q[j:j + 3], k[j:j + 3].transpose(-2, -1)), sc[j:j + 3])
            assert torch.equal(ops.attn_matmul(p[j:j + 3], v[j:j + 3]), ctx[j:j + 3])
        assert torch.equal(ops.attn_matmul(q[:, 3:5], k[:, 3:5].transpose(-2, -1)), sc[:, 3:5])          # two of the heads
        assert torch.equal(ops.attn_matmul(q[:, :, 5:23], k.transpose(-2, -1)), sc[:, :, 5:23])          # some of the queries
        assert torch.equal(ops.attn_matmul(p[:, :, 5:23], v), ctx[:, :, 5:23])
        assert torch.equal(ops.attn_matmul(q, k.transpose(-2, -1)), sc)                                  # and run to run


def test_replayed_blocks_route_their_batched_matmuls_to_the_kernel():
    """Inside `forward.invariant_linears` (the replay of a block) `@`, `torch.matmul` and `torch.bmm` on 16-bit 3-D / 4-D
    tensors (and, since round 6, fp32 ones) run on vlmc_attn_matmul; 2-D products and calls with gradients stay with the library."""
    from vlmc import forward, ops
    g = torch.Generator(device=DEV).manual_seed(2)
    q, k, v = _t5(2, 9, 11, 4, 64, g)
    before = dict(forward.stats)
    with torch.no_grad(), forward.invariant_linears([]):
        s1 = q @ k.transpose(3, 2)
        s2 = torch.matmul(q, k.transpose(3, 2))
        s3 = torch.bmm(q.reshape(8, 9, 64), k.reshape(8, 11, 64).transpose(1, 2)).view(2, 4, 9, 11)
        s4 = q.float() @ k.float().transpose(3, 2)                       # fp32: the fp32 kernel (round 6: the reference's fp32 Q-Former)
        w = torch.randn(64, 64, device=DEV).bfloat16()
        s5 = q @ w                                                       # 4-D @ 2-D: library
    assert forward.stats["attn_kernel"] - before["attn_kernel"] == 4
    assert forward.stats["attn_library"] - before["attn_library"] == 1
    want = ops.attn_matmul(q, k.transpose(3, 2))
    assert torch.equal(s1, want) and torch.equal(s2, want) and torch.equal(s3, want)
    assert s4.dtype == torch.float32 and s5.shape == (2, 4, 9, 64)
    with forward.invariant_linears([]):                                  # gradients enabled: untouched
        qq = q.clone().requires_grad_()
        (qq @ k.transpose(3, 2)).sum().backward()
    assert forward.stats["attn_kernel"] - before["attn_kernel"] == 4 and qq.grad is not None
    assert "__matmul__" not in torch.Tensor.__dict__


def test_bad_arguments_fail_loudly():
    from vlmc import _lib, ops
    a = torch.randn(2, 2, 4, 8, device=DEV).half()
    with pytest.raises(TypeError):
        ops.attn_matmul(a.double(), a.double().transpose(-1, -2))             # (fp32 is taken since round 6; fp64 is not)
    with pytest.raises(TypeError):
        ops.attn_matmul(a, a.float().transpose(-1, -2))                      # mixed dtypes
    lib = _lib.load()
    rc = lib.vlmc_attn_matmul(a.data_ptr(), a.data_ptr(), a.data_ptr(), _lib.F16, 1, 1, 4, 4, 8, 0, 0, 8, 0, 0, 2, 2, 0, 0, 4, None)
    assert rc == _lib.VLMC_EINVAL and b"contiguous" in lib.vlmc_last_error()
    rc = lib.vlmc_attn_matmul(a.data_ptr(), a.data_ptr(), a.data_ptr(), _lib.F32, 1, 1, 4, 4, 8, 0, 0, 8, 0, 0, 2, 2, 0, 0, 4, None)
    assert rc == _lib.VLMC_EINVAL and b"contiguous" in lib.vlmc_last_error()      # fp32: B neither k- nor n-contiguous
    rc = lib.vlmc_attn_matmul(a.data_ptr(), a.data_ptr(), a.data_ptr(), 7, 1, 1, 4, 4, 8, 0, 0, 8, 0, 0, 1, 8, 0, 0, 4, None)
    assert rc == _lib.VLMC_EINVAL                                                 # unknown dtype
