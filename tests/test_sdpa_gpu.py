"""vlmc_sdpa_fwd (csrc/sdpa.hip): the fused attention `F.scaled_dot_product_attention(q, k, v)` stands for inside a replayed
block -- eva_vit.py:129-168 / modeling_t5.py:520-640 written fused.  Held against the unfused fp32 form with the probabilities
rounded to the operand dtype (what the kernel computes), for the shapes of the models (ViT-g: 16 heads x 257 tokens x 88; T5:
32 heads x 64 / 16 tokens x 64, cross attention 16 x 64), strided q / k / v views, ragged tails; and for batch invariance."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _ref(q, k, v, scale=None):
    d = q.shape[-1]
    s = (q.float() @ k.float().transpose(-2, -1)) * (d ** -0.5 if scale is None else scale)
    p = torch.softmax(s, dim=-1).to(q.dtype).float()
    return (p @ v.float())


def _qkv(B, H, Tq, Tk, d, dtype, seed, spread=1.0):
    g = torch.Generator(device=DEV).manual_seed(seed)
    mk = lambda T: (torch.randn(B, H, T, d, generator=g, device=DEV) * spread).to(dtype)
    return mk(Tq), mk(Tk), mk(Tk)


SHAPES = [(3, 16, 257, 257, 88, torch.float16), (2, 32, 64, 64, 64, torch.bfloat16), (2, 32, 16, 64, 64, torch.bfloat16),
          (5, 4, 16, 16, 64, torch.bfloat16), (2, 3, 1, 1, 8, torch.float16), (1, 2, 33, 47, 40, torch.float16),
          (2, 2, 100, 288, 96, torch.bfloat16), (1, 2, 70, 256, 128, torch.float16), (1, 3, 31, 255, 32, torch.bfloat16),
          (2, 5, 96, 96, 128, torch.float16), (1, 1, 200, 17, 72, torch.float16),
          (1, 5, 16, 16, 64, torch.bfloat16),            # 5 heads, 4 per workgroup: the last workgroup has empty slots
          (1, 3, 40, 64, 64, torch.float16),             # 3 heads, 2 per workgroup
          (1, 2, 900, 100, 64, torch.bfloat16),          # many queries: a head's blocks on three workgroups
          (2, 2, 33, 129, 32, torch.float16),            # head_dim 32 in the 64-wide instantiation, 129 keys
          (2, 12, 48, 257, 64, torch.float16),           # the Q-Former's cross-attention to the ViT's 257 tokens (head_dim 64: 288 keys fit since round 5)
          (1, 2, 20, 288, 64, torch.bfloat16)]


@pytest.mark.parametrize("B,H,Tq,Tk,d,dtype", SHAPES)
def test_sdpa_matches_the_unfused_fp32_form(B, H, Tq, Tk, d, dtype):
    from vlmc import ops
    q, k, v = _qkv(B, H, Tq, Tk, d, dtype, seed=Tq * 131 + Tk)
    out = ops.sdpa(q, k, v)
    assert out.shape == (B, H, Tq, d) and out.dtype == dtype
    want = _ref(q, k, v)
    # one rounding of O to the dtype, P rounded to the dtype: half an ulp of the output plus the P rounding carried through
    eps = 2.0 ** -10 if dtype is torch.float16 else 2.0 ** -7
    err = (out.float() - want).abs()
    bound = eps * (want.abs() + v.float().abs().amax(dim=2, keepdim=True)) + 1e-6
    assert bool((err <= bound).all()), float((err / bound).max())
    # and against torch's own fused kernel, loosely (another order of operations)
    lib = F.scaled_dot_product_attention(q, k, v).float()
    assert float((out.float() - lib).abs().max()) <= 8 * eps * float(v.float().abs().max())


def test_sdpa_reads_strided_views_in_place_and_takes_a_scale():
    """q, k, v as the model files make them: slices of one qkv projection, [B, T, 3, H, d] -> transpose(1, 2)."""
    from vlmc import ops
    B, T, H, d = 4, 257, 16, 88
    g = torch.Generator(device=DEV).manual_seed(3)
    qkv = torch.randn(B, T, 3 * H * d, generator=g, device=DEV).to(torch.float16)
    q, k, v = (t.reshape(B, T, H, d).transpose(1, 2) for t in qkv.reshape(B, T, 3, H * d).unbind(2))
    assert not q.is_contiguous()
    out = ops.sdpa(q, k, v)
    assert torch.equal(out, ops.sdpa(q.contiguous(), k.contiguous(), v.contiguous()))
    # the result is laid out [B, T, H, d]: the reshape every model file does next is a view
    y = out.transpose(1, 2).reshape(B, T, H * d)
    assert y.data_ptr() == out.data_ptr()
    got = ops.sdpa(q, k, v, scale=0.05)
    want = _ref(q, k, v, scale=0.05)
    assert float((got.float() - want).abs().max()) < 4e-3
    # extreme scores: no overflow, the row maximum is subtracted
    q2 = q * 40
    o2 = ops.sdpa(q2, k, v)
    assert bool(torch.isfinite(o2).all())
    assert float((o2.float() - _ref(q2, k, v)).abs().max()) < 2e-2


def test_sdpa_is_batch_invariant():
    """A sample's heads alone, in a group, at another position of the batch: the same bits."""
    from vlmc import ops
    for (B, H, Tq, Tk, d, dtype) in [(9, 16, 257, 257, 88, torch.float16), (12, 32, 16, 64, 64, torch.bfloat16)]:
        q, k, v = _qkv(B, H, Tq, Tk, d, dtype, seed=5)
        whole = ops.sdpa(q, k, v)
        for j in (0, 4, B - 1):
            alone = ops.sdpa(q[j:j + 1], k[j:j + 1], v[j:j + 1])
            assert torch.equal(alone[0], whole[j]), (j, Tq)
        perm = torch.randperm(B, device=DEV)
        assert torch.equal(ops.sdpa(q[perm], k[perm], v[perm]), whole[perm])
        # one head alone
        assert torch.equal(ops.sdpa(q[:, 3:4], k[:, 3:4], v[:, 3:4])[:, 0], whole[:, 3])


def test_calls_the_kernel_does_not_take_are_refused_or_left_to_torch(monkeypatch):
    from vlmc import forward, ops
    q, k, v = _qkv(1, 2, 8, 300, 64, torch.float16, seed=1)                  # more keys than LDS holds
    assert ops.sdpa(q, k, v, _try=True) is None
    with pytest.raises(TypeError):
        ops.sdpa(q, k, v)
    with pytest.raises(TypeError):
        ops.sdpa(q.float(), k.float(), v.float())
    monkeypatch.setenv("VLMC_LINEAR_FWD", "1")
    q, k, v = _qkv(2, 4, 20, 20, 64, torch.bfloat16, seed=2)
    s0 = dict(forward.stats)
    with torch.no_grad(), forward.invariant_matmuls():
        a = F.scaled_dot_product_attention(q, k, v)
        b = F.scaled_dot_product_attention(q, k, v, is_causal=True)              # the kernel's causal form
        c = F.scaled_dot_product_attention(q, k, v, attn_mask=torch.zeros(20, 20, device=DEV, dtype=q.dtype))    # torch's
        d = F.scaled_dot_product_attention(q, k, v, dropout_p=0.1)              # torch's
    assert forward.stats["sdpa_kernel"] == s0["sdpa_kernel"] + 2
    assert torch.equal(a, ops.sdpa(q, k, v)) and torch.equal(b, ops.sdpa(q, k, v, causal=True)) and d.shape == a.shape
    assert b.shape == a.shape and c.shape == a.shape
    assert F.scaled_dot_product_attention.__module__ != forward.__name__        # the patch is gone


def test_lds_dma_and_register_staging_give_the_same_bits(tmp_path):
    """K and V reach LDS by LDS-DMA when their rows are 16-byte aligned, through registers otherwise (`VLMC_SDPA_DMA=0` forces
    the latter; read once per process, hence the child processes); a misaligned view takes the register route by itself."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = f"""
import sys, torch
sys.path.insert(0, {os.path.join(root, 'vlm-compression_amd')!r})
from vlmc import ops
g = torch.Generator(device='cuda:0').manual_seed(11)
outs = []
for B, H, Tq, Tk, d, dt in [(3, 16, 257, 257, 88, torch.float16), (4, 8, 16, 64, 64, torch.bfloat16), (2, 4, 40, 100, 40, torch.float16)]:
    q, k, v = ((torch.randn(B, H, T, d, generator=g, device='cuda:0')).to(dt) for T in (Tq, Tk, Tk))
    outs.append(ops.sdpa(q, k, v).cpu())
torch.save(outs, sys.argv[1])
"""
    res = []
    for dma in ("1", "0"):
        out = tmp_path / f"dma{dma}.pt"
        r = subprocess.run([sys.executable, "-c", code, str(out)], env=dict(os.environ, VLMC_SDPA_DMA=dma), capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        res.append(torch.load(out))
    for a, b in zip(*res):
        assert torch.equal(a, b)
    # rows that are not 16-byte aligned (a view that starts 4 elements into a buffer): the same values as an aligned copy
    from vlmc import ops
    B, H, T, d = 2, 4, 50, 64
    buf = torch.randn(3, B * H * T * d + 8, device=DEV).to(torch.float16)
    q, k, v = (buf[i, 4:4 + B * H * T * d].view(B, H, T, d) for i in range(3))
    assert k.data_ptr() % 16 != 0
    assert torch.equal(ops.sdpa(q, k, v), ops.sdpa(q.clone(), k.clone(), v.clone()))


@pytest.mark.parametrize("B,H,Tq,Tk,d,dtype", [(3, 32, 96, 96, 128, torch.float16), (2, 4, 50, 50, 64, torch.bfloat16), (2, 16, 257, 257, 88, torch.float16),
                                               (2, 3, 16, 16, 32, torch.float16), (1, 2, 40, 100, 64, torch.bfloat16), (1, 2, 130, 70, 96, torch.float16)])
def test_causal_attention_matches_the_masked_fp32_form(B, H, Tq, Tk, d, dtype):
    """is_causal=True (the self-attention of decoder-only towers, modeling_llama.py): key j counts for query i iff j <= i, the
    mask aligned to the top left as torch's is, also when Tq != Tk."""
    from vlmc import ops
    q, k, v = _qkv(B, H, Tq, Tk, d, dtype, seed=Tq + 3 * Tk)
    out = ops.sdpa(q, k, v, causal=True)
    s = (q.float() @ k.float().transpose(-2, -1)) * d ** -0.5
    keep = torch.ones(Tq, Tk, dtype=torch.bool, device=DEV).tril()
    p = torch.softmax(s.masked_fill(~keep, float("-inf")), dim=-1).to(dtype).float()
    want = p @ v.float()
    eps = 2.0 ** -10 if dtype is torch.float16 else 2.0 ** -7
    err = (out.float() - want).abs()
    bound = eps * (want.abs() + v.float().abs().amax(dim=2, keepdim=True)) + 1e-6
    assert bool((err <= bound).all()), float((err / bound).max())
    lib = F.scaled_dot_product_attention(q, k, v, is_causal=True).float()
    assert float((out.float() - lib).abs().max()) <= 8 * eps * float(v.float().abs().max())
    # batch-invariant like the plain form
    assert torch.equal(ops.sdpa(q[1:2] if B > 1 else q, k[1:2] if B > 1 else k, v[1:2] if B > 1 else v, causal=True)[0], out[1 if B > 1 else 0])
