"""`vlmc_row_mean` (csrc/row_reduce.hip): the fp32 mean over the hidden dimension inside T5LayerNorm / LlamaRMSNorm during a
replay -- against fp64, and against ITSELF for any number of rows in the launch (torch's own reduction is configured by the
number of outputs: the reason this kernel exists)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("n", [1, 3, 64, 255, 256, 1408, 2048, 4096, 5120, 11008])
def test_mean_against_fp64_and_for_any_row_count(n):
    from vlmc import ops
    g = torch.Generator(device=DEV).manual_seed(n)
    x = torch.randn(515, n, generator=g, device=DEV).pow(2)
    got = ops.row_mean(x, keepdim=True)
    want = x.double().mean(-1, keepdim=True)
    assert got.shape == (515, 1) and got.dtype == torch.float32
    assert float(((got.double() - want).abs() / want.abs().clamp_min(1e-30)).max()) < 2e-6
    for rows in (1, 4, 64):                                     # a row's mean does not depend on what else is in the launch
        assert torch.equal(ops.row_mean(x[:rows], keepdim=True), got[:rows])
    assert torch.equal(ops.row_mean(x[100:104].reshape(2, 2, n)), got[100:104, 0].reshape(2, 2))
    wide = torch.randn(9, 2 * n + 3, generator=g, device=DEV)
    assert torch.equal(ops.row_mean(wide[:, 1:n + 1]), ops.row_mean(wide[:, 1:n + 1].contiguous()))     # strided, unaligned rows


def test_replay_routes_the_norms_mean_to_the_kernel_and_nothing_else():
    from vlmc import forward, ops
    x = torch.randn(2, 4, 2048, device=DEV).bfloat16()
    before = dict(forward.stats)
    with torch.no_grad(), forward.invariant_linears([]):
        v = x.to(torch.float32).pow(2).mean(-1, keepdim=True)                 # transformers' T5LayerNorm
        w = torch.mean(x.float(), dim=-1)
        lib1 = x.float().mean()                                               # full reduction: torch's
        lib2 = x.float().mean(1)                                              # another dimension: torch's
        lib3 = x.mean(-1)                                                     # bf16: torch's
    assert forward.stats["mean_kernel"] - before["mean_kernel"] == 2
    assert torch.equal(v, ops.row_mean(x.float().pow(2), keepdim=True)) and torch.equal(w, ops.row_mean(x.float()))
    assert lib1.dim() == 0 and lib2.shape == (2, 2048) and lib3.dtype == torch.bfloat16
    assert "mean" not in torch.Tensor.__dict__
    xg = x.float().requires_grad_()
    with forward.invariant_linears([]):
        xg.mean(-1).sum().backward()                                          # with gradients: untouched
    assert forward.stats["mean_kernel"] - before["mean_kernel"] == 2 and xg.grad is not None


@pytest.mark.parametrize("which,T", [("vit", 257), ("enc0", 40), ("enc", 160), ("dec", 16), ("dec", 4)])
def test_reference_op_blocks_at_model_width_are_batch_invariant(which, T):
    """One block of the stand-in that follows the reference's op sequence (eva_vit.py:129-168, modeling_t5.py:520-640) at
    InstructBLIP-FlanT5-XL's width: 32 samples in one forward == 32 forwards of one sample, bit for bit -- linears on
    vlmc_linear_fwd, attention products on vlmc_attn_matmul, the norms' mean on vlmc_row_mean.  (T = 4: a 4-token answer --
    torch's own mean reduction gives those rows other bits alone than stacked.)"""
    from vlmc import forward, synthetic as S
    from lavis.compression.pruners import calibration as cal
    torch.manual_seed(0)
    g = torch.Generator(device=DEV).manual_seed(1)
    n = 32
    if which == "vit":
        blk = S.ViTBlock(1408, 6144, 16, reference_ops=True).to(DEV).half().eval()
        xs = [(torch.randn(1, T, 1408, generator=g, device=DEV) * 0.5).half() for _ in range(n)]
        kws = [{} for _ in xs]
        call = lambda x, kw: blk(x, None)
    else:
        blk = S.T5Block(2048, 5120, 32, 64, which == "dec", True, which == "enc0").to(DEV).bfloat16().eval()
        xs = [(torch.randn(1, T, 2048, generator=g, device=DEV) * 0.5).bfloat16() for _ in range(n)]
        kws = [dict(encoder_hidden_states=(torch.randn(1, 72, 2048, generator=g, device=DEV) * 0.5).bfloat16()) if which == "dec" else {}
               for _ in xs]
        call = lambda x, kw: blk(x, **kw)[0]
    S.randomize_(blk, seed=3)
    subset = cal.find_layers(blk)
    before = dict(forward.stats)
    with torch.no_grad(), forward.invariant_linears(subset.values()):
        one = torch.cat([call(x, kw) for x, kw in zip(xs, kws)])
        allg = call(torch.cat(xs), {k: torch.cat([kw[k] for kw in kws]) for k in kws[0]})
    assert torch.equal(one, allg)
    assert forward.stats["attn_kernel"] + forward.stats["attn_fused"] > before["attn_kernel"] + before["attn_fused"] and \
        forward.stats["attn_library"] == before["attn_library"]
    if which != "vit":
        assert forward.stats["mean_kernel"] > before["mean_kernel"]
