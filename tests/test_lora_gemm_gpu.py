"""GPU parity of the fused masked-LoRA GEMMs (C ABI: vlmc_sparse_lora_prep / _fwd / _bwd_input / _bwd_weight,
csrc/lora_gemm.hip) -- the `r > 0 and not merged` branch of lavis/peft/src/peft/tuners/lora.py:359-380 and its autograd
with neither W_eff nor G = dY^T x in memory.

What is exact and what has a tolerance:
* the GENERATED weight tiles: with X = I the forward returns W_eff^T and with dY = I the backward returns W_eff, each
  entry ONE product by 1.0 -- compared bit for bit with `vlmc_lora_effective_weight` (pinned to the reference's goldens
  by tests/test_sparse_lora_gpu.py) and with the CPU oracle: >= 99.9 % of the entries identical, never more than two ulps of
  the larger addend apart (the rank-r sum runs on the 16-bit MFMA here, on an fp32 fma chain there: delta can cross a
  rounding boundary, and wd(W + delta) can then round the other way);
* outputs and input gradients: fp32 accumulation of 16-bit products in another order than the library GEMM's:
  rtol 2e-2 (bf16) / 4e-3 (fp16) of the row's scale;
* adapter gradients: rounded to the autocast dtype at the end like the reference's: one ulp + summation order."""
import pytest
import torch

from oracle import sparse_lora as OL

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
CODE = {torch.float16: 1, torch.bfloat16: 2}
ULP = {torch.float16: 2.0 ** -10, torch.bfloat16: 2.0 ** -7}
V7B = [(4096, 4096), (11008, 4096), (4096, 11008)]          # q/k/v/o, gate/up, down (modeling_llama.py:160,204-206,253)


def _layer(out_f, in_f, r, wd, seed=0, bias=False):
    g = torch.Generator().manual_seed(seed + out_f * 7 + in_f * 3 + r)
    W = (torch.randn(out_f, in_f, generator=g) * 0.05).to(wd)
    A = torch.randn(r, in_f, generator=g) * 0.1
    B = torch.randn(out_f, r, generator=g) * 0.1
    M = torch.rand(out_f, in_f, generator=g) > 0.5
    b = (torch.randn(out_f, generator=g) * 0.1).to(wd) if bias else None
    return W, A, B, M, b


def _fused(x, W, A, B, M, b, s, sparse):
    from vlmc import sparse_lora as SL
    assert SL.fused_supported(W, A, CODE[W.dtype]), "this case must take the fused kernels"
    return SL._FusedSparseLoRALinear.apply(x, W, A, B, M, b, float(s), bool(sparse), CODE[W.dtype])


def _unfused(x, W, A, B, M, b, s, sparse):
    from vlmc import sparse_lora as SL
    return SL._SparseLoRALinear.apply(x, W, A, B, M, b, float(s), bool(sparse), CODE[W.dtype])


def _same_bits_or_one_ulp(got, ref, wd, name, W, min_exact=0.999):
    """Two ulps of the LARGER of the two addends: W_eff = wd(W + delta) can cancel; a delta one ulp off stays one ulp of delta
    off, and the sum's own rounding may then fall the other way."""
    got, ref = got.float(), ref.float()
    exact = (got == ref).float().mean().item()
    assert exact >= min_exact, f"{name}: only {exact:.5f} of the entries bit-identical"
    tol = (ref.abs() + W.float().to(ref.device).abs()).clamp_min(2.0 ** -14) * ULP[wd] * 2.002
    worst = ((got - ref).abs() - tol).max().item()
    assert worst <= 0, f"{name}: an entry differs by more than two ulps of its addends ({worst})"
    return exact


@pytest.mark.parametrize("wd", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("sparse", [True, False])
@pytest.mark.parametrize("shape", [(128, 64, 16), (192, 320, 8), (704, 256, 3), (256, 1408, 4)])
@pytest.mark.parametrize("s", [1.0, 2.0, 16 / 3])
def test_generated_tiles_are_the_effective_weight(shape, wd, sparse, s):
    """X = I -> Y = W_eff^T (K-major generator);  dY = I -> dX = W_eff (transposing-read image)."""
    from vlmc import sparse_lora as SL
    out_f, in_f, r = shape
    W, A, B, M, _ = _layer(out_f, in_f, r, wd)
    Wd, Ad, Bd, Md = (t.to(DEV) for t in (W, A, B, M))
    want = SL.effective_weight(Wd, Ad, Bd, Md, s, SL.FWD_SPARSE if sparse else SL.FWD_MASKED, CODE[wd])
    eye_in = torch.eye(in_f, dtype=wd, device=DEV)
    y = _fused(eye_in, Wd, Ad, Bd, Md, None, s, sparse)
    _same_bits_or_one_ulp(y.t(), want, wd, "forward", W)
    eye_out = torch.eye(out_f, dtype=wd, device=DEV).requires_grad_(True)
    x = torch.zeros(out_f, in_f, dtype=wd, device=DEV, requires_grad=True)
    yy = _fused(x, Wd, Ad, Bd, Md, None, s, sparse)
    (gx,) = torch.autograd.grad(yy, x, eye_out.detach())
    _same_bits_or_one_ulp(gx, want, wd, "backward", W)
    # and the CPU oracle on the autocast-rounded factors (the reference's `B @ A` under autocast)
    ref = OL.effective_weight(W, A.to(wd).float(), B.to(wd).float(), M, s, sparse)
    _same_bits_or_one_ulp(y.t().cpu(), ref, wd, "forward vs oracle", W, min_exact=0.995)


@pytest.mark.parametrize("wd", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("sparse", [True, False])
@pytest.mark.parametrize("shape,M", [((128, 64, 16), 1), ((192, 320, 8), 77), ((256, 256, 16), 256), ((704, 1408, 4), 300), ((1408, 704, 2), 513)])
def test_forward_backward_against_the_unfused_kernels(shape, M, wd, sparse):
    out_f, in_f, r = shape
    W, A, B, Mk, b = _layer(out_f, in_f, r, wd, bias=True)
    g = torch.Generator().manual_seed(M)
    x = torch.randn(M, in_f, generator=g).to(wd)
    gy = (torch.randn(M, out_f, generator=g) * 0.1).to(wd)
    res = []
    for fn in (_fused, _unfused):
        Wd, Md, bd = W.to(DEV), Mk.to(DEV), b.to(DEV).requires_grad_(True)
        Ad, Bd = A.to(DEV).requires_grad_(True), B.to(DEV).requires_grad_(True)
        xd = x.to(DEV).requires_grad_(True)
        y = fn(xd, Wd, Ad, Bd, Md, bd, 16 / r, sparse)
        y.backward(gy.to(DEV))
        res.append([t.float().cpu() for t in (y.detach(), xd.grad, Ad.grad, Bd.grad, bd.grad)])
    rt = 2e-2 if wd == torch.bfloat16 else 4e-3
    for name, got, ref in zip(("y", "gx", "gA", "gB", "gbias"), *res):
        scale = ref.abs().max().item() + 1e-6
        torch.testing.assert_close(got, ref, rtol=rt, atol=rt * scale * 0.25, msg=lambda m, n=name: f"{n}: {m}")


@pytest.mark.parametrize("wd", [torch.float16, torch.bfloat16])
def test_weight_gradients_against_the_oracle_autograd(wd):
    """dA, dB of the fused epilogue against autograd on the oracle's expression fed the SAME rounded G = dY^T x."""
    out_f, in_f, r, M = 320, 192, 8, 100
    W, A, B, Mk, _ = _layer(out_f, in_f, r, wd, seed=5)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(M, in_f, generator=g).to(wd)
    gy = (torch.randn(M, out_f, generator=g) * 0.1).to(wd)
    for sparse in (True, False):
        Ad, Bd = A.to(DEV).requires_grad_(True), B.to(DEV).requires_grad_(True)
        y = _fused(x.to(DEV), W.to(DEV), Ad, Bd, Mk.to(DEV), None, 2.0, sparse)
        y.backward(gy.to(DEV))
        # the reference's chain: G in the GEMM dtype; (B16 @ A16) in the autocast dtype, its gradients rounded to it
        G = (gy.float().t() @ x.float()).to(wd)
        A16, B16 = A.to(wd).float().requires_grad_(True), B.to(wd).float().requires_grad_(True)
        delta = (B16 @ A16).to(wd)
        delta.retain_grad()
        weff = (W + delta * 2.0) * Mk if sparse else W * Mk + delta * 2.0
        weff.backward(G)
        # (a sum of ~200 signed terms: an entry of G that rounds the other way on the CPU moves a small sum by ulps of its TERMS)
        for got, ref in ((Ad.grad.cpu(), A16.grad.to(wd).float()), (Bd.grad.cpu(), B16.grad.to(wd).float())):
            torch.testing.assert_close(got, ref, rtol=4 * ULP[wd], atol=8 * ULP[wd] * ref.abs().max().item())


def _oracle_autograd(x, gy, W, A, B, Mk, s, sparse, wd):
    """Forward and gradients of lora.py:359-380 under 16-bit autocast by autograd on the ORACLE's expression (oracle/sparse_lora.py:
    effective_weight), CPU: the fp32 adapters enter as autocast casts them (rounded to `wd`; their gradients come back rounded to
    it), `B @ A` leaves the autocast matmul in `wd`, the main product accumulates in fp32 and is rounded once.
    Returns (y, dX, dA, dB) as fp32 tensors."""
    xl = x.clone().requires_grad_(True)
    A16, B16 = A.to(wd).float().requires_grad_(True), B.to(wd).float().requires_grad_(True)
    weff = OL.effective_weight(W, A16, B16, Mk, s, sparse)
    assert weff.dtype == wd
    y = (xl.float() @ weff.float().t()).to(wd)
    y.backward(gy)
    return y.detach().float(), xl.grad.float(), A16.grad.to(wd).float(), B16.grad.to(wd).float()


@pytest.mark.parametrize("out_f,in_f", V7B)
@pytest.mark.parametrize("sparse", [True, False])
def test_vicuna_7b_shapes(out_f, in_f, sparse):
    """All seven SparseLoRA linears of a Vicuna-7B layer (three distinct shapes), r = 16, fp16 autocast, one RESSA
    micro-batch of 16 x (32 + 64) tokens: generated tiles exact; outputs, dX, dA, dB against the unfused kernels (every entry) and
    against the oracle's autograd (64 rows and 64 columns of the layer, each a complete sub-problem)."""
    from vlmc import sparse_lora as SL
    wd, r, M, s = torch.float16, 16, 16 * 96, 1.0
    W, A, B, Mk, _ = _layer(out_f, in_f, r, wd)
    Wd, Md = W.to(DEV), Mk.to(DEV)
    want = SL.effective_weight(Wd, A.to(DEV), B.to(DEV), Md, s, SL.FWD_SPARSE if sparse else SL.FWD_MASKED, 1)
    y = _fused(torch.eye(in_f, dtype=wd, device=DEV), Wd, A.to(DEV), B.to(DEV), Md, None, s, sparse)
    _same_bits_or_one_ulp(y.t(), want, wd, "forward tiles", W)
    del y
    x0 = torch.zeros(out_f, in_f, dtype=wd, device=DEV, requires_grad=True)
    (gx,) = torch.autograd.grad(_fused(x0, Wd, A.to(DEV), B.to(DEV), Md, None, s, sparse), x0, torch.eye(out_f, dtype=wd, device=DEV))
    _same_bits_or_one_ulp(gx, want, wd, "backward tiles", W)
    del gx, x0, want
    g = torch.Generator().manual_seed(3)
    x = torch.randn(M, in_f, generator=g).to(wd)
    gy = (torch.randn(M, out_f, generator=g) * 0.05).to(wd)
    res = []
    for fn in (_fused, _unfused):
        Ad, Bd = A.to(DEV).requires_grad_(True), B.to(DEV).requires_grad_(True)
        xd = x.to(DEV).requires_grad_(True)
        y = fn(xd, Wd, Ad, Bd, Md, None, s, sparse)
        y.backward(gy.to(DEV))
        res.append([t.float().cpu() for t in (y.detach(), xd.grad, Ad.grad, Bd.grad)])
    for name, got, ref in zip(("y", "gx", "gA", "gB"), *res):
        scale = ref.abs().max().item()
        torch.testing.assert_close(got, ref, rtol=4e-3, atol=2e-3 * scale, msg=lambda m, n=name: f"{n}: {m}")
    # ---- and DIRECTLY against the oracle's autograd, on slices that are complete sub-problems -----------------
    # rows J of W: y[:, J] and dB[J] depend on nothing else; columns I of W: x[:, I] @ W_eff[:, I]^T is one addend of y whose
    # gradients with respect to x[:, I] and A[:, I] are dX[:, I] and dA[:, I] of the whole layer.
    y, gx, gA, gB = res[0]
    gi = torch.Generator().manual_seed(out_f + in_f)
    J = torch.randperm(out_f, generator=gi)[:64].sort().values
    yo, _, _, gBo = _oracle_autograd(x, gy[:, J], W[J], A, B[J], Mk[J], s, sparse, wd)
    torch.testing.assert_close(y[:, J], yo, rtol=4e-3, atol=2e-3 * yo.abs().max().item(), msg=lambda m: f"y[:, J] vs oracle: {m}")
    torch.testing.assert_close(gB[J], gBo, rtol=4 * ULP[wd], atol=8 * ULP[wd] * gBo.abs().max().item(), msg=lambda m: f"dB[J] vs oracle: {m}")
    I = torch.randperm(in_f, generator=gi)[:64].sort().values
    _, gxo, gAo, _ = _oracle_autograd(x[:, I], gy, W[:, I], A[:, I], B, Mk[:, I], s, sparse, wd)
    torch.testing.assert_close(gx[:, I], gxo, rtol=4e-3, atol=2e-3 * gxo.abs().max().item(), msg=lambda m: f"dX[:, I] vs oracle: {m}")
    torch.testing.assert_close(gA[:, I], gAo, rtol=4 * ULP[wd], atol=8 * ULP[wd] * gAo.abs().max().item(), msg=lambda m: f"dA[:, I] vs oracle: {m}")


def test_drop_in_module_takes_the_fused_route_under_autocast_and_nothing_weight_sized_is_saved():
    from lavis.peft.src.peft.tuners.lora import Linear
    from vlmc import sparse_lora as SL
    lin = Linear(256, 384, r=8, lora_alpha=16, bias=True).to(DEV)
    lin.weight.data = lin.weight.data.half()
    lin.bias.data = lin.bias.data.half()
    lin.mask = torch.rand(384, 256, device=DEV) > 0.5
    lin.sparse = True
    with torch.no_grad():
        lin.lora_B.weight.normal_(0, 0.05)
    x = torch.randn(4, 24, 256, device=DEV, requires_grad=True)
    with torch.autocast("cuda", dtype=torch.float16):
        y = lin(x)
    assert type(y.grad_fn).__name__ == "_FusedSparseLoRALinearBackward"
    saved = [t for t in y.grad_fn.saved_tensors if t.numel() >= lin.weight.numel() and t.data_ptr() not in (lin.weight.data_ptr(), lin.mask.data_ptr())]
    assert not saved, "W_eff (or anything of its size) is kept for backward"
    y.float().pow(2).mean().backward()
    assert x.grad is not None and lin.lora_A.weight.grad is not None and lin.lora_B.weight.grad is not None
    # outside autocast, or with a width that is no multiple of 64, the unfused kernels take over
    assert not SL.fused_supported(lin.weight, lin.lora_A.weight, 0)
    assert not SL.fused_supported(torch.empty(96, 80, dtype=torch.float16), lin.lora_A.weight, 1)


def test_entry_points_refuse_what_they_do_not_implement():
    from vlmc import _lib
    lib = _lib.load()
    W = torch.zeros(128, 128, dtype=torch.float16, device=DEV)
    M = torch.ones(128, 128, dtype=torch.bool, device=DEV)
    prep = torch.empty(lib.vlmc_sparse_lora_prep_bytes(128, 128), dtype=torch.uint8, device=DEV)
    x = torch.zeros(8, 128, dtype=torch.float16, device=DEV)
    y = torch.empty(8, 128, dtype=torch.float16, device=DEV)
    args = lambda dtype, out_f, in_f, r, ac: (x.data_ptr(), 8, 128, W.data_ptr(), dtype, out_f, in_f, 128, M.data_ptr(), prep.data_ptr(), r, 1.0, 1, ac,
                                              None, y.data_ptr(), 128, None)
    assert lib.vlmc_sparse_lora_fwd(*args(_lib.F32, 128, 128, 8, 1)) == _lib.VLMC_EINVAL
    assert lib.vlmc_sparse_lora_fwd(*args(_lib.F16, 128, 128, 17, 1)) == _lib.VLMC_EINVAL
    assert lib.vlmc_sparse_lora_fwd(*args(_lib.F16, 128, 128, 8, 2)) == _lib.VLMC_EINVAL
    assert lib.vlmc_sparse_lora_fwd(*args(_lib.F16, 96, 128, 8, 1)) == _lib.VLMC_EINVAL
    assert b"multiples of 64" in lib.vlmc_last_error()
