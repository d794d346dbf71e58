"""GPU parity of the SparseLoRA kernels (C ABI: vlmc_lora_effective_weight, vlmc_lora_grad) and
of the drop-in `lora.Linear` against the reference's golden vectors and the CPU oracle.

Tolerances (floating point): fp32 -> rtol 1e-4 on outputs/gradients (library GEMMs accumulate in
a different order than the CPU), the effective weight itself to 1e-6; bf16 -> the effective /
merged weights must agree with the reference's on >= 99.5 % of the entries bit for bit and
never differ by more than one bf16 ulp (the rank-r fp32 accumulation order can move a value
across a rounding boundary), outputs and gradients to rtol 3e-2."""
import contextlib

import pytest
import torch

import golden_io
from oracle import sparse_lora as OL

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
G = golden_io.load("sparse_lora")
CASES = {"fp32": (torch.float32, None), "bf16": (torch.bfloat16, None), "bf16_autocast": (torch.bfloat16, torch.bfloat16),
         "fp32_autocast_bf16": (torch.float32, torch.bfloat16)}


def _close_lowp(got, ref, name, min_exact=0.995):
    got, ref = got.float().cpu(), ref.float()
    exact = (got == ref).float().mean().item()
    assert exact >= min_exact, f"{name}: only {exact:.4f} of the entries bit-identical"
    ulp = ref.abs().clamp_min(1e-30) * 2.0 ** -7
    assert bool(((got - ref).abs() <= ulp + 1e-12).all()), f"{name}: differs by more than one bf16 ulp"


def _module(wd, sparse):
    from lavis.peft.src.peft.tuners.lora import Linear
    lin = Linear(96, 80, r=int(G["r"]), lora_alpha=int(G["alpha"]), bias=True)
    with torch.no_grad():
        lin.weight.copy_(G["W"]); lin.bias.copy_(G["b"])
        lin.lora_A.weight.copy_(G["A"]); lin.lora_B.weight.copy_(G["B"])
    lin.weight.data = lin.weight.data.to(wd)
    lin.bias.data = lin.bias.data.to(wd)
    lin.mask = G["M"].clone()
    lin.sparse = sparse
    return lin.to(DEV)


@pytest.mark.parametrize("cname", list(CASES))
@pytest.mark.parametrize("sparse", [True, False])
def test_linear_forward_backward_matches_reference(cname, sparse):
    from vlmc import sparse_lora as SL
    wd, ac = CASES[cname]
    lin = _module(wd, sparse)
    key = f"{cname}/sparse{int(sparse)}"
    code = {None: 0, torch.float16: 1, torch.bfloat16: 2}[ac]
    weff = SL.effective_weight(lin.weight.data, lin.lora_A.weight.data, lin.lora_B.weight.data, lin.mask, lin.scaling,
                               SL.FWD_SPARSE if sparse else SL.FWD_MASKED, code)
    if wd == torch.float32 and ac is None:
        torch.testing.assert_close(weff.cpu(), G[f"{key}/weff"], rtol=1e-6, atol=1e-7)
    else:
        _close_lowp(weff, G[f"{key}/weff"], key + "/weff")
    x = G["X"].to(wd if ac is None else torch.float32).to(DEV).requires_grad_(True)
    ctx = torch.autocast("cuda", dtype=ac) if ac is not None else contextlib.nullcontext()
    with ctx:
        y = lin(x)
    assert y.dtype == wd                                          # lora.py:379-380
    y.backward(G["GY"].to(y.dtype).to(DEV))
    lowp = not (wd == torch.float32 and ac is None)
    tol = dict(rtol=3e-2, atol=3e-3) if lowp else dict(rtol=1e-4, atol=1e-5)   # fp32: GEMM summation order only
    torch.testing.assert_close(y.detach().float().cpu(), G[f"{key}/y"].float(), **tol)
    torch.testing.assert_close(x.grad.float().cpu(), G[f"{key}/gx"].float(), **tol)
    torch.testing.assert_close(lin.lora_A.weight.grad.cpu(), G[f"{key}/gA"], **tol)
    torch.testing.assert_close(lin.lora_B.weight.grad.cpu(), G[f"{key}/gB"], **tol)
    assert lin.weight.grad is None and lin.lora_A.weight.grad.dtype == torch.float32


@pytest.mark.parametrize("cname", ["fp32", "bf16"])
@pytest.mark.parametrize("sparse", [True, False])
def test_dense_path_and_merge_match_reference(cname, sparse):
    wd, _ = CASES[cname]
    lin = _module(wd, sparse)
    key = f"{cname}/sparse{int(sparse)}"
    with torch.no_grad():
        yd = lin(G["X"].to(wd).to(DEV), dense=True)
    tol = dict(rtol=3e-2, atol=3e-3) if wd != torch.float32 else dict(rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(yd.float().cpu(), G[f"{key}/y_dense"].float(), **tol)
    lin.merge()
    if wd == torch.float32:
        torch.testing.assert_close(lin.weight.data.cpu(), G[f"{key}/merged"], rtol=1e-6, atol=1e-7)
    else:
        _close_lowp(lin.weight.data, G[f"{key}/merged"], key + "/merged")
    assert bool((lin.lora_B.weight.data == 0).all())              # reset_peft(), lora.py:393


@pytest.mark.parametrize("shape", [(300, 520, 16), (64, 1000, 8), (4096, 4096, 16), (33, 72, 3), (5120, 2048, 33)])
@pytest.mark.parametrize("wd", [torch.bfloat16, torch.float16, torch.float32])
def test_kernels_vs_oracle_random_shapes(shape, wd):
    """effective weight (4 modes) and adapter gradients against the CPU oracle's autograd."""
    from vlmc import sparse_lora as SL
    out_f, in_f, r = shape
    g = torch.Generator().manual_seed(out_f + in_f + r)
    W = (torch.randn(out_f, in_f, generator=g) * 0.05).to(wd)
    A = torch.randn(r, in_f, generator=g) * 0.1
    B = torch.randn(out_f, r, generator=g) * 0.1
    M = torch.rand(out_f, in_f, generator=g) > 0.5
    Gw = (torch.randn(out_f, in_f, generator=g) * 0.1).to(wd)
    s = 16 / r
    Wd, Ad, Bd, Md, Gd = (t.to(DEV) for t in (W, A, B, M, Gw))
    for sparse in (True, False):
        Ar, Br = A.clone().requires_grad_(True), B.clone().requires_grad_(True)
        want = OL.effective_weight(W, Ar, Br, M, s, sparse)
        want.backward(Gw)
        got = SL.effective_weight(Wd, Ad, Bd, Md, s, SL.FWD_SPARSE if sparse else SL.FWD_MASKED, 0)
        gA, gB = SL.lora_grads(Gd, Ad, Bd, Md, s, sparse, autocast=0)
        merged = SL.effective_weight(Wd, Ad, Bd, Md, s, SL.MERGE_SPARSE if sparse else SL.MERGE_MASKED, 0)
        want_m = OL.merge(W, A, B, M, s, sparse)
        if wd == torch.float32:
            torch.testing.assert_close(got.cpu(), want.detach(), rtol=1e-5, atol=1e-6)
            torch.testing.assert_close(merged.cpu(), want_m, rtol=1e-5, atol=1e-6)
            gt = dict(rtol=1e-4, atol=1e-5)
        else:
            _close_lowp(got, want.detach(), "weff", min_exact=0.99)
            _close_lowp(merged, want_m, "merged", min_exact=0.99)
            gt = dict(rtol=2e-2, atol=2e-3 * (in_f / 500) ** 0.5)
        torch.testing.assert_close(gA.cpu(), Ar.grad, **gt)
        torch.testing.assert_close(gB.cpu(), Br.grad, **gt)


def test_lora_model_replacement_and_training_step():
    """LoraModel swaps targeted linears, shares weights, freezes everything but lora_*; one AdamW
    step on the sparse path changes only A and B and keeps pruned positions of W_eff at zero."""
    import torch.nn as nn
    from lavis.peft.src.peft.tuners.lora import Linear, LoraConfig, LoraModel
    from vlmc import sparse_lora as SL

    class Tiny(nn.Module):
        def __init__(self):
            super().__init__()
            self.q = nn.Linear(64, 64, bias=False)
            self.fc = nn.Linear(64, 32)
            self.head = nn.Linear(32, 8)

        def forward(self, x, dense=False):
            return self.head(torch.relu(self.fc(self.q(x, dense=dense), dense=dense)))

    base = Tiny().to(DEV)
    wq = base.q.weight
    model = LoraModel(LoraConfig(r=4, lora_alpha=16, target_modules=["q", "fc"], lora_dropout=0.0), base)
    assert isinstance(base.q, Linear) and isinstance(base.fc, Linear) and not isinstance(base.head, Linear)
    assert base.q.weight is wq
    trainable = [n for n, p in model.named_parameters() if p.requires_grad]
    assert trainable and all("lora_" in n for n in trainable)
    for m in (base.q, base.fc):
        m.mask = (torch.rand_like(m.weight) > 0.5)
        m.sparse = True
        with torch.no_grad():
            m.lora_B.weight.normal_(0, 0.05)
    opt = torch.optim.AdamW([p for p in model.parameters() if p.requires_grad], lr=1e-2)
    x = torch.randn(16, 64, device=DEV)
    before = {n: p.detach().clone() for n, p in model.named_parameters()}
    loss = model(x).pow(2).mean()
    loss.backward()
    opt.step()
    for n, p in model.named_parameters():
        changed = not torch.equal(p.detach(), before[n])
        assert changed == ("lora_" in n), n
    weff = SL.effective_weight(base.q.weight.data, base.q.lora_A.weight.data, base.q.lora_B.weight.data, base.q.mask,
                               base.q.scaling, SL.FWD_SPARSE, 0)
    assert bool((weff[~base.q.mask] == 0).all())
