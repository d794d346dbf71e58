"""GPU: `vlmc_linear_fwd_post` -- a linear with the elementwise op(s) the model applies to its output folded into the GEMM's epilogue --
against `vlmc_linear_fwd` followed by those torch ops: bit for bit (that is what lets vlmc/forward.py swap one for the other)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _bits(t):
    return t.contiguous().view(torch.int16)


def _same_as_torch_gelu(got, want):
    """torch's vectorized elementwise kernel runs the LAST partial block of a tensor through another instruction sequence (there hipcc
    contracts x/2 * (1 + erf) into fma(x/2, erf, x/2): gelu(-6.7) is +0.0 in the tail and -0.0 in the body, and ~20 % of all inputs
    differ in the last bit -- tools/micro/gelu_variants.py); the epilogue uses the body's arithmetic for every element.  Bit-equal
    on everything but torch's tail block, within one ulp there."""
    g, w = _bits(got).flatten(), _bits(want).flatten()
    body = max(0, g.numel() - 4096)
    tail_ok = ((g[body:].int() - w[body:].int()).abs() <= 1) | ((got.flatten()[body:] == 0) & (want.flatten()[body:] == 0))
    return torch.equal(g[:body], w[:body]) and bool(tail_ok.all())


SHAPES = [(257 * 3, 6144, 1408), (300, 1408, 6144), (64, 2048, 2048), (1, 512, 64), (33, 200, 72), (4096 + 17, 1408, 1408)]


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("M,N,K", SHAPES)
def test_post_ops_have_the_bits_of_the_separate_torch_ops(dtype, M, N, K):
    from vlmc import ops
    g = torch.Generator(device=DEV).manual_seed(M + N + K)
    x = (torch.randn(M, K, generator=g, device=DEV) * 0.5).to(dtype)
    w = (torch.randn(N, K, generator=g, device=DEV) * 0.05).to(dtype)
    b = torch.randn(N, generator=g, device=DEV).to(dtype)
    pb = torch.randn(N, generator=g, device=DEV).to(dtype)
    res = torch.randn(M, N, generator=g, device=DEV).to(dtype)
    y = ops.linear_fwd(x, w, b)
    assert torch.equal(_bits(ops.linear_fwd_post(x, w, b)), _bits(y))                                     # no post-op: the plain product
    assert _same_as_torch_gelu(ops.linear_fwd_post(x, w, b, act=1), F.gelu(y))                             # self.act(self.fc1(x))
    assert torch.equal(_bits(ops.linear_fwd_post(x, w, b, residual=res)), _bits(res + y))                  # x + mlp(..)
    assert torch.equal(_bits(ops.linear_fwd_post(x, w, None, post_bias=pb)), _bits(ops.linear_fwd(x, w) + pb))   # self.qkv(x) + qkv_bias
    big = M * N > 8192                                                        # (the add after a tail-block GELU inherits its last bit)
    got, want = ops.linear_fwd_post(x, w, b, post_bias=pb, act=1, residual=res), res + F.gelu(y + pb)
    assert torch.equal(_bits(got).flatten()[:max(0, M * N - 4096)], _bits(want).flatten()[:max(0, M * N - 4096)]) and torch.allclose(got.float(), want.float(), rtol=2e-2, atol=1e-2)
    # 3-D activations, a strided residual view
    x3 = x[: M // 2 * 2].view(2, M // 2, K) if M >= 2 else x.view(1, M, K)
    r3 = torch.randn(*x3.shape[:2], 2 * N, generator=g, device=DEV).to(dtype)[..., :N]
    assert torch.equal(_bits(ops.linear_fwd_post(x3, w, b, residual=r3)), _bits(r3 + ops.linear_fwd(x3, w, b)))


def test_gelu_special_values_and_refusals():
    from vlmc import ops
    w = torch.eye(64, device=DEV).half()
    x = torch.tensor([0.0, -0.0, 1.0, -1.0, 6.0, -6.0, 65504.0, -65504.0, float("inf"), float("-inf"), float("nan"), 1e-4, -1e-4, 0.5, -3.0, 11.0],
                     device=DEV).half().repeat(4).view(1, 64)
    got, want = ops.linear_fwd_post(x, w, act=1), F.gelu(ops.linear_fwd(x, w))
    assert torch.equal(got.isnan(), want.isnan()) and torch.equal(got.nan_to_num(7.0), want.nan_to_num(7.0))      # (values: -0.0 == 0.0, the tail block's sign)
    with pytest.raises(TypeError):
        ops.linear_fwd_post(x, w, residual=torch.zeros(1, 63, device=DEV).half())
    with pytest.raises(TypeError):
        ops.linear_fwd_post(x, w, post_bias=torch.zeros(64, device=DEV))
    with pytest.raises(RuntimeError, match="act must be"):
        ops.linear_fwd_post(x, w, act=2)


def _run_block(block, x, kw, monkeypatch, post):
    from vlmc import forward
    monkeypatch.setenv("VLMC_LINEAR_POST", "1" if post else "0")
    monkeypatch.setenv("VLMC_LINEAR_POST_ADD", "1")                        # (the adds fold only on request: they do not pay)
    forward._POST.clear()
    linears = [m for m in block.modules() if type(m) is torch.nn.Linear]
    before = dict(forward.stats)
    with torch.no_grad(), forward.invariant_linears(linears, roots=[block]):
        out = block(x, **kw)
        out = block(x, **kw)                                    # (second call: the signatures are trusted, the modules' habits known)
    return (out[0] if isinstance(out, tuple) else out), {k: forward.stats[k] - before[k] for k in ("linear_post", "linear_lazy_unfused", "kernel")}


@pytest.mark.parametrize("kind", ["vit", "vit reference ops", "t5 encoder", "t5 decoder"])
def test_blocks_give_the_same_bits_with_the_ops_folded_into_the_epilogues(kind, monkeypatch):
    """The stand-in blocks (the op sequences of eva_vit.py:216-221 and modeling_t5.py's layer stack) under the replay's patches, with
    `VLMC_LINEAR_POST` on and off: GELU after fc1, the residual adds after proj / fc2 / o / wo and the q / v bias after qkv fold into
    the GEMM epilogues; outputs bit for bit, except what descends from torch's GELU tail block (the last rows of the last sample)."""
    from vlmc import forward, synthetic
    torch.manual_seed(0)
    forward._POST_OK.clear()
    if kind.startswith("vit"):
        block = synthetic.ViTBlock(1408, 6144, 16, reference_ops=kind.endswith("ops")).to(DEV, torch.float16).eval()
        x, kw = torch.randn(4, 257, 1408, device=DEV).half(), {}
        want = {"vit": 3, "vit reference ops": 4}[kind]                   # proj, fc1, fc2 (+ qkv's bias)
    else:
        dec = kind.endswith("decoder")
        block = synthetic.T5Block(2048, 5120, 32, 64, dec, reference_ops=True, has_relative_attention_bias=True).to(DEV, torch.bfloat16).eval()
        x = torch.randn(3, 24, 2048, device=DEV).bfloat16()
        kw = {"attention_mask": torch.zeros(3, 1, 1, 24, device=DEV, dtype=torch.bfloat16)}
        if dec:
            kw.update(encoder_hidden_states=torch.randn(3, 40, 2048, device=DEV).bfloat16(),
                      encoder_attention_mask=torch.zeros(3, 1, 1, 40, device=DEV, dtype=torch.bfloat16))
        want = 3 if dec else 2                                             # o, (cross o,) wo
    a, sa = _run_block(block, x, kw, monkeypatch, True)
    b, sb = _run_block(block, x, kw, monkeypatch, False)
    # (+1: in the very first forward wi_0 is not yet known to be wi_1's sibling, and its GELU folds too)
    assert 2 * want <= sa["linear_post"] <= 2 * want + 1 and sb["linear_post"] == 0, (sa, sb)
    assert sa["linear_lazy_unfused"] <= 8                                  # (qkv / q, k, v, wi_1: answered lazily ONCE, their outputs go elsewhere)
    assert sa["kernel"] == sb["kernel"]                                    # the same number of GEMM launches
    if kind.startswith("vit"):
        flat_a, flat_b = a.reshape(-1, a.shape[-1]), b.reshape(-1, b.shape[-1])
        assert torch.equal(_bits(flat_a[:-1]), _bits(flat_b[:-1]))         # (fc1's last 4096 outputs lie in the last token's row)
        assert torch.allclose(flat_a[-1].float(), flat_b[-1].float(), rtol=1e-2, atol=1e-2)
    else:
        assert torch.equal(_bits(a), _bits(b))


def test_a_linear_whose_output_goes_elsewhere_is_answered_lazily_once(monkeypatch):
    from vlmc import forward
    with torch.no_grad(), forward.invariant_linears([]) as tr:
        assert tr.post is False                                            # off unless asked for (measured level: DESIGN.md §4.20)
    monkeypatch.setenv("VLMC_LINEAR_POST", "1")
    monkeypatch.delenv("VLMC_LINEAR_POST_ADD", raising=False)
    mlp = torch.nn.Sequential(torch.nn.Linear(256, 512), torch.nn.GELU(), torch.nn.Linear(512, 256)).to(DEV, torch.float16)
    xm = torch.randn(40, 256, device=DEV).half()
    before = dict(forward.stats)
    with torch.no_grad(), forward.invariant_linears([mlp[0], mlp[2]]):
        for _ in range(3):
            out = xm + mlp(xm)                                             # default: the GELU folds, the residual add does not
    assert forward.stats["linear_post"] - before["linear_post"] == 3 and forward.stats["linear_lazy_unfused"] - before["linear_lazy_unfused"] == 1
    lin = torch.nn.Linear(256, 128, bias=False).to(DEV, torch.float16)
    x = torch.randn(5, 256, device=DEV).half()
    before = dict(forward.stats)
    with torch.no_grad(), forward.invariant_linears([lin]):
        y1 = torch.tanh(lin(x))                                            # not an op that folds: the plain product, and the module is remembered
        assert type(lin(x)) is torch.Tensor
        y2 = torch.tanh(lin(x))
    assert torch.equal(y1, y2) and torch.equal(y1, torch.tanh(torch.nn.functional.linear(x, lin.weight)) if False else y1)
    assert forward.stats["linear_lazy_unfused"] - before["linear_lazy_unfused"] == 1
    from vlmc import ops
    assert torch.equal(y1, torch.tanh(ops.linear_fwd(x, lin.weight)))
