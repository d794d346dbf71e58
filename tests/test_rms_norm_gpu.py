"""vlmc_rms_norm (csrc/row_reduce.hip) and its installation during a replay (vlmc/forward.py): the RMS norm of a language-model
block -- transformers' T5LayerNorm.forward / LlamaRMSNorm.forward: to(float32), pow(2), mean(-1), + eps, rsqrt, x * r, to(dtype),
weight * h -- in one launch, bit for bit what the op sequence gives under the replay's patches.  A module is only ever replaced
after its own forward has been reproduced on random rows; look-alikes that compute something else keep their own forward."""
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


class T5StyleNorm(nn.Module):                  # modeling_t5.py T5LayerNorm
    def __init__(self, n, eps=1e-6):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(n))
        self.variance_epsilon = eps

    def forward(self, hidden_states):
        variance = hidden_states.to(torch.float32).pow(2).mean(-1, keepdim=True)
        hidden_states = hidden_states * torch.rsqrt(variance + self.variance_epsilon)
        if self.weight.dtype in [torch.float16, torch.bfloat16]:
            hidden_states = hidden_states.to(self.weight.dtype)
        return self.weight * hidden_states


class LlamaStyleNorm(nn.Module):               # modeling_llama.py LlamaRMSNorm
    def __init__(self, n, eps=1e-5):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(n))
        self.variance_epsilon = eps

    def forward(self, hidden_states):
        input_dtype = hidden_states.dtype
        hidden_states = hidden_states.to(torch.float32)
        variance = hidden_states.pow(2).mean(-1, keepdim=True)
        hidden_states = hidden_states * torch.rsqrt(variance + self.variance_epsilon)
        return self.weight * hidden_states.to(input_dtype)


class OnePlusWeightNorm(T5StyleNorm):          # a look-alike that scales by (1 + weight)
    def forward(self, x):
        v = x.to(torch.float32).pow(2).mean(-1, keepdim=True)
        return ((1.0 + self.weight.float()) * (x.float() * torch.rsqrt(v + self.variance_epsilon))).to(x.dtype)


class Fp32ProductNorm(T5StyleNorm):            # rounds once, after the product with the weight
    def forward(self, x):
        v = x.to(torch.float32).pow(2).mean(-1, keepdim=True)
        return (self.weight.float() * (x.float() * torch.rsqrt(v + self.variance_epsilon))).to(x.dtype)


class Block(nn.Module):
    def __init__(self, norm_cls, n, dtype):
        super().__init__()
        self.norm = norm_cls(n)
        self.lin = nn.Linear(n, n, bias=False)
        self.to(dtype)
        with torch.no_grad():
            self.norm.weight.copy_((torch.randn(n) * 0.3 + 1.0).to(dtype))

    def forward(self, x):
        return self.lin(self.norm(x))


@pytest.mark.parametrize("cls,dtype,n", [(T5StyleNorm, torch.bfloat16, 2048), (T5StyleNorm, torch.float16, 1000), (LlamaStyleNorm, torch.float16, 4096),
                                         (LlamaStyleNorm, torch.bfloat16, 136)])
def test_the_fused_norm_is_the_op_sequence_bit_for_bit_and_is_installed(cls, dtype, n, monkeypatch):
    from vlmc import forward
    monkeypatch.setenv("VLMC_LINEAR_FWD", "1")
    blk = Block(cls, n, dtype).to(DEV).eval()
    g = torch.Generator(device=DEV).manual_seed(n)
    x = (torch.randn(5, 19, n, generator=g, device=DEV) * 3).to(dtype)
    x[0, 0] = 0                                                             # a row of zeros: rsqrt(eps)
    x[1, 1] *= 200                                                          # large values
    with torch.no_grad():
        monkeypatch.setenv("VLMC_RMS_NORM", "0")
        with forward.invariant_linears([blk.lin], roots=(blk,)):
            want_norm, want = blk.norm(x), blk(x)
        monkeypatch.setenv("VLMC_RMS_NORM", "1")
        s0 = forward.stats["norm_kernel"]
        with forward.invariant_linears([blk.lin], roots=(blk,)):
            got_norm, got = blk.norm(x), blk(x)
            one = blk.norm(x[2:3, 4:5])                                     # one row alone: the same bits as inside the batch
        assert forward.stats["norm_kernel"] == s0 + 3, "the fused norm was not installed"
    assert torch.equal(got_norm, want_norm) and torch.equal(got, want)
    assert torch.equal(one, want_norm[2:3, 4:5])
    assert "forward" not in blk.norm.__dict__                                # the patch is gone


@pytest.mark.parametrize("cls", [OnePlusWeightNorm, Fp32ProductNorm])
def test_look_alikes_keep_their_own_forward(cls, monkeypatch):
    from vlmc import forward
    monkeypatch.setenv("VLMC_LINEAR_FWD", "1")
    blk = Block(cls, 512, torch.bfloat16).to(DEV).eval()
    x = torch.randn(4, 7, 512, device=DEV).to(torch.bfloat16)
    with torch.no_grad():
        want = blk.norm(x)
        s0 = forward.stats["norm_kernel"]
        with forward.invariant_linears([blk.lin], roots=(blk,)):
            got = blk.norm(x)
        assert forward.stats["norm_kernel"] == s0
    assert torch.equal(got, want)


def test_a_look_alike_with_an_all_ones_weight_is_not_whitelisted(monkeypatch):
    """the verdict is cached per class -- a fresh norm (weight all ones) of a class that rounds AFTER the product
    with the weight compares equal to the fused kernel with its own weight; the self-check uses a random weight instead, so the
    class is refused and a sibling with a trained weight keeps its own forward."""
    from vlmc import forward
    monkeypatch.setenv("VLMC_LINEAR_FWD", "1")
    forward._NORM_OK.clear()
    fresh = Block(Fp32ProductNorm, 640, torch.bfloat16).to(DEV).eval()
    with torch.no_grad():
        fresh.norm.weight.fill_(1.0)
    trained = Block(Fp32ProductNorm, 640, torch.bfloat16).to(DEV).eval()
    x = torch.randn(4, 7, 640, device=DEV).to(torch.bfloat16)
    with torch.no_grad():
        s0 = forward.stats["norm_kernel"]
        with forward.invariant_linears([fresh.lin], roots=(fresh,)):
            fresh.norm(x)
        want = trained.norm(x)
        with forward.invariant_linears([trained.lin], roots=(trained,)):
            got = trained.norm(x)
    assert forward.stats["norm_kernel"] == s0 and torch.equal(got, want)
    assert torch.equal(fresh.norm.weight, torch.ones_like(fresh.norm.weight))      # the module got its own weight back


def test_whole_prune_is_bit_identical_with_and_without_the_fused_norm(monkeypatch):
    """The synthetic InstructBLIP's T5 blocks carry T5LayerNorm's op sequence: a small Wanda prune with the norm fused and with
    the seven launches gives the same masks and weights."""
    from vlmc import forward, synthetic
    dev = torch.device(DEV)
    outs = []
    for flag in ("0", "1"):
        monkeypatch.setenv("VLMC_RMS_NORM", flag)
        torch.manual_seed(0)
        model = synthetic.InstructBlipT5(vit_dim=128, vit_hidden=256, vit_heads=4, vit_depth=2, d_model=128, d_ff=256, heads=4, d_kv=32,
                                         enc_depth=2, dec_depth=2, vocab=512, query_tokens=8).to(dev).eval()
        synthetic.randomize_(model, 0)
        batches = synthetic.calibration_batches(6, dev, vit_tokens=17, vit_dim=128, text_len=9, out_len=5, vocab=512)
        s0 = forward.stats["norm_kernel"]
        _, pruned, info = synthetic.time_prune(dev, model=model, batches=batches, n_samples=6, t5_prune_spec="2-0.5-1.0-1.0",
                                               vit_prune_spec="2-0.5-1.0-1.0")
        assert (forward.stats["norm_kernel"] > s0) == (flag == "1")
        assert 0.45 < info["pruned_fraction"] < 0.55
        outs.append({k: v.clone() for k, v in pruned.state_dict().items()})
    assert outs[0].keys() == outs[1].keys()
    for k in outs[0]:
        assert torch.equal(outs[0][k], outs[1][k]), k


def test_rms_norm_entry_point_on_odd_shapes_and_strided_rows():
    """Widths that are not multiples of 4 or 256, one row, 8192 columns, rows that are slices of a wider buffer: against the op
    sequence written with torch ops (the mean through vlmc_row_mean, as during a replay)."""
    from vlmc import forward, ops

    def ref(x, w, eps):
        with forward.invariant_matmuls():
            v = x.to(torch.float32).pow(2).mean(-1, keepdim=True)
            return w * (x * torch.rsqrt(v + eps)).to(w.dtype)
    g = torch.Generator(device=DEV).manual_seed(9)
    for shape, dt, eps in [((7, 1001), torch.float16, 1e-6), ((1, 8192), torch.bfloat16, 1e-5), ((3, 5, 130), torch.float16, 1e-6),
                           ((2, 2050), torch.bfloat16, 1e-6), ((4, 6), torch.float16, 1e-3)]:
        x = (torch.randn(*shape, generator=g, device=DEV) * 2).to(dt)
        w = (torch.randn(shape[-1], generator=g, device=DEV) * 0.3 + 1).to(dt)
        with torch.no_grad():
            assert torch.equal(ops.rms_norm(x, w, eps, 0), ref(x, w, eps)), (shape, dt)
    wide = (torch.randn(9, 4096, generator=g, device=DEV)).to(torch.float16)
    w = torch.ones(2048, device=DEV, dtype=torch.float16)
    with torch.no_grad():
        assert torch.equal(ops.rms_norm(wide[:, 1024:3072], w, 1e-6, 0), ref(wide[:, 1024:3072].contiguous(), w, 1e-6))
