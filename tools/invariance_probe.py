"""Which op of a reference-op block gives a sample other bits in a group than alone?  Runs the blocks of
vlmc/synthetic.py (reference_ops=True) at model width on n samples one by one and stacked, with the replay's patches active,
and compares every intermediate tensor bit for bit.  `python tools/invariance_probe.py [n=128]`"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vlm-compression_amd"))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402
from vlmc import forward, synthetic as S  # noqa: E402
from lavis.compression.pruners import calibration as cal  # noqa: E402

dev = "cuda:0"
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
torch.manual_seed(0)
g = torch.Generator(device=dev).manual_seed(1)

trace = None


def note(name, t):
    if trace is not None:
        trace.append((name, t.detach().clone()))
    return t


# ---- traced re-statements of the stand-in's forward (same ops in the same order) -----------------------------------------
def rms(mod, x, tag):
    v = note(tag + ".mean", x.to(torch.float32).pow(2).mean(-1, keepdim=True))
    h = x * torch.rsqrt(v + 1e-6)
    h = h.to(mod.weight.dtype)
    return note(tag + ".out", mod.weight * h)


def t5attn(mod, x, tag, mask=None, kv=None):
    B, T, _ = x.shape
    src = x if kv is None else kv

    def shape(t):
        return t.view(B, -1, mod.heads, mod.d_kv).transpose(1, 2)
    q, k, v = shape(note(tag + ".q", mod.q(x))), shape(note(tag + ".k", mod.k(src))), shape(note(tag + ".v", mod.v(src)))
    scores = note(tag + ".scores", torch.matmul(q, k.transpose(3, 2)))
    pb = torch.zeros((1, mod.heads, T, k.shape[2]), device=scores.device, dtype=scores.dtype) if not mod.has_relative_attention_bias \
        else mod.compute_bias(T, k.shape[2], scores.device)
    if mask is not None:
        pb = pb + mask
    scores += pb
    note(tag + ".biased", scores)
    attn = note(tag + ".softmax", F.softmax(scores.float(), dim=-1).type_as(scores))
    y = note(tag + ".ctx", torch.matmul(attn, v)).transpose(1, 2).contiguous().view(B, -1, mod.heads * mod.d_kv)
    return note(tag + ".o", mod.o(y))


def t5block(blk, x, enc=None, mask=None, emask=None):
    sa = blk.layer[0]
    x = note("res1", x + t5attn(sa.SelfAttention, rms(sa.layer_norm, x, "ln1"), "self", mask=mask))
    if blk.is_decoder:
        ca = blk.layer[1]
        x = note("res2", x + t5attn(ca.EncDecAttention, rms(ca.layer_norm, x, "ln2"), "cross", mask=emask, kv=enc))
    ff = blk.layer[-1]
    h = rms(ff.layer_norm, x, "ln3")
    d = ff.DenseReluDense
    a, b = note("wi0", d.wi_0(h)), note("wi1", d.wi_1(h))
    m = note("gated", F.gelu(a) * b)
    return note("out", x + note("wo", d.wo(m)))


def vitblock(blk, x):
    at = blk.attn
    h = note("norm1", blk.norm1(x))
    B, N, C = h.shape
    qkv_bias = torch.cat((at.q_bias, torch.zeros_like(at.v_bias, requires_grad=False), at.v_bias))
    qkv = note("qkv", at.qkv(h) + qkv_bias).reshape(B, N, 3, at.heads, -1).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    q = note("qscaled", q * at.scale)
    attn = note("scores", q @ k.transpose(-2, -1))
    attn = note("softmax", attn.softmax(dim=-1))
    y = note("ctx", attn @ v).transpose(1, 2).reshape(B, N, -1)
    x = note("res1", x + note("proj", at.proj(y)))
    h2 = note("norm2", blk.norm2(x))
    f1 = note("fc1", blk.mlp.fc1(h2))
    return note("out", x + note("fc2", blk.mlp.fc2(note("gelu", F.gelu(f1)))))


def compare(name, fn, xs, kws):
    """fn(x, **kw) traced; per-sample vs stacked"""
    global trace
    per = []
    for x, kw in zip(xs, kws):
        trace = []
        fn(x, **kw)
        per.append(trace)
    trace = []
    fn(torch.cat(xs), **{k: torch.cat([kw[k] for kw in kws]) for k in kws[0]})
    stacked = trace
    trace = None
    b0 = xs[0].shape[0]
    bad = []
    for i, (nm, t) in enumerate(stacked):
        want = torch.cat([p[i][1] for p in per])
        if t.shape != want.shape:
            want = want.reshape(t.shape)
        if not torch.equal(t, want):
            diff = (t != want)
            bad.append((nm, int(diff.sum()), t.numel()))
    print(f"{name}: {len(xs)} samples of batch {b0}: " + ("every intermediate tensor identical" if not bad else
          "FIRST DIFFERENCE at " + ", ".join(f"{nm} ({c} of {tot})" for nm, c, tot in bad[:4])), flush=True)


with torch.no_grad():
    vit = S.ViTBlock(1408, 6144, 16, reference_ops=True).to(dev).half().eval()
    enc0 = S.T5Block(2048, 5120, 32, 64, False, True, True).to(dev).bfloat16().eval()
    enc = S.T5Block(2048, 5120, 32, 64, False, True, False).to(dev).bfloat16().eval()
    dec = S.T5Block(2048, 5120, 32, 64, True, True, False).to(dev).bfloat16().eval()
    for blk in (vit, enc0, enc, dec):
        S.randomize_(blk, seed=3)
    for blk, label in ((vit, "vit"), (enc0, "t5 encoder block 0"), (enc, "t5 encoder"), (dec, "t5 decoder")):
        subset = cal.find_layers(blk)
        with forward.invariant_linears(subset.values()):
            if label == "vit":
                xs = [(torch.randn(1, 257, 1408, generator=g, device=dev) * 0.5).half() for _ in range(n)]
                compare(label, lambda x: vitblock(vit, x), xs, [{} for _ in xs])
            else:
                for T in ((64, 40, 160) if "encoder" in label else (16, 4)):
                    xs = [(torch.randn(1, T, 2048, generator=g, device=dev) * 0.5).bfloat16() for _ in range(n)]
                    if "decoder" in label:
                        kws = [dict(enc=(torch.randn(1, 72, 2048, generator=g, device=dev) * 0.5).bfloat16()) for _ in xs]
                    else:
                        kws = [{} for _ in xs]
                    compare(f"{label} T={T}", lambda x, **kw: t5block(blk, x, **kw), xs, kws)
