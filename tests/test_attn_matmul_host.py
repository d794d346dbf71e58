"""CPU: the host side of `vlmc_attn_matmul` -- which `torch.matmul` calls it takes, with which strides (the kernel reads the
operands in place through them), and the attribute patches that route a replayed block's batched matmuls to it."""
import pytest
import torch

from vlmc import forward, ops


def _emulate(a, b, plan):
    """What the kernel computes, from the plan's strides alone (fp64 gather): C[b0, b1, m, n] = sum_k A[..] B[..]."""
    batch, M, N, K, sa0, sa1, sam, sb0, sb1, sbk, sbn = plan
    nb0, nb1 = batch
    A = torch.as_strided(a, (nb0, nb1, M, K), (sa0, sa1, sam, 1)).double()
    B = torch.as_strided(b, (nb0, nb1, K, N), (sb0, sb1, sbk, sbn)).double()
    return A @ B


CASES = {
    "eva qk^T (permuted fused qkv)": lambda: (lambda qkv: (qkv[0], qkv[1].transpose(-2, -1)))(
        torch.randn(3, 9, 3 * 4 * 8).reshape(3, 9, 3, 4, 8).permute(2, 0, 3, 1, 4).half()),
    "eva attn @ v": lambda: (torch.randn(3, 4, 9, 9).half(), torch.randn(3, 9, 3, 4, 8).permute(2, 0, 3, 1, 4).half()[2]),
    "t5 scores": lambda: (torch.randn(2, 5, 32).view(2, 5, 4, 8).transpose(1, 2).bfloat16(),
                          torch.randn(2, 7, 32).view(2, 7, 4, 8).transpose(1, 2).transpose(3, 2).bfloat16()),
    "t5 attn @ v": lambda: (torch.randn(2, 4, 5, 7).bfloat16(), torch.randn(2, 7, 32).view(2, 7, 4, 8).transpose(1, 2).bfloat16()),
    "3-d bmm": lambda: (torch.randn(6, 5, 7).half(), torch.randn(6, 7, 3).half()),
    "broadcast batch": lambda: (torch.randn(1, 4, 5, 7).half(), torch.randn(3, 1, 7, 6).half()),
    "k == 1": lambda: (torch.randn(2, 3, 5, 1).half(), torch.randn(2, 3, 1, 4).half()),
    "n == 1": lambda: (torch.randn(2, 3, 5, 6).half(), torch.randn(2, 3, 6, 1).half()),
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_plan_addresses_the_operands_in_place(name):
    a, b = CASES[name]()
    plan = ops.attn_matmul_plan(a, b, _cuda_only=False)
    assert plan is not None, name
    want = torch.matmul(a.double(), b.double())
    got = _emulate(a, b, plan).reshape(want.shape)
    assert torch.equal(got, want)
    assert plan[9] == 1 or plan[10] == 1                         # B contiguous along k or along n


def test_plan_refuses_what_the_kernel_does_not_compute():
    h = torch.float16
    a, b = torch.randn(2, 3, 5, 8).to(h), torch.randn(2, 3, 8, 4).to(h)
    assert ops.attn_matmul_plan(a, b) is None                                            # CPU tensors
    assert ops.attn_matmul_plan(a.float(), b.float(), _cuda_only=False) is not None      # fp32: its own kernel since round 6 (the fp32 Q-Former)
    assert ops.attn_matmul_plan(a.double(), b.double(), _cuda_only=False) is None        # fp64 stays with the library
    assert ops.attn_matmul_plan(a, b.bfloat16(), _cuda_only=False) is None               # mixed dtypes
    assert ops.attn_matmul_plan(a[0, 0], b[0, 0], _cuda_only=False) is None              # 2-D: a linear, not attention
    assert ops.attn_matmul_plan(a, b[0], _cuda_only=False) is None                       # ranks differ
    assert ops.attn_matmul_plan(a.transpose(-1, -2), b.transpose(-1, -2)[..., :5, :].transpose(-1, -2).transpose(-1, -2),
                                _cuda_only=False) is None                               # a not contiguous along k
    assert ops.attn_matmul_plan(a, torch.randn(2, 3, 8, 8).to(h)[..., ::2], _cuda_only=False) is None   # b strided both ways
    assert ops.attn_matmul_plan(a, torch.randn(4, 3, 8, 4).to(h), _cuda_only=False) is None             # batch 2 vs 4
    assert ops.attn_matmul_plan(torch.randn(2, 3, 0, 8).to(h), b, _cuda_only=False) is None             # empty


def test_matmul_patches_are_installed_and_removed():
    base = torch._C.TensorBase
    orig = (torch.matmul, torch.bmm)
    assert "__matmul__" not in torch.Tensor.__dict__
    x, y = torch.randn(2, 3, 4), torch.randn(2, 4, 5)
    want = x @ y
    before = dict(forward.stats)
    with forward.invariant_matmuls():
        assert torch.matmul is not orig[0] and torch.bmm is not orig[1]
        assert all(n in torch.Tensor.__dict__ for n in ("__matmul__", "matmul", "bmm"))
        with forward.invariant_matmuls():                        # nests
            pass
        assert torch.matmul is not orig[0]
        with torch.no_grad():
            for got in (x @ y, torch.matmul(x, y), torch.bmm(x, y), x.matmul(y), x.bmm(y)):
                assert torch.equal(got, want)                    # CPU fp32: the original computes, the call is counted
        out = torch.empty(2, 3, 5)
        torch.matmul(x, y, out=out)                              # keyword forms go straight through
        assert torch.equal(out, want)
        xg = x.clone().requires_grad_()
        (xg @ y).sum().backward()                                # with gradients: untouched
        assert xg.grad is not None
    assert (torch.matmul, torch.bmm) == orig
    assert not any(n in torch.Tensor.__dict__ for n in ("__matmul__", "matmul", "bmm"))
    assert torch.Tensor.__matmul__ is base.__matmul__
    assert forward.stats["attn_library"] - before["attn_library"] == 5 and forward.stats["attn_kernel"] == before["attn_kernel"]


def test_switch_turns_the_patches_off(monkeypatch):
    monkeypatch.setenv("VLMC_ATTN_MATMUL", "0")
    orig = torch.matmul
    with forward.invariant_matmuls():
        assert torch.matmul is orig and "__matmul__" not in torch.Tensor.__dict__


def test_other_threads_see_the_original_functions_through_the_patches():
    """the attribute patches are process-wide, their routing is the installing thread's only -- a product
    issued by another thread while a replay holds the patches is neither counted nor routed, and a second thread entering the
    context meanwhile runs unpatched and leaves the first one's patches in place."""
    import threading
    x, y = torch.randn(2, 3, 4), torch.randn(2, 4, 5)
    want = x @ y
    seen = {}

    def other():
        before = dict(forward.stats)
        with torch.no_grad():
            seen["eq"] = torch.equal(x @ y, want) and torch.equal(torch.matmul(x, y), want) and torch.equal(x.float().mean(-1), x.mean(-1))
            with forward.invariant_matmuls():                     # a second holder: runs unpatched, must not unpatch the first
                seen["eq2"] = torch.equal(torch.bmm(x, y), want)
        seen["counted"] = forward.stats["attn_library"] - before["attn_library"] + forward.stats["attn_kernel"] - before["attn_kernel"]

    with forward.invariant_matmuls():
        patched = torch.matmul
        t = threading.Thread(target=other)
        t.start()
        t.join()
        assert torch.matmul is patched and "__matmul__" in torch.Tensor.__dict__      # still installed for this thread
        before = forward.stats["attn_library"]
        with torch.no_grad():
            x @ y
        assert forward.stats["attn_library"] == before + 1                             # and still routing it
    assert seen == {"eq": True, "eq2": True, "counted": 0}
    assert "__matmul__" not in torch.Tensor.__dict__


def test_function_mode_routes_the_same_calls_without_assigning_anything(monkeypatch):
    """`VLMC_TORCH_FUNCTION_MODE=1` (the cross-check route VERDICT r5 item 8 asked to be tried): the routing through a scoped
    TorchFunctionMode -- no attribute of torch is touched, the same calls are counted, an exception leaves nothing behind."""
    monkeypatch.setenv("VLMC_TORCH_FUNCTION_MODE", "1")
    orig = (torch.matmul, torch.bmm, torch.softmax, torch.mean)
    x, y = torch.randn(2, 3, 4), torch.randn(2, 4, 5)
    want = x @ y
    before = dict(forward.stats)
    with pytest.raises(ZeroDivisionError):
        with forward.invariant_matmuls():
            assert (torch.matmul, torch.bmm, torch.softmax, torch.mean) == orig and "__matmul__" not in torch.Tensor.__dict__
            with torch.no_grad():
                for got in (x @ y, torch.matmul(x, y), torch.bmm(x, y), x.matmul(y), x.bmm(y)):
                    assert torch.equal(got, want)
                assert torch.equal(torch.softmax(x, -1), x.softmax(-1)) and torch.equal(x.mean(-1), torch.mean(x, -1))
            with forward.invariant_matmuls():                        # nests
                with torch.no_grad():
                    x @ y
            1 / 0
    assert forward.stats["attn_library"] - before["attn_library"] == 6 and forward.stats["attn_kernel"] == before["attn_kernel"]
    with torch.no_grad():
        x @ y                                                       # outside: not routed, not counted
    assert forward.stats["attn_library"] - before["attn_library"] == 6

