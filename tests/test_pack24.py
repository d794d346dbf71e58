"""The packed 2:4 layout (include/vlmc.h: vlmc_pack_24 / vlmc_unpack_24; SURVEY.md §8(f)3 "optional packed 2:4 format").  The
reference has no such format, so the pins are: the numpy restatement (oracle/pack24.py) on a hand-written example and on the masks
the REFERENCE's n:m rule produced (tests/golden/wanda_unit.npz), the kernels against the restatement bit for bit, and the round
trip at model size."""
import numpy as np
import pytest
import torch

import golden_io
from oracle import pack24


def test_layout_on_a_hand_written_example():
    w = np.arange(1, 17, dtype=np.uint16).reshape(1, 16)
    keep = np.array([[1, 0, 0, 1,  0, 1, 1, 0,  0, 0, 1, 1,  1, 1, 0, 0]], dtype=bool)
    values, meta, bad = pack24.pack(w, keep)
    assert bad == 0
    assert values.tolist() == [[1, 4, 6, 7, 11, 12, 13, 14]]
    # codes i0 | i1 << 2: (0,3) -> 12, (1,2) -> 9, (2,3) -> 14, (0,1) -> 4; two groups per byte, the first in the low nibble
    assert meta.tolist() == [[12 | 9 << 4, 14 | 4 << 4]]
    w2, k2 = pack24.unpack(values, meta)
    assert np.array_equal(k2, keep) and np.array_equal(w2, np.where(keep, w, 0))
    assert pack24.pack(w, np.ones_like(keep))[2] == 4                     # not a 2:4 mask: every group counted


def test_masks_of_the_references_nm_rule_are_packable():
    g = golden_io.load("wanda_unit")
    names = sorted({k.rsplit("/", 1)[0] for k in g if k.endswith("/mask")})
    seen = 0
    for name in names:
        keep = g[f"{name}/mask"].numpy().astype(bool)
        if keep.ndim != 2 or keep.shape[1] % 8 or not (keep.reshape(keep.shape[0], -1, 4).sum(-1) == 2).all():
            continue
        bits = g[f"{name}/Wn"].to(torch.bfloat16).view(torch.int16).numpy().view(np.uint16)      # the reference's pruned weight
        values, meta, bad = pack24.pack(bits, keep)
        w2, k2 = pack24.unpack(values, meta)
        assert bad == 0 and np.array_equal(k2, keep) and np.array_equal(w2, np.where(keep, bits, 0)) and np.array_equal(w2, bits)
        seen += 1
    assert seen >= 1, "wanda_unit.npz holds the reference's 2:4 prunes (g4/*_2_4_*)"


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_kernels_equal_the_restatement_and_round_trip(dtype):
    from vlmc import ops
    dev = "cuda:0"
    g = torch.Generator(device=dev).manual_seed(0)
    for out_f, in_f in ((5, 8), (33, 64), (128, 1408), (64, 2048)):
        W = torch.randn(out_f, in_f, generator=g, device=dev).to(dtype)
        W[0, :4] = 0                                                       # kept zeros stay kept: positions come from the mask
        sq = torch.rand(in_f, generator=g, device=dev) + 0.5
        Wp = W.clone()
        keep, _ = ops.wanda_select(Wp, sq, "nm", n=2, m=4)
        values, meta = ops.pack_24(Wp, keep)
        ov, om, bad = pack24.pack(Wp.cpu().view(torch.int16).numpy().view(np.uint16), keep.cpu().numpy())
        assert bad == 0
        assert np.array_equal(values.cpu().view(torch.int16).numpy().view(np.uint16), ov) and np.array_equal(meta.cpu().numpy(), om)
        w2, k2 = ops.unpack_24(values, meta)
        assert torch.equal(k2, keep) and torch.equal(w2.view(torch.int16), Wp.view(torch.int16))
        assert values.numel() * 2 + meta.numel() == W.numel() * 2 * 9 // 16
    with pytest.raises(ValueError, match="not a 2:4 mask"):
        ops.pack_24(W, torch.ones_like(W, dtype=torch.bool))
    with pytest.raises(TypeError):
        ops.pack_24(W.float(), keep)
    with pytest.raises(ValueError, match="never writes"):
        ops.unpack_24(values, torch.zeros_like(meta))                      # code 0: i0 == i1


@pytest.mark.gpu
def test_state_dict_of_a_pruned_model_packs_and_unpacks():
    """a 2:4 Wanda prune of the toy T5 tower -> state dict -> packed -> torch.save / load -> unpacked: weights and masks bit for bit,
    everything that is not a 2:4 pair untouched"""
    import io
    from vlmc import formats, ops
    dev = "cuda:0"
    torch.manual_seed(0)
    lin = {f"blk.{i}.{n}": torch.nn.Linear(64, 96 if n == "a" else 64, bias=False).to(dev, torch.bfloat16) for i in range(2) for n in "ab"}
    state = {}
    for name, m in lin.items():
        keep, _ = ops.wanda_select(m.weight.data, torch.ones(64, device=dev), "nm", n=2, m=4)
        state[name + ".weight"], state[name + ".mask"] = m.weight.data, keep
    state["blk.0.norm.weight"] = torch.ones(64, device=dev, dtype=torch.bfloat16)
    state["dense.weight"] = torch.randn(8, 64, device=dev).bfloat16()                  # no mask: stays
    state["odd.weight"], state["odd.mask"] = torch.randn(8, 64, device=dev).bfloat16(), torch.rand(8, 64, device=dev) > 0.5   # unstructured
    packed = formats.pack_state_dict_24(state)
    assert sum(k.endswith("weight_packed24") for k in packed) == 4 and "odd.mask" in packed and "blk.0.a.mask" not in packed
    buf = io.BytesIO()
    torch.save(packed, buf)
    dense_bytes = sum(v.numel() * v.element_size() for k, v in state.items() if k.startswith("blk.") and "norm" not in k)
    packed_bytes = sum(v.numel() * v.element_size() for k, v in packed.items() if "24" in k)
    assert packed_bytes * 8 == dense_bytes * 3                              # 9 / 16 of the weights, 3 / 8 of weight + bool mask
    buf.seek(0)
    back = formats.unpack_state_dict_24(torch.load(buf), device=dev)
    assert back.keys() == state.keys()
    for k in state:
        assert torch.equal(back[k], state[k]), k
