"""TEST INFRASTRUCTURE (oracle): the packed 2:4 layout of include/vlmc.h (vlmc_pack_24 / vlmc_unpack_24), restated in numpy.

The reference has no packed format (SURVEY.md §8(f)3 lists it as optional; nothing under /root/reference reads one), so there is
no reference output to pin this against: the layout is this build's own, defined in include/vlmc.h, and what the tests hold the
kernels to is (a) this restatement, (b) the round trip unpack(pack(W, mask)) == W * mask, mask recovered, (c) the masks the
reference's n:m rule produces (tests/golden/wanda_unit.npz n:m cases) being packable."""
import numpy as np


def pack(w_bits: np.ndarray, keep: np.ndarray):
    """w_bits [out, in] uint16 (the weights' bit patterns), keep [out, in] bool -> (values [out, in/2] uint16, meta [out, in/8] uint8,
    number of groups that do not keep exactly two)"""
    out_f, in_f = w_bits.shape
    assert in_f % 8 == 0
    g = keep.reshape(out_f, in_f // 4, 4)
    bad = int((g.sum(-1) != 2).sum())
    values = np.zeros((out_f, in_f // 2), dtype=np.uint16)
    codes = np.zeros((out_f, in_f // 4), dtype=np.uint8)
    wg = w_bits.reshape(out_f, in_f // 4, 4)
    for r in range(out_f):
        for c in range(in_f // 4):
            idx = [j for j in range(4) if g[r, c, j]][:2]
            idx += [0] * (2 - len(idx))
            values[r, 2 * c], values[r, 2 * c + 1] = wg[r, c, idx[0]], wg[r, c, idx[1]]
            codes[r, c] = idx[0] | idx[1] << 2
    meta = (codes[:, 0::2] | (codes[:, 1::2] << 4)).astype(np.uint8)
    return values, meta, bad


def unpack(values: np.ndarray, meta: np.ndarray):
    """-> (w_bits [out, in] uint16, keep [out, in] bool)"""
    out_f, half = values.shape
    in_f = half * 2
    w = np.zeros((out_f, in_f), dtype=np.uint16)
    keep = np.zeros((out_f, in_f), dtype=bool)
    for r in range(out_f):
        for c in range(in_f // 4):
            code = (int(meta[r, c // 2]) >> (4 * (c % 2))) & 15
            i0, i1 = code & 3, code >> 2
            w[r, 4 * c + i0], w[r, 4 * c + i1] = values[r, 2 * c], values[r, 2 * c + 1]
            keep[r, 4 * c + i0] = keep[r, 4 * c + i1] = True
    return w, keep
