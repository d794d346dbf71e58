"""npz <-> torch helpers for golden fixtures (bf16 has no numpy dtype: stored as
its int16 bit pattern under the key `<name>::bf16`)."""
from __future__ import annotations

import os

import numpy as np
import torch

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def pack(d: dict) -> dict:
    out = {}
    for k, v in d.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu()
            if v.dtype == torch.bfloat16:
                out[k + "::bf16"] = v.contiguous().view(torch.int16).numpy()
            else:
                out[k] = v.contiguous().numpy()
        else:
            out[k] = np.asarray(v)
    return out


def unpack(npz) -> dict:
    out = {}
    for k in npz.files:
        a = npz[k]
        if k.endswith("::bf16"):
            out[k[:-6]] = torch.from_numpy(a.copy()).view(torch.bfloat16)
        elif a.dtype.kind in "US":
            out[k] = a
        elif a.dtype.kind in "fiub" and a.ndim > 0:
            out[k] = torch.from_numpy(a.copy())
        else:
            out[k] = a.item() if a.ndim == 0 else a
    return out


def save(name: str, d: dict):
    np.savez_compressed(os.path.join(GOLDEN_DIR, name + ".npz"), **pack(d))


def load(name: str) -> dict:
    with np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False) as z:
        return unpack(z)
