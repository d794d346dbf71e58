"""ctypes access to oracle/_build/libwanda_oracle.so (the C restatement, oracle/wanda_oracle.c).
TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py."""
import ctypes
import os

import numpy as np
import torch

_LIB = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_build", "libwanda_oracle.so")
_DT = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2}
_lib = None


def available():
    return os.path.exists(_LIB)


def _load():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(_LIB)
        _lib.wo_select.restype = ctypes.c_int
    return _lib


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr())


def act_sqnorm(x: torch.Tensor) -> np.ndarray:
    x = x.reshape(-1, x.shape[-1]).contiguous()
    out = np.empty(x.shape[1], dtype=np.float32)
    _load().wo_act_sqnorm(_ptr(x), _DT[x.dtype], ctypes.c_int64(x.shape[0]), ctypes.c_int64(x.shape[1]),
                          out.ctypes.data_as(ctypes.c_void_p))
    return out


def scaler_update(s: np.ndarray, n0: int, normsq: np.ndarray, batch: int = 1):
    s = np.ascontiguousarray(s, dtype=np.float32).copy()
    normsq = np.ascontiguousarray(normsq, dtype=np.float32).reshape(-1, s.size)
    _load().wo_scaler_update(s.ctypes.data_as(ctypes.c_void_p), ctypes.c_int64(s.size), ctypes.c_int64(n0),
                             normsq.ctypes.data_as(ctypes.c_void_p), ctypes.c_int64(normsq.shape[0]), ctypes.c_int64(batch))
    return s, n0 + normsq.shape[0] * batch


def select(W: torch.Tensor, scaler_row: np.ndarray, mode: str, *, k=0, n=0, m=0, apply_zero=True):
    """Returns (mask bool [out,in] True=keep, W zeroed copy, importance_score)."""
    Wn = W.detach().clone().contiguous()
    out_f, in_f = Wn.shape
    mask = np.empty((out_f, in_f), dtype=np.uint8)
    ssum = ctypes.c_double(0.0)
    s = np.ascontiguousarray(scaler_row, dtype=np.float32)
    rc = _load().wo_select(_ptr(Wn), _DT[Wn.dtype], ctypes.c_int64(out_f), ctypes.c_int64(in_f),
                           s.ctypes.data_as(ctypes.c_void_p), {"row": 0, "matrix": 1, "nm": 2}[mode], ctypes.c_int64(k),
                           n, m, int(apply_zero), mask.ctypes.data_as(ctypes.c_void_p), ctypes.byref(ssum))
    assert rc == 0
    return mask.astype(bool), Wn, ssum.value / (out_f * in_f)
