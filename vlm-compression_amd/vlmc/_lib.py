"""ctypes loader for libvlmc_hip.so (the C ABI declared in include/vlmc.h)."""
from __future__ import annotations

import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VLMC_LIB", os.path.join(_HERE, "libvlmc_hip.so"))   # VLMC_LIB: tuning builds
HEADER_PATH = os.path.join(os.path.dirname(os.path.dirname(_HERE)), "include", "vlmc.h")

VLMC_OK, VLMC_EINVAL, VLMC_EHIP, VLMC_EWORKSPACE, VLMC_ENOTPD = 0, -1, -2, -3, -4
F32, F16, BF16 = 0, 1, 2
SEL_ROW, SEL_MATRIX, SEL_NM = 0, 1, 2
SCORE_W, SCORE_S, SCORE_ABSW_S = 0, 1, 2

_c = ctypes
_p, _i, _i64, _sz = _c.c_void_p, _c.c_int, _c.c_int64, _c.c_size_t



class StatJob(_c.Structure):          # vlmc_stat_job
    _fields_ = [("x", _p), ("normsq", _p), ("in_features", _i64), ("tokens", _i64), ("row_stride", _i64),
                ("call_stride", _i64), ("normsq_stride", _i64), ("call_tokens", _p)]


class UpdateJob(_c.Structure):        # vlmc_update_job
    _fields_ = [("scaler_row", _p), ("normsq", _p), ("sqrt_out", _p), ("in_features", _i64), ("normsq_stride", _i64)]


class SelectJob(_c.Structure):        # vlmc_select_job
    _fields_ = [("W", _p), ("out_features", _i64), ("in_features", _i64), ("ldw", _i64), ("sqrt_scaler", _p), ("k", _i64),
                ("mask", _p), ("score_partials", _p), ("workspace", _p), ("workspace_bytes", _sz)]


class LinearJob(_c.Structure):        # vlmc_linear_job
    _fields_ = [("W", _p), ("bias", _p), ("Y", _p), ("N", _i64), ("ldw", _i64), ("ldy", _i64)]


class ScoreJob(_c.Structure):         # vlmc_score_job
    _fields_ = [("W", _p), ("S", _p), ("prev_keep", _p), ("keep", _p), ("numel", _i64), ("protect_k", _i64),
                ("scope", _c.c_int32), ("dtype", _c.c_int32)]


# name -> (restype, argtypes); must list every function include/vlmc.h declares
# (tests/test_abi.py parses the header and checks both directions).
SIGNATURES = {
    "vlmc_abi_version": (_i, []),
    "vlmc_last_error": (_c.c_char_p, []),
    "vlmc_set_launch_events": (None, [_p, _p]),
    "vlmc_act_sqnorm": (_i, [_p, _i, _i64, _i64, _i64, _i64, _i64, _p, _p]),
    "vlmc_wanda_scaler_update": (_i, [_p, _i64, _i64, _p, _i64, _i64, _p, _p]),
    "vlmc_act_sqnorm_batch": (_i, [_p, _i, _i, _i64, _p]),
    "vlmc_wanda_scaler_update_batch": (_i, [_p, _i, _i64, _i64, _i64, _p]),
    "vlmc_wanda_select_batch": (_i, [_p, _i, _i, _i, _i, _i, _i, _p]),
    "vlmc_wanda_select_workspace": (_sz, [_i, _i64, _i64]),
    "vlmc_wanda_select_partials": (_i64, [_i, _i64, _i64]),
    "vlmc_lora_effective_weight": (_i, [_p, _i, _i64, _i64, _i64, _p, _p, _i, _c.c_float, _p, _i, _i, _p, _i64, _p]),
    "vlmc_lora_grad_workspace": (_sz, [_i64, _i64, _i]),
    "vlmc_lora_grad": (_i, [_p, _i, _i64, _i64, _i64, _p, _p, _i, _c.c_float, _p, _i, _i, _p, _p, _p, _sz, _p]),
    "vlmc_sparse_lora_prep_bytes": (_sz, [_i64, _i64]),
    "vlmc_sparse_lora_prep": (_i, [_p, _p, _i64, _i64, _i, _i, _p, _p]),
    "vlmc_sparse_lora_fwd": (_i, [_p, _i64, _i64, _p, _i, _i64, _i64, _i64, _p, _p, _i, _c.c_float, _i, _i, _p, _p, _i64, _p]),
    "vlmc_sparse_lora_bwd_input": (_i, [_p, _i64, _i64, _p, _i, _i64, _i64, _i64, _p, _p, _i, _c.c_float, _i, _i, _p, _i64, _p]),
    "vlmc_sparse_lora_bwd_weight_workspace": (_sz, [_i64, _i64, _i64]),
    "vlmc_sparse_lora_bwd_weight": (_i, [_p, _i64, _p, _i64, _i64, _i, _i64, _i64, _p, _p, _i, _c.c_float, _i, _i, _p, _p, _p, _sz, _p]),
    "vlmc_act_moments": (_i, [_p, _i, _i64, _i64, _i64, _i64, _i64, _p, _p, _p, _p]),
    "vlmc_dsnot_stats_update": (_i, [_p, _p, _p, _i64, _i64, _i64, _p, _p, _p, _p, _i64, _i64, _p, _p]),
    "vlmc_dsnot_refine": (_i, [_p, _i, _i64, _i64, _i64, _p, _p, _p, _p, _i, _i, _i, _i, _c.c_float, _c.c_float, _i, _p, _p, _p]),
    "vlmc_dsnot_apply": (_i, [_p, _i, _i64, _i64, _i64, _p, _p, _p, _i, _i, _i, _p]),
    "vlmc_reorder_indices": (_i, [_p, _i, _i64, _i64, _i64, _p, _i64, _p]),
    "vlmc_pack_24": (_i, [_p, _i, _i64, _i64, _i64, _p, _i64, _p, _p, _p, _p]),
    "vlmc_unpack_24": (_i, [_p, _p, _i, _i64, _i64, _p, _i64, _p, _i64, _p, _p]),
    "vlmc_linear_fwd": (_i, [_p, _p, _p, _i, _i64, _i64, _i64, _i64, _i64, _p, _i64, _p]),
    "vlmc_linear_fwd_group": (_i, [_p, _p, _i, _i, _i64, _i64, _i64, _p]),
    "vlmc_linear_fwd_rows": (_i, [_p, _p, _i, _i, _i64, _i64, _i64, _p, _i64, _p]),
    "vlmc_linear_fwd_gather": (_i, [_p, _p, _p, _i, _i64, _i64, _i64, _i64, _p, _i64, _p, _p, _i64, _i64, _p]),
    "vlmc_attn_matmul": (_i, [_p, _p, _p, _i] + [_i64] * 15 + [_p]),
    "vlmc_gelu": (_i, [_p, _p, _i64, _i, _i, _p]),
    "vlmc_row_mean": (_i, [_p, _i64, _i64, _i64, _p, _p]),
    "vlmc_softmax_rows": (_i, [_p, _i, _i64, _i64, _i64, _p, _i, _i64, _p]),
    "vlmc_attn_max_keys": (_i, [_i64]),
    "vlmc_attn_fwd": (_i, [_p, _p, _p, _p, _i] + [_i64] * 5 + [_p, _p, _p, _i, _c.c_float, _p, _p, _p, _p, _p]),
    "vlmc_attn_fwd_lens": (_i, [_p, _p, _p, _p, _i] + [_i64] * 5 + [_p, _p, _p, _i, _c.c_float, _p, _p, _p, _p, _p, _p, _p]),
    "vlmc_rms_norm": (_i, [_p, _i, _i64, _i64, _i64, _p, _c.c_float, _i, _p, _i64, _p]),
    "vlmc_sdpa_max_keys": (_i, [_i64]),
    "vlmc_sdpa_fwd": (_i, [_p, _p, _p, _p, _i] + [_i64] * 17 + [_c.c_float, _i, _p]),
    "vlmc_hessian_workspace": (_sz, [_i, _i64, _i64]),
    "vlmc_hessian_accum": (_i, [_p, _i, _i64, _i64, _i64, _p, _i64, _c.c_float, _c.c_float, _p, _sz, _p]),
    "vlmc_symmetrize_lower": (_i, [_p, _i64, _i64, _p]),
    "vlmc_chol_block": (_i, [_p, _i64, _i, _p, _i64, _p, _i64, _p, _i, _p]),
    "vlmc_chol_inverse_workspace": (_sz, [_i64]),
    "vlmc_chol_inverse": (_i, [_p, _i64, _i64, _p, _i64, _p, _i64, _p, _p, _sz, _i, _p]),
    "vlmc_sparsegpt_sweep": (_i, [_p, _i64, _i64, _i64, _p, _i64, _p, _i64, _i, _i, _p, _i64, _p, _i64, _p]),
    "vlmc_sparsegpt_trailing_update": (_i, [_p, _i64, _i64, _i64, _p, _i64, _p, _i64, _i64, _p]),
    "vlmc_sparsegpt_select_workspace_bytes": (_i64, []),
    "vlmc_sparsegpt_prune_blocks": (_i, [_p, _i64, _i64, _i64, _p, _i64, _i64, _i, _i, _i, _p, _p, _p, _i64, _p, _i64, _p, _p]),
    "vlmc_sparsegpt_select_sweep": (_i, [_p, _i64, _i64, _p, _i64, _i, _p, _p, _p, _i64, _p, _i64, _p, _p]),
    "vlmc_score_select_workspace": (_sz, [_i, _i]),
    "vlmc_score_select": (_i, [_p, _i, _p, _i, _i, _i, _p, _sz, _p]),
    "vlmc_wanda_select": (_i, [_p, _i, _i64, _i64, _i64, _p, _i, _i64, _i, _i, _i, _p, _p, _p, _sz, _p]),
}

_lib = None


class VlmcError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"vlmc error {code}: {msg}")
        self.code = code


def load():
    """Load the library once; raise loudly if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build the HIP kernels first "
            "(python -c 'import __graft_entry__ as g; g.build()' or make -C vlm-compression_amd/csrc)")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype, fn.argtypes = res, args
    if lib.vlmc_abi_version() != header_abi_version():
        raise ImportError("libvlmc_hip.so ABI version does not match include/vlmc.h; rebuild")
    _lib = lib
    return lib


def header_abi_version() -> int:
    with open(HEADER_PATH) as f:
        return int(re.search(r"#define\s+VLMC_ABI_VERSION\s+(\d+)", f.read()).group(1))


def header_functions():
    """Function names declared in include/vlmc.h."""
    with open(HEADER_PATH) as f:
        src = re.sub(r"/\*.*?\*/", "", f.read(), flags=re.S)
    return sorted(set(re.findall(r"\b(vlmc_[a-z0-9_]+)\s*\(", src)))


def check(rc: int):
    if rc != VLMC_OK:
        raise VlmcError(rc, load().vlmc_last_error().decode())
