"""`VLMC_CROSSCHECK=name[,name...]` -- ONE switch for the alternative code paths that exist to cross-check the default ones.

Every kernel or host route that was replaced by a faster one is still in the library as the check of its successor (the tests
compare them bit for bit); each has an individual `VLMC_*` variable that the code reads where it decides.  This module turns a
comma-separated list of NAMES into those variables at import time (before the C library reads its own), so that a user who
wants "the slow, obviously-correct routes" does not need to know twenty spellings:

    VLMC_CROSSCHECK=all                      every route below
    VLMC_CROSSCHECK=gemm_staged,select_multi a subset

An individual variable set by the user wins over the list."""
from __future__ import annotations

import os

ROUTES = {
    # name: ({variable: value}, what it selects)
    "gemm_staged": ({"VLMC_GEMM_RING": "0"}, "register-staged GEMM kernel instead of the LDS-DMA rings"),
    "gemm_lockstep": ({"VLMC_GEMM_PINGPONG": "0"}, "both waves of a SIMD in lockstep (no ping-pong)"),
    "gemm_half_lines": ({"VLMC_GEMM_WIDE": "0"}, "K-steps of 32 with half-line requests"),
    "gemm_one_tile_per_workgroup": ({"VLMC_GEMM_PERSIST": "0"}, "no persistent workgroups"),
    "gemm_plain_schedule": ({"VLMC_GEMM_EDGE": "0"}, "edge tiles scheduled like whole ones"),
    "gemm_square_tiles": ({"VLMC_GEMM_SMALL_TILES": "0"}, "128 x 128 tiles also where a launch makes few workgroups of them"),
    "gemm_shallow_ring": ({"VLMC_GEMM_WIDE_SLOTS": "2"}, "one double step of loads in flight also for the tiles of at most 64 x 64"),
    "linear_single": ({"VLMC_LINEAR_GROUP": "0"}, "every linear its own launch"),
    "linear_library": ({"VLMC_LINEAR_FWD": "0"}, "the GEMM library for the blocks' linears"),
    "f32_library": ({"VLMC_LINEAR_F32": "0"}, "fp32 modules (the reference's Q-Former) left to the GEMM library and torch's GELU: run per sample, never stacked"),
    "select_multi": ({"VLMC_MATRIX_FUSED": "0", "VLMC_SELECT_MIXED": "0"}, "multi-launch matrix-wide / per-width row selects"),
    "dsnot_radix": ({"VLMC_DSNOT_RADIX_ONLY": "1"}, "DSnoT list heads by the exact radix route only"),
    "dsnot_simulate": ({"VLMC_DSNOT_LISTS": "0"}, "DSnoT by the per-cycle arg-min kernel"),
    "replay_per_sample": ({"VLMC_BATCH_REPLAY": "1", "VLMC_TOWER_BATCH": "0"}, "the reference's one-sample-per-forward loop"),
    "replay_eager": ({"VLMC_GRAPH_REPLAY": "0"}, "no HIP graphs anywhere in the replay"),
    "tail_full": ({"VLMC_SKIP_DEAD_TAIL": "0"}, "statistics pass runs every block to its end"),
    "compare_at_once": ({"VLMC_LATER_EQUAL": "0"}, "remembered tower inputs compared at once"),
    "tower_eager_trace": ({"VLMC_TOWER_BATCHED_TRACE": "0"}, "the forward that traces a finished tower's wiring runs the tower for its one sample"),
    "tower_two_traces": ({"VLMC_TOWER_TRACES": "2"}, "a finished tower's wiring is used after two identical traces"),
    "tower_no_prediction": ({"VLMC_TOWER_PREDICT": "0"}, "finished towers learn every sample's block-0 arguments from an aborted forward"),
    "tower_rerun": ({"VLMC_TOWER_MEMO": "0", "VLMC_TOWER_GRAPH": "0"}, "finished towers are run again in every capture phase"),
    "sgpt_one_by_one": ({"VLMC_SGPT_CONCURRENT": "0", "VLMC_SGPT_STACK": "0"}, "one Hessian / one linear at a time"),
    "sgpt_select_multi": ({"VLMC_SGPT_SELECT_SWEEP": "0"}, "SparseGPT block threshold by the multi-launch radix select, sweep with the mask handed in"),
    "sgpt_library": ({"VLMC_SGPT_SYRK": "0", "VLMC_SGPT_DIRECT_FACTOR": "0", "VLMC_CHOL_GRAPH": "0"}, "library GEMM Hessian, the reference's three-step factor chain"),
    "sgpt_per_call": ({"VLMC_SGPT_DEFER": "0"}, "one Hessian update per hook call"),
    "sgpt_chain": ({"VLMC_SGPT_PERSISTENT": "0"}, "the factorization as a chain of launches per 128 columns instead of one persistent launch"),
    "norm_op_sequence": ({"VLMC_RMS_NORM": "0"}, "the language-model blocks' RMS norms run as the model files' seven launches"),
    "sdpa_register_staging": ({"VLMC_SDPA_DMA": "0"}, "K and V of a head staged into LDS through registers instead of LDS-DMA"),
    "sdpa_library": ({"VLMC_SDPA": "0"}, "F.scaled_dot_product_attention inside a replayed block left to torch"),
    "attn_library": ({"VLMC_ATTN_MATMUL": "0", "VLMC_ROW_MEAN": "0"}, "the blocks' batched matmuls and the norms' mean left to torch during the replay"),
    "attn_transposing_write": ({"VLMC_ATTN_TR": "0"}, "attn @ v with the operand transposed while writing LDS (no ds_read_b64_tr_b16)"),
    "lora_unfused": ({"VLMC_LORA_FUSED": "0"}, "SparseLoRA through a materialised W_eff, library GEMMs and a G = dY^T x in memory"),
    "gelu_torch": ({"VLMC_GELU": "0"}, "torch's GELU kernel (another instruction sequence in a tensor's last partial block)"),
    "attn_unfused": ({"VLMC_ATTN_FUSED": "0"}, "an attention written op by op runs op by op (vlmc_attn_matmul, torch elementwise, vlmc_softmax_rows) instead of vlmc_attn_fwd"),
    "replay_equal_shapes": ({"VLMC_PAD_RAGGED": "0", "VLMC_TOWER_PAD": "0"}, "ragged samples forwarded in groups of equal shape, never padded"),
    "towers_per_length": ({"VLMC_TOWER_PAD": "0"}, "a finished tower runs one stacked pass per token count"),
    "wiring_per_signature": ({"VLMC_TOWER_SHARE_WIRING": "0"}, "a finished tower is traced once per exact argument signature"),
    "capture_per_sample_behind_pruned": ({"VLMC_CAPTURE_MERGED_PRUNED": "0"}, "ragged batches with a pruned tower on the way (the decoder's phase) are captured one forward per sample"),
    "rows_no_slices": ({"VLMC_ROW_SLICES": "0"}, "a token slice of a padded fp32 stack (the Q-Former's query / text halves) is multiplied with its padding rows"),
    "memo_copies": ({"VLMC_MEMO_COPY": "1"}, "what a finished tower remembers of a capture phase (its block-0 arguments, its outputs) is copied instead of kept with its version"),
    "patches_as_function_mode": ({"VLMC_TORCH_FUNCTION_MODE": "1"}, "the replay's routing of matmul / softmax / gelu / mean / sdpa through a scoped TorchFunctionMode instead of attribute patches"),
    "group_stacks": ({"VLMC_GROUP_VIEWS": "0"}, "a merged forward's group is handed copies (torch.cat) of every block output of a finished tower, not views of the padded pass"),
    "host_ctypes": ({"VLMC_FAST": "0"}, "every launch through the ctypes route (no compiled host path)"),
}


def apply(environ=os.environ):
    """Translate `VLMC_CROSSCHECK` into the individual variables (those already set are left alone).  Returns the names applied."""
    spec = environ.get("VLMC_CROSSCHECK", "").strip()
    if not spec:
        return []
    names = list(ROUTES) if spec == "all" else [n.strip() for n in spec.split(",") if n.strip()]
    unknown = [n for n in names if n not in ROUTES]
    if unknown:
        raise ValueError(f"VLMC_CROSSCHECK: unknown route(s) {unknown}; known: {', '.join(ROUTES)}")
    for n in names:
        for k, v in ROUTES[n][0].items():
            environ.setdefault(k, v)
    return names
