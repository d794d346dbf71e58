"""Every `VLMC_*` environment variable this build reads, in ONE place ("57 switches read in Python and 22 getenv in csrc").

Three kinds, and only the first is an interface:

PUBLIC (ten): what a user of the drop-in pruners may want to set.  Everything works with none of them set.

CROSSCHECK: the alternative routes that exist so that the tests can hold every kernel / engine route against the one it replaced, on
the GPU, inside whole prunes (`VLMC_CROSSCHECK=name[,name..]` names them: vlmc/crosscheck.py).  Not a second backend: nothing in the
product selects them, and a route that was measured and lost is deleted, not parked here (round 6 removed the GEMM epilogue folds,
the padding length buckets and the SparseGPT look-ahead).

INTERNAL: measurement aids of bench.py / tools/, tuning overrides of single kernels, test hooks.

`tests/test_abi.py::test_every_environment_switch_is_registered` greps the tree: a `VLMC_*` variable that is read anywhere but not
listed here fails the build's CPU tests."""
from __future__ import annotations

PUBLIC = {
    "VLMC_LIB": "path of another build of libvlmc_hip.so (default: the in-tree one next to this package)",
    "VLMC_FAST": "0: every launch through ctypes instead of the compiled host path vlmc/_fast (same C ABI, same kernels)",
    "VLMC_BATCH_REPLAY": "calibration samples per block forward of the replay (default 128 = all; 1 = the reference's per-sample loop)",
    "VLMC_REPLAY_TOKENS": "activation rows per block forward at most (default 65536)",
    "VLMC_SHARD_CALIB": "0: under torch.distributed every rank is a replica that sees all samples (the reference's behaviour); default: samples sharded, statistics all-gathered",
    "VLMC_RELEASE_MEMORY": "1: free cached device memory after every tower (the reference's `torch.cuda.empty_cache()` habit; costs time)",
    "VLMC_CROSSCHECK": "comma-separated names of cross-check routes (vlmc/crosscheck.py), or `all`",
    "VLMC_STRICT": "1: an exception inside an optional fast path (the stacked capture) is raised instead of falling back to the reference's route",
    "VLMC_VERBOSE": "1: the pruners print what the reference prints per linear",
    "VLMC_PHASE_TIMERS": "1: synchronising per-phase timers (vlmc/phases.py; bench.py's sub-totals)",
}

# every variable vlmc/crosscheck.py's ROUTES set (kept in step by the test)
CROSSCHECK = {
    "VLMC_GEMM_RING", "VLMC_GEMM_PINGPONG", "VLMC_GEMM_WIDE", "VLMC_GEMM_PERSIST", "VLMC_GEMM_EDGE", "VLMC_GEMM_SMALL_TILES", "VLMC_GEMM_WIDE_SLOTS",
    "VLMC_LINEAR_GROUP", "VLMC_LINEAR_FWD", "VLMC_LINEAR_F32", "VLMC_MATRIX_FUSED", "VLMC_SELECT_MIXED", "VLMC_DSNOT_RADIX_ONLY", "VLMC_DSNOT_LISTS",
    "VLMC_TOWER_BATCH", "VLMC_GRAPH_REPLAY", "VLMC_SKIP_DEAD_TAIL", "VLMC_LATER_EQUAL", "VLMC_TOWER_BATCHED_TRACE", "VLMC_TOWER_TRACES",
    "VLMC_TOWER_PREDICT", "VLMC_TOWER_MEMO", "VLMC_TOWER_GRAPH", "VLMC_SGPT_CONCURRENT", "VLMC_SGPT_STACK", "VLMC_SGPT_SELECT_SWEEP", "VLMC_SGPT_SYRK",
    "VLMC_SGPT_DIRECT_FACTOR", "VLMC_CHOL_GRAPH", "VLMC_SGPT_DEFER", "VLMC_SGPT_PERSISTENT", "VLMC_RMS_NORM", "VLMC_SDPA_DMA", "VLMC_SDPA",
    "VLMC_ATTN_MATMUL", "VLMC_ROW_MEAN", "VLMC_ATTN_TR", "VLMC_LORA_FUSED", "VLMC_GELU", "VLMC_ATTN_FUSED", "VLMC_PAD_RAGGED", "VLMC_TOWER_PAD",
    "VLMC_TOWER_SHARE_WIRING", "VLMC_SGPT_BLOCK_LOOP", "VLMC_ROW_MAP", "VLMC_CAPTURE_MERGED", "VLMC_CAPTURE_MERGED_RAGGED", "VLMC_SOFTMAX",
    "VLMC_CAPTURE_MERGED_PRUNED", "VLMC_ROW_SLICES", "VLMC_MEMO_COPY", "VLMC_TORCH_FUNCTION_MODE", "VLMC_GROUP_VIEWS",
}

INTERNAL = {
    # measurement aids
    "VLMC_SIMULATE_WORLD": "bench.py --calib-local: one process rehearses rank 0 of W (warns: the masks are not a real prune's)",
    "VLMC_BENCH_ONE_DEVICE": "bench.py: rehearse the N > 1 code path with every rank on cuda:0 over gloo",
    "VLMC_DEBUG_TOWERS": "print what every finished tower decides per calibration forward",
    "VLMC_GC_FREEZE": "0: do not freeze the garbage collector's old generation for the duration of a prune",
    "VLMC_CAPTURE_STREAMS": "side streams of the per-sample capture route (default 4)",
    "VLMC_CAPTURE_MERGED_MIN": "fewest calibration batches per rank for the stacked capture (default 24)",
    # tuning overrides of single kernels / routes (A/B runs of tools/)
    "VLMC_GEMM_BIG_TILES": "tiles needed to pick 256 x 256 (default 200)", "VLMC_GEMM_SHAPE": "force one small tile shape",
    "VLMC_DSNOT_NW": "waves per workgroup of the DSnoT list kernel", "VLMC_SQNORM_VEC": "vector width of act_sqnorm",
    "VLMC_F32_TILE": "128 / 64 / 32: one tile size for every fp32 GEMM launch (same bits)",
    "VLMC_LORA_BQ": "activation rows per generated W_eff tile", "VLMC_LORA_DBG": "lora_gemm ablation bits (results invalid)",
    "VLMC_LORA_RECOMPUTE": "1: regenerate W_eff in the unfused backward instead of keeping it", "VLMC_ATTN_DMA": "0: K / V of vlmc_attn_fwd staged through registers",
    "VLMC_SGPT_SWEEP_STREAMS": "streams for the sweeps of a block's independent linears (default 4)",
    "VLMC_SGPT_SHARD_LAYERS": "0: under torch.distributed every rank prunes every linear", "VLMC_SGPT_SYRK_F32": "1: fp32 activations take the nine-plane SYRK",
    "VLMC_SGPT_SORT_THRESHOLD": "1: the block threshold by torch.sort (the reference's op)", "VLMC_SGPT_PERSISTENT_WGS": "workgroups of the persistent factorization",
    "VLMC_SGPT_INVERSE_FORK": "1: the inverse's rows on a second stream inside the graph (measured slower)", "VLMC_CHOL_PANEL_GEMM": "0: library triangular solve for the panel",
    # test hooks
    "VLMC_SGPT_SELECT_FORCE_FAIL": "tests: make the one-launch sweep's grid barrier fail at a given level",
}


def all_names():
    return set(PUBLIC) | set(CROSSCHECK) | set(INTERNAL)
