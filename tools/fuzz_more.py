import sys, os
sys.path.insert(0, "tests"); sys.path.insert(0, "vlm-compression_amd"); sys.path.insert(0, ".")
import torch
import test_fuzz_gpu as F
import test_global_gpu as Gt
fails = 0
for seed in range(100, 160):
    for dt in (torch.bfloat16, torch.float16, torch.float32):
        try:
            F.test_row_and_matrix_select_random_cases(seed, dt)
        except AssertionError as e:
            fails += 1; print("FAIL select", seed, dt, str(e)[:200])
for seed in range(100, 140):
    try:
        F.test_batched_select_random_job_mixes(seed)
    except AssertionError as e:
        fails += 1; print("FAIL batch", seed, str(e)[:200])
for seed in range(100, 160):
    for mode in ("weight", "score", "absw_score"):
        for layout in ("global", "per_model", "layerwise"):
            try:
                Gt.test_score_select_matches_oracle(seed, mode, layout)
            except AssertionError as e:
                fails += 1; print("FAIL score", seed, mode, layout, str(e)[:200])
for seed in range(100, 180):
    try:
        F.test_mixed_width_row_select_random_job_mixes(seed)
    except AssertionError as e:
        fails += 1; print("FAIL mixed rows", seed, str(e)[:200])
print("done, failures:", fails)

# DSnoT: the list-head kernel against the per-cycle kernel (bit-identical events) and, for small rows, the CPU oracle
import numpy as np
import test_dsnot_gpu as D
rng = np.random.default_rng(2024)
dfails = 0
for case in range(60):
    m = int(rng.choice([0, 0, 4, 8]))
    n = 0 if m == 0 else int(rng.integers(1, m))
    in_f = int(rng.integers(2, 400)) * 8
    if m:
        in_f = (in_f // m) * m
    out_f = int(rng.integers(1, 40))
    dt = [torch.bfloat16, torch.float16, torch.float32][case % 3]
    try:
        D.test_list_kernel_emits_the_same_events_as_the_cycle_kernel((out_f, in_f), (n, m), dt)
    except AssertionError as e:
        dfails += 1; print("FAIL dsnot", out_f, in_f, n, m, dt, str(e)[:200])
print("dsnot done, failures:", dfails)
