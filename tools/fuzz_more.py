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
print("done, failures:", fails)
