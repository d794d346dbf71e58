#!/bin/bash
# Everything profiles/r03_* quotes, in one call on the GPU box (run from the repo root); outputs under gpurun_out/r03/.
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd $R && mkdir -p gpurun_out/r03
WITH_PMC=1 bash tools/collect_r03.sh > gpurun_out/r03/collect.log 2>&1
python bench.py --calib-local 16 > gpurun_out/r03/bench_default.json 2> gpurun_out/r03/bench_default.err
python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --kernel-pass 0 --reference-ops 0 > gpurun_out/r03/bench_steps20.json 2>/dev/null
for L in 32 64; do python bench.py --steps 3 --warmup 1 --cpu-seconds 0 --kernel-pass 0 --reference-ops 0 --calib-local $L > gpurun_out/r03/floor_$L.json 2>/dev/null; done
python tools/bench_gemm.py > gpurun_out/r03/bench_gemm.md 2>&1
python tools/e2e_prune.py wanda dsnot sparsegpt wanda@vicuna dsnot@vicuna > gpurun_out/r03/e2e_modes.log 2>&1
python tools/sgpt_profile.py 2:4 > gpurun_out/r03/sgpt_24.log 2>&1
python tools/ressa_step.py --layers 32 --batch 16 --steps 3 > gpurun_out/r03/ressa_16.log 2>&1
python tools/ressa_step.py --layers 32 --batch 4 --steps 3 > gpurun_out/r03/ressa_4.log 2>&1
python tools/bench_methods.py > gpurun_out/r03/bench_methods.md 2>&1
ls -la gpurun_out/r03
