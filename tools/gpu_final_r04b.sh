# Round 4, end of the round: everything the profiles/ tables of the final build are made from (run on the GPU box from the repo root).
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
WITH_PMC=1 bash tools/collect_r04.sh > gpurun_out/collect_r04.log 2>&1
python bench.py --steps 20 --warmup 5 > gpurun_out/r04/bench_line_steps20.json 2> gpurun_out/r04/bench_line_steps20.err
python bench.py --calib-local 16 --cpu-seconds 0 --kernel-pass 0 --reference-ops 0 > gpurun_out/r04/bench_line_floor16.json 2> gpurun_out/r04/bench_line_floor16.err
python tools/tower_times.py 1 2 4 8 2>&1 | grep -v amdgpu | grep -E "^world|^   [vt]" > gpurun_out/r04/tower_times.txt
python tools/bench_sdpa.py 2>&1 | grep "|" > gpurun_out/r04/bench_sdpa.md
BENCH_GEMM_ONLY_LINEAR=1 python tools/bench_gemm.py 2>&1 | grep -E "^\||^\(" > gpurun_out/r04/bench_gemm.md
python tools/ressa_step.py --layers 32 --batch 16 --steps 3 2>&1 | grep -v amdgpu | tail -6 > gpurun_out/r04/ressa_step.txt
tail -3 gpurun_out/r04/bench_line_steps20.err
ls gpurun_out/r04
