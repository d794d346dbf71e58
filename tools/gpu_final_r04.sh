cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
WITH_PMC=1 bash tools/collect_r04.sh > gpurun_out/collect_r04.log 2>&1
python bench.py --steps 20 --warmup 5 > gpurun_out/r04/bench_line_steps20.json 2> gpurun_out/r04/bench_line_steps20.err
python bench.py --calib-local 16 --cpu-seconds 0 --kernel-pass 0 --reference-ops 0 > gpurun_out/r04/bench_line_floor16.json 2> gpurun_out/r04/bench_line_floor16.err
python tools/tower_times.py 1 2 4 8 2>&1 | grep -v amdgpu | grep -E "^world|^   " > gpurun_out/r04/tower_times.txt
python tools/bench_attn.py 2>&1 | grep "|" > gpurun_out/r04/bench_attn.md
python tools/chain_probe2.py 1408 2048 5120 6144 2>&1 | grep "^n =" > gpurun_out/r04/chain_probe2.txt
tail -3 gpurun_out/r04/bench_line_steps20.err
ls gpurun_out/r04
