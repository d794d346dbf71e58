cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out/r04c
python bench.py --steps 20 --warmup 5 > gpurun_out/r04c/bench_line_steps20.json 2> gpurun_out/r04c/bench_line_steps20.err
python bench.py --calib-local 16 --cpu-seconds 0 --kernel-pass 0 --reference-ops 0 > gpurun_out/r04c/bench_line_floor16.json 2> gpurun_out/r04c/bench_line_floor16.err
python tools/tower_times.py 1 2 4 8 2>&1 | grep -v amdgpu | grep -E "^world|^   [vt]" > gpurun_out/r04c/tower_times.txt
tail -2 gpurun_out/r04c/bench_line_steps20.err
