set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_row_mean_gpu.py tests/test_fastpath_gpu.py tests/test_replay_invariance_gpu.py -x -q > gpurun_out/t_mean.log 2>&1 || { tail -40 gpurun_out/t_mean.log; exit 1; }
tail -2 gpurun_out/t_mean.log
python tools/invariance_probe.py 128 2>&1 | grep -v amdgpu | tail -3
python bench.py --steps 5 --warmup 2 --cpu-seconds 0 --kernel-pass 0 > gpurun_out/bench_r04b.json 2> gpurun_out/bench_r04b.err
python - <<PY
import json
d=json.load(open("gpurun_out/bench_r04b.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["frac"])
r=d["config"]["reference_ops"]; print({k:r[k] for k in r if k!="what"})
PY
