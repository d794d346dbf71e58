#!/bin/bash
# Round-5 final check on the GPU box (run from the repo root): the driver's three steps -- GPU tests, smoke, default bench -- into gpurun_out/r05/
set -o pipefail
mkdir -p gpurun_out/r05
timeout -k 10 700 python -m pytest tests -q -m gpu > gpurun_out/r05/final_tests.log 2>&1; echo "tests rc=$?" | tee -a gpurun_out/r05/final_tests.log
tail -n 3 gpurun_out/r05/final_tests.log
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r05/final_smoke.log 2>&1; echo "smoke rc=$?"; tail -n 2 gpurun_out/r05/final_smoke.log
timeout -k 10 600 python bench.py > gpurun_out/r05/final_bench.json 2> gpurun_out/r05/final_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r05/final_bench.json").read().strip().splitlines()[-1])
print(d["value"], d["unit"], d["ms_per_step"], "ms; roofline", d["roofline"]["frac"], "; reference_ops", d["config"]["reference_ops"]["seconds_per_prune"],
      d["config"]["reference_ops"]["mask_agreement_grouped_vs_per_sample"], "; workload:", d["config"]["workload"][:160])
PY
