#!/bin/bash
# Where does a K-step of the ping-pong GEMM go?  Diagnostic builds of the library with parts of the step removed
# (VLMC_GEMM_DBG: 1 no ring loads in the steady state, 2 no fragment reads, 4 no MFMAs, 8 no epilogue; results are garbage), timed by
# tools/bench_gemm.py.  Run on the GPU box from the repo root; the builds go to /tmp.
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}/vlm-compression_amd/csrc" || exit 1
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -Wno-unused-function -Wno-inline-asm"
for dbg in ${DBGS:-0 1 2 3 4 5 7 8 15}; do
  /opt/rocm/bin/hipcc $FLAGS -DVLMC_GEMM_DBG=$dbg -c gemm_nt.hip -o /tmp/gemm_dbg$dbg.o || exit 1
  objs=$(ls build/*.o | grep -v gemm_nt)
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $objs /tmp/gemm_dbg$dbg.o -o /tmp/libvlmc_dbg$dbg.so || exit 1
  echo "== VLMC_GEMM_DBG=$dbg"
  VLMC_LIB=/tmp/libvlmc_dbg$dbg.so python ../../tools/bench_gemm.py 2>/dev/null | grep -E "vit.fc1|t5enc.wo|vit.qkv" | cut -d'|' -f2-5
done
