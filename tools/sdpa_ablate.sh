#!/bin/bash
# Where the time of vlmc_sdpa_fwd goes: diagnostic builds of csrc/sdpa.hip with parts switched off (results are garbage), timed on
# the ViT-g shape.  Run on the GPU box from the repo root.
set -e
cd /root/repo/vlm-compression_amd/csrc
for dbg in 0 1 2 4 6 7 8 16 18 31; do
  mkdir -p /tmp/sdpa_dbg$dbg
  for f in *.hip api.cpp; do
    o=/tmp/sdpa_dbg$dbg/${f%.*}.o
    if [ "$f" = "sdpa.hip" ]; then
      /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wno-unused-function -Wno-inline-asm -DVLMC_SDPA_DBG=$dbg -c $f -o $o
    else
      cp build/${f%.*}.o $o
    fi
  done
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 /tmp/sdpa_dbg$dbg/*.o -o /tmp/libvlmc_dbg$dbg.so
  echo "== VLMC_SDPA_DBG=$dbg"
  VLMC_LIB=/tmp/libvlmc_dbg$dbg.so python3 - <<'PY'
import os, sys, statistics
sys.path.insert(0, "/root/repo/vlm-compression_amd")
import torch
from vlmc import ops
dev = "cuda:0"
for B in (128, 1):
    H, T, d = 16, 257, 88
    qkv = (torch.randn(B, T, 3 * H * d, device=dev) * 0.5).to(torch.float16)
    q, k, v = (t.reshape(B, T, H, d).transpose(1, 2) for t in qkv.reshape(B, T, 3, H * d).unbind(2))
    ts = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10): ops.sdpa(q, k, v)
        b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / 10)
    print(f"   B={B}: {statistics.median(ts) * 1e3:.1f} us")
PY
done
