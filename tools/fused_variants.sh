#!/bin/bash
# Diagnostic builds of the fused matrix-wide select (run HERE, hipcc cross-compiles): phase stamps and store ablations.
#   bash tools/fused_variants.sh   ->  vlm-compression_amd/vlmc/libvlmc_diag_{stamps,exp1,exp2,exp3,exp4}.so
set -e
cd "$(dirname "$0")/../vlm-compression_amd/csrc"
make -s
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -Wno-unused-function -Wno-inline-asm"
mkdir -p build_diag
others=$(ls build/*.o | grep -v wanda_select)
for v in "stamps:-DVLMC_FUSED_STAMPS" "scan:-DVLMC_FUSED_STAMPS -DVLMC_FUSED_STAMP_SCAN" "exp1:-DVLMC_FUSED_EXP=1" "exp2:-DVLMC_FUSED_EXP=2" "exp3:-DVLMC_FUSED_EXP=3" "exp4:-DVLMC_FUSED_EXP=4"; do
  name=${v%%:*}; flag=${v#*:}
  /opt/rocm/bin/hipcc $F $flag -c wanda_select.hip -o build_diag/wanda_select_$name.o &
done
wait
for name in stamps scan exp1 exp2 exp3 exp4; do
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $others build_diag/wanda_select_$name.o -o ../vlmc/libvlmc_diag_$name.so
done
ls -la ../vlmc/libvlmc_diag_*.so
