#!/bin/bash
# Build variant libraries of the pass kernel (one per set of -D flags) for A/B timing.
# Usage: tools/ab_variants.sh name1 "-DFOO=1 -DBAR=2" name2 "..."   -> gpurun_out/ab/libhiprz_<name>.so
set -eo pipefail
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=$R/build/ab; mkdir -p $OUT
cd $R/rayzath_amd/csrc
FLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -I../../include -I. --offload-arch=gfx950 -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-rdc"
while [ $# -ge 2 ]; do
  name=$1; defs=$2; shift 2
  ( /opt/rocm/bin/hipcc $FLAGS $defs -c hiprz_api.hip -o $OUT/api_$name.o -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "Function Name|     VGPRs:|Occupancy" | sed 's/.*remark: *//; s/\[-Rpass.*//' | grep -A2 "rz_pass_kernelILb0ELb0ELi1ELb1" | tr '\n' ' '; echo " <- $name"
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $OUT/libhiprz_$name.so hiprz_host.o $OUT/api_$name.o ) &
done
wait
ls $OUT/*.so
