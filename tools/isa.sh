#!/bin/bash
# gfx950 assembly of one translation unit: usage: [UNIT=hiprz_launch_trace] tools/isa.sh out.s [extra hipcc flags...]
out=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
/opt/rocm/bin/hipcc -O3 -std=c++17 -ffp-contract=off -fno-fast-math -I$root/include -I$root/rayzath_amd/csrc --offload-arch=gfx950 \
  -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-rdc --cuda-device-only -S "$@" $root/rayzath_amd/csrc/${UNIT:-hiprz_launch_trace}.hip -o $out
