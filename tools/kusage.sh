#!/bin/bash
# VGPRs / scratch / occupancy of every kernel whose mangled name matches $1 (default: all), from the compiler's remarks.
# usage: [UNIT=hiprz_launch_batch|hiprz_launch_trace|hiprz_launch_shade|hiprz_sort] tools/kusage.sh [regex] [extra hipcc flags...]
pat=${1:-.}; shift
root=$(cd "$(dirname "$0")/.." && pwd)
out=$(mktemp -d)
/opt/rocm/bin/hipcc -O3 -std=c++17 -ffp-contract=off -fno-fast-math -I$root/include -I$root/rayzath_amd/csrc --offload-arch=gfx950 \
  -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-rdc --cuda-device-only -Rpass-analysis=kernel-resource-usage "$@" \
  -c $root/rayzath_amd/csrc/${UNIT:-hiprz_launch_batch}.hip -o $out/dev.o 2> $out/usage.txt
python3 - "$out/usage.txt" "$pat" <<'PY'
import re, sys
t = open(sys.argv[1]).read()
for b in re.split(r'remark: [^\n]*Function Name: ', t)[1:]:
    name = b.split('\n')[0].split(' ')[0]
    if re.search(sys.argv[2], name):
        g = lambda k: re.search(k + r': (\d+)', b).group(1)
        v, sc, oc = g('VGPRs'), g(r'ScratchSize \[bytes/lane\]'), g(r'Occupancy \[waves/SIMD\]')
        print(f"{name[:84]:84s} vgpr {v:>3s} scratch {sc:>4s} occ {oc}")
PY
rm -rf $out
