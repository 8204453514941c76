#!/bin/bash
# session 36: how many key bits the ray sort needs on the device's SAH trees and with the new walk levels (8 = one radix pass, 16 = two, 24 = three)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r03; mkdir -p $OUT; cd $R
export HIPRZ_TRUST_DEVICE_TREES=1
for cfg in C D E; do
  for bits in 8 16 24; do
      f=$OUT/s36_${cfg}_bits$bits.json
      HIPRZ_SORT_BITS=$bits timeout -k 10 120 python3 bench.py --config $cfg --no-cpu-baseline --min-seconds 0.5 > $f 2> $f.err || { tail -5 $f.err; exit 1; }
      python3 -c "import json; d=json.load(open('$f')); r=d['roofline']; print('$cfg bits $bits value', round(d['value'],1), 'single', round(d['single_stream']['value'],1), round(d['single_stream']['ms_per_step'],3), 'ms/step trace', round(r['avg_launch_us'],1))"
  done
done
