#!/bin/bash
# session 21: waves per SIMD of the cooperative trace kernel on the device's SAH trees
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r03; mkdir -p $OUT; cd $R
export HIPRZ_TRUST_DEVICE_TREES=1
for cfg in D C E; do
  for w in 4 5 6; do
      f=$OUT/s21_${cfg}_waves$w.json
      HIPRZ_TRACE_WAVES=$w timeout -k 10 120 python3 bench.py --config $cfg --no-cpu-baseline --min-seconds 0.5 --streams 1 > $f 2> $f.err || { tail -5 $f.err; exit 1; }
      python3 -c "import json; d=json.load(open('$f')); r=d['roofline']; print('$cfg waves $w', round(d['single_stream']['value'],1), 'Mrays/s', round(d['single_stream']['ms_per_step'],3), 'ms/step trace', round(r['avg_launch_us'],1))"
  done
done
