#!/bin/bash
# session 14: the price of a node step beyond 4 triangle tests (leaves of 8)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r03; mkdir -p $OUT; cd $R
export HIPRZ_TRUST_DEVICE_TREES=1
for cfg in D C E; do
  for cost in 6 8 12 16 32; do
      f=$OUT/s14_${cfg}_cost${cost}.json
      HIPRZ_SAH_LEAF=8 HIPRZ_SAH_COST=$cost timeout -k 10 120 python3 bench.py --config $cfg --tree 3 --no-cpu-baseline --min-seconds 0.5 --streams 1 > $f 2> $f.err || { tail -5 $f.err; exit 1; }
      python3 -c "import json; d=json.load(open('$f')); r=d['roofline']; print('$cfg leaf 8 cost $cost', round(d['single_stream']['value'],1), 'Mrays/s', round(d['single_stream']['ms_per_step'],3), 'ms/step trace', round(r['avg_launch_us'],1))"
  done
done
