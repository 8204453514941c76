#!/bin/bash
# session 13: the surface-area build's two parameters on C, D, E (device builder, tree 3): leaf size and the price of a node step
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r03; mkdir -p $OUT; cd $R
export HIPRZ_TRUST_DEVICE_TREES=1
for cfg in D C E; do
  for leaf in 8 4 2; do
    for cost in 4 2 1 0.5; do
      f=$OUT/s13_${cfg}_leaf${leaf}_cost${cost}.json
      HIPRZ_SAH_LEAF=$leaf HIPRZ_SAH_COST=$cost timeout -k 10 120 python3 bench.py --config $cfg --tree 3 --no-cpu-baseline --min-seconds 0.5 --streams 1 > $f 2> $f.err || { tail -5 $f.err; exit 1; }
      python3 -c "import json; d=json.load(open('$f')); r=d['roofline']; print('$cfg leaf $leaf cost $cost', round(d['single_stream']['value'],1), 'Mrays/s', round(d['single_stream']['ms_per_step'],3), 'ms/step trace', round(r['avg_launch_us'],1))"
    done
  done
done
