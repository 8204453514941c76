#!/bin/bash
# session 24: the cooperative walk's round parameters on the device's SAH trees (they were tuned on the reference trees)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r03; mkdir -p $OUT; cd $R
export HIPRZ_TRUST_DEVICE_TREES=1
for cfg in D C E; do
  for kh in "4 65" "2 65" "3 65" "6 65" "8 65" "16 65" "4 16" "4 32" "8 32" "8 16"; do
      set -- $kh
      f=$OUT/s24_${cfg}_k$1_h$2.json
      HIPRZ_WALK_K=$1 HIPRZ_WALK_H=$2 timeout -k 10 120 python3 bench.py --config $cfg --no-cpu-baseline --min-seconds 0.5 --streams 1 > $f 2> $f.err || { tail -5 $f.err; exit 1; }
      python3 -c "import json; d=json.load(open('$f')); r=d['roofline']; print('$cfg K $1 H $2', round(d['single_stream']['value'],1), 'Mrays/s', round(d['single_stream']['ms_per_step'],3), 'ms/step trace', round(r['avg_launch_us'],1))"
  done
done
