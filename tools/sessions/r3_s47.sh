#!/bin/bash
# session 47: the per-wave resident kernel (now the whole-frame default of C and D): rounds of the mesh walk, the instance level
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r03; mkdir -p $OUT; cd $R
export HIPRZ_TRUST_DEVICE_TREES=1
for cfg in C D; do
  for v in "4 1" "2 1" "3 1" "6 1" "8 1" "4 0" "4 2"; do
      set -- $v
      f=$OUT/s47_${cfg}_k$1_a$2.json
      HIPRZ_WALK_K=$1 HIPRZ_WALK_ADVANCE=$2 timeout -k 10 120 python3 bench.py --config $cfg --no-cpu-baseline --min-seconds 0.5 > $f 2> $f.err || { tail -5 $f.err; exit 1; }
      python3 -c "import json; d=json.load(open('$f')); print('$cfg K $1 advance $2: value', round(d['value'],1), 'single', round(d['single_stream']['value'],1), round(d['single_stream']['ms_per_step'],3))"
  done
done
