#!/bin/bash
# session 32: E with the world level in place — the instance level and the mesh rounds again
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r03; mkdir -p $OUT; cd $R
export HIPRZ_TRUST_DEVICE_TREES=1
for v in "1 4" "2 4" "4 4" "8 4" "1 2" "1 6" "1 8"; do
  set -- $v
  f=$OUT/s32_E_inst$1_k$2.json
  HIPRZ_WALK_ADVANCE=$1 HIPRZ_WALK_K=$2 timeout -k 10 120 python3 bench.py --config E --no-cpu-baseline --min-seconds 0.5 --streams 1 > $f 2> $f.err || { tail -5 $f.err; exit 1; }
  python3 -c "import json; d=json.load(open('$f')); r=d['roofline']; print('E instance $1 K $2', round(d['single_stream']['value'],1), 'Mrays/s', round(d['single_stream']['ms_per_step'],3), 'ms/step trace', round(r['avg_launch_us'],1), 'shade+shadow', round(r.get('shade_kernel_avg_launch_us') or 0,1))"
done
