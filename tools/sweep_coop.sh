#!/bin/bash
# On the GPU box: sweep the cooperative walk's round bounds (HIPRZ_WALK_H holders that end a node phase, HIPRZ_WALK_K node steps per
# round).  Usage: tools/sweep_coop.sh "D E" "8,4 8,8 16,8"
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
for cfg in $1; do
  for hk in $2; do
    h=${hk%,*}; k=${hk#*,}
    HIPRZ_WALK_H=$h HIPRZ_WALK_K=$k timeout -k 10 280 python3 $R/bench.py --config $cfg --steps 5 --warmup 1 --repeats 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('$cfg', 'H=$h K=$k', round(d['value'], 1), 'Mrays/s', round(d['ms_per_step'], 3), 'ms/step', r['kernel'].split(' ')[0], round(r['avg_launch_us'], 1), 'us; shade+shadow', round(r['shade_kernel_avg_launch_us'] or 0, 1), 'frac', round(r['frac'], 3))" || echo "$cfg $hk failed"
  done
done
