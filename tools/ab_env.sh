#!/bin/bash
# On the GPU box: one config under a list of environment settings.  Usage: tools/ab_env.sh E "HIPRZ_SHADOW_SORT=0" "HIPRZ_SHADOW_SORT=1" ...
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
cfg=$1; shift
for setting in "$@"; do
  env $setting timeout -k 10 280 python3 $R/bench.py --config $cfg --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('$cfg', '$setting', round(d['value'], 1), 'Mrays/s', round(d['ms_per_step'], 3), 'ms/step', r['kernel'].split(' ')[0], round(r['avg_launch_us'], 1), 'us; shade+shadow', round(r['shade_kernel_avg_launch_us'] or 0, 1))" || echo "$cfg $setting failed"
done
