#!/bin/bash
# On the GPU box: closest-hit walk variants (hiprz_set_traversal_mode) on one config.  Usage: tools/ab_modes.sh E 3 0 4 6
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
cfg=$1; shift
for mode in "$@"; do
  timeout -k 10 280 python3 $R/bench.py --config $cfg --steps 4 --warmup 1 --no-cpu-baseline --traversal $mode --pipeline 1 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('$cfg mode $mode', round(d['value'], 1), 'Mrays/s', round(d['ms_per_step'], 3), 'ms/step', r['kernel'].split(' ')[0], round(r['avg_launch_us'], 1), 'us; shade', r['shade_kernel_avg_launch_us'])" || echo "$cfg mode $mode failed"
done
