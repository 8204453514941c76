#!/bin/bash
# On the GPU box: the skip-link walk in the reference's child order against front to back, per config.  Usage: tools/ab_order.sh C D E
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
for cfg in "$@"; do
  for order in 0 1; do
    timeout -k 10 280 python3 $R/bench.py --config $cfg --steps 10 --warmup 2 --no-cpu-baseline --walk-order $order 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('$cfg order $order', round(d['value'], 1), 'Mrays/s', round(d['ms_per_step'], 3), 'ms/step', r['kernel'].split(' ')[0], round(r['avg_launch_us'], 1), 'us frac', round(r['frac'], 3))" || exit 1
  done
done
