#!/bin/bash
# On the GPU box: ray reordering off / on with each sort-key layout (HIPRZ_SORT_KEY), per config.  Usage: tools/ab_sort.sh C D E
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
run() {
  timeout -k 10 280 python3 $R/bench.py --config $cfg --steps 6 --warmup 1 --no-cpu-baseline "$@" 2>/dev/null | python3 -c "
import sys, json, os
d = json.loads(sys.stdin.read()); r = d['roofline']
print('$cfg', ' '.join(sys.argv[1:]), 'key', os.environ.get('HIPRZ_SORT_KEY', '-'), round(d['value'], 1), 'Mrays/s', round(d['ms_per_step'], 3), 'ms/step', r['kernel'].split(' ')[0], round(r['avg_launch_us'], 1), 'us')" "$@" || exit 1
}
for cfg in "$@"; do
  run --ray-sort 0 || exit 1
  for key in 0 1 2; do HIPRZ_SORT_KEY=$key run --ray-sort 1 || exit 1; done
done
