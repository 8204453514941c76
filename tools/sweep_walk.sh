#!/bin/bash
# On the GPU box: sweep the mesh-walk round bounds (HIPRZ_WALK_K node steps / HIPRZ_WALK_L triangles per lane per round) and the
# register budget (HIPRZ_TRACE_WAVES) of the skip-link trace kernel on one config.  Usage: tools/sweep_walk.sh D [bench args]
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
cfg=$1; shift
run() {
  timeout -k 10 200 python3 $R/bench.py --config $cfg --steps 6 --warmup 1 --no-cpu-baseline "$@" 2>/dev/null | python3 -c "
import sys, json, os
d = json.loads(sys.stdin.read()); r = d['roofline']
print('$cfg', 'K', os.environ.get('HIPRZ_WALK_K', '-'), 'L', os.environ.get('HIPRZ_WALK_L', '-'), 'waves', os.environ.get('HIPRZ_TRACE_WAVES', '-'), round(d['value'], 1), 'Mrays/s', r['kernel'].split(' ')[0], round(r['avg_launch_us'], 1), 'us')" || exit 1
}
for w in 4 6; do
  for k in 2 4 8; do
    for l in 4 8 16; do
      HIPRZ_TRACE_WAVES=$w HIPRZ_WALK_K=$k HIPRZ_WALK_L=$l run "$@" || exit 1
    done
  done
done
