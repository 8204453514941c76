#!/bin/bash
# On the GPU box: the default library and every variant library under build/ab/ (tools/ab_variants.sh) side by side — frame digests +
# executed tests (tools/ab_digest.py: must be equal) and bench figures per config in the default packaging and the split pipeline.
# usage: tools/ab_libs.sh [configs...]   (default C D E)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
CONFIGS=${@:-C D E}
for lib in "" $(ls build/ab/libhiprz_*.so 2>/dev/null); do
  echo "== lib ${lib:-default}"
  HIPRZ_LIB=${lib:+$R/$lib} timeout -k 10 300 python tools/ab_digest.py $CONFIGS 2>&1 | grep -v amdgpu.ids
  for c in $CONFIGS; do
    for p in -1 1; do
      HIPRZ_LIB=${lib:+$R/$lib} timeout -k 10 200 python bench.py --config $c --pipeline $p --steps 10 --warmup 2 --repeats 3 --min-seconds 1 --no-cpu-baseline --streams 1 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$c pipeline $p', round(d['value'],1), 'Mrays/s', round(d['ms_per_step'],3), 'ms/step', r['kernel'].split(' ')[0], round(r['avg_launch_us'],1), 'us', 'trace alone', (r.get('traversal_kernel') or {}).get('avg_launch_us'), 'shade+shadow', r.get('shade_kernel_avg_launch_us'))"
    done
  done
done
