#!/bin/bash
# On the GPU box: one bench figure per variant library under build/ab/ (tools/ab_variants.sh) with extra environment.  usage: [ENV=..] tools/ab_libs_env.sh CONFIG [bench args]
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
CFG=$1; shift
for lib in "" $(ls build/ab/libhiprz_*.so 2>/dev/null); do
  HIPRZ_LIB=${lib:+$R/$lib} timeout -k 10 200 python bench.py --config $CFG --steps 10 --warmup 3 --repeats 3 --min-seconds 2 --no-cpu-baseline --streams 1 "$@" 2>/dev/null < /dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$CFG ${lib:-default}', round(d['value'],1), 'Mrays/s', round(d['ms_per_step'],3), 'ms/step', round(r['avg_launch_us'],1), 'us')"
done
