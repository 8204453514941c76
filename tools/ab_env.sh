#!/bin/bash
# On the GPU box: one bench figure per (config, value of an environment switch).  usage: tools/ab_env.sh VAR "v1 v2 ..." "configs" [bench args]
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
VAR=$1; VALUES=$2; CONFIGS=$3; shift 3
for c in $CONFIGS; do
  for v in $VALUES; do
    env $VAR=$v timeout -k 10 200 python bench.py --config $c --steps 20 --warmup 3 --repeats 5 --min-seconds 2 --no-cpu-baseline "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; h=d.get('hosts_default_packaging') or {}; print('$c $VAR=$v', round(d['value'],1), 'Mrays/s', round(d['ms_per_step'],3), 'ms/step', r['kernel'].split(' ')[0], round(r['avg_launch_us'],1), 'us | hosts default', round(h.get('value',0),1), round(h.get('ms_per_step',0),3))"
  done
done
