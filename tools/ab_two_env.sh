#!/bin/bash
# On the GPU box: bench.py of one config under "VAR=val VAR2=val2" settings given as quoted words.  usage: tools/ab_two_env.sh CONFIG "A=1 B=2" "A=3" ... 
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
CFG=$1; shift
for setting in "$@"; do
  env $setting timeout -k 10 200 python bench.py --config $CFG --steps 20 --warmup 3 --repeats 5 --min-seconds 2 --no-cpu-baseline --streams 1 2>/dev/null < /dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$CFG [$setting]', round(d['value'],1), 'Mrays/s', round(d['ms_per_step'],3), 'ms/step', round(r['avg_launch_us'],1), 'us')"
done
