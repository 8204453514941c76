#!/bin/bash
# On the GPU box: trace kernel register budgets (HIPRZ_TRACE_WAVES = 4 | 5 | 6) on configs C, D, E.
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/ab_waves; mkdir -p $O
for c in ${@:-D C E}; do for w in ${WAVES:-4 5 6}; do
  HIPRZ_TRACE_WAVES=$w timeout -k 10 200 python $R/bench.py --config $c --steps 10 --warmup 3 --repeats 3 --no-cpu-baseline > $O/${c}_$w.json 2> $O/${c}_$w.err || { echo "$c $w FAILED"; tail -3 $O/${c}_$w.err; exit 1; }
  python3 -c "import json; d=json.load(open('$O/${c}_$w.json')); r=d['roofline']; print('$c waves $w  %8.1f Mrays/s  %7.3f ms/step  trace %7.1f us  shade(+shadow) %7.1f us' % (d['value'], d['ms_per_step'], r['avg_launch_us'], r.get('shade_kernel_avg_launch_us') or 0))"
done; done
