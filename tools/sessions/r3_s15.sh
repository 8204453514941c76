#!/bin/bash
# session 15: the hosts' default trees (HIPRZ_TREE_AUTO): the whole GPU suite, then the bench lines
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r03; mkdir -p $OUT; cd $R
timeout -k 10 1000 python -m pytest tests -m gpu -q -x > $OUT/pytest15.log 2>&1 || { grep -E "^(FAILED|ERROR)|Error|assert " $OUT/pytest15.log | tail -20; tail -5 $OUT/pytest15.log; exit 1; }
tail -2 $OUT/pytest15.log
for c in B C D E; do
  timeout -k 10 400 python3 bench.py --config $c --no-cpu-baseline --min-seconds 2 > $OUT/s15_bench_$c.json 2> $OUT/s15_bench_$c.err || { tail -5 $OUT/s15_bench_$c.err; exit 1; }
  python3 -c "import json; d=json.load(open('$OUT/s15_bench_$c.json')); r=d['roofline']; print('$c', round(d['value'],1), 'Mrays/s', d['value_from'], round(d['ms_per_step'],3), 'ms/step; single', round(d['single_stream']['value'],1), r['kernel'].split(' ')[0], round(r['avg_launch_us'],1), 'us frac', round(r['frac'],3), d['config']['mesh_trees'])"
done
