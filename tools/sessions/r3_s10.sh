#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r03; mkdir -p $OUT; cd $R
timeout -k 10 300 python -m pytest tests/test_boundary_gpu.py -m gpu -q > $OUT/pytest10.log 2>&1 || { grep -E "^(FAILED|ERROR)|Error|assert " $OUT/pytest10.log | tail -20; }
grep -E "passed|failed" $OUT/pytest10.log | tail -2
for c in B C D E; do
  timeout -k 10 400 python3 bench.py --config $c > $OUT/bench_$c.json 2> $OUT/bench_$c.err || { tail -5 $OUT/bench_$c.err; exit 1; }
  python3 -c "import json; d=json.load(open('$OUT/bench_$c.json')); r=d['roofline']; print('$c', round(d['value'],1), 'Mrays/s', d['value_from'], round(d['ms_per_step'],3), 'ms/step; single', round(d['single_stream']['value'],1), r['kernel'].split(' ')[0], round(r['avg_launch_us'],1), 'us frac', round(r['frac'],3), 'valu_busy', round(r.get('valu_busy') or 0,3), 'lanes', round(r.get('lanes_active') or 0,3), 'e2e', round(d['end_to_end']['value'],1), 'cpu', round(d['cpu_baseline']['value'],1))"
done
