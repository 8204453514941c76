#!/bin/bash
# session 31: the bench lines of C, D, E with the SQ figures of this tree (profiles/sq_*.json regenerated after the kernel-name fix of tools/pmc_summary.py), and B's default line
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r03; mkdir -p $OUT; cd $R
for c in B C D E; do
  if [ $c = B ]; then timeout -k 10 400 python3 bench.py > $OUT/bench_$c.json 2> $OUT/bench_$c.err || { tail -5 $OUT/bench_$c.err; exit 1; }
  else timeout -k 10 400 python3 bench.py --config $c > $OUT/bench_$c.json 2> $OUT/bench_$c.err || { tail -5 $OUT/bench_$c.err; exit 1; }; fi
  python3 -c "import json; d=json.load(open('$OUT/bench_$c.json')); r=d['roofline']; print('$c', round(d['value'],1), 'Mrays/s', d['value_from'], round(d['ms_per_step'],3), 'ms/step; single', round(d['single_stream']['value'],1), round(d['single_stream']['ms_per_step'],3), 'e2e', round(d['end_to_end']['value'],1), r['kernel'].split(' ')[0], round(r['avg_launch_us'],1), 'us achieved', round(r['achieved']), 'frac', round(r['frac'],3), 'valu_busy', r.get('valu_busy'), 'lanes', r.get('lanes_active'), r.get('valu_instr_per_wave'), 'traffic', round((r.get('traffic') or 0)/1e6), 'cpu', round(d['cpu_baseline']['value'],1), 'spp', round(d['spp_per_s'],1))"
done
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')"
