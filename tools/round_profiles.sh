#!/bin/bash
# On the GPU box: the round's profile set — for every config the SQ counter passes (-> profiles/sq_<config>.json), the rocprofv3 kernel stats
# of the bench command, the HBM traffic from the FETCH_SIZE / WRITE_SIZE passes (-> profiles/traffic_<config>.json), and then the bench line
# itself, which quotes both.  Summaries land in gpurun_out/$ROUND/ (ROUND=r04 by default).
# usage: [ROUND=r04] tools/round_profiles.sh [configs...]
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${ROUND:-r04}; mkdir -p $OUT
for c in ${@:-B C D E}; do
  echo "== config $c"
  bash $R/tools/pmc_sq.sh $c --config $c > $OUT/sq_$c.log 2>&1 || { tail -5 $OUT/sq_$c.log; exit 1; }
  cp $R/gpurun_out/sq_$c/summary.txt $OUT/sq_counters_$c.txt; cp $R/gpurun_out/sq_$c/sq.json $OUT/sq_$c.json; cp $R/gpurun_out/sq_$c/sq.json $R/profiles/sq_$c.json
  echo "progress: $c SQ counters done"
  bash $R/tools/profile_gpu.sh $c --config $c --no-cpu-baseline --min-seconds 1 > $OUT/profile_$c.log 2>&1 || { tail -5 $OUT/profile_$c.log; exit 1; }
  cp $R/gpurun_out/prof_$c/trace/trace_kernel_stats.csv $OUT/kernel_stats_$c.csv
  cp $R/gpurun_out/prof_$c/hbm_pmc.txt $OUT/hbm_pmc_$c.txt
  cp $R/gpurun_out/prof_$c/traffic.json $OUT/traffic_$c.json; cp $R/gpurun_out/prof_$c/traffic.json $R/profiles/traffic_$c.json
  python3 $R/tools/kstats.py $OUT/kernel_stats_$c.csv 6
  echo "progress: $c kernel stats + HBM counters done"
  (cd $R && timeout -k 10 400 python3 bench.py --config $c > $OUT/bench_$c.json 2> $OUT/bench_$c.err) || { tail -5 $OUT/bench_$c.err; exit 1; }
  python3 -c "import json; d=json.load(open('$OUT/bench_$c.json')); r=d['roofline']; print('$c', round(d['value'],1), 'Mrays/s', round(d['ms_per_step'],3), 'ms/step', r['kernel'].split(' ')[0], round(r['avg_launch_us'],1), 'us frac', round(r['frac'],3), 'valu_busy', r.get('valu_busy'), 'lanes', r.get('lanes_active'), 'traffic MB', (r.get('traffic') or 0)/1e6)"
done
