#!/bin/bash
# On the GPU box: the round's profile set — for every config the bench line, the rocprofv3 kernel stats of the same command, the HBM
# traffic from the FETCH_SIZE / WRITE_SIZE passes, and SQ counter passes.  Summaries land in gpurun_out/$ROUND/ (ROUND=r03 by default).
# usage: [ROUND=r03] tools/round_profiles.sh [configs...]
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${ROUND:-r03}; mkdir -p $OUT
for c in ${@:-B C D E}; do
  echo "== config $c"
  bash $R/tools/profile_gpu.sh $c --config $c > $OUT/profile_$c.log 2>&1 || { tail -5 $OUT/profile_$c.log; exit 1; }
  cp $R/gpurun_out/prof_$c/bench.json $OUT/bench_$c.json
  cp $R/gpurun_out/prof_$c/trace/trace_kernel_stats.csv $OUT/kernel_stats_$c.csv
  cp $R/gpurun_out/prof_$c/hbm_pmc.txt $OUT/hbm_pmc_$c.txt
  cp $R/gpurun_out/prof_$c/traffic.json $OUT/traffic_$c.json
  python3 $R/tools/kstats.py $OUT/kernel_stats_$c.csv 6
  bash $R/tools/pmc_sq.sh $c --config $c > /dev/null 2>&1 && cp $R/gpurun_out/sq_$c/summary.txt $OUT/sq_counters_$c.txt
  echo "progress: $c done"
done
