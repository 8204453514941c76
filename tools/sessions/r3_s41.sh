#!/bin/bash
# session 41: the per-wave resident kernel on WHOLE frames again, now that the walk's instance level is in place (HIPRZ_WAVE_RESIDENT_MAX lifts the shard-size limit)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r03; mkdir -p $OUT; cd $R
export HIPRZ_TRUST_DEVICE_TREES=1
for cfg in C D; do
  for lim in 16384 40000; do
    for st in 1 2; do
      f=$OUT/s41_${cfg}_lim${lim}_streams$st.json
      HIPRZ_WAVE_RESIDENT_MAX=$lim timeout -k 10 120 python3 bench.py --config $cfg --no-cpu-baseline --min-seconds 1 --streams $st > $f 2> $f.err || { tail -5 $f.err; exit 1; }
      python3 -c "
import json; d=json.load(open('$f')); print('$cfg limit $lim streams $st: value', round(d['value'],1), round(d['ms_per_step'],3), 'single', round(d['single_stream']['value'],1), round(d['single_stream']['ms_per_step'],3), d['config'].get('pipeline'))"
    done
  done
done
