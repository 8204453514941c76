#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r03; mkdir -p $OUT; cd $R
export HIPRZ_TRUST_DEVICE_TREES=1
for rep in 1 2; do
for cfg in D C E; do
  for v in old new; do
      lib=""; [ $v = old ] && lib=$R/build/old/rayzath_amd/csrc/libhiprz.so
      f=$OUT/s29_${cfg}_${v}_$rep.json
      HIPRZ_LIB=$lib timeout -k 10 120 python3 bench.py --config $cfg --no-cpu-baseline --min-seconds 1 --streams 1 > $f 2> $f.err || { tail -5 $f.err; exit 1; }
      python3 -c "import json; d=json.load(open('$f')); r=d['roofline']; print('$cfg $v', round(d['single_stream']['value'],1), 'Mrays/s', round(d['single_stream']['ms_per_step'],3), 'ms/step trace', round(r['avg_launch_us'],1))"
  done
done
done
