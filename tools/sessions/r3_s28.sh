#!/bin/bash
# session 28: the whole GPU suite; then D and C on the library of the commit before the world-level change against the current one, same box
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r03; mkdir -p $OUT; cd $R
timeout -k 10 1000 python -m pytest tests -m gpu -q -x > $OUT/pytest28.log 2>&1 || { grep -E "^(FAILED|ERROR)|Error|assert " $OUT/pytest28.log | tail -20; tail -5 $OUT/pytest28.log; exit 1; }
tail -2 $OUT/pytest28.log
export HIPRZ_TRUST_DEVICE_TREES=1
for rep in 1 2; do
for cfg in D C; do
  for v in old new; do
      lib=""; [ $v = old ] && lib=$R/build/old/rayzath_amd/csrc/libhiprz.so
      f=$OUT/s28_${cfg}_${v}_$rep.json
      HIPRZ_LIB=$lib timeout -k 10 120 python3 bench.py --config $cfg --no-cpu-baseline --min-seconds 1 --streams 1 > $f 2> $f.err || { tail -5 $f.err; exit 1; }
      python3 -c "import json; d=json.load(open('$f')); r=d['roofline']; print('$cfg $v', round(d['single_stream']['value'],1), 'Mrays/s', round(d['single_stream']['ms_per_step'],3), 'ms/step trace', round(r['avg_launch_us'],1))"
  done
done
done
