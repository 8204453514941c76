#!/bin/bash
# session 37: EXPERIMENT — 4-wide records in the closest-hit mesh walk (HIPRZ_WIDE_NODES=1): frames against the binary walk, then D C E
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r03; mkdir -p $OUT; cd $R
HIPRZ_WIDE_NODES=1 timeout -k 10 600 python -m pytest tests/test_device_build_gpu.py tests/test_world_levels_gpu.py tests/test_trees_gpu.py -m gpu -q -x > $OUT/s37_tests.log 2>&1 || { grep -E "^(FAILED|ERROR)|Error|assert " $OUT/s37_tests.log | tail -10; tail -3 $OUT/s37_tests.log; }
tail -1 $OUT/s37_tests.log
export HIPRZ_TRUST_DEVICE_TREES=1
for cfg in D C E; do
  for w in 0 1; do
      f=$OUT/s37_${cfg}_wide$w.json
      if [ $w = 1 ]; then export HIPRZ_WIDE_NODES=1; else unset HIPRZ_WIDE_NODES; fi
      timeout -k 10 120 python3 bench.py --config $cfg --no-cpu-baseline --min-seconds 0.5 --streams 1 > $f 2> $f.err || { tail -5 $f.err; exit 1; }
      python3 -c "import json; d=json.load(open('$f')); r=d['roofline']; print('$cfg wide $w', round(d['single_stream']['value'],1), 'Mrays/s', round(d['single_stream']['ms_per_step'],3), 'ms/step trace', round(r['avg_launch_us'],1), r['kernel'][:40])"
  done
done
