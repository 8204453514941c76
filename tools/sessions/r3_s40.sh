#!/bin/bash
# session 40: 4-wide records, second cut — finite reciprocals, overflowed rays walked again by a small second launch
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r03; mkdir -p $OUT; cd $R
timeout -k 10 1000 python -m pytest tests -m gpu -q > $OUT/pytest41.log 2>&1; grep -E "^FAILED" $OUT/pytest41.log | cut -c 1-160 | head; tail -1 $OUT/pytest41.log
HIPRZ_WIDE_NODES=1 timeout -k 10 600 python -m pytest tests/test_device_build_gpu.py tests/test_world_levels_gpu.py tests/test_trees_gpu.py tests/test_full_size_gpu.py -m gpu -q > $OUT/pytest41_forced.log 2>&1; echo "forced on for every scene (D included: its stacks overflow, the second launch walks those rays):"; grep -E "^FAILED" $OUT/pytest41_forced.log | cut -c 1-160 | head; tail -1 $OUT/pytest41_forced.log
export HIPRZ_TRUST_DEVICE_TREES=1
for cfg in D C E; do
  for v in 0 default 1; do
      f=$OUT/s40_${cfg}_wide$v.json
      unset HIPRZ_WIDE_NODES; [ $v != default ] && export HIPRZ_WIDE_NODES=$v
      timeout -k 10 120 python3 bench.py --config $cfg --no-cpu-baseline --min-seconds 0.5 > $f 2> $f.err || { tail -5 $f.err; exit 1; }
      python3 -c "
import json; d=json.load(open('$f')); r=d['roofline']; print('$cfg wide $v value', round(d['value'],1), 'single', round(d['single_stream']['value'],1), round(d['single_stream']['ms_per_step'],3), 'ms/step trace', round(r['avg_launch_us'],1))"
  done
done
