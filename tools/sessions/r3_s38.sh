#!/bin/bash
# session 38: EXPERIMENT — 4-wide records, difference-first slab test, 8 / 16 stack entries (2 / 4 KiB of LDS per wave)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r03; mkdir -p $OUT; cd $R
for lib in "" $R/build/ab/libhiprz_stack16.so; do
HIPRZ_LIB=$lib HIPRZ_WIDE_NODES=1 timeout -k 10 600 python -m pytest tests/test_device_build_gpu.py tests/test_world_levels_gpu.py tests/test_trees_gpu.py -m gpu -q > $OUT/s38_tests.log 2>&1
echo "tests with wide records (lib '$lib'):"; grep -E "^FAILED" $OUT/s38_tests.log | cut -c 1-150 | head -8; tail -1 $OUT/s38_tests.log
done
export HIPRZ_TRUST_DEVICE_TREES=1
for cfg in D C E; do
  for v in binary wide8 wide16; do
      f=$OUT/s38_${cfg}_$v.json
      unset HIPRZ_WIDE_NODES; lib=""
      [ $v != binary ] && export HIPRZ_WIDE_NODES=1
      [ $v = wide16 ] && lib=$R/build/ab/libhiprz_stack16.so
      HIPRZ_LIB=$lib timeout -k 10 120 python3 bench.py --config $cfg --no-cpu-baseline --min-seconds 0.5 --streams 1 > $f 2> $f.err || { tail -5 $f.err; exit 1; }
      python3 -c "import json; d=json.load(open('$f')); r=d['roofline']; print('$cfg $v', round(d['single_stream']['value'],1), 'Mrays/s', round(d['single_stream']['ms_per_step'],3), 'ms/step trace', round(r['avg_launch_us'],1))"
  done
done
