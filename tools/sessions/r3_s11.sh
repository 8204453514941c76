#!/bin/bash
# session 11: the device's binned surface-area builder — tests, then config D / E walks on each kind of tree
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
OUT=$R/gpurun_out/r03; mkdir -p $OUT
timeout -k 10 500 python -m pytest tests/test_device_build_gpu.py -x -q -s > $OUT/s11_tests.log 2>&1 || { tail -30 $OUT/s11_tests.log; exit 1; }
tail -3 $OUT/s11_tests.log
grep -o "'build mesh trees (device)': [0-9.]*" $OUT/s11_tests.log | tail -4
for cfg in D E; do
  for tree in 0 1 2 3; do
    timeout -k 10 200 python3 bench.py --config $cfg --tree $tree --no-cpu-baseline --min-seconds 2 > $OUT/s11_bench_${cfg}_tree$tree.json 2> $OUT/s11_bench_${cfg}_tree$tree.err || { tail -5 $OUT/s11_bench_${cfg}_tree$tree.err; exit 1; }
    python3 -c "import json; d=json.load(open('$OUT/s11_bench_${cfg}_tree$tree.json')); print('$cfg tree $tree', round(d['value'],1), 'Mrays/s', round(d['ms_per_step'],3), 'ms/step single', d.get('single_stream'))"
  done
done
