#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r03; mkdir -p $OUT; cd $R
timeout -k 10 600 python -m pytest tests/test_full_size_gpu.py tests/test_trees_gpu.py tests/test_parity_gpu.py -q -x > $OUT/pytest4.log 2>&1 || { tail -30 $OUT/pytest4.log; exit 1; }
tail -2 $OUT/pytest4.log
for c in D C E; do
  echo "== probe loads in the cooperative walks, config $c (pairs 128-byte aligned in both)"
  bash tools/ab_run.sh --config $c --min-seconds 0 --streams 1 --repeats 3 2>&1 | tee $OUT/ab_walk_prefetch_$c.txt
done
