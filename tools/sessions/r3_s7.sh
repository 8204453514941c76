#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r03; mkdir -p $OUT; cd $R
timeout -k 10 900 python -m pytest tests/test_parity_gpu.py tests/test_full_size_gpu.py tests/test_trees_gpu.py tests/test_device_build_gpu.py tests/test_cuda_compat_gpu.py -q -s > $OUT/pytest7.log 2>&1 || { grep -E "^(FAILED|ERROR)|Error|assert " $OUT/pytest7.log | tail -30; }
grep -E "undecided|passed|failed" $OUT/pytest7.log | tail -6
for c in D C E; do
  echo "== filtered box test in the cooperative walks, config $c"
  bash tools/ab_run.sh --config $c --min-seconds 0 --streams 1 --repeats 3 2>&1 | tee $OUT/ab_filtered_box_$c.txt
done
echo "== shards of 8, D: SAH trees, device-built trees"
timeout -k 10 300 python tools/shard_scaling.py --config D --shards 1,8 --steps 10 --tree 1 2>> $OUT/shards7.err | cut -c 1-330
timeout -k 10 300 python tools/shard_scaling.py --config D --shards 1,8 --steps 10 --tree 2 2>> $OUT/shards7.err | cut -c 1-330
