#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r03; mkdir -p $OUT; cd $R
timeout -k 10 900 python -m pytest tests/test_device_build_gpu.py tests/test_trees_gpu.py tests/test_parity_gpu.py tests/test_full_size_gpu.py -q -s > $OUT/pytest5.log 2>&1 || { grep -E "^(FAILED|ERROR)|Error|assert " $OUT/pytest5.log | tail -30; }
grep -E "mesh trees|world tree|passed|failed" $OUT/pytest5.log | tail -12
for c in D C E; do
  echo "== probe loads in the cooperative walks, config $c (pairs 128-byte aligned in both)"
  bash tools/ab_run.sh --config $c --min-seconds 0 --streams 1 --repeats 3 2>&1 | tee $OUT/ab_walk_prefetch_$c.txt
done
echo "== shards of D, C: per-wave resident chains for shards of at most 4608 waves (default)"
timeout -k 10 400 python tools/shard_scaling.py --config D,C --shards 1,2,4,8 --steps 10 > $OUT/shards_DC_wave_resident.jsonl 2> $OUT/shards5.err || { tail -5 $OUT/shards5.err; exit 1; }
cut -c 1-400 $OUT/shards_DC_wave_resident.jsonl
echo "== ... allowed up to 16384 waves (shards of 2 and 4 too)"
HIPRZ_WAVE_RESIDENT_MAX=16384 timeout -k 10 400 python tools/shard_scaling.py --config D,C --shards 2,4 --steps 10 > $OUT/shards_DC_wave_resident_16k.jsonl 2>> $OUT/shards5.err || { tail -5 $OUT/shards5.err; exit 1; }
cut -c 1-400 $OUT/shards_DC_wave_resident_16k.jsonl
