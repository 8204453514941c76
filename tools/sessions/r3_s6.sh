#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r03; mkdir -p $OUT; cd $R
for c in D C E; do
  echo "== probe loads in the cooperative walks, config $c (pairs 128-byte aligned in both)"
  bash tools/ab_run.sh --config $c --min-seconds 0 --streams 1 --repeats 3 2>&1 | tee $OUT/ab_walk_prefetch_$c.txt
done
for lib in noprobe probe; do
echo "== whole frame (N = 1) D, C: split pipeline ($lib)"
HIPRZ_LIB=build/ab/libhiprz_$lib.so HIPRZ_WAVE_RESIDENT_MAX=0 timeout -k 10 300 python tools/shard_scaling.py --config D,C --shards 1 --steps 10 2>> $OUT/shards6.err | cut -c 1-200
echo "== whole frame (N = 1) D, C: per-wave resident chains ($lib)"
HIPRZ_LIB=build/ab/libhiprz_$lib.so HIPRZ_WAVE_RESIDENT_MAX=1000000 timeout -k 10 300 python tools/shard_scaling.py --config D,C --shards 1 --steps 10 2>> $OUT/shards6.err | cut -c 1-200
done
echo "== shards of 8, no probes: split, resident"
HIPRZ_LIB=build/ab/libhiprz_noprobe.so HIPRZ_WAVE_RESIDENT_MAX=0 timeout -k 10 300 python tools/shard_scaling.py --config D,C --shards 8 --steps 10 2>> $OUT/shards6.err | cut -c 1-330
HIPRZ_LIB=build/ab/libhiprz_noprobe.so timeout -k 10 300 python tools/shard_scaling.py --config D,C --shards 8 --steps 10 2>> $OUT/shards6.err | cut -c 1-330
