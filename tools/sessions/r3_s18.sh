#!/bin/bash
# session 18: which part of the new shard map costs B / C their eighth-of-a-frame time: the padded columns or the row offsets
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r03; mkdir -p $OUT; cd $R
for v in default stripes quincunx; do
  lib=""; [ $v != default ] && lib=$R/build/ab/libhiprz_$v.so
  HIPRZ_LIB=$lib timeout -k 10 300 python tools/shard_scaling.py --config B,C --shards 8 --steps 10 --tree 4 > $OUT/s18_$v.jsonl 2> $OUT/s18_$v.err || { tail -5 $OUT/s18_$v.err; exit 1; }
  python3 -c "
import json
for l in open('$OUT/s18_$v.jsonl'):
    d = json.loads(l); print('$v', d['config'], d['shards'], 'slowest', d['ms_per_step_slowest_shard'], 'mean', d['ms_per_step_mean'])"
done
