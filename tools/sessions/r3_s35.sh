#!/bin/bash
# session 35: B as a shard of 8 — which tiles share a compute unit (execution order of the owned tiles permuted by a coprime stride)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r03; mkdir -p $OUT; cd $R
for s in 0 3 7 37 127 251 509; do
  HIPRZ_TILE_STRIDE=$s timeout -k 10 300 python tools/shard_scaling.py --config B --shards 1,8 --steps 20 --tree 4 > $OUT/s35_B_stride$s.jsonl 2> $OUT/s35.err || { tail -5 $OUT/s35.err; exit 1; }
  python3 -c "
import json
for l in open('$OUT/s35_B_stride$s.jsonl'):
    d = json.loads(l); print('stride $s', d['config'], d['shards'], 'slowest', d['ms_per_step_slowest_shard'], 'mean', d['ms_per_step_mean'])"
done
