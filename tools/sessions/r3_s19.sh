#!/bin/bash
# session 19: config B as an eighth of a frame on the per-wave resident kernel (scene not staged in LDS) against the LDS-resident one
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r03; mkdir -p $OUT; cd $R
for l in -1 0; do
  timeout -k 10 300 python tools/shard_scaling.py --config B --shards 1,4,8 --steps 10 --tree 4 --lds-scene $l > $OUT/s19_lds$l.jsonl 2> $OUT/s19_lds$l.err || { tail -5 $OUT/s19_lds$l.err; exit 1; }
  python3 -c "
import json
for l in open('$OUT/s19_lds$l.jsonl'):
    d = json.loads(l); print('lds_scene $l', d['config'], d['shards'], 'slowest', d['ms_per_step_slowest_shard'], 'mean', d['ms_per_step_mean'])"
done
