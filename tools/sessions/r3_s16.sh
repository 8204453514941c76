#!/bin/bash
# session 16: the shard map with row offsets (hiprz_shard.hpp): the GPU suite, then every shard of 1 / 2 / 4 / 8 of B C D E under the hosts' default trees
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r03; mkdir -p $OUT; cd $R
timeout -k 10 1000 python -m pytest tests -m gpu -q -x > $OUT/pytest16.log 2>&1 || { grep -E "^(FAILED|ERROR)|Error|assert " $OUT/pytest16.log | tail -20; tail -5 $OUT/pytest16.log; exit 1; }
tail -2 $OUT/pytest16.log
timeout -k 10 900 python tools/shard_scaling.py --config B,C,D,E --shards 1,2,4,8 --steps 10 --tree 4 > $OUT/shards_row_offsets.jsonl 2> $OUT/shards_row_offsets.err || { tail -5 $OUT/shards_row_offsets.err; exit 1; }
python3 -c "
import json
for l in open('$OUT/shards_row_offsets.jsonl'):
    d = json.loads(l); print(d['config'], d['shards'], 'slowest', d['ms_per_step_slowest_shard'], 'mean', d['ms_per_step_mean'], 'speedup', d['kernel_side_speedup'])"
