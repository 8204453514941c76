#!/bin/bash
# session 20: the whole GPU suite on the tree as it stands, then every shard of 1 / 2 / 4 / 8 of B C D E under the hosts' defaults
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r03; mkdir -p $OUT; cd $R
timeout -k 10 1000 python -m pytest tests -m gpu -q -x > $OUT/pytest20.log 2>&1 || { grep -E "^(FAILED|ERROR)|Error|assert " $OUT/pytest20.log | tail -20; tail -5 $OUT/pytest20.log; exit 1; }
tail -2 $OUT/pytest20.log
timeout -k 10 900 python tools/shard_scaling.py --config B,C,D,E --shards 1,2,4,8 --steps 10 --tree 4 > $OUT/shards_final.jsonl 2> $OUT/shards_final.err || { tail -5 $OUT/shards_final.err; exit 1; }
python3 -c "
import json
for l in open('$OUT/shards_final.jsonl'):
    d = json.loads(l); print(d['config'], d['shards'], 'slowest', d['ms_per_step_slowest_shard'], 'mean', d['ms_per_step_mean'], 'speedup', d['kernel_side_speedup'])"
timeout -k 10 600 python tools/shard_scaling.py --config B,C,D --shards 1,8 --steps 10 --tree 4 --streams 2 > $OUT/shards_final_two_streams.jsonl 2>> $OUT/shards_final.err || { tail -5 $OUT/shards_final.err; exit 1; }
python3 -c "
import json
for l in open('$OUT/shards_final_two_streams.jsonl'):
    d = json.loads(l); print('two streams', d['config'], d['shards'], 'slowest', d['ms_per_step_slowest_shard'], 'mean', d['ms_per_step_mean'], 'speedup', d['kernel_side_speedup'])"
