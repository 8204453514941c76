#!/bin/bash
# session 30: the tree as it stands — whole GPU suite, the profile set of C D E, every shard of 1 / 2 / 4 / 8
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r03; mkdir -p $OUT; cd $R
timeout -k 10 1000 python -m pytest tests -m gpu -q -x > $OUT/pytest30.log 2>&1 || { grep -E "^(FAILED|ERROR)|Error|assert " $OUT/pytest30.log | tail -20; tail -5 $OUT/pytest30.log; exit 1; }
tail -2 $OUT/pytest30.log
bash tools/round_profiles.sh C D E || exit 1
timeout -k 10 900 python tools/shard_scaling.py --config B,C,D,E --shards 1,2,4,8 --steps 10 --tree 4 > $OUT/shards_final2.jsonl 2> $OUT/shards_final2.err || { tail -5 $OUT/shards_final2.err; exit 1; }
python3 -c "
import json
for l in open('$OUT/shards_final2.jsonl'):
    d = json.loads(l); print(d['config'], d['shards'], 'slowest', d['ms_per_step_slowest_shard'], 'mean', d['ms_per_step_mean'], 'speedup', d['kernel_side_speedup'])"
