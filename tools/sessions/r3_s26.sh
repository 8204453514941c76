#!/bin/bash
# session 26: after the instance-advance default — the whole GPU suite, the profile set of C and D again, every shard of 1 / 8 of C and D
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r03; mkdir -p $OUT; cd $R
timeout -k 10 1000 python -m pytest tests -m gpu -q -x > $OUT/pytest26.log 2>&1 || { grep -E "^(FAILED|ERROR)|Error|assert " $OUT/pytest26.log | tail -20; tail -5 $OUT/pytest26.log; exit 1; }
tail -2 $OUT/pytest26.log
bash tools/round_profiles.sh C D || exit 1
timeout -k 10 600 python tools/shard_scaling.py --config C,D --shards 1,2,4,8 --steps 10 --tree 4 > $OUT/shards_CD_advance.jsonl 2> $OUT/shards_CD_advance.err || { tail -5 $OUT/shards_CD_advance.err; exit 1; }
python3 -c "
import json
for l in open('$OUT/shards_CD_advance.jsonl'):
    d = json.loads(l); print(d['config'], d['shards'], 'slowest', d['ms_per_step_slowest_shard'], 'mean', d['ms_per_step_mean'], 'speedup', d['kernel_side_speedup'])"
