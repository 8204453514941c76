#!/bin/bash
# session 17: empty local tiles leave the resident kernels at once — shard tests, then every shard of 4 / 8
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r03; mkdir -p $OUT; cd $R
timeout -k 10 600 python -m pytest tests/test_parity_gpu.py tests/test_boundary_gpu.py tests/test_full_size_gpu.py -m gpu -q -x > $OUT/pytest17.log 2>&1 || { grep -E "^(FAILED|ERROR)|Error|assert " $OUT/pytest17.log | tail -20; tail -5 $OUT/pytest17.log; exit 1; }
tail -2 $OUT/pytest17.log
timeout -k 10 900 python tools/shard_scaling.py --config B,C,D,E --shards 1,4,8 --steps 10 --tree 4 > $OUT/shards_row_offsets2.jsonl 2> $OUT/shards_row_offsets2.err || { tail -5 $OUT/shards_row_offsets2.err; exit 1; }
python3 -c "
import json
for l in open('$OUT/shards_row_offsets2.jsonl'):
    d = json.loads(l); print(d['config'], d['shards'], 'slowest', d['ms_per_step_slowest_shard'], 'mean', d['ms_per_step_mean'], 'speedup', d['kernel_side_speedup'], d['ms_per_shard'] if d['shards']==8 else '')"
