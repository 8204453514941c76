#!/bin/bash
# session 33: E on two streams with the world level in place; C and D shards after the wave-batch kernel got the one-step world level; parity subset
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r03; mkdir -p $OUT; cd $R
timeout -k 10 300 python3 bench.py --config E --streams 2 --no-cpu-baseline --min-seconds 2 > $OUT/s33_E_streams2.json 2> $OUT/s33.err || { tail -5 $OUT/s33.err; exit 1; }
python3 -c "import json; d=json.load(open('$OUT/s33_E_streams2.json')); print('E two streams', round(d['value'],1), d['value_from'], round(d['ms_per_step'],3), 'single', round(d['single_stream']['value'],1))"
timeout -k 10 600 python tools/shard_scaling.py --config C,D --shards 1,2,4,8 --steps 10 --tree 4 > $OUT/s33_shards_CD.jsonl 2>> $OUT/s33.err || { tail -5 $OUT/s33.err; exit 1; }
python3 -c "
import json
for l in open('$OUT/s33_shards_CD.jsonl'):
    d = json.loads(l); print(d['config'], d['shards'], 'slowest', d['ms_per_step_slowest_shard'], 'mean', d['ms_per_step_mean'], 'speedup', d['kernel_side_speedup'])"
timeout -k 10 600 python -m pytest tests/test_parity_gpu.py tests/test_full_size_gpu.py -m gpu -q -x 2>&1 | tail -2
for c in C D; do
  timeout -k 10 400 python3 bench.py --config $c --no-cpu-baseline --min-seconds 2 > $OUT/s33_bench_$c.json 2>> $OUT/s33.err || { tail -5 $OUT/s33.err; exit 1; }
  python3 -c "import json; d=json.load(open('$OUT/s33_bench_$c.json')); print('$c', round(d['value'],1), 'Mrays/s', d['value_from'], round(d['ms_per_step'],3), 'single', round(d['single_stream']['value'],1))"
done
