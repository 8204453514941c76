#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r03; mkdir -p $OUT; cd $R
timeout -k 10 900 python -m pytest tests/test_boundary_gpu.py tests/test_parity_gpu.py tests/test_cuda_compat_gpu.py tests/test_device_build_gpu.py tests/test_adapter.py tests/test_cpp_host.py -m gpu -q > $OUT/pytest9.log 2>&1 || { grep -E "^(FAILED|ERROR)|Error|assert |Cannot find" $OUT/pytest9.log | tail -30; }
grep -E "passed|failed" $OUT/pytest9.log | tail -3
echo "== bench B (value on two streams, single stream beside it)"
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $OUT/bench_B_s9.json 2> $OUT/bench9.err || { tail -5 $OUT/bench9.err; exit 1; }
python3 -c "import json; d=json.load(open('$OUT/bench_B_s9.json')); print(round(d['value'],1), d['value_from'], 'single', round(d['single_stream']['value'],1), 'frac', round(d['roofline']['frac'],3))"
echo "== self-launched 2 ranks on one GPU, two streams each, gathered frame verified"
timeout -k 10 300 python bench.py --gpus 2 --steps 10 --warmup 2 --rehearse-on-one-gpu --verify-gather --min-seconds 1 > $OUT/rehearse2_streams.json 2> $OUT/rehearse2_streams.err || { tail -20 $OUT/rehearse2_streams.err; exit 1; }
python3 -c "import json; d=json.load(open('$OUT/rehearse2_streams.json')); print(round(d['value'],1), d['value_from'], d['n_gpus'], 'single', round(d['single_stream']['value'],1))"
echo "== shards with two streams per shard: B, C, D"
timeout -k 10 600 python tools/shard_scaling.py --config B,C,D --shards 1,8 --steps 10 --streams 2 > $OUT/shards_two_streams.jsonl 2> $OUT/shards9.err || { tail -5 $OUT/shards9.err; exit 1; }
python3 -c "
import json
for l in open('$OUT/shards_two_streams.jsonl'):
    d = json.loads(l); print(d['config'], d['shards'], 'slowest', d['ms_per_step_slowest_shard'], 'mean', d['ms_per_step_mean'], 'speedup', d['kernel_side_speedup'])"
