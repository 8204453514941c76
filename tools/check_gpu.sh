#!/bin/bash
# GPU tests + a short bench of every config (on the GPU box).  Usage: tools/check_gpu.sh [configs...]
set -eo pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/check
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $R/gpurun_out/check/pytest.log 2>&1 || { tail -40 $R/gpurun_out/check/pytest.log; exit 1; }
tail -2 $R/gpurun_out/check/pytest.log
for c in ${@:-B C D E}; do
  timeout -k 10 300 python bench.py --config $c --steps 20 --warmup 3 --no-cpu-baseline --min-seconds 1 --streams 1 > $R/gpurun_out/check/bench_$c.json 2> $R/gpurun_out/check/bench_$c.err || { tail -5 $R/gpurun_out/check/bench_$c.err; exit 1; }
  python3 -c "import json,sys; d=json.load(open('$R/gpurun_out/check/bench_$c.json')); r=d['roofline']; print('$c', round(d['value'],1), 'Mrays/s', round(d['ms_per_step'],3), 'ms', r['kernel'], round(r['avg_launch_us'],1), 'us frac', round(r['frac'],3), 'reference-order frac', (r.get('reference_algorithm') or {}).get('frac'), 'shade us', r.get('shade_kernel_avg_launch_us'))"
done
