#!/bin/bash
# round-2 baseline on today's box: GPU tests, bench for every config, SQ counters of the shipped config-B kernel
set -eo pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/r2base
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $R/gpurun_out/r2base/pytest.log 2>&1 || { tail -30 $R/gpurun_out/r2base/pytest.log; exit 1; }
tail -3 $R/gpurun_out/r2base/pytest.log
for c in B C D E; do
  timeout -k 10 300 python bench.py --config $c --steps 20 --warmup 3 --no-cpu-baseline > $R/gpurun_out/r2base/bench_$c.json 2> $R/gpurun_out/r2base/bench_$c.err || { tail -5 $R/gpurun_out/r2base/bench_$c.err; exit 1; }
  python3 -c "import json,sys; d=json.load(open('$R/gpurun_out/r2base/bench_$c.json')); r=d['roofline']; print('$c', round(d['value'],1), 'Mrays/s', round(d['ms_per_step'],3), 'ms', r['kernel'], round(r['avg_launch_us'],1), 'us frac', round(r['frac'],3), 'exec', (r.get('executed') or {}).get('frac'))"
done
bash tools/pmc_sq.sh B --config B
