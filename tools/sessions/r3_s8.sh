#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r03; mkdir -p $OUT; cd $R
timeout -k 10 1000 python -m pytest tests -m gpu -q > $OUT/pytest8.log 2>&1 || { grep -E "^(FAILED|ERROR)|Error|assert |Cannot find" $OUT/pytest8.log | tail -30; }
grep -E "passed|failed" $OUT/pytest8.log | tail -3
echo "== compat integrator (flags 31) on config E: fused kernel per pass, split pipeline"
for p in 0 -1; do
timeout -k 10 300 python bench.py --config E --mode 31 --pipeline $p --steps 3 --warmup 1 --repeats 3 --min-seconds 0 --no-cpu-baseline --streams 1 > $OUT/bench_E_compat31_pipeline$p.json 2>> $OUT/bench8.err || { tail -3 $OUT/bench8.err; }
python3 -c "import json; d=json.load(open('$OUT/bench_E_compat31_pipeline$p.json')); print('pipeline $p:', round(d['value'],1), 'Mrays/s', round(d['ms_per_step'],2), 'ms/step', d['config']['pipeline'])"
done
