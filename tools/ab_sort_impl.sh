#!/bin/bash
# On the GPU box: bench.py of one config under the radix sort's variants (HIPRZ_SORT_IMPL: 0 = round 3's kernels, 1 = runs scatter + LDS-atomic
# count, 5 = runs scatter + match count).  usage: tools/ab_sort_impl.sh OUTDIR CONFIG impl...
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/$1; CFG=$2; shift 2
mkdir -p $OUT
for impl in "$@"; do
  HIPRZ_SORT_IMPL=$impl timeout -k 10 240 python3 $R/bench.py --config $CFG --no-cpu-baseline --streams 1 --steps 10 --warmup 3 --repeats 3 --min-seconds 2 > $OUT/bench_${CFG}_sort$impl.json 2> $OUT/bench_${CFG}_sort$impl.err < /dev/null || { echo "impl $impl failed"; tail -3 $OUT/bench_${CFG}_sort$impl.err; exit 1; }
  python3 -c '
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("impl", sys.argv[2], "value", d["value"], d["unit"], "ms_per_step", d["ms_per_step"])
' $OUT/bench_${CFG}_sort$impl.json $impl
done
