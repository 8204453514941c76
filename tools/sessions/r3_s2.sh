#!/bin/bash
# round 3, GPU session 2: tests after the boundary changes, B shards with the LDS-stack walk, SQ counters (with GRBM_GUI_ACTIVE) for B and D
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r03; mkdir -p $OUT; cd $R
timeout -k 10 500 python -m pytest tests -m gpu -x -q > $OUT/pytest2.log 2>&1 || { tail -40 $OUT/pytest2.log; exit 1; }
tail -2 $OUT/pytest2.log
echo "== B shards, traversal 1 (LDS stack) in the resident kernel"; timeout -k 10 200 python tools/shard_scaling.py --config B --shards 1,8 --traversal 1 > $OUT/shards_B_trav1.jsonl 2> $OUT/shards2.err || { tail -5 $OUT/shards2.err; exit 1; }
cut -c 1-330 $OUT/shards_B_trav1.jsonl
for c in B D; do
  echo "== SQ counters $c"
  bash tools/pmc_sq.sh $c --config $c > $OUT/sq_$c.log 2>&1 || { tail -5 $OUT/sq_$c.log; exit 1; }
  cp gpurun_out/sq_$c/summary.txt $OUT/sq_counters_$c.txt; cp gpurun_out/sq_$c/sq.json $OUT/sq_$c.json
  python3 -c "import json; d=json.load(open('$OUT/sq_$c.json')); [print(k, {a: round(b, 3) if isinstance(b, float) else b for a, b in v.items() if a != 'instantiation'}) for k, v in d.items()]"
done
