#!/bin/bash
# session 34: leaves of more than 8 triangles from the device SAH build (the walks consume a leaf 8 triangles per round)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r03; mkdir -p $OUT; cd $R
for cfg in D C E; do
  for leaf in 8 12 16 24 32; do
      f=$OUT/s34_${cfg}_leaf$leaf.json
      HIPRZ_SAH_LEAF=$leaf timeout -k 10 120 python3 bench.py --config $cfg --no-cpu-baseline --min-seconds 0.5 --streams 1 > $f 2> $f.err || { tail -5 $f.err; exit 1; }
      python3 -c "import json; d=json.load(open('$f')); r=d['roofline']; print('$cfg leaf $leaf', round(d['single_stream']['value'],1), 'Mrays/s', round(d['single_stream']['ms_per_step'],3), 'ms/step trace', round(r['avg_launch_us'],1))"
  done
done
for leaf in 8 16 32; do
HIPRZ_SAH_LEAF=$leaf timeout -k 10 300 python tools/shard_scaling.py --config D --shards 8 --steps 10 --tree 4 > $OUT/s34_shards_D_leaf$leaf.jsonl 2> $OUT/s34.err || { tail -5 $OUT/s34.err; exit 1; }
python3 -c "
import json
for l in open('$OUT/s34_shards_D_leaf$leaf.jsonl'):
    d = json.loads(l); print('leaf $leaf', d['config'], d['shards'], 'slowest', d['ms_per_step_slowest_shard'], 'mean', d['ms_per_step_mean'], d['ms_per_shard'])"
done
