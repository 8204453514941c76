#!/bin/bash
# session 27: world level of the cooperative walks — lanes step through world-tree nodes until they hold a leaf (HIPRZ_WORLD_ADVANCE further steps per round)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r03; mkdir -p $OUT; cd $R
export HIPRZ_TRUST_DEVICE_TREES=1
for w in 0 1 2 4 8 64; do
  for a in 0 1; do
      f=$OUT/s27_E_world${w}_inst$a.json
      HIPRZ_WORLD_ADVANCE=$w HIPRZ_WALK_ADVANCE=$a timeout -k 10 120 python3 bench.py --config E --no-cpu-baseline --min-seconds 0.5 --streams 1 > $f 2> $f.err || { tail -5 $f.err; exit 1; }
      python3 -c "import json; d=json.load(open('$f')); r=d['roofline']; print('E world $w instance $a', round(d['single_stream']['value'],1), 'Mrays/s', round(d['single_stream']['ms_per_step'],3), 'ms/step trace', round(r['avg_launch_us'],1), 'shade+shadow', round(r.get('shade_kernel_avg_launch_us') or 0,1))"
  done
done
HIPRZ_WORLD_ADVANCE=64 timeout -k 10 600 python -m pytest tests/test_parity_gpu.py tests/test_trees_gpu.py tests/test_shading_inputs_gpu.py -m gpu -q -x 2>&1 | tail -3
