#!/bin/bash
# On the GPU box: config B with the batch kernel's 4- and 5-wave builds.
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/ab_batch; mkdir -p $O
for w in ${WAVES:-5 4}; do
  HIPRZ_BATCH_WAVES=$w timeout -k 10 200 python $R/bench.py --config B --steps 20 --warmup 3 --repeats 5 --no-cpu-baseline > $O/B_$w.json 2> $O/B_$w.err || { echo "$w FAILED"; tail -3 $O/B_$w.err; exit 1; }
  python3 -c "import json; d=json.load(open('$O/B_$w.json')); r=d['roofline']; print('B waves $w  %8.1f Mrays/s  %7.3f ms/step  batch kernel %7.1f us' % (d['value'], d['ms_per_step'], r['avg_launch_us']))"
done
