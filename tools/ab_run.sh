#!/bin/bash
# On the GPU box: time every variant library in gpurun_out/ab/ on one config. Usage: tools/ab_run.sh [bench args]
R=${GRAFT_REPO_ROOT:-$(pwd)}
for lib in $R/build/ab/libhiprz_*.so; do
  name=$(basename $lib .so)
  HIPRZ_LIB=$lib python3 $R/bench.py --steps 10 --no-cpu-baseline "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', round(d['value'],1), 'Mrays/s', round(d['roofline']['avg_launch_us'],1), 'us')"
done
