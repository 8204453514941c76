#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r03; mkdir -p $OUT; cd $R
echo "== node-step sensitivity on D (trace kernel: base / arithmetic twice / one more dependent fetch)"
bash tools/ab_run.sh --config D --min-seconds 0 --streams 1 --repeats 3 2>&1 | tee $OUT/ab_node_step_sensitivity_D.txt
echo "== the same on C"
bash tools/ab_run.sh --config C --min-seconds 0 --streams 1 --repeats 3 2>&1 | tee $OUT/ab_node_step_sensitivity_C.txt
