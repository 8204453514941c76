#!/bin/bash
# SQ counter passes for the pass kernel (on the GPU box). Usage: tools/pmc_sq.sh <tag> [bench args]
set -eo pipefail
TAG=${1:-sq}; shift || true
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/sq_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
A="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VALU_TRANS_F32"
B="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM_RD SQ_INSTS_BRANCH SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE"  # GRBM slots are independent of the SQ block's 8
rocprofv3 --pmc $A --kernel-trace --output-format csv -d $OUT/a -o pmc -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --streams 1 --min-seconds 0 --repeats 1 "$@" > $OUT/a.log 2>&1
rocprofv3 --pmc $B --kernel-trace --output-format csv -d $OUT/b -o pmc -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --streams 1 --min-seconds 0 --repeats 1 "$@" > $OUT/b.log 2>&1
python3 $R/tools/pmc_summary.py $OUT/a/pmc_counter_collection.csv $OUT/b/pmc_counter_collection.csv | tee $OUT/summary.txt
python3 $R/tools/sq_json.py $OUT/summary.txt $OUT/sq.json "${ROUND:-r04}: rocprofv3 --pmc passes of bench.py $* --streams 1 on an MI355X box of the builder's pool"
