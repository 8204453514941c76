#!/bin/bash
# Memory-side counters of the trace kernel (on the GPU box; the TA_* set aborts rocprofv3 on this image and is left out). Usage: tools/pmc_mem.sh <tag> [bench args]
set -eo pipefail
TAG=${1:-mem}; shift || true
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/mem_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_TAG_STALL_sum TCC_BUSY_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES"; do
  i=$((i+1))
  echo "pass $i: $set"
  timeout -k 10 240 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -o pmc -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --streams 1 --min-seconds 0 --repeats 1 "$@" > $OUT/p$i.log 2>&1 || tail -3 $OUT/p$i.log
done
python3 $R/tools/pmc_summary.py $OUT/p*/pmc_counter_collection.csv | grep -E "trace_[a-z_]*kernel<false, false" | tee $OUT/summary.txt
