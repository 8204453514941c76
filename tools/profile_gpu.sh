#!/bin/bash
# Runs on the GPU box (via gpurun): bench + rocprofv3 kernel trace + PMC passes for HBM traffic.  The profiled runs use --streams 1: only
# whole-frame launches on one stream, the ones the bench line's roofline describes.
# Usage: tools/profile_gpu.sh <tag> [bench args...]
set -eo pipefail
TAG=${1:-r01}; shift || true
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py "$@" > $OUT/bench.json 2> $OUT/bench.err || { tail -20 $OUT/bench.err; exit 1; }
cat $OUT/bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --streams 1 --min-seconds 0 --repeats 1 "$@" > $OUT/trace.log 2>&1
echo trace done
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o pmc -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --streams 1 --min-seconds 0 --repeats 1 "$@" > $OUT/pmc_fetch.log 2>&1
echo fetch done
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o pmc -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --streams 1 --min-seconds 0 --repeats 1 "$@" > $OUT/pmc_write.log 2>&1
echo write done
python3 $R/tools/hbm_traffic.py $TAG $OUT $OUT/hbm_pmc.txt $OUT/traffic.json > /dev/null
python3 $R/tools/kstats.py $OUT/trace/trace_kernel_stats.csv 8
cat $OUT/hbm_pmc.txt
