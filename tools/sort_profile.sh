#!/bin/bash
# On the GPU box: per-kernel times of the radix sort (rocprofv3 kernel trace of tools/sort_bench.py).  usage: tools/sort_profile.sh OUTDIR [n ...]
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/$1
shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o sp -- python3 $R/tools/sort_bench.py "$@" > $OUT/bench.json 2> $OUT/err.log
F=$(find $OUT -name "*kernel_stats.csv" | head -1)
if [ -z "$F" ]; then echo "no kernel stats written"; tail -5 $OUT/err.log; exit 1; fi
python3 -c '
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print(r["Name"].replace("hiprz::(anonymous namespace)::", "")[:60].ljust(60), r["Calls"].rjust(5), "avg %8.1f us  min %8.1f  max %8.1f" % (float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
' $F
