#!/bin/bash
# On the GPU box: rocprofv3 kernel stats of a short one-stream bench.py run.  usage: [env...] tools/kernel_profile.sh OUTDIR bench-args...
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/$1
shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o kp -- python3 $R/bench.py --no-cpu-baseline --streams 1 --steps 10 --warmup 2 --repeats 1 --min-seconds 0 "$@" > $OUT/bench.json 2> $OUT/err.log < /dev/null
F=$(find $OUT -name "*kernel_stats.csv" | head -1)
if [ -z "$F" ]; then echo "no kernel stats written"; tail -5 $OUT/err.log; exit 1; fi
python3 -c '
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if float(r["Percentage"]) < 0.3: continue
    print(r["Name"].replace("hiprz::(anonymous namespace)::", "").replace("hiprz::", "")[:64].ljust(64), r["Calls"].rjust(5), "avg %8.1f us  total %8.1f ms  %5.1f %%" % (float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, float(r["Percentage"])))
' $F
tail -c 600 $OUT/bench.json | grep -o '"ms_per_step": [0-9.]*'
