#!/usr/bin/env python3
"""Print a rocprofv3 kernel_stats.csv as a compact table."""
import csv, re, sys
for r in list(csv.DictReader(open(sys.argv[1])))[: int(sys.argv[2]) if len(sys.argv) > 2 else 10]:
    n = re.sub(r"\(anonymous namespace\)::|hiprz::|void ", "", r["Name"]).split("(")[0][-60:]
    print(f"{n:60s} calls={r['Calls']:>5s} avg={float(r['AverageNs'])/1000:9.1f} us  total={float(r['TotalDurationNs'])/1e6:9.2f} ms {float(r['Percentage']):5.1f}%")
