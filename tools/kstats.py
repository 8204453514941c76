#!/usr/bin/env python3
"""Print a rocprofv3 kernel_stats.csv as a compact table."""
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[: int(sys.argv[2]) if len(sys.argv) > 2 else 10]:
    n = r["Name"].split("(")[0][-62:]
    print(f"{n:62s} calls={r['Calls']:>4s} avg={float(r['AverageNs'])/1000:9.1f} us  {float(r['Percentage']):5.1f}%")
