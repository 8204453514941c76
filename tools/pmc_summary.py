#!/usr/bin/env python3
"""Mean of every PMC counter per kernel (rocprofv3 --pmc ... --output-format csv)."""
import collections
import csv
import sys

agg = collections.defaultdict(list)
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"].split("(")[0][-56:]
        agg[(name, r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(agg.items()):
    if any(t in k for t in ("rz_pass_kernel<false, false", "rz_trace_kernel<false, false", "rz_shade_kernel<false, false", "rz_batch_kernel<false", "rz_wave_batch_kernel<false",
                            "rz_trace_skip_kernel<false, false", "rz_trace_coop_kernel<false, false", "rz_shadow_kernel<false, false", "rz_shadow_coop_kernel<false, false", "rz_shadow_packet_kernel<false, false", "rz_radix")) or len(sys.argv) > 3:
        print(f"{k:42s} {c:28s} n={len(v):3d} mean={sum(v)/len(v):.6g}")
