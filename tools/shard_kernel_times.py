#!/usr/bin/env python3
"""From a rocprofv3 --kernel-trace CSV of tools/shard_scaling.py: average duration of the batch kernel per grid size (= per shard
count), next to the wall time per step the script printed.  usage: shard_kernel_times.py trace_kernel_trace.csv"""
import collections, csv, sys
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "rz_batch_kernel" in r["Kernel_Name"] or "rz_pass_add" in r["Kernel_Name"]:
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("hiprz::", "")
        agg[(name, int(r["Grid_Size_X"]) if "Grid_Size_X" in r else int(r["Grid_Size"]))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for (name, grid), v in sorted(agg.items(), key=lambda kv: -kv[0][1]):
    v = v[len(v) // 4:]
    print(f"{name:44s} grid {grid:8d} threads  n={len(v):3d}  avg {sum(v) / len(v):8.1f} us  min {min(v):8.1f}  max {max(v):8.1f}")
