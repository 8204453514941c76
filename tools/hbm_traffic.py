#!/usr/bin/env python3
"""HBM traffic per launch of every rz_* kernel from the two rocprofv3 PMC passes of tools/profile_gpu.sh
(`--pmc FETCH_SIZE` and `--pmc WRITE_SIZE`, collected separately: the TCC block cannot hold both).

Correction (MI355X_MICROARCH.md, HBM): on gfx950 FETCH_SIZE counts a wide coalesced read stream at half its bytes, WRITE_SIZE is
exact for 16-B-per-lane stores; both are in KiB.  hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 — exact for the state streams,
an upper bound for scattered 32-B node fetches.

usage: tools/hbm_traffic.py <config> <prof dir of profile_gpu.sh> <out txt> <out json>"""
import collections
import csv
import json
import sys

cfg, prof, out_txt, out_json = sys.argv[1:5]


def means(path, counter):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter and "rz_" in r["Kernel_Name"]:
            agg[r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("hiprz::", "").split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in agg.items()}


fetch = means(f"{prof}/pmc_fetch/pmc_counter_collection.csv", "FETCH_SIZE")
write = means(f"{prof}/pmc_write/pmc_counter_collection.csv", "WRITE_SIZE")
doc = {"config": cfg, "correction": "hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024: on gfx950 FETCH_SIZE counts a wide coalesced read stream at half its "
       "bytes (MI355X_MICROARCH.md, HBM); exact for the 16-B/lane state streams, an upper bound for scattered node fetches"}
lines = []
for k in sorted(fetch, key=lambda k: -fetch[k][0]):
    if k not in write:
        continue
    f, n = fetch[k]
    w, _ = write[k]
    total = (2 * f + w) * 1024
    lines.append(f"{k:60s} FETCH_SIZE {f:14.1f} KiB  WRITE_SIZE {w:14.1f} KiB  corrected HBM bytes/launch {total / 1e6:10.1f} MB  (n={n})")
    short = k.split("<")[0]
    # the uncounted, steady-state instantiation (FIRST = false, COUNT = false) is the one the bench times: it has the most launches
    if short not in doc or n > doc[short]["launches_sampled"]:
        doc[short] = {"FETCH_SIZE_KiB": f, "WRITE_SIZE_KiB": w, "hbm_bytes_per_launch": total, "launches_sampled": n, "instantiation": k}
open(out_txt, "w").write("\n".join(lines) + "\n")
json.dump(doc, open(out_json, "w"), indent=1)
print("\n".join(lines))
