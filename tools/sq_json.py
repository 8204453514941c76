#!/usr/bin/env python3
"""Compute-side figures of every rz_* kernel from a tools/pmc_sq.sh summary -> JSON that bench.py puts into the bench line
(profiles/sq_<config>.json, like profiles/traffic_<config>.json for the HBM bytes).

  valu_busy     share of the chip's vector-issue capacity the kernel used: SQ_ACTIVE_INST_VALU counts quad-cycles in which a wave had a
                VALU instruction executing, summed over all waves; GRBM_GUI_ACTIVE counts the cycles the kernel was on the chip, summed
                over the 8 XCDs (MI355X_MICROARCH.md: DVFS / SQ PMC units) -> 4 * ACTIVE_INST_VALU / (1024 SIMDs * GUI_ACTIVE / 8)
  lanes_active  SQ_THREAD_CYCLES_VALU / (64 * SQ_ACTIVE_INST_VALU): mean share of the 64 lanes enabled in a VALU instruction
usage: tools/sq_json.py summary.txt out.json ["round 4, rocprofv3 --pmc passes of bench.py --config B --streams 1"]"""
import collections, json, re, sys

k = collections.defaultdict(dict)
for line in open(sys.argv[1]):
    m = re.match(r"(.*?)\s+((?:SQ|GRBM)_\w+)\s+n=\s*\d+ mean=([\d.e+]+)", line)
    if m:
        k[m.group(1).strip()][m.group(2)] = float(m.group(3))
doc = {}
for name, c in k.items():
    if not c.get("SQ_ACTIVE_INST_VALU") or not c.get("SQ_WAVES"):
        continue
    short = re.sub(r"^.*?(rz_\w+).*$", r"\1", name)
    rec = {"instantiation": name, "valu_instr_per_wave": c["SQ_INSTS_VALU"] / c["SQ_WAVES"],
           "lanes_active": c["SQ_THREAD_CYCLES_VALU"] / (64 * c["SQ_ACTIVE_INST_VALU"]),
           "waiting_share_of_wave_cycles": c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], "vmem_reads_per_wave": c["SQ_INSTS_VMEM_RD"] / c["SQ_WAVES"]}
    if c.get("GRBM_GUI_ACTIVE"):
        # rocprofv3 hands out GRBM_GUI_ACTIVE summed over the 8 XCDs, i.e. the MEAN XCD's active cycles after / 8: the XCD that finishes last
        # was active longer than that, and a wave's "instruction executing" quad-cycles overlap the next wave's at instruction boundaries, so
        # the raw quotient of a saturated pipe reads up to ~1.03.  It is a share of a capacity: reported clamped to 1, the raw figure beside it
        # (read both to +-5 %).
        raw = 4 * c["SQ_ACTIVE_INST_VALU"] / (1024 * c["GRBM_GUI_ACTIVE"] / 8)
        rec["valu_busy"] = min(raw, 1.0)
        rec["valu_busy_raw"] = raw
        rec["gui_active_cycles_per_xcd"] = c["GRBM_GUI_ACTIVE"] / 8
    if short not in doc or c["SQ_WAVES"] > doc[short].get("_waves", 0):  # the steady-state instantiation has the biggest grids
        rec["_waves"] = c["SQ_WAVES"]
        doc[short] = rec
for v in doc.values():
    v.pop("_waves", None)
if len(sys.argv) > 3:   # where and when the passes were collected: bench.py quotes it (`roofline.counters_from`)
    doc["_meta"] = {"collected": sys.argv[3]}
json.dump(doc, open(sys.argv[2], "w"), indent=1)
print(json.dumps(doc, indent=1))
