#!/usr/bin/env python3
"""Derived figures from a tools/pmc_sq.sh summary (profiles/rNN/sq_counters_X.txt): per kernel the VALU instructions per wave, the
share of lanes active in them (SQ_THREAD_CYCLES_VALU / (64 * SQ_ACTIVE_INST_VALU)), VALU pipe busy share (SQ_ACTIVE_INST_VALU * 4
... / SQ_BUSY_CYCLES per SIMD is not exposed; the wave-level ratio ACTIVE_INST_VALU / WAVE_CYCLES is printed instead), and the
share of wave cycles spent waiting.  usage: tools/sq_derived.py file..."""
import collections, re, sys
for path in sys.argv[1:]:
    k = collections.defaultdict(dict)
    for line in open(path):
        m = re.match(r"(.*?)\s+(SQ_\w+)\s+n=\s*\d+ mean=([\d.e+]+)", line)
        if m:
            k[m.group(1).strip()][m.group(2)] = float(m.group(3))
    print(f"# {path}")
    for name, c in k.items():
        if "SQ_WAVES" not in c or not c.get("SQ_ACTIVE_INST_VALU"):
            continue
        print(f"{name:44s} VALU instr/wave {c['SQ_INSTS_VALU'] / c['SQ_WAVES']:9.0f}  lanes active {c['SQ_THREAD_CYCLES_VALU'] / (64 * c['SQ_ACTIVE_INST_VALU']):5.3f}"
              f"  VALU issue cycles / wave cycles {c['SQ_ACTIVE_INST_VALU'] / c['SQ_WAVE_CYCLES']:5.3f}  waiting / wave cycles {c['SQ_WAIT_ANY'] / c['SQ_WAVE_CYCLES']:5.3f}"
              f"  SALU/VALU {c['SQ_INSTS_SALU'] / c['SQ_INSTS_VALU']:4.2f}  VMEM reads/wave {c['SQ_INSTS_VMEM_RD'] / c['SQ_WAVES']:7.0f}")
