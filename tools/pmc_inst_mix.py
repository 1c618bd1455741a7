#!/usr/bin/env python3
"""Dynamic instruction mix per kernel from a rocprofv3 --pmc counter_collection.csv (SQ_INSTS_VALU / _MFMA / _SALU / _LDS / SQ_WAVES ...):
per kernel name, summed over its dispatches: instructions per wave and VALU per MFMA.   python tools/pmc_inst_mix.py counter_collection.csv"""
import csv, re, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(float))
n = defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    n[k].add(r["Dispatch_Id"])
rows = []
for k, c in acc.items():
    w = c.get("SQ_WAVES", 0.0) or 1.0
    rows.append((c.get("SQ_INSTS_VALU", 0.0), k, len(n[k]), w, c))
tot = sum(r[0] for r in rows) or 1.0
print("%-60s %6s %9s %8s %8s %8s %8s %8s %9s" % ("kernel", "disp", "waves", "VALU/wv", "MFMA/wv", "SALU/wv", "LDS/wv", "CVT/wv", "VALU/MFMA"))
for v, k, d, w, c in sorted(rows, reverse=True)[:40]:
    m = c.get("SQ_INSTS_MFMA", 0.0)
    print("%-60s %6d %9.0f %8.0f %8.0f %8.0f %8.0f %8.0f %9.2f   %4.1f%% of all VALU" % (
        k[:60], d, w, v / w, m / w, c.get("SQ_INSTS_SALU", 0.0) / w, c.get("SQ_INSTS_LDS", 0.0) / w, c.get("SQ_INSTS_VALU_CVT", 0.0) / w,
        (v - m) / m if m else float("nan"), 100.0 * v / tot))
