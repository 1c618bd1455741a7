#!/usr/bin/env python3
"""Kernel time of ONE training step by kernel family (the families of tools/pmc_step_traffic.py), from a rocprofv3 --kernel-trace --stats
summary of a bench.py run (4 adam_k launches = one step), beside the family's HBM traffic (profiles/<tag>_step_traffic.json):
    python tools/family_time.py profiles/r06_bench_b32_single_stream_kernel_stats.csv [profiles/r06_step_traffic.json] [copy TB/s]"""
import csv, json, sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_step_traffic import fam_of, FAMILIES

rows = [r for r in csv.DictReader(open(sys.argv[1])) if not r["Name"].startswith("__amd")]
steps = sum(int(r["Calls"]) for r in rows if r["Name"].startswith("adam_k")) / 4.0
traffic = json.load(open(sys.argv[2])) if len(sys.argv) > 2 else {}
rate = float(sys.argv[3]) if len(sys.argv) > 3 else 5.9
t, c = {}, {}
for r in rows:
    f = fam_of(r["Name"])
    t[f] = t.get(f, 0.0) + float(r["TotalDurationNs"]) / steps / 1e6
    c[f] = c.get(f, 0.0) + int(r["Calls"]) / steps
tot = sum(t.values())
print("steps %.0f, kernel time per step %.2f ms" % (steps, tot))
print("%-14s %8s %6s %9s %8s %12s" % ("family", "ms/step", "%", "launches", "GB/step", "ms at copy"))
for f, _ in FAMILIES:
    if f not in t:
        continue
    tr = traffic.get(f)
    gb = (tr["read_MB_per_step"] + tr["write_MB_per_step"]) / 1e3 if tr else float("nan")
    print("%-14s %8.2f %6.1f %9.1f %8.1f %12.2f" % (f, t[f], 100 * t[f] / tot, c[f], gb, gb / rate))
