#!/usr/bin/env python3
"""Two rocprofv3 --kernel-trace --stats summaries (kernel_stats.csv) side by side, per step (4 adam_k launches = one step):
    python tools/compare_kernel_stats.py A_kernel_stats.csv B_kernel_stats.csv [min_us_per_step]"""
import csv, re, sys


def load(path):
    rows = [r for r in csv.DictReader(open(path)) if not r["Name"].startswith("__amd")]
    steps = sum(int(r["Calls"]) for r in rows if r["Name"].startswith("adam_k")) / 4.0
    out = {}
    for r in rows:
        n = re.sub(r"\(.*", "", r["Name"]).replace("void ", "")
        t, c = out.get(n, (0.0, 0.0))
        out[n] = (t + float(r["TotalDurationNs"]) / steps / 1e3, c + int(r["Calls"]) / steps)
    return out, steps


a, sa = load(sys.argv[1])
b, sb = load(sys.argv[2])
thr = float(sys.argv[3]) if len(sys.argv) > 3 else 20.0
ta, tb = sum(v[0] for v in a.values()), sum(v[0] for v in b.values())
print("steps %.0f / %.0f; kernel time per step %.1f us -> %.1f us (%+.1f)" % (sa, sb, ta, tb, tb - ta))
keys = sorted(set(a) | set(b), key=lambda k: -abs(b.get(k, (0, 0))[0] - a.get(k, (0, 0))[0]))
for k in keys:
    va, vb = a.get(k, (0.0, 0.0)), b.get(k, (0.0, 0.0))
    if abs(vb[0] - va[0]) < thr:
        continue
    print("%+9.1f us/step  %9.1f -> %9.1f  calls %6.1f -> %6.1f  %s" % (vb[0] - va[0], va[0], vb[0], va[1], vb[1], k[:100]))
