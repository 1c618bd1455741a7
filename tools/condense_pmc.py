#!/usr/bin/env python3
"""rocprofv3 --pmc counter_collection.csv (one row per dispatch and counter) -> one row per kernel: launches and the mean of
every counter over its launches.   python3 tools/condense_pmc.py raw_counter_collection.csv > profiles/rNN_pmc_xxx.csv"""
import collections
import csv
import sys


def main():
    acc = collections.OrderedDict()
    counters = []
    for r in csv.DictReader(open(sys.argv[1])):
        k, c = r["Kernel_Name"], r["Counter_Name"]
        if c not in counters:
            counters.append(c)
        d = acc.setdefault(k, {})
        s = d.setdefault(c, [0.0, 0])
        s[0] += float(r["Counter_Value"])
        s[1] += 1
    w = csv.writer(sys.stdout, quoting=csv.QUOTE_NONNUMERIC, lineterminator="\n")
    print("kernel,launches," + ",".join(counters))
    for k, d in acc.items():
        n = max(v[1] for v in d.values())
        w.writerow([k, n] + [round(d[c][0] / d[c][1], 1) if c in d else "" for c in counters])


if __name__ == "__main__":
    main()
