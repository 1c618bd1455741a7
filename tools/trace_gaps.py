#!/usr/bin/env python3
"""GPU idle time inside the timed steps, from a rocprofv3 kernel trace:
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tr -o tr -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-roofline
    python3 tools/trace_gaps.py gpurun_out/tr/*/tr_kernel_trace.csv [steps]
Takes the last `steps` x (kernels per step) dispatches, merges their [start, end] intervals over all queues and prints the
span, the time at least one kernel was running, the time two were running (side-stream overlap), the idle remainder, and
the histogram of idle gaps."""
import csv
import sys

import numpy as np


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    ks, ke = "Start_Timestamp", "End_Timestamp"
    ev = sorted((int(r[ks]), int(r[ke]), r["Kernel_Name"]) for r in rows)
    # the timed region = the trailing part of the trace with a stable kernels-per-step count: find the period from the Adam launches
    adam = [i for i, e in enumerate(ev) if "adam_k" in e[2]]
    per_step = 4                                      # four optimiser steps per training step (calls A-D)
    idx = adam[-per_step * steps - 1] + 1             # first dispatch after the Adam that ends step -(steps+1)
    ev = ev[idx:adam[-1] + 1]
    t0, t1 = ev[0][0], max(e[1] for e in ev)
    pts = sorted([(s, 1) for s, _, _ in ev] + [(e, -1) for _, e, _ in ev])
    depth, last, busy1, busy2, gaps = 0, t0, 0, 0, []
    for t, d in pts:
        if depth >= 1:
            busy1 += t - last
        if depth >= 2:
            busy2 += t - last
        if depth == 0 and t > last:
            gaps.append(t - last)
        depth += d
        last = t
    span = t1 - t0
    ksum = sum(e - s for s, e, _ in ev)
    print("dispatches %d over %d steps: span %.2f ms/step, sum of kernel durations %.2f ms/step" % (len(ev), steps, span / steps * 1e-6, ksum / steps * 1e-6))
    print("  >=1 kernel running %.1f %%, >=2 running %.1f %%, idle %.1f %% (%.2f ms/step in %d gaps/step)" % (
        100 * busy1 / span, 100 * busy2 / span, 100 * (span - busy1) / span, (span - busy1) / steps * 1e-6, len(gaps) // steps))
    g = np.array(gaps) * 1e-3
    for lo, hi in ((0, 1), (1, 2), (2, 5), (5, 10), (10, 50), (50, 1e9)):
        m = (g >= lo) & (g < hi)
        print("  gaps %4g-%-4g us: %6d/step, %.2f ms/step" % (lo, hi if hi < 1e9 else float("inf"), m.sum() // steps, g[m].sum() / steps * 1e-3))


if __name__ == "__main__":
    main()
