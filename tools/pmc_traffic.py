#!/usr/bin/env python3
"""Fold rocprofv3 --pmc passes of `bench.py --kernels-only` into profiles/pmc_traffic.json.

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_f -o f -- python3 bench.py --kernels-only
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_w -o w -- python3 bench.py --kernels-only
    python tools/pmc_traffic.py gpurun_out/pmc_f/f_counter_collection.csv gpurun_out/pmc_w/w_counter_collection.csv

MI355X_MICROARCH.md §HBM: FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports exactly half of the bytes of
a wide coalesced streaming read, so it is doubled; WRITE_SIZE is exact for 16-B-per-lane streaming stores."""
import collections
import csv
import json
import os
import sys

# kernel -> (name pattern, FETCH_SIZE factor): x2 is calibrated for 16-B-per-lane streams (the WT kernels: the corrected
# value reproduces their algorithmic bytes to 0.5 %); the conv loader issues 4-B-per-lane loads, for which the guide
# gives no calibration — the raw value (x1) is reported (halo-tile estimate: 1.33 x input), x2 would be an upper bound
KERNELS = {"conv": ("conv_fwd_k<3, 2, 5", 1.0), "wt_fwd": ("gram_partial_k", 2.0), "wt_bwd": ("gram_bwd_k", 2.0)}


def per_launch(path, counter):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}


def main():
    fetch = per_launch(sys.argv[1], "FETCH_SIZE")
    write = per_launch(sys.argv[2], "WRITE_SIZE")
    out = {}
    for key, (pat, fac) in KERNELS.items():
        f = [v for k, v in fetch.items() if pat in k]
        w = [v for k, v in write.items() if pat in k]
        if f and w:
            out[key] = {"hbm_read_bytes": fac * f[0] * 1024.0, "hbm_write_bytes": w[0] * 1024.0,
                        "hbm_bytes": fac * f[0] * 1024.0 + w[0] * 1024.0, "fetch_size_kib_raw": f[0], "write_size_kib_raw": w[0],
                        "note": "rocprofv3 --pmc, separate passes; FETCH_SIZE KiB x%g, WRITE_SIZE KiB x1 (MI355X_MICROARCH.md HBM)" % fac}
    dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "pmc_traffic.json")
    json.dump(out, open(dst, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
