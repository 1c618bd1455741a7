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
# The x3 kernels run on four layer shapes each in `bench.py --kernels-only` (conv3 of up1..up4, forward and data gradient
# share one kernel name): their entry is the mean over all those launches, to be compared with the mean algorithmic bytes.
# Round 3: the weight gradient is wgrad_r_k (16-byte loads per lane: the x2 calibration applies), the 16-channel layers and the
# DWT micro-benchmark kernels are listed too (dwt2_stream_k: 4-/8-byte loads per lane — uncalibrated, raw value reported).
KERNELS = {"x3_conv": ("conv_x3_k<3", 1.0), "x3_wgrad": ("wgrad_r_k<2, 2", 2.0), "conv": ("conv_fwd_k<3, 2, 5", 1.0),
           "c16_fwd": ("conv_fwd_k<3, 3, 5", 1.0), "c16_wgrad": ("wgrad_r_k<1, 1", 2.0),
           "wt_fwd": ("gram_partial_k", 2.0), "wt_bwd": ("gram_bwd_k", 2.0),
           "dwt_haar": ("dwt2_stream_k<0", 1.0), "dwt_db2": ("dwt2_stream_k<1", 1.0)}


def per_launch(path, counter, pattern):
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(path))
            if r["Counter_Name"] == counter and pattern in r["Kernel_Name"]]
    return (sum(vals) / len(vals), len(vals)) if vals else (None, 0)


def main():
    out = {}
    for key, (pat, fac) in KERNELS.items():
        f, nf = per_launch(sys.argv[1], "FETCH_SIZE", pat)
        w, nw = per_launch(sys.argv[2], "WRITE_SIZE", pat)
        if f is not None and w is not None:
            out[key] = {"hbm_read_bytes": fac * f * 1024.0, "hbm_write_bytes": w * 1024.0,
                        "hbm_bytes": fac * f * 1024.0 + w * 1024.0, "fetch_size_kib_raw": f, "write_size_kib_raw": w,
                        "launches_averaged": nf,
                        "note": "rocprofv3 --pmc, separate passes; FETCH_SIZE KiB x%g, WRITE_SIZE KiB x1 (MI355X_MICROARCH.md HBM)" % fac}
    dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "pmc_traffic.json")
    json.dump(out, open(dst, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
