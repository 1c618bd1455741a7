#!/usr/bin/env python3
"""Fold rocprofv3 --pmc passes of `bench.py --kernels-only` into profiles/pmc_traffic.json.

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_f -o f -- python3 bench.py --kernels-only
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_w -o w -- python3 bench.py --kernels-only
    python tools/pmc_traffic.py gpurun_out/pmc_f/f_counter_collection.csv gpurun_out/pmc_w/w_counter_collection.csv

MI355X_MICROARCH.md §HBM: FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half of the bytes of a wide coalesced
streaming read, other access widths are uncalibrated: "calibrate on a known byte count in your own access pattern".  So the factors
are not chosen per kernel (round 3 did) but DERIVED, per access width, from the two copy kernels of the same run
(csrc/pointwise.hip: copy_w16_k / copy_w4_k move a known 134 MB in + 134 MB out at 16 / 4 bytes per lane, rotating over 5 operand
sets so that nothing is served from the Infinity Cache):
    factor_read[w] = bytes read by copy_w<w>_k / (FETCH_SIZE KiB x 1024),   factor_write[w] likewise from WRITE_SIZE
and every kernel is corrected with the factors of ITS load / store width.  The file is stamped with the git revision and the source
hash of the library it was measured on; bench.py refuses it when the loaded library differs."""
import csv
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "wt-pse-code_amd")]

# kernel -> (name pattern, load width, store width) in bytes per lane
KERNELS = {"x3_conv": (("conv_x3_k<3", "conv_x3r_k<"), 4, 4),      # tile loader: dword buffer loads; epilogue: dword buffer stores
           "x3_wgrad": (("wgrad_r_k<2, 2",), 16, 16),
           "conv": (("conv_fwd_k<3, 2, 5",), 4, 4),
           "c16_fwd": (("conv_fwd_k<3, 3, 5", "conv_fwd_k<3, 4, 5"), 4, 4), "c16_wgrad": (("wgrad_r_k<1, 1",), 16, 16),
           "wt_fwd": (("gram_partial_k",), 16, 16), "wt_bwd": (("gram_bwd_k",), 16, 16),
           "dwt_haar": (("dwt2_stream_k<0",), 4, 4), "dwt_db2": (("dwt2_stream_k<1",), 4, 4),
           "copy_w16": (("copy_w16_k",), 16, 16), "copy_w4": (("copy_w4_k",), 4, 4)}


def per_launch(path, counter, patterns):
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(path))
            if r["Counter_Name"] == counter and any(p in r["Kernel_Name"] for p in patterns)]
    return (sum(vals) / len(vals), len(vals)) if vals else (None, 0)


def main():
    fpath, wpath = sys.argv[1], sys.argv[2]
    copy_bytes = float(sys.argv[3]) if len(sys.argv) > 3 else 32 * 16 * 256 * 256 * 4.0     # one direction of bench.py's copy probes
    fac_r, fac_w = {}, {}
    for w in (16, 4):
        f, _ = per_launch(fpath, "FETCH_SIZE", KERNELS["copy_w%d" % w][0])
        s, _ = per_launch(wpath, "WRITE_SIZE", KERNELS["copy_w%d" % w][0])
        if not f or not s:
            sys.exit("no copy_w%d_k launches in the counter files: run `bench.py --kernels-only` of this revision" % w)
        fac_r[w], fac_w[w] = copy_bytes / (f * 1024.0), copy_bytes / (s * 1024.0)
    out = {}
    for key, (pats, lw, sw) in KERNELS.items():
        f, nf = per_launch(fpath, "FETCH_SIZE", pats)
        w, nw = per_launch(wpath, "WRITE_SIZE", pats)
        if f is not None and w is not None:
            out[key] = {"hbm_read_bytes": fac_r[lw] * f * 1024.0, "hbm_write_bytes": fac_w[sw] * w * 1024.0,
                        "hbm_bytes": fac_r[lw] * f * 1024.0 + fac_w[sw] * w * 1024.0, "fetch_size_kib_raw": f, "write_size_kib_raw": w,
                        "launches_averaged": nf,
                        "note": "rocprofv3 --pmc, separate passes; FETCH_SIZE KiB x %.3f (the %d-byte-per-lane copy of this run), "
                                "WRITE_SIZE KiB x %.3f (%d-byte copy)" % (fac_r[lw], lw, fac_w[sw], sw)}
    from wtpse_hip import build
    try:
        head = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], text=True).strip()
    except Exception:
        head = os.environ.get("WTPSE_GIT_HEAD", "unknown")
    out["_calibration"] = {"copy_bytes_per_direction": copy_bytes, "fetch_factor": {str(k): v for k, v in fac_r.items()},
                           "write_factor": {str(k): v for k, v in fac_w.items()},
                           "note": "factor = known bytes of copy_w<N>_k / raw counter; applied per kernel by its load / store width"}
    out["_stamp"] = {"git_head": os.environ.get("WTPSE_GIT_HEAD", head), "source_hash": build.source_hash()}
    dst = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    json.dump(out, open(dst, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
