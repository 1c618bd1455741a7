#!/usr/bin/env python3
"""HBM bytes of ONE training step by kernel family, from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE: separate runs, as
MI355X_MICROARCH.md prescribes) of `bench.py --steps S --warmup W --roi-presteps 0 --no-cpu-baseline --no-kernel-roofline`.

    python tools/pmc_step_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <steps in the run> [pmc_traffic.json]

Calibration: the per-access-width factors of profiles/pmc_traffic.json (`_calibration`: derived from the copy kernels of the same
library in tools/pmc_traffic.py) — FETCH_SIZE under-reports wide streaming reads on gfx950; kernels are mapped to the width of their
dominant loads / stores (KERNEL_WIDTH below, 4 bytes per lane unless listed).  -> JSON on stdout: per family MB per step, the step's total,
and the time that total takes at the box's own streaming-copy rate (the HBM floor of the step)."""
import csv
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAMILIES = (("x3_conv", ("conv_x3_k", "conv_x3r_k")), ("x3_wgrad", ("wgrad_r_k<2", "wgrad_r_k<1, 2", "conv_wgrad_x3_k", "wgrad_fold4_k")),
            ("conv16", ("conv_fwd_k<3, 3", "conv_fwd_k<3, 4", "wgrad_r_k<1, 1")), ("bn_backward", ("bn_bwd_",)), ("heads", ("head_",)),
            ("conv_fp32", ("conv_fwd_k", "conv_wgrad_k", "wgrad_reduce_k")), ("pool_upsample", ("maxpool2", "upsample2x")),
            ("wt_loss", ("gram_", "mmd_", "wt_")), ("adam_pack", ("adam_k", "pack_")), ("other", ("",)))
WIDE = ("wgrad_r_k", "gram_", "bn_bwd_apply_k<true>", "adam_k", "axpy_v_k", "relu_mask_v", "copy_w16", "upsample2x_bwd_v_k", "maxpool2_bwd",
        "affine_act_k<true>", "wgrad_fold4_k", "amax_k", "zero")          # kernels whose streaming accesses are 16 bytes per lane


def fam_of(name):
    for f, pats in FAMILIES:
        if any(p in name for p in pats):
            return f
    return "other"


def main():
    fpath, wpath, steps = sys.argv[1], sys.argv[2], float(sys.argv[3])
    cal = json.load(open(sys.argv[4] if len(sys.argv) > 4 else os.path.join(ROOT, "profiles", "pmc_traffic.json")))["_calibration"]
    fr, fw = {int(k): v for k, v in cal["fetch_factor"].items()}, {int(k): v for k, v in cal["write_factor"].items()}
    tot = defaultdict(lambda: [0.0, 0.0, 0])
    for path, counter, idx, fac in ((fpath, "FETCH_SIZE", 0, fr), (wpath, "WRITE_SIZE", 1, fw)):
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] != counter:
                continue
            n = r["Kernel_Name"]
            w = 16 if any(p in n for p in WIDE) else 4
            t = tot[fam_of(n)]
            t[idx] += float(r["Counter_Value"]) * 1024.0 * fac[w]
            if idx == 0:
                t[2] += 1
    out = {f: {"read_MB_per_step": v[0] / steps / 1e6, "write_MB_per_step": v[1] / steps / 1e6, "launches_per_step": v[2] / steps}
           for f, v in sorted(tot.items(), key=lambda kv: -(kv[1][0] + kv[1][1]))}
    total = sum(v[0] + v[1] for v in tot.values()) / steps
    out["_total"] = {"GB_per_step": total / 1e9, "steps_in_run": steps,
                     "note": "all steps of the run are averaged (priming, warm-up and timed steps run the same launches)"}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
