#!/usr/bin/env python3
"""3x3 weight-gradient kernels at the benchmark's layer shapes (HIP events, interleaved in one process):
fp32-input MFMA (conv.hip) | x3 through LDS (conv_x3.hip) | x3 register-resident (wgrad_r.hip).

    python tools/microbench_wgrad.py [B] [--filter name]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.environ.get("WTPSE_PKG_DIR") or os.path.join(ROOT, "wt-pse-code_amd"), os.path.join(ROOT, "tools")]     # (WTPSE_PKG_DIR: another build of the package, same-box A/B)
import torch  # noqa: E402
from wtpse_hip import ops  # noqa: E402
from microbench import SHAPES, timeit, DEV  # noqa: E402

EXTRA = [(16, 0, 16, 256, 3, "deepwt 16->16 (bias)"), (16, 0, 16, 512, 3, "16->16 @512 (B/2)")]


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 32
    flt = sys.argv[sys.argv.index("--filter") + 1] if "--filter" in sys.argv else ""
    tot = {"fp32": 0.0, "x3": 0.0, "r": 0.0, "best_old": 0.0}
    for c0, c1, co, H, k, name in SHAPES + EXTRA:
        if k != 3 or (flt and flt not in name):
            continue
        b = B // 2 if H == 512 else B
        cin = c0 + c1
        if not ops.wgrad_r_supported(cin, co, k, c0 if c1 else 16, H):
            continue
        x0 = torch.randn(b, c0, H, H, device=DEV)
        x1 = torch.randn(b, c1, H, H, device=DEV) if c1 else None
        dy = torch.randn(b, co, H, H, device=DEV)
        pro0 = torch.rand(c0, 2, device=DEV)
        pro1 = torch.rand(c1, 2, device=DEV) if c1 else None
        dw = torch.empty(co, cin, 3, 3, device=DEV)
        db = torch.empty(co, device=DEV) if "bias" in name else None
        flops = 2.0 * cin * co * 9 * H * H * b
        byts = 4.0 * (cin + co) * H * H * b
        t32, _ = timeit(lambda: ops.conv_wgrad(dy, x0, x1, k, dw, db, pro0, 3, False, pro1), 10)
        line = "%-22s %3d+%-3d->%-3d @%3d  fp32 %7.1f us %5.1f TF" % (name, c0, c1, co, H, t32, flops / t32 / 1e6)
        best_old = t32
        if db is None and ops.wgrad_x3_supported(cin, co, k, c0 if c1 else 8):
            t3, _ = timeit(lambda: ops.conv_wgrad_x3(dy, x0, x1, k, dw, pro0, 3, False, pro1), 10)
            line += " | x3-lds %7.1f us %5.1f TF" % (t3, flops / t3 / 1e6)
            best_old = min(best_old, t3)
            tot["x3"] += t3
        else:
            line += " | x3-lds       -            "
        tr, trmin = timeit(lambda: ops.conv_wgrad_r(dy, x0, x1, dw, db, pro0, 3, False, pro1), 10)
        ns = ops.lib().query("wtpse_wgrad_r_slabs", b, H, H, cin, co)
        line += " | x3-reg %7.1f us (min %7.1f) %5.1f TF  %5.2f TB/s  %4d slabs  (%.2fx)" % (tr, trmin, flops / tr / 1e6, byts / tr / 1e6, ns, best_old / tr)
        if db is None and H != 16:       # (16-wide maps: two images per step, materialised dY only)
            yb = torch.randn(b, co, H, H, device=DEV)
            coef = torch.rand(co, 3, device=DEV)
            ta, _ = timeit(lambda: ops.conv_wgrad_r_bn(dy, yb, coef, x0, x1, dw, pro0, 3, False, pro1), 10)
            line += " | +BN-apply on load %7.1f us" % ta
        tot["fp32"] += t32; tot["r"] += tr; tot["best_old"] += best_old
        print(line, flush=True)
    print("sum: fp32 %.0f us | best of the old kernels %.0f us | x3-reg %.0f us" % (tot["fp32"], tot["best_old"], tot["r"]))


if __name__ == "__main__":
    main()
