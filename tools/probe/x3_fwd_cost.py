#!/usr/bin/env python3
"""Where the forward conv_x3_k loses against the data gradient on the same shape: prologue / concat / stats / bias toggled."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "wt-pse-code_amd"), os.path.join(ROOT, "tools"), os.path.join(ROOT, "tests")]
import torch
from wtpse_hip import ops
from microbench import timeit, DEV
from test_conv_x3_gpu import pack_x3

B = 32
for C, H in ((64, 128), (128, 64), (32, 256)):
    x0 = torch.randn(B, C // 2, H, H, device=DEV); x1 = torch.randn(B, C // 2, H, H, device=DEV)
    xa = torch.randn(B, C, H, H, device=DEV)
    w = torch.randn(C, C, 3, 3) * 0.05
    packed, xf, xd = pack_x3(w)
    bias = torch.zeros(C, device=DEV)
    p0 = torch.rand(C // 2, 2, device=DEV) + 0.5; p1 = torch.rand(C // 2, 2, device=DEV) + 0.5; pa = torch.rand(C, 2, device=DEV) + 0.5
    fl = 2.0 * C * C * 9 * H * H * B
    wp = packed.data_ptr() + 2 * xf
    rows = [
        ("one input, plain", lambda: ops.conv_fwd_x3(xa, None, wp, None, C, 3)),
        ("one input, bias", lambda: ops.conv_fwd_x3(xa, None, wp, bias, C, 3)),
        ("one input, stats", lambda: ops.conv_fwd_x3(xa, None, wp, None, C, 3, want_stats=True)),
        ("one input, ReLU on load", lambda: ops.conv_fwd_x3(xa, None, wp, None, C, 3, None, 1)),
        ("one input, affine+ReLU on load", lambda: ops.conv_fwd_x3(xa, None, wp, None, C, 3, pa, 1)),
        ("concat, plain", lambda: ops.conv_fwd_x3(x0, x1, wp, None, C, 3)),
        ("concat, affine+ReLU on load", lambda: ops.conv_fwd_x3(x0, x1, wp, None, C, 3, p0, 3, pro1=p1)),
        ("concat, affine+ReLU, bias, stats (in-step)", lambda: ops.conv_fwd_x3(x0, x1, wp, bias, C, 3, p0, 3, want_stats=True, pro1=p1)),
        ("data gradient, split", lambda: ops.conv_fwd_x3(xa, None, packed.data_ptr() + 2 * xd, None, C, 3, split=C // 2)),
    ]
    print("C=%d @%d" % (C, H))
    for name, fn in rows:
        t, mn = timeit(fn, 10)
        print("  %-46s %7.1f us (min %7.1f)  %6.1f TF" % (name, t, mn, fl / t / 1e6), flush=True)
