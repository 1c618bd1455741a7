#!/usr/bin/env python3
"""x3 data gradient on fp32 dY (split on load) vs on a dY split once per tensor (wtpse_split3_pack + wtpse_conv_fwd_x3_pre), and
the cost of the split pass, on the network's layers (B=32)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "wt-pse-code_amd"), os.path.join(ROOT, "tests")]
from test_kernels_gpu import rnd, ops, DEV   # noqa: E402
from test_conv_x3_gpu import pack_x3         # noqa: E402

o = ops()
B = 32
LAYERS = [("down1.conv2", 32, 32, 128), ("down2.conv2", 64, 64, 64), ("down3.conv2", 128, 128, 32), ("up1.conv3", 256, 256, 32),
          ("up2.conv1", 256, 128, 32), ("up2.conv3", 128, 128, 64), ("up3.conv1", 128, 64, 64), ("up3.conv3", 64, 64, 128),
          ("up4.conv1", 64, 32, 128), ("up4.conv3", 32, 32, 256)]


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


tot = [0.0, 0.0, 0.0]
for name, cin, cout, hw in LAYERS:
    w = rnd(cout, cin, 3, 3, seed=1, scale=0.1)
    dy = torch.randn(B, cout, hw, hw, device=DEV)
    packed, _, xd = pack_x3(w)
    wp = packed.data_ptr() + 2 * xd
    dys = o.split3_pack(dy)
    t0 = timed(lambda: o.conv_fwd_x3(dy, None, wp, None, cin, 3))
    t1 = timed(lambda: o.conv_fwd_x3_pre(dys, (hw, hw), wp, None, cin))
    t2 = timed(lambda: o.split3_pack(dy))
    tot = [tot[0] + t0, tot[1] + t1, tot[2] + t2]
    print("%-12s dgrad %3d->%3d @%3d: split on load %7.1f us | pre-split %7.1f us (%.2fx) | split pass %6.1f us (%.0f GB/s)" % (
        name, cout, cin, hw, t0, t1, t0 / t1, t2, dy.numel() * 10 / t2 * 1e-3))
print("sum: %.0f us -> %.0f us (+ %.0f us if the split were a pass of its own)" % tuple(tot))
