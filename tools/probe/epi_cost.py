#!/usr/bin/env python3
"""What the epilogue options of the x2h forward cost per launch: no statistics / statistics partials / statistics finished in the
launch, on layers of the step (HIP events, two operand sets)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "wt-pse-code_amd")]
import torch
from wtpse_hip import ops, nn as E
DEV = torch.device("cuda")
B = 32
LAYERS = [("up1.conv3", 128, 128, 256, 32), ("up2.conv3", 64, 64, 128, 64), ("up3.conv3", 32, 32, 64, 128), ("up4.conv3", 16, 16, 32, 256),
          ("down2.conv2", 64, 0, 64, 64), ("down3.conv2", 128, 0, 128, 32), ("up3.conv1", 128, 0, 64, 64)]


def timeit(fns, reps=20):
    for f in fns:
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        fns[i % len(fns)]()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps


for name, c0, c1, co, H in LAYERS:
    class Holder(E.HipNet):
        def __init__(self):
            super().__init__()
            self.conv, self.bn = E.ConvP(c0 + c1, co, 3), E.BNP(co)
            self._finish_init()
    net = Holder().to(DEV)
    net.train()
    net.ensure_ready(repack=True)
    sets = []
    for i in range(2):
        a0 = E.Act(torch.randn(B, c0, H, H, device=DEV), torch.rand(c0, 2, device=DEV) + 0.5, True)
        a1 = E.Act(torch.randn(B, c1, H, H, device=DEV), torch.rand(c1, 2, device=DEV) + 0.5, True) if c1 else None
        E.act_amax(a0); E.act_amax(a1)
        sets.append((a0, a1))
    with ops.fwd_scope(DEV):
        t0 = timeit([(lambda s=s: E._conv(net.conv, s[0], s[1], False, False)) for s in sets])
        t1 = timeit([(lambda s=s: E._conv(net.conv, s[0], s[1], False, True)) for s in sets])
        t2 = timeit([(lambda s=s: E.convbn_fwd(net.conv, net.bn, s[0], s[1], True, True, want_tape=False)) for s in sets])
    fl = 2.0 * (c0 + c1) * co * 9 * H * H * B
    print("%-12s %3d+%-3d->%-3d @%3d | no statistics %6.1f us %5.1f TF | + partials %6.1f us | + fold in the launch %6.1f us" % (
        name, c0, c1, co, H, t0, fl / t0 / 1e6, t1, t2), flush=True)
