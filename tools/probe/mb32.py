#!/usr/bin/env python3
"""A/B of the 32-channel-block launches of the step on 256-pixel vs 128-pixel tiles (wtpse_x3_small_wide), as the step launches them:
forward with BatchNorm+ReLU prologue, bias, statistics + in-launch finalize where the grid allows; data gradient with the
BatchNorm-backward epilogue.  HIP events, operands rotating through > 512 MB for the 256x256 layers."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "wt-pse-code_amd")]
import torch
from wtpse_hip import ops, nn as E

DEV = torch.device("cuda")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
LAYERS = [("up4.conv3", 16, 16, 32, 256, 3), ("up4.conv1", 64, 0, 32, 128, 3), ("down1.conv1", 16, 0, 32, 128, 3),
          ("down1.conv2", 32, 0, 32, 128, 3), ("up3.conv2(1x1)", 64, 0, 32, 64, 1)]


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


for name, c0, c1, co, H, k in LAYERS:
    class Holder(E.HipNet):
        def __init__(self):
            super().__init__()
            self.conv, self.bn = E.ConvP(c0 + c1, co, k), E.BNP(co)
            self.below, self.bnb = E.ConvP(c0 + c1, c0 + c1, 3), E.BNP(c0 + c1)
            self._finish_init()
    net = Holder().to(DEV)
    net.train()
    net.ensure_ready(repack=True)
    nset = 4 if H >= 256 else 2
    sets = []
    for i in range(nset):
        x0 = torch.randn(B, c0, H, H, device=DEV)
        x1 = torch.randn(B, c1, H, H, device=DEV) if c1 else None
        a0 = E.Act(x0, torch.rand(c0, 2, device=DEV) + 0.5, True)
        a1 = E.Act(x1, torch.rand(c1, 2, device=DEV) + 0.5, True) if c1 else None
        dy = torch.randn(B, co, H, H, device=DEV)
        sets.append((a0, a1, dy))
    res = {}
    for mode in (0, 1):
        ops.lib().query("wtpse_x3_small_wide", mode)
        for a0, a1, dy in sets:
            E.act_amax(a0); E.act_amax(a1)
        fwd = timeit([(lambda a0=a0, a1=a1: E.convbn_fwd(net.conv, net.bn, a0, a1, True, True, want_tape=False)) for a0, a1, dy in sets])
        net.begin_backward()
        for a0, a1, dy in sets:
            ops.amax_of(dy)
        dg = timeit([(lambda dy=dy: E._dgrad(net.conv, dy, c0 if c1 else None)) for a0, a1, dy in sets]) if k == 3 or c0 + c1 >= 64 else float("nan")
        res[mode] = (fwd, dg)
    ops.lib().query("wtpse_x3_small_wide", 0)
    fl = 2.0 * (c0 + c1) * co * k * k * H * H * B
    print("%-16s %3d+%-3d->%-3d k%d @%3d | 256-px tiles: fwd %6.1f us %5.1f TF, dgrad %6.1f us | 128-px tiles: fwd %6.1f us %5.1f TF, dgrad %6.1f us" % (
        name, c0, c1, co, k, H, res[0][0], fl / res[0][0] / 1e6, res[0][1], res[1][0], fl / res[1][0] / 1e6, res[1][1]), flush=True)
    del sets, net
