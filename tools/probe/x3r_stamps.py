#!/usr/bin/env python3
"""Diagnostic build only (conv_x3r_k with s_memtime stamps, WTPSE_X3_DBG): where a workgroup's lifetime goes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "wt-pse-code_amd")]
import torch
from wtpse_hip import ops, nn as E
DEV = torch.device("cuda")
B = 32
LAYERS = [("up2.conv3", 64, 64, 128, 64), ("up1.conv3", 128, 128, 256, 32), ("up3.conv3", 32, 32, 64, 128), ("down2.conv2", 64, 0, 64, 64), ("up3.conv1", 128, 0, 64, 64)]
for name, c0, c1, co, H in LAYERS:
    class Holder(E.HipNet):
        def __init__(self):
            super().__init__()
            self.conv, self.bn = E.ConvP(c0 + c1, co, 3), E.BNP(co)
            self._finish_init()
    net = Holder().to(DEV); net.train(); net.ensure_ready(repack=True)
    a0 = E.Act(torch.randn(B, c0, H, H, device=DEV), torch.rand(c0, 2, device=DEV) + 0.5, True)
    a1 = E.Act(torch.randn(B, c1, H, H, device=DEV), torch.rand(c1, 2, device=DEV) + 0.5, True) if c1 else None
    E.act_amax(a0); E.act_amax(a1)
    nwg = 1 << 16
    dbg = torch.zeros(nwg * 16, dtype=torch.int64, device=DEV)
    with ops.fwd_scope(DEV):
        for _ in range(3):
            E.convbn_fwd(net.conv, net.bn, a0, a1, True, True, want_tape=False)
        torch.cuda.synchronize()
        os.environ["WTPSE_X3_DBG"] = hex(dbg.data_ptr())
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        E.convbn_fwd(net.conv, net.bn, a0, a1, True, True, want_tape=False)
        e1.record()
        torch.cuda.synchronize()
        del os.environ["WTPSE_X3_DBG"]
    d = dbg.view(-1, 4).cpu().double()
    d = d[d.sum(1) > 0]
    tot = d.sum(1)
    nch = (c0 + c1) // 16
    print("%-12s %3d+%-3d->%-3d @%3d  %6.1f us | waves %5d | cycles per wave: prologue %6.0f  loop %7.0f (per chunk %5.0f, MFMA floor 3456)  of it at the chunk barrier %6.0f (%4.1f %%)  epilogue %6.0f | lifetime %7.0f" % (
        name, c0, c1, co, H, 1e3 * e0.elapsed_time(e1), d.shape[0], d[:, 0].mean(), d[:, 1].mean(), d[:, 1].mean() / nch, d[:, 2].mean(), 100 * d[:, 2].mean() / d[:, 1].mean(), d[:, 3].mean(), tot.mean()), flush=True)
