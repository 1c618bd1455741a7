#!/usr/bin/env python3
"""Why is the x2h forward / data-gradient family ~19 % slower inside the step than in isolated launches?
The same launches (one U-Net's layers on the x3 kernels, as the step dispatches them), timed four ways with HIP events:
  A  isolated: each layer 20 times back to back on ONE operand set (bench.py's `roofline.achieved`: warm L2 / Infinity Cache)
  B  isolated, operands rotating through sets that add up to > 600 MB per layer (cold caches, same clocks)
  C  in network order: layer 1 .. layer N, one operand set each, the pass repeated (every launch finds its operands cold — the
     footprint of a pass is GBs — but the chip never idles between layers: the step's access pattern and duty cycle)
  D  as C for 3 seconds (sustained: what DVFS settles at)
  E  as A, but the launches the STEP makes: forward with the BatchNorm statistics finished in the launch (conv_fwd_bnf), data gradient
     with the BatchNorm-backward epilogue of the layer below (mask load of its raw output, two reductions, coefficient fold)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "wt-pse-code_amd")]
import torch
import bench
from wtpse_hip import ops, nn as E

DEV = torch.device("cuda")
B, H = 32, 256


def ev_time(fn):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1)


layers = []
for c0, c1, co, div, k, name in bench.UNET_LAYERS:
    Hc, cin = H // div, c0 + c1

    class Holder(E.HipNet):
        def __init__(self):
            super().__init__()
            self.conv = E.ConvP(cin, co, k)
            self._finish_init()
    net = Holder().to(DEV)
    net.ensure_ready(repack=True)
    if net.conv.xf_off < 0:
        continue
    per_set = 4.0 * B * (cin + co) * Hc * Hc
    nset = max(2, min(8, int(700e6 / per_set) + 1))
    sets = []
    for i in range(nset):
        x0 = torch.randn(B, c0, Hc, Hc, device=DEV)
        x1 = torch.randn(B, c1, Hc, Hc, device=DEV) if c1 else None
        a0 = E.Act(x0, torch.rand(c0, 2, device=DEV) + 0.5, True)
        a1 = E.Act(x1, torch.rand(c1, 2, device=DEV) + 0.5, True) if c1 else None
        dy = torch.randn(B, co, Hc, Hc, device=DEV)
        E.act_amax(a0); E.act_amax(a1); ops.amax_of(dy)
        sets.append((a0, a1, dy))
    fl = 2.0 * cin * co * k * k * Hc * Hc * B
    bn = E.BNP(co).to(DEV)
    object.__setattr__(bn, "_root", net)
    below = E.Tape()
    cb = c1 if c1 else cin                  # the BatchNorm'd tensor the gradient flows into: the second half of a concat, else all of it
    below.y = torch.randn(B, cb, Hc, Hc, device=DEV)
    below.ss = torch.rand(cb, 2, device=DEV) + 0.5
    below.mean = torch.randn(cb, device=DEV) * 0.1
    below.invstd = torch.rand(cb, device=DEV) + 0.5
    below.relu = True
    below.bn = E.BNP(cb).to(DEV)
    layers.append((name, net, sets, c0 if c1 else None, fl, net.conv.xd_off >= 0 and cin > 4, bn, below))

fwd = lambda L, s: E._conv(L[1].conv, L[2][s][0], L[2][s][1], False, True)
dgr = lambda L, s: E._dgrad(L[1].conv, L[2][s][2], L[3])
def fwd_step(L, s):
    return E.convbn_fwd(L[1].conv, L[6], L[2][s][0], L[2][s][1], True, True, want_tape=False)


def dgr_step(L, s):
    root = L[1]
    if not hasattr(root, "_gtarget"):
        root.begin_backward()
    # the gradient buffers of the fabricated BatchNorm below: anything of the right size
    gw = torch.empty_like(L[7].bn.weight); gb = torch.empty_like(L[7].bn.bias)
    real = root.gview
    root.gview = lambda p: gw if p is L[7].bn.weight else gb if p is L[7].bn.bias else real(p)
    try:
        return E._dgrad(L[1].conv, L[2][s][2], L[3], below0=None if L[3] is not None else L[7], below1=L[7] if L[3] is not None else None)
    finally:
        root.gview = real


PER = {}
for what, fn, sel in (("forward", fwd, lambda L: True), ("data gradient", dgr, lambda L: L[5]), ("forward, step form", fwd_step, lambda L: True),
                      ("data gradient, step form", dgr_step, lambda L: L[5])):
    Ls = [L for L in layers if sel(L)]
    flop = sum(L[4] for L in Ls)
    for L in Ls:
        for s in range(len(L[2])):
            fn(L, s)
    scope = ops.fwd_scope(DEV)
    scope.__enter__()
    per = [ev_time(lambda L=L: [fn(L, 0) for _ in range(20)]) / 20 for L in Ls]
    PER[what] = {L[0]: t for L, t in zip(Ls, per)}
    A = sum(per)
    Bt = sum(ev_time(lambda L=L: [fn(L, i % len(L[2])) for i in range(20)]) / 20 for L in Ls)
    C = ev_time(lambda: [[fn(L, p % len(L[2])) for L in Ls] for p in range(10)]) / 10
    t0 = time.time(); n = 0
    while time.time() - t0 < 3.0:
        for p in range(10):
            for L in Ls:
                fn(L, p % len(L[2]))
        n += 10
        torch.cuda.synchronize()
    D = ev_time(lambda: [[fn(L, p % len(L[2])) for L in Ls] for p in range(10)]) / 10
    scope.__exit__()
    print("%-24s %2d layers %6.0f GFLOP | A isolated warm %7.3f ms %5.1f TF | B isolated cold %7.3f ms %5.1f TF | C network order %7.3f ms %5.1f TF | "
          "D after 3 s sustained %7.3f ms %5.1f TF" % (what, len(Ls), flop / 1e9, A, flop / A / 1e9, Bt, flop / Bt / 1e9, C, flop / C / 1e9, D, flop / D / 1e9), flush=True)

print("per layer, isolated warm (us): forward plain | step form || data gradient plain | step form")
for L in layers:
    n = L[0]
    g = lambda k: ("%7.1f" % (1e3 * PER[k][n])) if n in PER[k] else "      -"
    print("%-36s %s | %s || %s | %s" % (n, g("forward"), g("forward, step form"), g("data gradient"), g("data gradient, step form")))
