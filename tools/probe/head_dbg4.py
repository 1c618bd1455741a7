#!/usr/bin/env python3
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "wt-pse-code_amd")]
import torch
from wtpse_hip import ops as o
DEV = torch.device("cuda")
def rnd(*s, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*s, generator=g) * scale
B, H, W = 1, 4, 8
x = rnd(B, 32, H, W, seed=61).double()
w1 = torch.eye(32).view(32, 32, 1, 1).double(); b1 = torch.full((32,), 50.0).double()
w2f = torch.round(rnd(8, 32, 1, 1, seed=66, scale=0.3).double() * 64) / 64; b2 = rnd(8, seed=67, scale=0.2).double()
dyf = torch.round(rnd(B, 8, H, W, seed=70).double() * 4) / 4
D = lambda t: t.detach().float().to(DEV).contiguous()
xd = D(x); xam = o.amax_of(xd)
def run(w2, dy, tag):
    got, _, h2d = o.head_fwd(xd, None, False, D(w1), D(b1), D(w2), D(b2), None, None, True, x_amax=xam)
    dpar = torch.full((1320,), float("nan"), device=DEV)
    dx = o.head_bwd(D(dy), xd, None, False, None, h2d, D(w1), D(w2), None, dpar, b1=D(b1), x_amax=xam).double().cpu()
    want = torch.einsum("mk,bmhw->bkhw", w2.view(8, 32), dy)
    print("%-28s max err %.3e  scale %.3e" % (tag, float((dx - want).abs().max()), float(want.abs().max())))
run(w2f, dyf, "full")
for m in range(8):
    dy = torch.zeros_like(dyf); dy[:, m] = dyf[:, m]
    run(w2f, dy, "dy row %d only" % m)
for m in (0, 3):
    dy = torch.zeros_like(dyf); dy[:, m] = 1.0
    run(w2f, dy, "dy row %d = 1" % m)
w2 = torch.zeros_like(w2f); w2[3] = 0.5
run(w2, dyf, "w2 row 3 = 0.5 only")
w2 = torch.zeros_like(w2f); w2[:, 5] = 0.5
run(w2, dyf, "w2 col 5 = 0.5 only")
