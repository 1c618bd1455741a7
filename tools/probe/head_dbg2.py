#!/usr/bin/env python3
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "wt-pse-code_amd")]
import torch, torch.nn.functional as F
from wtpse_hip import ops as o
DEV = torch.device("cuda")
def rnd(*s, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*s, generator=g) * scale
B, H, W = 2, 16, 16
for variant in ("plain", "w2 exact", "dy exact", "both exact", "both exact, mask off", "mask off"):
    x = rnd(B, 32, H, W, seed=61).double()
    w1 = rnd(32, 32, 1, 1, seed=64, scale=0.3).double(); b1 = rnd(32, seed=65, scale=0.2).double()
    w2 = rnd(8, 32, 1, 1, seed=66, scale=0.3).double(); b2 = rnd(8, seed=67, scale=0.2).double()
    dy = rnd(B, 8, H, W, seed=70).double()
    if "w2 exact" in variant or "both" in variant: w2 = torch.round(w2 * 64) / 64
    if "dy exact" in variant or "both" in variant: dy = torch.round(dy * 4) / 4
    if "mask off" in variant: b1 = b1 + 50.0
    for t in (w1, b1, w2, b2): t.requires_grad_(True)
    xa = x.clone().requires_grad_(True)
    h1 = F.relu(F.conv2d(xa, w1, b1)); h2 = F.conv2d(h1, w2, b2)
    h2.backward(dy)
    D = lambda t: t.detach().float().to(DEV).contiguous()
    xd = D(x); xam = o.amax_of(xd)
    got, h1d, h2d = o.head_fwd(xd, None, False, D(w1), D(b1), D(w2), D(b2), None, None, True, x_amax=xam, want_h1=True)
    dpar = torch.full((1320,), float("nan"), device=DEV)
    dx = o.head_bwd(D(dy), xd, None, False, h1d, h2d, D(w1), D(w2), None, dpar, b1=D(b1), x_amax=xam)
    rel = lambda g, w: float((g.double().cpu() - w).norm() / w.norm())
    print("%-22s out %.2e dx %.2e dW1 %.2e db1 %.2e dW2 %.2e db2 %.2e" % (variant, rel(got, h2.detach()), rel(dx, xa.grad), rel(dpar[:1024], w1.grad.reshape(-1)),
          rel(dpar[1024:1056], b1.grad), rel(dpar[1056:1312], w2.grad.reshape(-1)), rel(dpar[1312:1320], b2.grad)))
