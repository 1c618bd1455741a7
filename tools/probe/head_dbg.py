#!/usr/bin/env python3
"""Error statistics of the fused heads against an fp64 torch evaluation (debugging aid)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "wt-pse-code_amd")]
import torch, torch.nn.functional as F
from wtpse_hip import ops as o
DEV = torch.device("cuda")
def rnd(*s, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*s, generator=g) * scale
for (B, H, W, nc, use_pro) in [(2, 8, 8, 1, True), (2, 8, 8, 1, False), (3, 16, 32, 1, False), (2, 16, 16, 0, True), (6, 64, 64, 1, True)]:
    three = nc > 0
    x = rnd(B, 32, H, W, seed=61).double()
    pro = torch.stack([rnd(32, seed=62) * 0.5 + 1.0, rnd(32, seed=63) * 0.3], 1).contiguous().double() if use_pro else None
    w1 = rnd(32, 32, 1, 1, seed=64, scale=0.3).double().requires_grad_(True); b1 = rnd(32, seed=65, scale=0.2).double().requires_grad_(True)
    w2 = rnd(8, 32, 1, 1, seed=66, scale=0.3).double().requires_grad_(True); b2 = rnd(8, seed=67, scale=0.2).double().requires_grad_(True)
    w3 = rnd(nc, 8, 1, 1, seed=68, scale=0.5).double().requires_grad_(True) if three else None
    b3 = rnd(nc, seed=69, scale=0.2).double().requires_grad_(True) if three else None
    xa = (F.relu(x * pro[:, 0].view(1, -1, 1, 1) + pro[:, 1].view(1, -1, 1, 1)) if use_pro else x.clone()).requires_grad_(True)
    h1 = F.relu(F.conv2d(xa, w1, b1))
    h2 = F.conv2d(h1, w2, b2)
    out = F.conv2d(F.relu(h2), w3, b3) if three else h2
    dy = rnd(*out.shape, seed=70).double()
    out.backward(dy)
    D = lambda t: t.detach().float().to(DEV).contiguous() if t is not None else None
    for mode in ("fixed", "table"):
        xd = D(x)
        xam = None if mode == "fixed" else (o.amax_of(xd) if not use_pro else o.act_bound(D(pro), o.amax_of(xd)))
        got, h1d, h2d = o.head_fwd(xd, D(pro), use_pro, D(w1), D(b1), D(w2), D(b2), D(w3), D(b3), True, x_amax=xam, want_h1=True)
        ns = 1320 + 9 * nc
        dpar = torch.full((ns,), float("nan"), device=DEV)
        dx = o.head_bwd(D(dy), xd, D(pro), use_pro, h1d, h2d, D(w1), D(w2), D(w3), dpar, b1=D(b1), x_amax=xam)
        def st(name, g, w):
            g = g.double().cpu(); w = w.detach()
            e = (g - w).abs()
            print("  %-8s max err %.3e  rel L2 %.3e  scale %.3e  nan %d" % (name, float(e.max()), float((g - w).norm() / w.norm()), float(w.abs().max()), int(torch.isnan(g).sum())))
        print(B, H, W, nc, use_pro, mode)
        st("out", got, out); st("h1", h1d, h1); st("h2", h2d, F.relu(h2) if three else h2); st("dx", dx, xa.grad)
        off = 0
        for nm, t in (("dW1", w1), ("db1", b1), ("dW2", w2), ("db2", b2)) + ((("dW3", w3), ("db3", b3)) if three else ()):
            k = t.numel(); st(nm, dpar[off:off + k], t.grad.reshape(-1)); off += k
