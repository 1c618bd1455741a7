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
B, H, W = 1, 4, 8
x = rnd(B, 32, H, W, seed=61).double()
w1 = torch.eye(32).view(32, 32, 1, 1).double(); b1 = torch.full((32,), 50.0).double()
w2 = torch.round(rnd(8, 32, 1, 1, seed=66, scale=0.3).double() * 64) / 64; b2 = rnd(8, seed=67, scale=0.2).double()
dy = torch.round(rnd(B, 8, H, W, seed=70).double() * 4) / 4
D = lambda t: t.detach().float().to(DEV).contiguous()
xd = D(x); xam = o.amax_of(xd)
got, _, h2d = o.head_fwd(xd, None, False, D(w1), D(b1), D(w2), D(b2), None, None, True, x_amax=xam)
dpar = torch.full((1320,), float("nan"), device=DEV)
dx = o.head_bwd(D(dy), xd, None, False, None, h2d, D(w1), D(w2), None, dpar, b1=D(b1), x_amax=xam).double().cpu()
want = torch.einsum("mk,bmhw->bkhw", w2.view(8, 32), dy)
err = (dx - want)
print("max err", float(err.abs().max()), "scale", float(want.abs().max()))
print("per-row (k) max err:", [("%.1e" % float(err[0, k].abs().max())) for k in range(32)])
print("per-pixel max err:", [("%.1e" % float(err[0, :, p // 8, p % 8].abs().max())) for p in range(32)])
k = int(err[0].abs().amax(dim=(1, 2)).argmax())
print("row", k, "got", dx[0, k].flatten()[:8].tolist(), "want", want[0, k].flatten()[:8].tolist())
print("w2 col", w2.view(8, 32)[:, k].tolist())
print("dy px0", dy[0, :, 0, 0].tolist())
e = err[0].reshape(32, -1)
print("mean over px:", [("%+.2e" % float(v)) for v in e.mean(1)])
print("std  over px:", [("%.1e" % float(v)) for v in e.std(1)])
cs = w2.view(8, 32).sum(0)
print("colsum W2   :", [("%+.2f" % float(v)) for v in cs])
for m in range(8):
    r = (e.mean(1) * w2.view(8, 32)[m]).sum() / (w2.view(8, 32)[m] ** 2).sum()
    print("  projection on W2 row", m, "%+.3e" % float(r))
