import os, sys
import numpy as np, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path[:0] = [ROOT, os.path.join(ROOT, "wt-pse-code_amd"), os.path.join(ROOT, "tests")]
from wtpse_hip import nn as E, ops
from oracle.filler import fill_state_dict
from oracle.inputs import make_noise
DEV = "cuda:0"
g = np.load(os.path.join(ROOT, "tests/golden/blocks.npz"))
SEED_W = 1000
import test_parity_gpu as T
SEED_W = T.SEED_W
B, H = 4, 16
res = {}
for terms in (3, 2):
    ops.lib().query("wtpse_x3_terms", terms)
    class Holder(E.HipNet):
        def __init__(self, blk):
            super().__init__(); self.blk = blk; self._finish_init()
    h = Holder(E.ConvUBlock(64, first=True)).to(DEV)
    fill_state_dict(h.blk, SEED_W + 20 + 2)
    h.ensure_ready(repack=True)
    x = make_noise(302, (B, 64, H // 2, H // 2)).to(DEV)
    prev = make_noise(402, (B, 32, H, H)).to(DEV)
    y, tape = E.convu_fwd(h.blk, x, prev, True)
    yd = y.dense()
    dy = make_noise(502, tuple(yd.shape)).to(DEV)
    h.begin_backward()
    dx, dprev = E.convu_bwd(h.blk, tape, dy)
    h.end_backward()
    torch.cuda.synchronize()
    res[terms] = (yd.cpu(), dprev.cpu(), tape.c3.y.cpu(), tape.c3.ss.cpu(), tape.c2.y.cpu(), tape.c2.ss.cpu())
ref_y = torch.from_numpy(g["convu_first.y"]); ref_dp = torch.from_numpy(g["convu_first.dprev"])
for terms in (3, 2):
    yd, dp, y3, ss3, y2, ss2 = res[terms]
    print("terms", terms, "y err", float((yd - ref_y).abs().max()), "dprev err", float((dp - ref_dp).abs().max()),
          "mask diff vs golden (conv3 out)", int(((yd > 0) != (ref_y > 0)).sum()))
    z2 = y2 * ss2[:, 0].view(1, -1, 1, 1) + ss2[:, 1].view(1, -1, 1, 1)
    print("   conv2/bn2 pre-activations within 1e-5 of 0:", int((z2.abs() < 1e-5).sum()), "min |z|", float(z2.abs().min()))
    z3 = y3 * ss3[:, 0].view(1, -1, 1, 1) + ss3[:, 1].view(1, -1, 1, 1)
    print("   conv3/bn3 pre-activations within 1e-5 of 0:", int((z3.abs() < 1e-5).sum()), "min |z|", float(z3.abs().min()))
m3 = res[3][0] > 0; m2 = res[2][0] > 0
print("conv3 mask flips between terms 3 and 2:", int((m3 != m2).sum()), (m3 != m2).nonzero()[:5].tolist())
z2_3 = res[3][4] * res[3][5][:, 0].view(1, -1, 1, 1) + res[3][5][:, 1].view(1, -1, 1, 1)
z2_2 = res[2][4] * res[2][5][:, 0].view(1, -1, 1, 1) + res[2][5][:, 1].view(1, -1, 1, 1)
fl = ((z2_3 > 0) != (z2_2 > 0))
print("conv2/bn2 (upsampled) mask flips:", int(fl.sum()), fl.nonzero()[:5].tolist(), [float(v) for v in z2_3[fl][:5]], [float(v) for v in z2_2[fl][:5]])
bad = (res[2][1] - ref_dp).abs() > 2e-4 + 1e-3 * ref_dp.abs()
idx = bad.nonzero()
print("bad dprev:", int(bad.sum()), "images", sorted(set(idx[:, 0].tolist())), "rows", sorted(set(idx[:, 2].tolist())), "cols", sorted(set(idx[:, 3].tolist())))
