"""Diagnostic: per-parameter relative L2 error of HIP gradients vs the CPU oracle (fp32 and fp64) for one update()."""
import sys, os
import numpy as np, torch, torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))      # a diagnostic (uses oracle/ as the CHECKER, like the tests it borrows from; nothing here ships)
sys.path[:0] = [ROOT, os.path.join(ROOT, "wt-pse-code_amd"), os.path.join(ROOT, "tests")]
from oracle import wtpse_cpu as O
from oracle.inputs import make_inputs, make_noise
from test_parity_gpu import build_nets, HP, is_prebn_bias
B, pb, H = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (3, 1, 32)
S0 = int(sys.argv[4]) if len(sys.argv) > 4 else 0          # seed offset: golden case ci uses 600+ci, 700+ci, ...
img, od, oc = make_inputs(600 + S0, B, H, H)
eps = make_noise(700 + S0, (B, 1, H, H))
main, shape, _, _ = build_nets(pb)
sd0 = {k: v.detach().cpu().clone() for k, v in main.state_dict().items()}
main.train(); main.zero_grad(); main.set_noise([eps])
out, _, _, ins, dom = main.update(img.cuda(), od.cuda(), two_stage_inputs=img.cuda(), two_step=True)
loss = F.binary_cross_entropy(torch.sigmoid(out), od.cuda()) + ins + dom
loss.backward()
res = {}
for dt in (torch.float32, torch.float64):
    sd = {k: (v.detach().clone().to(dt).requires_grad_(not O.is_buffer(k)) if v.is_floating_point() else v.clone()) for k, v in sd0.items()}
    o, _, _, i2, d2 = O.wt_pse_update(sd, HP, img.to(dt), od.to(dt), img.to(dt), True, eps.to(dt), 3, pb)
    l = O.seg_loss_od(o, od.to(dt)) + i2 + d2
    l.backward()
    res[dt] = {k: sd[k].grad.double() for k in sd if not O.is_buffer(k) and sd[k].grad is not None}
    print(dt, "out err", float((out.detach().cpu().double() - o.detach().double()).abs().max()), "loss", float(loss.detach()), float(l.detach()))
rows = []
for k, p in main.named_parameters():
    if is_prebn_bias(k) or p.grad is None: continue
    g = p.grad.cpu().double(); r32 = res[torch.float32][k]; r64 = res[torch.float64][k]
    n64 = float(r64.norm()) + 1e-30
    rows.append((float((g - r64).norm()) / n64, float((r32 - r64).norm()) / n64, k, n64))
rows.sort(reverse=True)
print("rel L2 err vs fp64 oracle:  HIP      oracle-fp32   param   |g|")
for r in rows[:25]: print("  %.3e  %.3e  %-50s %.3e" % r)
print("median HIP %.3e  median fp32 %.3e" % (np.median([r[0] for r in rows]), np.median([r[1] for r in rows])))

# ---- call B: student update on the same (unstepped) teacher
print("=== shape update")
shape.train(); shape.zero_grad(); main.zero_grad()
sds0 = {k: v.detach().cpu().clone() for k, v in shape.state_dict().items()}
kd, ins_t, ins_off, ins_diag, dom_s = shape.update(main, img.cuda(), od.cuda(), two_stage_inputs=img.cuda(), two_step=True)
(kd + ins_t + dom_s).backward()
res = {}
for dt in (torch.float32, torch.float64):
    mk = lambda s0: {k: (v.detach().clone().to(dt).requires_grad_(not O.is_buffer(k)) if v.is_floating_point() else v.clone()) for k, v in s0.items()}
    sdm, sds = mk(sd0), mk(sds0)
    r = O.shape_update(sds, sdm, HP, img.to(dt), od.to(dt), img.to(dt), True, make_noise(800 + S0, (B, 1, H, H)).to(dt), make_noise(900 + S0, (B, 1, H, H)).to(dt), pb)
    (r[0] + r[1] + r[4]).backward()
    res[dt] = {k: sds[k].grad.double() for k in sds if not O.is_buffer(k) and sds[k].grad is not None}
    print(dt, "kd", float(kd.detach()), float(r[0].detach()), "ins", float(ins_t.detach()), float(r[1].detach()), "dom", float(dom_s.detach()), float(r[4].detach()))
rows = []
for k, p in shape.named_parameters():
    if is_prebn_bias(k) or p.grad is None: continue
    g = p.grad.cpu().double(); r32 = res[torch.float32][k]; r64 = res[torch.float64][k]
    n64 = float(r64.norm()) + 1e-30
    rows.append((float((g - r64).norm()) / n64, float((r32 - r64).norm()) / n64, k, n64))
rows.sort(reverse=True)
for r in rows[:30]: print("  %.3e  %.3e  %-50s %.3e" % r)
print("median HIP %.3e  median fp32 %.3e" % (np.median([r[0] for r in rows]), np.median([r[1] for r in rows])))
