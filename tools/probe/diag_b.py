"""Call B (student update) gradients at B=6, 32x32: HIP vs fp64 oracle — is the deviation a common scale factor?"""
import sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))      # a diagnostic (uses oracle/ as the CHECKER, like the tests it borrows from; nothing here ships)
sys.path[:0] = [ROOT, os.path.join(ROOT, "wt-pse-code_amd"), os.path.join(ROOT, "tests")]
from oracle import wtpse_cpu as O
from oracle.inputs import make_inputs, make_noise
from test_parity_gpu import build_nets, HP, is_prebn_bias
B, pb, H = 6, 2, 32
img, od, oc = make_inputs(600, B, H, H)
main, shape, _, _ = build_nets(pb)
sd0 = {k: v.detach().cpu().clone() for k, v in main.state_dict().items()}
sds0 = {k: v.detach().cpu().clone() for k, v in shape.state_dict().items()}
main.train(); shape.train(); shape.zero_grad()
terms = sys.argv[1] if len(sys.argv) > 1 else "kd,ins,dom"
kd, ins_t, ins_off, ins_diag, dom_s = shape.update(main, img.cuda(), od.cuda(), two_stage_inputs=img.cuda(), two_step=True)
loss = 0
if "kd" in terms: loss = loss + kd
if "ins" in terms: loss = loss + ins_t
if "dom" in terms: loss = loss + dom_s
loss.backward()
dt = torch.float64
mk = lambda s0: {k: (v.detach().clone().to(dt).requires_grad_(not O.is_buffer(k)) if v.is_floating_point() else v.clone()) for k, v in s0.items()}
sdm, sds = mk(sd0), mk(sds0)
r = O.shape_update(sds, sdm, HP, img.to(dt), od.to(dt), img.to(dt), True, make_noise(800, (B, 1, H, H)).to(dt), make_noise(900, (B, 1, H, H)).to(dt), pb)
l = 0
if "kd" in terms: l = l + r[0]
if "ins" in terms: l = l + r[1]
if "dom" in terms: l = l + r[4]
l.backward()
print("terms", terms, "kd %.9g vs %.9g  ins %.9g vs %.9g  dom %.9g vs %.9g" % (float(kd), float(r[0]), float(ins_t), float(r[1]), float(dom_s), float(r[4])))
rows = []
for k, p in shape.named_parameters():
    if is_prebn_bias(k) or p.grad is None or sds[k].grad is None: continue
    g = p.grad.cpu().double().reshape(-1); g64 = sds[k].grad.double().reshape(-1)
    alpha = float(g @ g64 / (g64 @ g64 + 1e-300))
    rows.append((k, float((g - g64).norm() / g64.norm()), alpha, float((g - alpha * g64).norm() / g64.norm())))
for k, e, a, res in rows[::6]:
    print("%-45s rel err %.3e  alpha %.6f  residual after scaling %.3e" % (k, e, a, res))
