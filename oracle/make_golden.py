#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE ITSELF on CPU (build container only).

    python oracle/make_golden.py            # writes tests/golden/

TEST INFRASTRUCTURE.  Imports /root/reference through oracle/ref_import.py, fills weights with
oracle/filler.py, feeds seeded inputs from oracle/inputs.py, injects sampling noise, and records
outputs (+ per-tensor checksums of gradients / updated parameters).  No reference code or weights
are written: fixtures hold seeds, shapes and expected outputs only.

The ~30 lines of `ref_iteration` restate the body of the hot loop (Trainer.py:766-914) around the
reference's own modules, because Trainer.py itself cannot be imported here (tensorboardX, medpy,
skimage, cv2 are not installed — SURVEY.md §8c).
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from oracle import ref_import  # noqa: E402
from oracle.filler import fill_state_dict  # noqa: E402
from oracle.inputs import make_inputs, make_noise, make_feature  # noqa: E402
from oracle.wtpse_cpu import checksum, DEFAULT_HPARAMS  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
SEED_W = 1234


def np_(t):
    return t.detach().cpu().numpy()


def grads_of(module, skip_none=True):
    out = {}
    for n, p in module.named_parameters():
        if p.grad is None:
            continue
        out[n] = checksum(p.grad)
    return out


def pack(prefix, d):
    return {prefix + k: v for k, v in d.items()}


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    os.makedirs(OUT, exist_ok=True)
    hreg, alg, shp = ref_import.load()
    hp = hreg.default_hparams("WT_PSE", "fundus")
    for k, v in DEFAULT_HPARAMS.items():
        assert hp[k] == v, (k, hp[k], v)

    def new_main(pb, two_step=False, hparams=None):
        m = alg.WT_PSE(n_channels=3, n_classes=1, hparams=hparams or hp, device="cpu", two_step=two_step,
                       per_domain_batch=pb, source_domain_num=3)
        fill_state_dict(m, SEED_W + (7 if two_step else 0))
        return m

    def new_shape(pb, oc=False):
        s = shp.ShapeVariationalDist_x(hp, "cpu", n_classes=1, number_source_domain=3, batch_size=pb)
        fill_state_dict(s, SEED_W + (11 if oc else 3))
        return s

    # ------------------------------------------------------------------ a-4 / a-5: WT loss + MMD
    fx = {}
    cases = []
    for ci, (B, pb, H, white, margin) in enumerate([
            (3, 1, 8, False, 0), (6, 2, 8, False, 0), (6, 2, 16, False, 0), (6, 2, 8, True, 0),
            (6, 2, 8, True, 0.05), (7, 2, 8, False, 0.01), (12, 4, 16, True, 0)]):
        hpm = dict(hp); hpm["margin"] = margin
        m = alg.WT_PSE(3, 1, hpm, "cpu", two_step=False, per_domain_batch=pb, source_domain_num=3)
        s = shp.ShapeVariationalDist_x(hpm, "cpu", n_classes=1, number_source_domain=3, batch_size=pb)
        z = make_feature(100 + ci, (B, 16, H, H), white).requires_grad_(True)
        ins, dom = m.compute_whitening_loss(z)
        off, dg, dom2 = s.compute_whitening_loss(z)
        (off + dg + dom2).backward()
        # also the raw upper-triangle vector and the pairwise MMD on it (a-5 alone)
        with torch.no_grad():
            f = z.view(B, 16, -1)
            g = torch.bmm(f, f.transpose(1, 2)).div(H * H - 1) + 1e-5 * torch.eye(16)
            iu = torch.triu_indices(16, 16, 1)
            v = (g * torch.ones(16, 16).triu(1))[:, iu[0], iu[1]]
        vv = v.clone().requires_grad_(True)
        d3 = m.mmd_operator.forward(vv)
        d3.backward()
        cases.append((B, pb, H, int(white), margin, 100 + ci))
        fx.update(pack(f"c{ci}_", dict(ins=np_(ins), dom=np_(dom), off=np_(off), diag=np_(dg), dom2=np_(dom2),
                                       gram=np_(g), v=np_(v), mmd=np_(d3), dmmd_dv=np_(vv.grad),
                                       dz=np_(z.grad) if H <= 8 else checksum(z.grad), dz_cs=checksum(z.grad))))
    fx["cases"] = np.array(cases, dtype=np.float64)
    np.savez_compressed(os.path.join(OUT, "wtloss.npz"), **fx)
    print("wtloss.npz", len(fx))

    # ------------------------------------------------------------------ a-1 / a-2 / a-3 / a-10 blocks
    fx = {}
    B, H = 4, 16
    blocks = [
        ("convd_first", lambda: alg.ConvD(3, 16, "bn", first=True), (B, 3, H, H), None),
        ("convd", lambda: alg.ConvD(16, 32, "bn"), (B, 16, H, H), None),
        ("convu_first", lambda: alg.ConvU(64, "bn", first=True), (B, 64, H // 2, H // 2), (B, 32, H, H)),
        ("convu", lambda: alg.ConvU(32, "bn"), (B, 64, H // 2, H // 2), (B, 16, H, H)),
    ]
    for bi, (name, ctor, xs, ps) in enumerate(blocks):
        mod = ctor()
        fill_state_dict(mod, SEED_W + 20 + bi)
        mod.train()
        x = make_noise(300 + bi, xs).requires_grad_(True)
        args = [x]
        if ps is not None:
            prev = make_noise(400 + bi, ps).requires_grad_(True)
            args.append(prev)
        y = mod(*args)
        wgt = make_noise(500 + bi, y.shape)
        (y * wgt).sum().backward()
        d = dict(y=np_(y), dx=np_(x.grad))
        if ps is not None:
            d["dprev"] = np_(prev.grad)
        for n, p in mod.named_parameters():
            d["g." + n] = np_(p.grad)
        for n, b in mod.named_buffers():
            d["buf." + n] = np_(b)
        mod.eval()
        with torch.no_grad():
            d["y_eval"] = np_(mod(*[a.detach() for a in args]))
        fx.update(pack(name + ".", d))
    wt = alg.DeepWT(3, 16, whitening=True)
    fill_state_dict(wt, SEED_W + 30)
    x = make_noise(310, (B, 3, H, H)).requires_grad_(True)
    zs = wt(x)
    sum((z * make_noise(510 + i, z.shape)).sum() for i, z in enumerate(zs)).backward()
    d = dict(z1=np_(zs[0]), z2=np_(zs[1]), z3=np_(zs[2]), dx=np_(x.grad))
    for n, p in wt.named_parameters():
        d["g." + n] = np_(p.grad)
    fx.update(pack("deepwt.", d))
    att = alg.attention_layer(1, 1)
    fill_state_dict(att, SEED_W + 31)
    a, pre = att(make_noise(311, (B, 1, H, H)))
    fx.update(pack("attention.", dict(sig=np_(a), pre=np_(pre))))
    np.savez_compressed(os.path.join(OUT, "blocks.npz"), **fx)
    print("blocks.npz", len(fx))

    # ------------------------------------------------------------------ a-7 / a-8 / a-9 whole-network calls
    fx = {}
    cases = []
    for ci, (B, pb, H) in enumerate([(3, 1, 32), (6, 2, 32), (7, 2, 32)]):
        img, od, oc = make_inputs(600 + ci, B, H, H)
        main, shape = new_main(pb), new_shape(pb)
        main_oc, shape_oc = new_main(pb, two_step=True), new_shape(pb, oc=True)
        # predict (eval)
        for mm in (main, shape, main_oc, shape_oc):
            mm.eval()
        with torch.no_grad():
            logit, att_pre = main.predict(shape, img)
            roi = (img + 1) * (torch.sigmoid(logit) > 0.75).float() - 1
            logit2, att2 = main_oc.predict(shape_oc, torch.stack((roi, roi), 0))
        d = dict(pred_logit=np_(logit), pred_att=np_(att_pre), pred2_logit=np_(logit2), pred2_att=np_(att2))
        # update (train), call A
        main.train(); shape.train()
        main.zero_grad()
        eps_a = make_noise(700 + ci, (B, 1, H, H))
        with ref_import.replay_noise([eps_a]):
            out, m1, m2, ins, dom = main.update(img, od, two_stage_inputs=img, sp_mask=od, two_step=True)
        loss = F.binary_cross_entropy(torch.sigmoid(out), od) + ins + dom
        loss.backward()
        d.update(upd_out=np_(out), upd_mask=np_(m1), upd_ins=np_(ins), upd_dom=np_(dom), upd_loss=np_(loss))
        d.update(pack("upd_g.", grads_of(main)))
        d.update({"upd_buf." + n: checksum(b.float()) for n, b in main.named_buffers()})
        # shape update (train), call B — on the same (not stepped) teacher
        shape.zero_grad(); main.zero_grad()
        eps_t, eps_s = make_noise(800 + ci, (B, 1, H, H)), make_noise(900 + ci, (B, 1, H, H))
        with ref_import.replay_noise([eps_t, eps_s]):
            kd, ins_t, ins_ij, ins_ii, dom_s = shape.update(main, img, od, two_stage_inputs=img, two_step=True)
        (kd + ins_t + dom_s).backward()
        d.update(shp_kd=np_(kd), shp_ins_total=np_(ins_t), shp_ins_off=np_(ins_ij), shp_ins_diag=np_(ins_ii),
                 shp_dom=np_(dom_s))
        d.update(pack("shp_g.", grads_of(shape)))
        cases.append((B, pb, H, 600 + ci, 700 + ci, 800 + ci, 900 + ci))
        fx.update(pack(f"c{ci}_", d))
    fx["cases"] = np.array(cases, dtype=np.float64)
    # seg-net only configuration (BASELINE.json configs[1]): whitening=False, shape_prior=False
    hp0 = dict(hp); hp0["whitening"] = False; hp0["shape_prior"] = False
    m0 = new_main(2, hparams=hp0)
    img, od, oc = make_inputs(650, 6, 32, 32)
    m0.train()
    out0 = m0.update(img, od, two_stage_inputs=img, two_step=True)
    F.binary_cross_entropy(torch.sigmoid(out0[0]), od).backward()
    fx.update(pack("segonly_", dict(out=np_(out0[0]))))
    fx.update(pack("segonly_g.", grads_of(m0)))
    m0.eval()
    with torch.no_grad():
        fx["segonly_pred"] = np_(m0.predict(None, img)[0])
    np.savez_compressed(os.path.join(OUT, "network.npz"), **fx)
    print("network.npz", len(fx))

    # ------------------------------------------------------------------ a-11: full A-D iterations
    fx = {}
    B, pb, H, iters = 6, 2, 32, 3
    nets = [new_main(pb), new_shape(pb), new_main(pb, two_step=True), new_shape(pb, oc=True)]
    opts = [torch.optim.Adam(n.parameters(), lr=5e-4, betas=(0.9, 0.99)) for n in nets]
    bce = torch.nn.BCELoss()
    losses = []
    for it in range(iters):
        img, od, oc = make_inputs(1000 + it, B, H, H)
        nz = {k: make_noise(1100 + 10 * it + j, (B, 1, H, H)) for j, k in enumerate(["a", "b_t", "b_s", "c", "d_t", "d_s"])}
        losses.append(ref_iteration(nets, opts, hp, bce, img, od, oc, nz))
    keys = sorted(losses[0])
    fx["loss_keys"] = np.array(keys)
    fx["losses"] = np.array([[l[k] for k in keys] for l in losses], dtype=np.float64)
    fx["meta"] = np.array([B, pb, H, iters, 1000, 1100], dtype=np.float64)
    for tag, n in zip(["od", "shape_od", "oc", "shape_oc"], nets):
        for k, v in n.state_dict().items():
            fx[f"{tag}.{k}"] = checksum(v.float())
    np.savez_compressed(os.path.join(OUT, "iteration.npz"), **fx)
    print("iteration.npz", len(fx))

    # ------------------------------------------------------------------ Dice (metrics.py:68-97), via the reference's metrics.py
    sys.path.insert(0, ref_import.REFERENCE_ROOT)
    import metrics as ref_metrics
    sys.path.remove(ref_import.REFERENCE_ROOT)
    r = np.random.RandomState(5)
    seg = r.uniform(size=(5, 24, 24)) > 0.6
    gt = r.uniform(size=(5, 24, 24)) > 0.5
    seg[4] = False; gt[4] = False
    np.savez_compressed(os.path.join(OUT, "dice.npz"), seg=seg, gt=gt,
                        dice=np.array([ref_metrics.dice_coefficient_numpy(s, g) for s, g in zip(seg, gt)]))
    print("dice.npz")


def ref_iteration(nets, opts, hp, bce, image, target_od, target_oc, nz):
    """Body of the hot loop around the reference's modules — Trainer.py:766-914."""
    model, model_shape, model_oc, model_shape_oc = nets
    optim, optim_shape, optim_oc, optim_shape_oc = opts
    for n in nets:
        n.train()
    image = image.clone()
    res = {}
    optim.zero_grad(); model.zero_grad()
    with ref_import.replay_noise([nz["a"]]):
        output, _, _, ins, dom = model.update(image, target_od, two_stage_inputs=image, sp_mask=target_od, two_step=True)
    loss_seg = bce(torch.sigmoid(output), target_od)
    loss_main = loss_seg + hp["instance_wt_gm"] * ins + hp["domain_wt_gm"] * dom
    loss_main.backward(); optim.step()
    res.update(seg_od=loss_seg.item(), ins_od=ins.item(), dom_od=dom.item(), main_od=loss_main.item())
    optim_shape.zero_grad(); model_shape.zero_grad()
    with ref_import.replay_noise([nz["b_t"], nz["b_s"]]):
        kd, ins_t, ins_ij, ins_ii, dom_s = model_shape.update(model, image, target_od, two_stage_inputs=image, two_step=True)
    loss_shape = kd + hp["instance_wt_gm"] * ins_t + hp["domain_wt_gm"] * dom_s
    loss_shape.backward(); optim_shape.step()
    res.update(kd_od=kd.item(), ins_shape_od=ins_t.item(), ins_ij_od=ins_ij.item(), ins_ii_od=ins_ii.item(),
               dom_shape_od=dom_s.item(), shape_od=loss_shape.item())
    od_pred = (torch.sigmoid(output) > 0.75).float().detach().float()
    optim_oc.zero_grad(); model_oc.zero_grad()
    image += 1
    image_roi = image * od_pred
    image_roi -= 1
    with ref_import.replay_noise([nz["c"]]):
        output_oc, _, _, ins_c, dom_c = model_oc.update(image_roi, target_oc, two_stage_inputs=image_roi, two_step=True)
    pw = torch.sum(od_pred) / torch.sum(od_pred * target_oc)
    if torch.isinf(pw) or torch.isnan(pw):
        pw = torch.tensor(1.)
    loss_seg_oc = F.binary_cross_entropy_with_logits(output_oc * od_pred, target_oc, pos_weight=pw)
    loss_main_oc = loss_seg_oc + hp["instance_wt_gm"] * ins_c + hp["domain_wt_gm"] * dom_c
    loss_main_oc.backward(); optim_oc.step()
    res.update(seg_oc=loss_seg_oc.item(), ins_oc=ins_c.item(), dom_oc=dom_c.item(), main_oc=loss_main_oc.item())
    optim_shape_oc.zero_grad(); model_shape_oc.zero_grad()
    with ref_import.replay_noise([nz["d_t"], nz["d_s"]]):
        kd2, ins_t2, _, _, dom_s2 = model_shape_oc.update(model_oc, image_roi, target_oc, two_stage_inputs=image_roi, two_step=True)
    loss_shape_oc = kd2 + hp["instance_wt_gm"] * ins_t2 + hp["domain_wt_gm"] * dom_s2
    loss_shape_oc.backward(); optim_shape_oc.step()
    res.update(kd_oc=kd2.item(), ins_shape_oc=ins_t2.item(), dom_shape_oc=dom_s2.item(), shape_oc=loss_shape_oc.item())
    return res


if __name__ == "__main__":
    if not ref_import.available():
        sys.exit("reference not present at %s — goldens can only be generated in the build container" % ref_import.REFERENCE_ROOT)
    main()
