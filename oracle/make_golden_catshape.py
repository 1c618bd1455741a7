#!/usr/bin/env python3
"""Generate tests/golden/catshape.npz by running the REFERENCE ITSELF on CPU (build container only) with
hparams['cat_shape'] = True (algorithms.py:1192,1253,1348: `outc` over cat(fuse_embedding, z_posterior)).

    python oracle/make_golden_catshape.py

TEST INFRASTRUCTURE, same rules as oracle/make_golden.py: the fixture holds seeds, shapes and expected outputs only.
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
from oracle.inputs import make_inputs, make_noise  # noqa: E402
from oracle.wtpse_cpu import checksum  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
SEED_W = 1234


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    hreg, alg, shp = ref_import.load()
    hp = dict(hreg.default_hparams("WT_PSE", "fundus"))
    hp["cat_shape"] = True
    fx, cases = {}, []
    for ci, (B, pb, H) in enumerate([(3, 1, 32), (6, 2, 32)]):
        img, od, _ = make_inputs(1600 + ci, B, H, H)
        main_ = alg.WT_PSE(n_channels=3, n_classes=1, hparams=hp, device="cpu", two_step=False, per_domain_batch=pb,
                           source_domain_num=3)
        fill_state_dict(main_, SEED_W + 40)
        shape = shp.ShapeVariationalDist_x(hp, "cpu", n_classes=1, number_source_domain=3, batch_size=pb)
        fill_state_dict(shape, SEED_W + 43)
        assert tuple(main_.outc[0].weight.shape) == (1, 9, 1, 1)
        main_.eval(); shape.eval()
        with torch.no_grad():
            logit, att_pre = main_.predict(shape, img)
        d = dict(pred_logit=logit.numpy(), pred_att=att_pre.numpy())
        main_.train()
        main_.zero_grad()
        eps = make_noise(1700 + ci, (B, 1, H, H))
        with ref_import.replay_noise([eps]):
            out, m1, _, ins, dom = main_.update(img, od, two_stage_inputs=img, sp_mask=od, two_step=True)
        loss = F.binary_cross_entropy(torch.sigmoid(out), od) + ins + dom
        loss.backward()
        d.update(upd_out=out.detach().numpy(), upd_mask=m1.numpy(), upd_ins=ins.detach().numpy(), upd_dom=dom.detach().numpy(),
                 upd_loss=loss.detach().numpy())
        for n, p in main_.named_parameters():
            if p.grad is not None:
                d["upd_g." + n] = checksum(p.grad)
        d["upd_g_full.outc.0.weight"] = main_.outc[0].weight.grad.numpy()
        cases.append((B, pb, H, 1600 + ci, 1700 + ci))
        fx.update({f"c{ci}_" + k: v for k, v in d.items()})
    fx["cases"] = np.array(cases, dtype=np.float64)
    np.savez_compressed(os.path.join(OUT, "catshape.npz"), **fx)
    print("catshape.npz", len(fx))


if __name__ == "__main__":
    main()
