"""Synthetic fundus-shaped batches: there is no dataset on the GPU box.

image  = U(-1,1) noise per pixel (range of Normalize_tf, custom_transforms.py:471-472) with a per-domain channel
         tint so the domain (MMD) term is non-degenerate; rows are domain-major [D0 x pb, D1 x pb, D2 x pb]
         exactly as get_multi_batch stacks them (Trainer.py:45-55)
target_od = filled disc, centre U(0.4,0.6)*size, radius U(0.25,0.4)*size; target_oc = concentric disc of 0.4-0.6x
         that radius; binary {0,1} floats (custom_transforms.py:480-494)
Generated on the host once, outside any timed region, then moved to HBM.
"""
import numpy as np
import torch


def make_batch(B, H, W, device, seed=1, domains=3):
    r = np.random.RandomState(seed)
    pb = max(B // domains, 1)
    img = r.uniform(-1.0, 1.0, size=(B, 3, H, W)).astype(np.float32)
    tint = r.uniform(-0.3, 0.3, size=(domains, 3, 1, 1)).astype(np.float32)
    gain = r.uniform(0.7, 1.0, size=(domains, 3, 1, 1)).astype(np.float32)
    dom = np.minimum(np.arange(B) // pb, domains - 1)
    img = np.clip(img * gain[dom] + tint[dom], -1.0, 1.0)
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    cy = r.uniform(0.4, 0.6, size=B) * H
    cx = r.uniform(0.4, 0.6, size=B) * W
    rad = r.uniform(0.25, 0.4, size=B) * min(H, W)
    rc = rad * r.uniform(0.4, 0.6, size=B)
    d2 = (yy[None] - cy[:, None, None]) ** 2 + (xx[None] - cx[:, None, None]) ** 2
    od = (d2 <= (rad ** 2)[:, None, None]).astype(np.float32)[:, None]
    oc = (d2 <= (rc ** 2)[:, None, None]).astype(np.float32)[:, None]
    to = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
    return to(img), to(od), to(oc)


def default_hparams(full=True):
    """The hparams the hot path reads, at the reference's defaults (hparams_registry.py:71-93)."""
    return {
        "whitening": bool(full), "shape_prior": bool(full), "shape_attention": True,
        "shape_attention_coeffient": 0.3, "cat_shape": False, "margin": 0, "shape_start": 0.5,
        "instance_wt_gm": 1, "domain_wt_gm": 1, "multi-turn": 1,
    }
