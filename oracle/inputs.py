"""Seeded inputs shared by the golden generator and the parity tests (TEST INFRASTRUCTURE).

Fixtures store seeds, not tensors: ``numpy.random.RandomState`` streams are frozen by
numpy's compatibility policy, so the tests regenerate bit-identical inputs.
Shapes follow SURVEY.md §8d: image in [-1, 1] (``Normalize_tf``, custom_transforms.py:471-472),
binary disc masks for OD / OC (custom_transforms.py:480-494), rows domain-major with a
per-domain channel tint so the MMD term is non-degenerate.
"""
import numpy as np
import torch


def make_inputs(seed, B, H, W, domains=3):
    r = np.random.RandomState(seed)
    img = r.uniform(-1.0, 1.0, size=(B, 3, H, W)).astype(np.float32)
    pb = max(B // domains, 1)
    tint = r.uniform(-0.3, 0.3, size=(domains, 3)).astype(np.float32)
    gain = r.uniform(0.7, 1.0, size=(domains, 3)).astype(np.float32)
    for b in range(B):
        d = min(b // pb, domains - 1)
        img[b] = np.clip(img[b] * gain[d][:, None, None] + tint[d][:, None, None], -1.0, 1.0)
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    od = np.zeros((B, 1, H, W), np.float32)
    oc = np.zeros((B, 1, H, W), np.float32)
    for b in range(B):
        cy, cx = r.uniform(0.4, 0.6, size=2) * (H, W)
        rad = r.uniform(0.25, 0.4) * min(H, W)
        d2 = (yy - cy) ** 2 + (xx - cx) ** 2
        od[b, 0] = (d2 <= rad * rad)
        rc = rad * r.uniform(0.4, 0.6)
        oc[b, 0] = (d2 <= rc * rc)
    return torch.from_numpy(img), torch.from_numpy(od), torch.from_numpy(oc)


def make_noise(seed, shape):
    return torch.from_numpy(np.random.RandomState(seed).standard_normal(size=tuple(shape)).astype(np.float32))


def make_feature(seed, shape, white=False):
    """Random 16-channel feature map for the WT-loss fixtures.  `white=True` gives channels that are
    nearly orthonormal under the (HW-1) divisor, so |G_ii - 1| and the off-diagonal terms sit near the
    clamp at 0 and pairwise MMD distances near the 1e-30 floor."""
    r = np.random.RandomState(seed)
    B, C, H, W = shape
    z = r.standard_normal(size=(B, C, H * W)).astype(np.float64)
    if white:
        for b in range(B):
            q, _ = np.linalg.qr(z[b].T)            # [HW, C] orthonormal columns
            z[b] = (q.T * np.sqrt(H * W - 1.0))    # rows have squared norm HW-1  -> G ~= I
        z += 1e-4 * r.standard_normal(size=z.shape)
    return torch.from_numpy(z.reshape(B, C, H, W).astype(np.float32))
