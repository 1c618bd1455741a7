"""CPU restatement of the reference's training input transforms (TEST INFRASTRUCTURE, never on the product path).

Reference: train.py:58-62  Compose([Resize(256), RandomScaleCrop(256), Normalize_tf(), ToTensor()]) with the classes of
custom_transforms.py (Resize :375-391, RandomScaleCrop :330-354, RandomCrop :139-176, Normalize_tf :455-499,
ToTensor :581-599).  The resampling itself lives in a third-party dependency, Pillow (`Image.resize`; the reference's
requirements do not pin a version, this image ships Pillow 12.2.0): its published algorithm (src/libImaging/Resample.c:
precompute_coeffs, normalize_coeffs_8bpc, ImagingResampleHorizontal/Vertical_8bpc; Geometry.c: ImagingScaleAffine for
NEAREST) is restated here in numpy integer arithmetic and pinned bit-exactly against Pillow itself and against the
reference classes (tests/test_transforms_cpu.py, tests/golden/transforms.npz).

The reference draws its random numbers from Python's `random` module; here they are explicit arguments
(`draws = (seed, fw, fh, x1, y1)`), see `draw_like_reference`.
"""
import math
import random

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def _bilinear(x):
    x = abs(x)
    return 1.0 - x if x < 1.0 else 0.0


def _bicubic(x):
    a = -0.5
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


FILTERS = {"bilinear": (_bilinear, 1.0), "bicubic": (_bicubic, 2.0)}


def precompute_coeffs(in_size, out_size, filt):
    """Resample.c precompute_coeffs + normalize_coeffs_8bpc -> (bounds [out,2] int32 (xmin, count), kk [out,ksize] int32)."""
    f, fsupport = FILTERS[filt]
    scale = float(in_size) / out_size
    filterscale = scale if scale >= 1.0 else 1.0
    support = fsupport * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        w = [f((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        for x in range(xmax):
            v = w[x] / ww if ww != 0.0 else w[x]
            p = v * (1 << PRECISION_BITS)
            kk[xx, x] = int(-0.5 + p) if v < 0 else int(0.5 + p)
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def _resample_axis(img, out_size, filt, axis):
    """One 8-bit pass along `axis` (0: vertical, 1: horizontal) of an [H,W,C] uint8 array."""
    a = np.moveaxis(img, axis, 0).astype(np.int64)            # [in, other, C]
    bounds, kk = precompute_coeffs(a.shape[0], out_size, filt)
    out = np.empty((out_size,) + a.shape[1:], np.uint8)
    for xx in range(out_size):
        xmin, n = bounds[xx]
        acc = np.full(a.shape[1:], 1 << (PRECISION_BITS - 1), np.int64)
        for x in range(n):
            acc += a[xmin + x] * int(kk[xx, x])
        out[xx] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return np.moveaxis(out, 0, axis)


def resample_u8(img, out_w, out_h, filt):
    """Pillow's ImagingResample on uint8 data: horizontal pass, then vertical, each rounded to 8 bits."""
    squeeze = img.ndim == 2
    a = img[:, :, None] if squeeze else img
    if a.shape[1] != out_w:
        a = _resample_axis(a, out_w, filt, 1)
    if a.shape[0] != out_h:
        a = _resample_axis(a, out_h, filt, 0)
    return a[:, :, 0] if squeeze else a


def nearest_index(in_size, out_size):
    """Geometry.c ImagingScaleAffine for Image.resize(..., NEAREST): source index of every output position."""
    a0 = float(in_size) / out_size
    xo = a0 * 0.5
    idx = np.zeros(out_size, np.int64)
    for x in range(out_size):
        xin = -1 if xo < 0.0 else int(xo)
        idx[x] = min(max(xin, 0), in_size - 1)
        xo += a0
    return idx


def nearest_u8(img, out_w, out_h):
    return img[nearest_index(img.shape[0], out_h)][:, nearest_index(img.shape[1], out_w)]


def draw_like_reference(rng, w=256, h=256, size=256):
    """The draws RandomScaleCrop/RandomCrop make, in their order (custom_transforms.py:342-346,167-168)."""
    seed = rng.random()
    nw, nh = w, h
    fw = fh = None
    if seed > 0.5:
        fw = rng.uniform(1, 1.5)
        fh = rng.uniform(1, 1.5)
        nw, nh = int(fw * w), int(fh * h)
    if nw == size and nh == size:
        return seed, nw, nh, 0, 0
    x1 = rng.randint(0, nw - size)
    y1 = rng.randint(0, nh - size)
    return seed, nw, nh, x1, y1


def train_transform(img, od, oc, draws, size=256):
    """img [H,W,3] uint8, od/oc [H,W] uint8 -> (image [3,size,size] f32, od [1,size,size] f32, oc [1,size,size] f32)."""
    img = resample_u8(img, size, size, "bicubic")            # Resize: Image.resize default filter = BICUBIC
    od = resample_u8(od, size, size, "bicubic")
    oc = resample_u8(oc, size, size, "bicubic")
    seed, nw, nh, x1, y1 = draws
    if seed > 0.5:
        img = resample_u8(img, nw, nh, "bilinear")
        od = nearest_u8(od, nw, nh)
        oc = nearest_u8(oc, nw, nh)
    img = img[y1:y1 + size, x1:x1 + size]
    od = od[y1:y1 + size, x1:x1 + size]
    image = img.astype(np.float32)
    image /= 127.5
    image -= 1.0
    m_od = (od <= 200).astype(np.float32)                     # Normalize_tf: > 200 -> background
    m_oc = (od <= 50).astype(np.float32)                      # the cup mask is cut from the disc image too (:488-489)
    return image.transpose(2, 0, 1), m_od[None], m_oc[None]
