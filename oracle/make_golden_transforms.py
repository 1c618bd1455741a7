#!/usr/bin/env python3
"""tests/golden/transforms.npz: the reference's own training transforms (custom_transforms.py via oracle/ref_import.py) on
small synthetic samples.  Run in the build container only:  python oracle/make_golden_transforms.py

The reference composes Resize(256), RandomScaleCrop(256), Normalize_tf(), ToTensor() (train.py:58-62); the fixture uses
size 64 to stay small — the classes take the size as a parameter."""
import os
import random
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_import  # noqa: E402


def synth_sample(rs, h, w):
    """A fundus-like sample: smooth RGB image, disc mask (0 cup / 128 rim / 255 background) and an unused cup mask."""
    yy, xx = np.mgrid[0:h, 0:w]
    img = np.stack([127 + 100 * np.sin(xx / 7.0 + c) * np.cos(yy / 9.0 - c) + rs.randint(-20, 20, (h, w)) for c in range(3)], -1)
    img = np.clip(img, 0, 255).astype(np.uint8)
    r = np.hypot(yy - h * 0.45, xx - w * 0.55)
    od = np.where(r < min(h, w) * 0.18, 0, np.where(r < min(h, w) * 0.33, 128, 255)).astype(np.uint8)
    oc = rs.randint(0, 256, (h, w)).astype(np.uint8)
    return img, od, oc


def main():
    tr = ref_import.load_transforms()
    from PIL import Image
    size = 64
    rs = np.random.RandomState(5)
    out = {"size": np.int64(size)}
    cases = [(90, 70, 11), (64, 64, 2), (120, 150, 5), (50, 48, 15), (64, 64, 23), (200, 180, 22), (77, 131, 42)]
    for i, (h, w, seed) in enumerate(cases):
        img, od, oc = synth_sample(rs, h, w)
        sample = {"image": Image.fromarray(img), "label_od": Image.fromarray(od), "label_oc": Image.fromarray(oc), "dc": 0}
        random.seed(seed)
        for t in (tr.Resize(size), tr.RandomScaleCrop(size), tr.Normalize_tf()):
            sample = t(sample)
        image = np.array(sample["image"]).astype(np.float32).transpose(2, 0, 1)        # ToTensor (:581-599)
        m_od = np.array(sample["label_od"]).astype(np.uint8).transpose(2, 0, 1).astype(np.float32)
        m_oc = np.array(sample["label_oc"]).astype(np.uint8).transpose(2, 0, 1).astype(np.float32)
        out["in%d_img" % i], out["in%d_od" % i], out["in%d_oc" % i] = img, od, oc
        out["seed%d" % i] = np.int64(seed)
        out["out%d_img" % i], out["out%d_od" % i], out["out%d_oc" % i] = image, m_od, m_oc
    out["n"] = np.int64(len(cases))
    dst = os.path.join(ROOT, "tests", "golden", "transforms.npz")
    np.savez_compressed(dst, **out)
    print("wrote", dst, os.path.getsize(dst), "bytes")


if __name__ == "__main__":
    main()
