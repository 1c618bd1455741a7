"""TEST INFRASTRUCTURE: a tiny synthetic copy of the reference's on-disk dataset layout (fundus_dataloader.py:41-44,164-179):

    <root>/Domain{1..4}/{train,test}/ROIs/image/<name>.png      RGB fundus crop
    <root>/Domain{1..4}/{train,test}/ROIs/mask/<name>.png       grey levels: 0 = cup, 128 = disc rim, 255 = background

with the filename prefixes the reference infers the dataset from (gd / nd = Drishti-GS, g / n = REFUGE, G / N / S = RIM-ONE,
V = REFUGE-val).  Deterministic in `seed`: the same bytes in the build container (where the fixture is generated from the
reference's own loader, oracle/make_golden_dataset.py) and wherever the tests rebuild it.  The real dataset is not on any box."""
import os

import numpy as np
from PIL import Image

NAMES = {1: ["gdrishtiGS_001", "ndrishtiGS_002", "gdrishtiGS_003"], 2: ["g0001", "n0002", "n0003", "g0004"],
         3: ["G-1-L", "N-2-R", "S-3-L"], 4: ["V0001", "V0002", "V0003"]}
SIZES = [(300, 280), (256, 256), (411, 333), (512, 512)]      # (width, height) of the source crops


def _sample(rs, w, h, rgb_mask):
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    cy, cx = rs.uniform(0.4, 0.6) * h, rs.uniform(0.4, 0.6) * w
    rad = rs.uniform(0.25, 0.4) * min(h, w)
    d2 = (yy - cy) ** 2 + (xx - cx) ** 2
    img = rs.randint(0, 256, (h, w, 3)).astype(np.float32) * 0.25
    img += 140.0 * np.exp(-d2 / (2 * (1.5 * rad) ** 2))[:, :, None] * np.array([1.0, 0.6, 0.3], np.float32)
    mask = np.full((h, w), 255, np.uint8)
    mask[d2 <= rad * rad] = 128
    mask[d2 <= (0.5 * rad) ** 2] = 0
    im = Image.fromarray(np.clip(img, 0, 255).astype(np.uint8), "RGB")
    mk = Image.fromarray(mask, "L")
    if rgb_mask:
        mk = mk.convert("RGB")                       # fundus_dataloader.py:193-194 converts such masks back to 'L'
    return im, mk


def make_tree(root, seed=5):
    """-> {(domain, phase): [basename, ...]} of what was written."""
    rs = np.random.RandomState(seed)
    written = {}
    k = 0
    for dom in (1, 2, 3, 4):
        for phase in ("train", "test"):
            di = os.path.join(root, "Domain%d" % dom, phase, "ROIs", "image")
            dm = os.path.join(root, "Domain%d" % dom, phase, "ROIs", "mask")
            os.makedirs(di, exist_ok=True)
            os.makedirs(dm, exist_ok=True)
            names = NAMES[dom] if phase == "train" else NAMES[dom][:2]
            for n in names:
                w, h = SIZES[k % len(SIZES)]
                im, mk = _sample(rs, w, h, rgb_mask=(k % 5 == 3))
                base = "%s_%s.png" % (n, phase)
                im.save(os.path.join(di, base))
                mk.save(os.path.join(dm, base))
                written.setdefault((dom, phase), []).append(base)
                k += 1
    # Domain5 (round 5): files whose names carry none of the known prefixes — the reference's reader prints "[ERROR:] Unknown dataset!"
    # and stops reading (fundus_dataloader.py:176-178); written last, so every other domain's bytes are what they were
    for phase in ("train",):
        di = os.path.join(root, "Domain5", phase, "ROIs", "image")
        dm = os.path.join(root, "Domain5", phase, "ROIs", "mask")
        os.makedirs(di, exist_ok=True)
        os.makedirs(dm, exist_ok=True)
        for n in ("x0001", "y0002"):
            im, mk = _sample(rs, 256, 256, rgb_mask=False)
            base = "%s_%s.png" % (n, phase)
            im.save(os.path.join(di, base))
            mk.save(os.path.join(dm, base))
            written.setdefault((5, phase), []).append(base)
    return written
