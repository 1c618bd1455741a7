"""The on-disk front of the training input pipeline: counterpart of the reference's FundusSegmentation dataset
(fundus_dataloader.py:16-202) and of Trainer.get_batch / get_multi_batch (Trainer.py:29-55) — SURVEY.md §8f row 3.

    Domain{id}/{phase}/ROIs/image/*.png, mask = the same path with 'image' -> 'mask'            (:41-44)
    dataset inferred from the file name: gd / nd -> DGS, g / n -> REF, G / N / S -> RIM, V -> REF_val   (:164-179)
    images decoded once into memory, RGB, LANCZOS-resized to 256 x 256; masks 'L', resized to 256 x 256
    (Image.resize's default filter) unless state == 'prediction'                                  (:180-199)
    train phase: __getitem__ ignores its index and draws one with np.random.choice per pool       (:86-99)

Host side and Python as in the reference (PNG decoding is PIL's on both sides); what it hands on are DECODED uint8 arrays, which
`DeviceInputPipeline` (input_pipeline.py) turns into the fp32 batch on the GPU.  Kept quirks: the Domain-4 centre crop is dead
code in the reference (`self.splitid[0] == '4'` compares an int with a str, :180) and is not performed; empty pools are removed the
reference's way — the first empty one, three times (:58-75) — so `dc` (the domain code) is the index among the pools that are left:
0 for every single-domain dataset train.py builds; a file with an unknown prefix prints the reference's error line and STOPS the
reading (:176-178).  One spelling differs without a difference: the reference tests `_target.mode is 'RGB'` (:193) — an identity
test that holds in CPython because PIL's mode strings and the literal are the same interned constant — and this file writes `==`;
the fixture's RGB masks (every fifth sample of oracle/fundus_tree.py) come out 'L' on both sides."""
import os
from glob import glob

import numpy as np
from PIL import Image

FLAGS = (("DGS", ("gd", "nd"), 2), ("REF", ("g", "n"), 1), ("RIM", ("G", "N", "S"), 1), ("REF_val", ("V",), 1))


def dataset_of(basename):
    """fundus_dataloader.py:166-178 (order matters: 'gd' / 'nd' are tested before 'g' / 'n')."""
    for key, prefixes, n in FLAGS:
        if basename[0:n] in prefixes:
            return key
    return None


class FundusTree:
    def __init__(self, base_dir, phase="train", splitid=(1,), state="train", size=256):
        self.phase, self.state, self.splitid, self.size = phase, state, list(splitid), int(size)
        self.image_list = []
        for i in self.splitid:
            image_dir = os.path.join(base_dir, "Domain" + str(i), phase, "ROIs/image/")
            for image_path in glob(image_dir + "*.png"):
                self.image_list.append({"image": image_path, "label": image_path.replace("image", "mask")})
        pools = {k: ([], [], []) for k, _, _ in FLAGS}
        for item in self.image_list:
            base = os.path.basename(item["image"])
            key = dataset_of(base)
            if key is None:
                # as the reference: an error line, and reading STOPS — the pools keep what was read so far (:176-178)
                print("[ERROR:] Unknown dataset!")
                break
            img = Image.open(item["image"]).convert("RGB").resize((self.size, self.size), Image.LANCZOS)
            target = Image.open(item["label"])
            if target.mode == "RGB":
                target = target.convert("L")
            if state != "prediction":
                target = target.resize((self.size, self.size))
            pools[key][0].append(img)
            pools[key][1].append(target)
            pools[key][2].append(item["image"].split("/")[-1])
        # the reference deletes the FIRST empty pool it meets, three times over (:58-75): with all four pools empty one stays
        # (then __len__ is 0 and drawing from it raises, there as here)
        for _ in range(3):
            for k in list(pools):
                if len(pools[k][0]) < 1:
                    del pools[k]
                    break
        self.pools = pools

    def __len__(self):
        return max((len(v[0]) for v in self.pools.values()), default=-1)

    def keys(self):
        return list(self.pools.keys())

    def get(self, index=0, rng=np.random):
        """-> [(image [H,W,3] uint8, mask [H,W] uint8, domain code, file name)] — one entry per pool, as __getitem__ returns one
        sample per pool.  Train phase: `index` is ignored and drawn with rng.choice(len(pool), 1)[0] per pool (:90)."""
        out = []
        for dc, (key, (imgs, masks, names)) in enumerate(self.pools.items()):
            i = int(rng.choice(len(imgs), 1)[0]) if self.phase != "test" else index
            out.append((np.array(imgs[i]), np.array(masks[i]), dc, names[i]))       # writable copies of the decoded pixels
        return out


def multi_batch(datasets, per_domain, rng=np.random):
    """Trainer.get_batch / get_multi_batch (Trainer.py:29-55): for every dataset (one per source domain), `per_domain` times
    `dataset[0]` (a fresh random index each) and its FIRST pool's sample.  -> (images, masks): lists of decoded arrays, domain-major,
    ready for DeviceInputPipeline (which applies Resize / RandomScaleCrop / Normalize_tf / ToTensor on the GPU)."""
    images, masks = [], []
    for ds in datasets:
        for _ in range(per_domain):
            img, mask, _, _ = ds.get(0, rng)[0]
            images.append(img)
            masks.append(mask)
    return images, masks
