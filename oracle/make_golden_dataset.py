"""BUILD CONTAINER ONLY: run the reference's OWN dataset class (fundus_dataloader.FundusSegmentation — it needs no shim: PIL, numpy,
torch and glob only) on the synthetic PNG tree of oracle/fundus_tree.py and record what it builds -> tests/golden/dataset.npz.

Recorded per (splitid, phase, state): the pool keys in order, per pool the file names with each image's / mask's size, mode and a
checksum of its pixels, and — train phase — the (pool, file name) sequence of 12 __getitem__(0) calls under np.random.seed(11).
Numbers and names only; no reference code."""
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path[:0] = [ROOT]
from oracle import ref_import  # noqa: E402
from oracle.fundus_tree import make_tree  # noqa: E402

CASES = [((1,), "train", "train"), ((2,), "train", "train"), ((3,), "train", "train"), ((4,), "test", "prediction"),
         ((1, 2), "train", "train"), ((3,), "test", "train"),
         ((5,), "train", "train")]        # unknown file prefixes only: the reader stops, three of the four empty pools are removed (:58-75)


def checksum(pil):
    a = np.asarray(pil).astype(np.int64)
    return np.array([a.sum(), (a * (1 + (np.arange(a.size).reshape(a.shape) % 251))).sum()], np.int64)


def main():
    sys.path.insert(0, ref_import.REFERENCE_ROOT)
    import fundus_dataloader as DL
    sys.path.remove(ref_import.REFERENCE_ROOT)
    assert os.path.dirname(os.path.abspath(DL.__file__)) == os.path.abspath(ref_import.REFERENCE_ROOT)
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        make_tree(tmp, seed=5)
        for ci, (split, phase, state) in enumerate(CASES):
            ds = DL.FundusSegmentation(base_dir=tmp, phase=phase, splitid=list(split), transform=None, state=state, label="OD")
            keys = list(ds.image_pool.keys())
            out["c%d_meta" % ci] = np.array(["|".join(str(s) for s in split), phase, state, "|".join(keys), str(len(ds))])
            for key in keys:
                names = ds.img_name_pool[key]
                if not names:          # a pool the reference left in place although it is empty (unknown-prefix tree)
                    out["c%d_%s_names" % (ci, key)] = np.array([], dtype="<U1")
                    continue
                out["c%d_%s_names" % (ci, key)] = np.array(names)
                out["c%d_%s_img" % (ci, key)] = np.stack([np.concatenate([np.array(im.size), checksum(im)]) for im in ds.image_pool[key]])
                out["c%d_%s_imgmode" % (ci, key)] = np.array([im.mode for im in ds.image_pool[key]])
                out["c%d_%s_mask" % (ci, key)] = np.stack([np.concatenate([np.array(m.size), checksum(m)]) for m in ds.label_pool[key]])
                out["c%d_%s_maskmode" % (ci, key)] = np.array([m.mode for m in ds.label_pool[key]])
            if phase == "train" and len(ds) > 0:
                np.random.seed(11)
                seq = []
                for _ in range(12):
                    for s in ds[0]:
                        hit = [n for n, im in zip(ds.img_name_pool[keys[s["dc"]]], ds.image_pool[keys[s["dc"]]]) if im is s["image"]]
                        seq.append("%d:%s" % (s["dc"], hit[0]))
                out["c%d_draws" % ci] = np.array(seq)
    dst = os.path.join(ROOT, "tests", "golden", "dataset.npz")
    np.savez_compressed(dst, **out)
    print("wrote", dst, os.path.getsize(dst), "bytes;", len(out), "arrays")


if __name__ == "__main__":
    main()
