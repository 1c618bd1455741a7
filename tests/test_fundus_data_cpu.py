"""wtpse_hip/fundus_data.py (the on-disk front of the input pipeline) against tests/golden/dataset.npz — what the reference's OWN
FundusSegmentation built from the same synthetic PNG tree (oracle/make_golden_dataset.py, oracle/fundus_tree.py): pool keys and
order, file -> dataset assignment, decoded sizes / modes / pixels (checksums), and the random index sequence of the train phase."""
import os

import numpy as np
import pytest

from oracle.fundus_tree import make_tree
from oracle.make_golden_dataset import CASES, checksum

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "dataset.npz")


@pytest.fixture(scope="module")
def tree(tmp_path_factory):
    root = str(tmp_path_factory.mktemp("fundus"))
    make_tree(root, seed=5)
    return root


@pytest.mark.parametrize("ci", range(len(CASES)))
def test_tree_loader_matches_reference(tree, ci):
    from PIL import Image
    from wtpse_hip.fundus_data import FundusTree
    g = np.load(GOLDEN)
    split, phase, state = CASES[ci]
    meta = [str(v) for v in g["c%d_meta" % ci]]
    assert meta[:3] == ["|".join(str(s) for s in split), phase, state]
    ds = FundusTree(tree, phase=phase, splitid=split, state=state)
    assert ds.keys() == [k for k in meta[3].split("|") if k]
    assert len(ds) == int(meta[4])
    for key in ds.keys():
        imgs, masks, names = ds.pools[key]
        ref_names = [str(n) for n in g["c%d_%s_names" % (ci, key)]]
        if not ref_names:
            assert not names
            continue
        assert sorted(names) == sorted(ref_names)                       # glob order is the file system's: compare per file
        order = [ref_names.index(n) for n in names]
        for j, r in enumerate(order):
            assert imgs[j].mode == str(g["c%d_%s_imgmode" % (ci, key)][r]) and masks[j].mode == str(g["c%d_%s_maskmode" % (ci, key)][r])
            assert np.array_equal(np.concatenate([np.array(imgs[j].size), checksum(imgs[j])]), g["c%d_%s_img" % (ci, key)][r]), names[j]
            assert np.array_equal(np.concatenate([np.array(masks[j].size), checksum(masks[j])]), g["c%d_%s_mask" % (ci, key)][r]), names[j]
    if phase == "train" and len(ds) > 0:        # (case 6, unknown prefixes only: one EMPTY pool is left, nothing can be drawn)
        # the same np.random stream draws the same pool positions; positions index the pools in glob order on both sides, so the
        # FILE drawn agrees when the two listings agree — which they do on one file system; compare positions through the names
        np.random.seed(11)
        seq = []
        for _ in range(12):
            for img, mask, dc, name in ds.get(0):
                seq.append((dc, name))
        ref = [str(s).split(":", 1) for s in g["c%d_draws" % ci]]
        assert [dc for dc, _ in seq] == [int(d) for d, _ in ref]
        pos = lambda dc, name: ds.pools[ds.keys()[dc]][2].index(name)
        ref_pos = [[str(n) for n in g["c%d_%s_names" % (ci, ds.keys()[int(d)])]].index(n) for d, n in ref]
        assert [pos(dc, n) for dc, n in seq] == ref_pos


def test_multi_batch_layout(tree):
    """Trainer.get_multi_batch: domain-major, per_domain samples from each single-domain dataset, decoded 256 x 256 uint8."""
    from wtpse_hip.fundus_data import FundusTree, multi_batch, dataset_of
    sets = [FundusTree(tree, "train", (i,)) for i in (1, 2, 4)]
    np.random.seed(3)
    images, masks = multi_batch(sets, 2)
    assert len(images) == len(masks) == 6
    assert all(im.shape == (256, 256, 3) and im.dtype == np.uint8 for im in images)
    assert all(m.shape == (256, 256) and m.dtype == np.uint8 for m in masks)
    assert [dataset_of(n) for n in ("gd1.png", "nd2.png", "g3.png", "n4.png", "G5.png", "N6.png", "S7.png", "V8.png", "x.png")] == \
        ["DGS", "DGS", "REF", "REF", "RIM", "RIM", "RIM", "REF_val", None]
