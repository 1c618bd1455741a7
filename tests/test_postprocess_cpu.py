"""validate.largest_fillhole / postprocess (the product's scipy.ndimage implementation of utils.py:267-329) against the oracle's
independent flood-fill restatement (oracle/postprocess_cpu.py; skimage is absent: see its header for what parity is pinned to),
on seeded random masks with ties, diagonal contacts, holes, holes touching the border, empty and full images."""
import numpy as np
import pytest

from oracle import postprocess_cpu as P


def _masks():
    rng = np.random.default_rng(7)
    out = [np.zeros((9, 13), np.uint8), np.ones((7, 5), np.uint8)]
    for h, w, p in ((16, 16, 0.3), (24, 17, 0.45), (32, 32, 0.55), (40, 28, 0.62), (11, 50, 0.5), (64, 64, 0.58)):
        for _ in range(6):
            out.append((rng.random((h, w)) < p).astype(np.uint8))
    # two blobs of EQUAL area (np.argmax takes the first in raster order), a ring with a hole, a ring open at the border
    m = np.zeros((20, 20), np.uint8)
    m[2:5, 12:15] = 1
    m[10:13, 3:6] = 1
    out.append(m)
    m = np.zeros((20, 20), np.uint8)
    m[4:12, 4:12] = 1
    m[6:10, 6:10] = 0
    m[7:9, 7:9] = 1           # an island inside the hole: a separate (smaller) component, removed, then the hole fills
    out.append(m)
    m = np.zeros((12, 12), np.uint8)
    m[0:6, 0:6] = 1
    m[0:4, 2:4] = 0           # a notch open to the border: NOT a hole
    out.append(m)
    # blobs that touch only diagonally: one component under 8-connectivity
    m = np.zeros((10, 10), np.uint8)
    m[1:4, 1:4] = 1
    m[4:7, 4:7] = 1
    m[8, 0] = 1
    out.append(m)
    return out


@pytest.mark.parametrize("i", range(len(_masks())))
def test_largest_fillhole_matches_oracle(i):
    from wtpse_hip import validate as V
    m = _masks()[i]
    want = P.get_largest_fillhole(m)
    got = V.largest_fillhole(m)
    assert got.shape == want.shape and np.array_equal(np.asarray(got, bool), want), i
    assert m.dtype == np.uint8 and np.array_equal(m, _masks()[i])          # the input is not modified


def test_label8_numbering_and_ties():
    lab, areas = P.label8(_masks()[-4])        # two 3x3 blobs: the upper-right one is met first in raster order
    assert areas == [9, 9] and lab[2, 12] == 1 and lab[10, 3] == 2
    keep = P.get_largest_fillhole(_masks()[-4])
    assert keep[2:5, 12:15].all() and not keep[10:13, 3:6].any()


def test_postprocessing_threshold_and_dice():
    import torch
    from wtpse_hip import validate as V
    from oracle import wtpse_cpu as O
    g = torch.Generator().manual_seed(5)
    logits = torch.randn(1, 48, 40, generator=g) * 3.0
    logits[0, 10:30, 8:30] += 4.0
    want = P.postprocessing(logits)
    got = V.postprocess(logits)
    assert got.dtype == np.uint8 and np.array_equal(got, want)
    gt = np.zeros((48, 40), np.uint8)
    gt[12:28, 10:28] = 1
    assert V.dice(got[0], gt) == O.dice_coefficient(want[0], gt)
