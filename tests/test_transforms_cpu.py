"""CPU side of the input-pipeline row: the oracle restatement of the reference's training transforms
against (1) fixtures generated from the reference's own custom_transforms classes (oracle/make_golden_transforms.py),
(2) Pillow itself, the third-party engine behind them; and the product's host-side table builder against the oracle."""
import os
import random
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "wt-pse-code_amd")]
from oracle import transforms_cpu as T  # noqa: E402


def test_oracle_vs_reference_fixtures(golden_dir):
    g = np.load(os.path.join(golden_dir, "transforms.npz"))
    size = int(g["size"])
    scaled = 0
    for i in range(int(g["n"])):
        draws = T.draw_like_reference(random.Random(int(g["seed%d" % i])), size, size, size)
        scaled += draws[0] > 0.5
        im, od, oc = T.train_transform(g["in%d_img" % i], g["in%d_od" % i], g["in%d_oc" % i], draws, size)
        assert np.array_equal(im, g["out%d_img" % i])      # bit-exact, floats included
        assert np.array_equal(od, g["out%d_od" % i])
        assert np.array_equal(oc, g["out%d_oc" % i])
    assert scaled >= 4                                     # the fixtures exercise the random up-scale branch


@pytest.mark.parametrize("case", [(40, 37, 16, 16, "bicubic"), (33, 50, 64, 64, "bilinear"), (64, 64, 80, 91, "bilinear"),
                                  (100, 80, 32, 32, "bicubic"), (16, 16, 24, 20, "bilinear"), (3, 5, 9, 2, "bicubic"),
                                  (200, 160, 64, 64, "bicubic")])
def test_oracle_vs_pillow(case):
    Image = pytest.importorskip("PIL.Image")
    h, w, oh, ow, f = case
    rs = np.random.RandomState(h * 1000 + w)
    a = rs.randint(0, 256, (h, w, 3)).astype(np.uint8)
    flt = Image.BICUBIC if f == "bicubic" else Image.BILINEAR
    assert np.array_equal(T.resample_u8(a, ow, oh, f), np.array(Image.fromarray(a).resize((ow, oh), flt)))
    m = rs.randint(0, 256, (h, w)).astype(np.uint8)
    assert np.array_equal(T.nearest_u8(m, ow, oh), np.array(Image.fromarray(m).resize((ow, oh), Image.NEAREST)))
    assert np.array_equal(T.resample_u8(m, ow, oh, "bicubic"), np.array(Image.fromarray(m).resize((ow, oh))))   # default filter


@pytest.mark.parametrize("case", [(800, 256, "bicubic"), (90, 64, "bicubic"), (48, 64, "bicubic"), (64, 94, "bilinear"),
                                  (256, 383, "bilinear"), (256, 256, "bilinear"), (257, 256, "bicubic")])
def test_product_tables_vs_oracle(case):
    from wtpse_hip import input_pipeline as P
    i, o, f = case
    b0, k0 = T.precompute_coeffs(i, o, f)
    b1, k1, ks = P.resample_table(i, o, f)
    assert np.array_equal(b0, b1) and np.array_equal(k0, k1) and ks == k0.shape[1]
    b2, k2, _ = P.resample_table(i, o, f, 5, 20)           # a crop window of the table
    assert np.array_equal(b0[5:25], b2) and np.array_equal(k0[5:25], k2)
    assert np.array_equal(T.nearest_index(i, o), P.nearest_table(i, o))
    rng_a, rng_b = random.Random(7), random.Random(7)
    for _ in range(20):
        d = T.draw_like_reference(rng_a, 256, 256, 256)
        assert P.draw(rng_b, 256) == tuple(d[1:])
