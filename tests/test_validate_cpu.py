"""ASD / HD95 (wtpse_hip/validate.py, restating medpy 0.5.2's published algorithm — parity unpinned, medpy is absent) against the
brute-force surface-distance oracle (oracle/metrics_cpu.py), the empty-prediction convention of Trainer.py:218-239, and the
per-epoch bookkeeping of Trainer.py:258-288 on stand-in networks (no GPU: the bookkeeping is host code)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "wt-pse-code_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

from oracle import metrics_cpu as M          # noqa: E402
from wtpse_hip import validate as V          # noqa: E402


def _disc(h, w, cy, cx, r):
    yy, xx = np.mgrid[:h, :w]
    return ((yy - cy) ** 2 + (xx - cx) ** 2 <= r * r).astype(np.uint8)


CASES = [(_disc(24, 24, 12, 12, 7), _disc(24, 24, 11, 13, 5)),            # nested discs
         (_disc(20, 28, 8, 9, 6), _disc(20, 28, 11, 17, 6)),              # overlapping, shifted
         (_disc(16, 16, 2, 2, 4), _disc(16, 16, 13, 12, 3)),              # disjoint, one touching the image border
         (np.pad(np.ones((5, 7), np.uint8), 4), _disc(13, 15, 6, 7, 2)),  # rectangle vs disc
         (np.eye(9, dtype=np.uint8), np.fliplr(np.eye(9, dtype=np.uint8)))]   # one-pixel-wide objects: every pixel is surface


@pytest.mark.parametrize("i", range(len(CASES)))
def test_asd_hd95_vs_bruteforce(i):
    a, b = CASES[i]
    for x, y in ((a, b), (b, a)):
        assert abs(V.asd(x, y) - M.asd(x.tolist(), y.tolist())) < 1e-12
        assert abs(V.hd95(x, y) - M.hd95(x.tolist(), y.tolist())) < 1e-12
    assert V.hd95(a, b) == V.hd95(b, a)                 # pooled over both directions: symmetric
    assert V.asd(a, a) == 0.0 and V.hd95(a, a) == 0.0


def test_random_masks_vs_bruteforce():
    rng = np.random.default_rng(5)
    for _ in range(20):
        a = (rng.random((12, 14)) > 0.6).astype(np.uint8)
        b = (rng.random((12, 14)) > 0.5).astype(np.uint8)
        assert abs(V.asd(a, b) - M.asd(a.tolist(), b.tolist())) < 1e-12
        assert abs(V.hd95(a, b) - M.hd95(a.tolist(), b.tolist())) < 1e-12


def test_empty_masks():
    z, d = np.zeros((8, 8), np.uint8), _disc(8, 8, 4, 4, 2)
    assert V.surface_metrics(z, d) == (100.0, 100.0)            # Trainer.py:218-221,229-231
    with pytest.raises(RuntimeError):                            # an empty LABEL is medpy's error, as in the reference
        V.surface_metrics(d, z)


def test_validator_bookkeeping(monkeypatch, tmp_path):
    """objective, best tracking, return tuple, checkpoint dict keys (Trainer.py:258-288) with validate_epoch stubbed."""
    seq = iter([dict(cup_dice=0.5, disc_dice=0.7, cup_hd=9.0, disc_hd=5.0, cup_asd=3.0, disc_asd=2.0, n=4),
                dict(cup_dice=0.6, disc_dice=0.6, cup_hd=8.0, disc_hd=6.0, cup_asd=2.5, disc_asd=2.5, n=4),
                dict(cup_dice=0.9, disc_dice=0.5, cup_hd=7.0, disc_hd=7.0, cup_asd=2.0, disc_asd=3.0, n=4)])
    monkeypatch.setattr(V, "validate_epoch", lambda *a: next(seq))

    class Net:
        def __init__(self, tag):
            self.tag = tag

        def state_dict(self):
            return {"w": self.tag}
    nets = [Net(i) for i in range(4)]
    import torch
    val = V.Validator("OD_OC", out_dir=str(tmp_path))
    r = val(0, *nets, None)
    assert r == (1, 0.5, 9.0, 3.0, 0.7, 5.0, 2.0) and val.best_epoch == 1 and abs(val.best_mean_dice - 0.6) < 1e-12
    assert val(1, *nets, None) == (0, 0, 0, 0, 0, 0, 0)          # mean 0.6 is not > 0.6
    r = val(2, *nets, None)
    assert r[0] == 1 and val.best_epoch == 3
    ck = torch.load(os.path.join(str(tmp_path), "checkpoint_3.pth.tar"))
    assert set(ck) == {"model", "model_shape", "model_oc", "model_oc_shape"} and ck["model_oc_shape"] == {"w": 3}
    assert len(open(os.path.join(str(tmp_path), "score.txt")).read().strip().splitlines()) == 2
    v2 = V.Validator("OD")
    seq2 = iter([dict(cup_dice=0.1, disc_dice=0.8, cup_hd=1.0, disc_hd=1.0, cup_asd=1.0, disc_asd=1.0, n=1)])
    monkeypatch.setattr(V, "validate_epoch", lambda *a: next(seq2))
    assert v2(0, *nets, None)[0] == 1 and abs(v2.best_mean_dice - 0.8) < 1e-12
