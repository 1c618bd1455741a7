"""Device-side input pipeline against the reference: bit-exact on the fixtures generated from the
reference's own transform classes, and against the CPU oracle at the real size with mixed input sizes."""
import os
import random
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "wt-pse-code_amd")]

pytestmark = pytest.mark.gpu


def test_pipeline_vs_reference_fixtures(golden_dir):
    from wtpse_hip.input_pipeline import DeviceInputPipeline, draw
    g = np.load(os.path.join(golden_dir, "transforms.npz"))
    size, n = int(g["size"]), int(g["n"])
    pipe = DeviceInputPipeline(size, "cuda")
    draws = [draw(random.Random(int(g["seed%d" % i])), size) for i in range(n)]
    image, od, oc = pipe([g["in%d_img" % i] for i in range(n)], [g["in%d_od" % i] for i in range(n)], draws)
    for i in range(n):
        assert np.array_equal(image[i].cpu().numpy(), g["out%d_img" % i]), i     # floats included: bit-exact
        assert np.array_equal(od[i].cpu().numpy(), g["out%d_od" % i]), i
        assert np.array_equal(oc[i].cpu().numpy(), g["out%d_oc" % i]), i


def test_pipeline_vs_oracle_256():
    from oracle import transforms_cpu as T
    from wtpse_hip.input_pipeline import DeviceInputPipeline, draw
    rs = np.random.RandomState(3)
    sizes = [(300, 300), (300, 300), (256, 256), (411, 333), (300, 300), (256, 200)]
    imgs = [rs.randint(0, 256, (h, w, 3)).astype(np.uint8) for h, w in sizes]
    ods = [rs.choice(np.array([0, 40, 50, 51, 128, 200, 201, 255], np.uint8), (h, w)) for h, w in sizes]
    rng = random.Random(99)
    draws = [draw(rng, 256) for _ in sizes]
    assert any(d[0] != 256 or d[1] != 256 for d in draws) and any(d[:2] == (256, 256) for d in draws)
    image, od, oc = DeviceInputPipeline(256, "cuda")(imgs, ods, draws)
    for i, d in enumerate(draws):
        seed = 1.0 if d[:2] != (256, 256) else 0.0
        im0, od0, oc0 = T.train_transform(imgs[i], ods[i], ods[i], (seed,) + d, 256)
        assert np.array_equal(image[i].cpu().numpy(), im0), i
        assert np.array_equal(od[i].cpu().numpy(), od0), i
        assert np.array_equal(oc[i].cpu().numpy(), oc0), i


def test_png_tree_to_device_batch(tmp_path):
    """End of §8f row 3: the synthetic PNG tree (oracle/fundus_tree.py; layout of fundus_dataloader.py:41-44) -> FundusTree /
    multi_batch (the reference's dataset + get_multi_batch on the host) -> DeviceInputPipeline: the fp32 batch equals, bit for bit,
    the CPU oracle's Resize / RandomScaleCrop / Normalize_tf / ToTensor on the same decoded samples with the same random draws."""
    from oracle import transforms_cpu as T
    from oracle.fundus_tree import make_tree
    from wtpse_hip.fundus_data import FundusTree, multi_batch
    from wtpse_hip.input_pipeline import DeviceInputPipeline, draw
    make_tree(str(tmp_path), seed=5)
    sets = [FundusTree(str(tmp_path), "train", (i,)) for i in (1, 2, 4)]       # datasetTrain = [1, 2, 4] (BASELINE.json configs[0])
    np.random.seed(21)
    images, masks = multi_batch(sets, 2)
    rng = random.Random(7)
    draws = [draw(rng, 256) for _ in images]
    image, od, oc = DeviceInputPipeline(256, "cuda")(images, masks, draws)
    assert image.shape == (6, 3, 256, 256) and od.shape == (6, 1, 256, 256)
    for i, d in enumerate(draws):
        seed = 1.0 if d[:2] != (256, 256) else 0.0
        im0, od0, oc0 = T.train_transform(images[i], masks[i], masks[i], (seed,) + d, 256)
        assert np.array_equal(image[i].cpu().numpy(), im0), i
        assert np.array_equal(od[i].cpu().numpy(), od0), i
        assert np.array_equal(oc[i].cpu().numpy(), oc0), i
    assert 0.0 < float(od.mean()) < 1.0 and float(oc.sum()) > 0        # disc and cup made it through the thresholds
