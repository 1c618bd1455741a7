"""Race detector for the training step: no kernel uses floating-point atomics and every reduction folds its partials in a
fixed order, so two runs from the same state must agree bit for bit — also with the weight gradients on their side stream
(wtpse_hip/nn.py::_wgrad_side).  A difference means a missing stream dependency or an unsynchronised scratch buffer."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "wt-pse-code_amd")]


@pytest.mark.gpu
def test_two_runs_bitwise_identical():
    import bench
    from wtpse_hip.step import TrainStep
    from wtpse_hip.synth import make_batch, default_hparams
    dev = torch.device("cuda:0")
    hp = default_hparams(True)
    B = 6

    def run():
        torch.manual_seed(0)
        nets = bench.build_nets(hp, B // 3, dev)
        for n in nets:
            n.seed_noise(1234)
        ts = TrainStep(*nets, hp, dp=None)
        for k in range(3):
            image, od, oc = make_batch(B, 64, 64, dev, seed=10 + k)
            res = ts.step(image, od, oc)
        torch.cuda.synchronize()
        return [n.flat_params().clone() for n in nets], {k: float(v) for k, v in res.items()}

    a, la = run()
    b, lb = run()
    assert all(v == v for v in la.values()), la
    assert la == lb
    for x, y in zip(a, b):
        assert torch.equal(x, y)
