#!/usr/bin/env python3
"""Runs K full training steps twice from the same initial state and compares every parameter bit for bit.
The step has no atomics, so any difference would mean a race (e.g. in the side-stream weight gradients)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "wt-pse-code_amd")]
import bench  # noqa: E402
from wtpse_hip.step import TrainStep  # noqa: E402
from wtpse_hip.synth import make_batch, default_hparams  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device("cuda:0")
hp = default_hparams(True)
B = 12


def run():
    torch.manual_seed(0)
    nets = bench.build_nets(hp, B // 3, dev)
    for n in nets:
        n.seed_noise(1234)
    ts = TrainStep(*nets, hp, dp=None)
    out = []
    for k in range(K):
        image, od, oc = make_batch(B, 128, 128, dev, seed=10 + k)
        res = ts.step(image, od, oc)
    torch.cuda.synchronize()
    return [n.flat_params().clone() for n in nets], {k: float(v) for k, v in res.items()}


a, la = run()
b, lb = run()
same = all(torch.equal(x, y) for x, y in zip(a, b))
print("losses run 1:", {k: round(v, 6) for k, v in la.items()})
print("bitwise identical parameters after %d steps: %s" % (K, same))
assert same and la == lb
assert all(v == v for v in la.values())
