#!/usr/bin/env python3
"""cProfile of the host side of TrainStep.step (where the ~1 850 launches per step cost their Python time)."""
import cProfile
import os
import pstats
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "wt-pse-code_amd")]
import bench  # noqa: E402
from wtpse_hip.step import TrainStep  # noqa: E402
from wtpse_hip.synth import make_batch, default_hparams  # noqa: E402

dev = torch.device("cuda:0")
hp = default_hparams(True)
B = 32
nets = bench.build_nets(hp, B // 3, dev)
ts = TrainStep(*nets, hp, dp=None)
image, od, oc = make_batch(B, 256, 256, dev, seed=1)
for _ in range(2):
    ts.step(image, od, oc)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(3):
    ts.step(image, od, oc)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
