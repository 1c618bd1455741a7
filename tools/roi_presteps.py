#!/usr/bin/env python3
"""How many training steps on the synthetic batch until the optic-disc prediction (the ROI calls C/D work on) is
non-empty?  Prints the fraction of pixels inside od_pred and the losses every few steps."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "wt-pse-code_amd")]
import torch  # noqa: E402

sys.argv = sys.argv[:1] + sys.argv[1:]
import bench  # noqa: E402
from wtpse_hip.step import TrainStep  # noqa: E402
from wtpse_hip.synth import make_batch, default_hparams  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
B, H = 32, 256
hp = default_hparams(True)
nets = bench.build_nets(hp, B // 3, dev)
ts = TrainStep(*nets, hp)
image, od, oc = make_batch(B, H, H, dev, seed=1)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
t0 = time.time()
for i in range(n):
    res = ts.step(image, od, oc)
    if i % 10 == 0 or i == n - 1:
        frac = float(ts.last_od_pred.mean())
        print("step %3d  od_pred fraction %.4f  target %.4f  seg_od %.4f seg_oc %.4f dom_oc %.2e kd_oc %.4f  (%.1fs)" % (
            i, frac, float(od.mean()), float(res["seg_od"]), float(res["seg_oc"]), float(res["dom_oc"]), float(res["kd_oc"]),
            time.time() - t0), flush=True)
