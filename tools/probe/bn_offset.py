#!/usr/bin/env python3
"""How much does the (sum, sum of squares) form of the train-mode BatchNorm statistics lose on maps whose mean dwarfs their
spread (the all -1 ROI of an empty disc prediction)?  conv -> BN of  offset + amp * noise  against fp64."""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "wt-pse-code_amd"), os.path.join(ROOT, "tests")]
from test_kernels_gpu import rnd, ops, pack, DEV   # noqa: E402

o = ops()
B, C, H, W = 8, 32, 64, 64
for offset, amp in ((0.0, 1.0), (-1.0, 1e-1), (-1.0, 1e-2), (-1.0, 1e-3), (-1.0, 0.0), (10.0, 1e-2)):
    xin = offset + amp * rnd(B, 16, H, W, seed=1)
    w = rnd(C, 16, 3, 3, seed=2, scale=0.3)
    bias = rnd(C, seed=3)
    gamma, beta = torch.ones(C), torch.zeros(C)
    y64 = F.conv2d(xin.double(), w.double(), bias.double(), padding=1)
    z64 = F.batch_norm(y64, None, None, gamma.double(), beta.double(), True, 0.1, 1e-5)
    z32 = F.batch_norm(F.conv2d(xin, w, bias, padding=1), None, None, gamma, beta, True, 0.1, 1e-5)
    packed, wf, _ = pack(w)
    y, _, stats = o.conv_fwd(xin.to(DEV), None, packed.data_ptr() + 4 * wf, bias.to(DEV), C, 3, want_stats=True)
    nbt = torch.zeros((), dtype=torch.long, device=DEV)
    ss, mean, invstd = o.bn_finalize(stats, B * H * W, gamma.to(DEV), beta.to(DEV), torch.zeros(C, device=DEV), torch.ones(C, device=DEV), nbt)
    z = o.affine_act(y, ss, False).cpu().double()
    var64 = y64.var((0, 2, 3), unbiased=False)
    e = lambda a: float((a - z64).abs().max() / z64.abs().max())
    ivs64 = 1.0 / torch.sqrt(var64 + 1e-5)
    print("offset %5.1f amp %7.0e: |mean|/std of y %9.1f   max |z - z64| / max|z64|: HIP %.2e, torch CPU fp32 %.2e   invstd rel err HIP %.2e" % (
        offset, amp, float((y64.mean((0, 2, 3)).abs() / var64.sqrt().clamp_min(1e-30)).median()), e(z), e(z32.double()),
        float(((invstd.cpu().double() - ivs64) / ivs64).abs().max())))
