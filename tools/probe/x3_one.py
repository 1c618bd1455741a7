#!/usr/bin/env python3
"""One conv launch shape, repeated, for rocprofv3 --pmc runs:  x3_one.py <x3|f32> <fwd|dgrad> C0 C1 Cout H [reps] [B]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "wt-pse-code_amd"), os.path.join(ROOT, "tools")]
import torch  # noqa: E402
from wtpse_hip import ops  # noqa: E402
from microbench import pack, DEV  # noqa: E402
from microbench_x3 import pack_x3  # noqa: E402

kind, direction = sys.argv[1], sys.argv[2]
c0, c1, co, H = (int(v) for v in sys.argv[3:7])
reps = int(sys.argv[7]) if len(sys.argv) > 7 else 10
B = int(sys.argv[8]) if len(sys.argv) > 8 else 32
k = 3
x0 = torch.randn(B, c0, H, H, device=DEV)
x1 = torch.randn(B, c1, H, H, device=DEV) if c1 else None
w = torch.randn(co, c0 + c1, k, k, device=DEV) * 0.05
bias = torch.zeros(co, device=DEV)
packed, wd_off = pack(w)
px, xd_off = pack_x3(w)
dy = torch.randn(B, co, H, H, device=DEV)
pro0 = torch.rand(c0, 2, device=DEV)
pro1 = torch.rand(c1, 2, device=DEV) if c1 else None
for _ in range(reps):
    if direction == "fwd":
        if kind == "x3":
            ops.conv_fwd_x3(x0, x1, px.data_ptr(), bias, co, k, pro0, 3, want_stats=True, pro1=pro1)
        else:
            ops.conv_fwd(x0, x1, packed.data_ptr(), bias, co, k, pro0, 3, want_stats=True, pro1=pro1)
    else:
        if kind == "x3":
            ops.conv_fwd_x3(dy, None, px.data_ptr() + 2 * xd_off, None, c0 + c1, k, split=(c0 if c1 else None))
        else:
            ops.conv_fwd(dy, None, packed.data_ptr() + 4 * wd_off, None, c0 + c1, k, split=(c0 if c1 else None))
torch.cuda.synchronize()
