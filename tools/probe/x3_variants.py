#!/usr/bin/env python3
"""Which fused piece of the forward x3 kernel costs what: plain / +bias / +prologue / +stats / all, on a few layer shapes."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "wt-pse-code_amd"), os.path.join(ROOT, "tools")]
import torch  # noqa: E402
from wtpse_hip import ops  # noqa: E402
from microbench import timeit, pack, DEV  # noqa: E402
from microbench_x3 import pack_x3  # noqa: E402

B = 32
for c0, c1, co, H in ((32, 32, 64, 128), (64, 0, 64, 64), (16, 16, 32, 256), (128, 128, 256, 32)):
    x0 = torch.randn(B, c0, H, H, device=DEV)
    x1 = torch.randn(B, c1, H, H, device=DEV) if c1 else None
    w = torch.randn(co, c0 + c1, 3, 3, device=DEV) * 0.05
    bias = torch.zeros(co, device=DEV)
    px, _ = pack_x3(w)
    pk, _ = pack(w)
    pro0 = torch.rand(c0, 2, device=DEV)
    pro1 = torch.rand(c1, 2, device=DEV) if c1 else None
    fl = 2.0 * (c0 + c1) * co * 9 * H * H * B
    res = []
    for name, kw in (("plain", dict()), ("bias", dict(bias=bias)), ("pro", dict(pro0=pro0, pro1=pro1, pro_relu=3)),
                     ("stats", dict(want_stats=True)), ("all", dict(bias=bias, pro0=pro0, pro1=pro1, pro_relu=3, want_stats=True))):
        b = kw.pop("bias", None)
        t3, _ = timeit(lambda: ops.conv_fwd_x3(x0, x1, px.data_ptr(), b, co, 3, **kw), 10)
        t32, _ = timeit(lambda: ops.conv_fwd(x0, x1, pk.data_ptr(), b, co, 3, **kw), 10)
        res.append("%s x3 %.1f us (%.0f TF) f32 %.1f us" % (name, t3, fl / t3 / 1e6, t32))
    print("%d+%d->%d @%d: " % (c0, c1, co, H) + " | ".join(res), flush=True)
