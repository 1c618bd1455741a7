#!/usr/bin/env python3
"""HBM bandwidth of the standalone 2-D DWT micro-benchmark (NOT part of WT-PSE).
Algorithmic bytes of an L-level transform: every level reads and writes its region once: 2 * 4 B * N * (1 + 1/4 + ... )."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "wt-pse-code_amd"), os.path.join(ROOT, "tools")]
import torch  # noqa: E402
from wtpse_hip import dwt  # noqa: E402
from microbench import timeit  # noqa: E402

for shape, levels in (((32, 16, 256, 256), 1), ((32, 16, 256, 256), 3), ((16, 16, 512, 512), 4)):
    x = torch.randn(*shape, device="cuda")
    n = x.numel()
    nbytes = 8.0 * n * sum(0.25 ** l for l in range(levels))
    for wv in ("haar", "db2"):
        c = dwt.dwt2(x, wv, levels)
        f, _ = timeit(lambda: dwt.dwt2(x, wv, levels), 20)
        b, _ = timeit(lambda: dwt.idwt2(c, wv, levels), 20)
        print("dwt2 %-4s %s levels=%d  analysis %7.1f us %6.0f GB/s | synthesis %7.1f us %6.0f GB/s   (%.0f MB algorithmic)" % (
            wv, list(shape), levels, f, nbytes / f / 1e3, b, nbytes / b / 1e3, nbytes / 1e6), flush=True)
