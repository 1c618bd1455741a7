#!/usr/bin/env python3
"""compute_whitening_loss forward as a call (wtpse_wt_loss_fwd) and its Gram stage alone (wtpse_wt_gram_fwd) on [32,16,256,256] and
[16,16,512,512], HIP events, five operand sets in rotation (> 512 MB).  WTPSE_PKG_DIR selects the package build (same-box A/B)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.environ.get("WTPSE_PKG_DIR") or os.path.join(ROOT, "wt-pse-code_amd")]
import torch
from wtpse_hip import ops

dev = torch.device("cuda")
L = ops.lib()
for B, H in ((32, 256), (16, 512)):
    zs = [torch.randn(B, 16, H, H, device=dev) for _ in range(5)]
    S = L.query("wtpse_wt_split", B, H * H, 0)
    partial = torch.empty(B * S * 256 + 64, device=dev)
    bufs = [torch.empty(B * 256, device=dev), torch.empty(B * 120, device=dev), torch.empty(B, device=dev), torch.empty(B, device=dev),
            torch.empty(B + 1, dtype=torch.float64, device=dev), torch.empty(B * 120, device=dev), torch.empty(3, device=dev)]
    pb = B // 3

    def call(z):
        L.call("wtpse_wt_loss_fwd", z.data_ptr(), B, 16, H * H, 1e-5, 0.0, 3, pb, partial.data_ptr(), *[b.data_ptr() for b in bufs], ops.stream_ptr())

    def gram(z):
        L.call("wtpse_wt_gram_fwd", z.data_ptr(), B, 16, H * H, 1e-5, partial.data_ptr(), *[b.data_ptr() for b in bufs[:4]], ops.stream_ptr())

    for name, fn in (("call (Gram + fold + MMD + final)", call), ("Gram + fold", gram)):
        for z in zs:
            fn(z)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(50):
            fn(zs[i % 5])
        e1.record()
        torch.cuda.synchronize()
        us = 1e3 * e0.elapsed_time(e1) / 50
        print("[%d,16,%d,%d] S=%d %-34s %6.1f us  %6.0f GB/s" % (B, H, H, S, name, us, B * 16 * H * H * 4 / us / 1e3), flush=True)
