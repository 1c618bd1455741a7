#!/usr/bin/env python3
"""conv_x3r_k with the BatchNorm-backward apply formed on load (wtpse_dgrad_x3_in) against apply pass + plain data gradient, and the
weight gradient's fused form against the plain one, per layer (B = 32).   gpurun -- python tools/probe/bnin_time.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "wt-pse-code_amd"), os.path.join(ROOT, "tests")]
from wtpse_hip import ops  # noqa: E402


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


def main():
    from test_conv_x3_gpu import pack_x3
    dev = torch.device("cuda:0")
    B = 32
    for name, K, rows, hw in (("down2.conv2", 64, 64, 64), ("down3.conv2", 128, 128, 32), ("down4.conv2", 256, 256, 16),
                              ("up1.conv3", 256, 256, 32), ("up2.conv3", 128, 128, 64), ("up3.conv3", 64, 64, 128),
                              ("up3.conv1", 64, 128, 64), ("up2.conv1", 128, 256, 32)):
        g = torch.randn(B, K, hw, hw, device=dev)
        y = torch.randn(B, K, hw, hw, device=dev)
        coef = torch.rand(K, 3, device=dev) + 0.5
        w = torch.randn(K, rows, 3, 3) * 0.05
        packed, _, xd = pack_x3(w)
        wptr = packed.data_ptr() + 2 * xd
        x = torch.randn(B, rows, hw, hw, device=dev)
        dw = torch.empty(K, rows, 3, 3, device=dev)
        dy = ops.bn_bwd_apply_coef(g, y, coef)
        ok = ops.x3_bnin_supported(B, hw, hw, rows)
        t_apply = timeit(lambda: ops.bn_bwd_apply_coef(g, y, coef))
        t_plain = timeit(lambda: ops.conv_fwd_x3(dy, None, wptr, None, rows, 3, None, 0, False, False, None, None))
        t_in = timeit(lambda: ops.dgrad_x3_in(g, y, coef, wptr, rows, None)) if ok else float("nan")
        t_w = timeit(lambda: ops.conv_wgrad_r(dy, x, None, dw, None))
        t_wbn = timeit(lambda: ops.conv_wgrad_r_bn(g, y, coef, x, None, dw))
        print("%-12s K %3d -> rows %3d @%3d: apply %6.1f us | dgrad plain %6.1f, on-load %6.1f (%+5.1f) | wgrad plain %6.1f, on-load %6.1f (%+5.1f) | "
              "sum %6.1f -> %6.1f" % (name, K, rows, hw, t_apply, t_plain, t_in, t_in - t_plain, t_w, t_wbn, t_wbn - t_w,
                                       t_apply + t_plain + t_w, t_in + t_wbn))


if __name__ == "__main__":
    main()
