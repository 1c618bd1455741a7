#!/usr/bin/env python3
"""Per-kernel timings at the benchmark's layer shapes (HIP events, one process, interleaved rounds).

    python tools/microbench.py [--batch 32] [--only fwd|wgrad|wt|pw] [--reps 10]

Prints one line per (kernel, shape): median / min microseconds and achieved TFLOP/s or GB/s.  Used to A/B kernel
changes between gpurun calls; numbers quoted in DESIGN.md come from bench.py + rocprofv3, not from here."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.environ.get("WTPSE_PKG_DIR") or os.path.join(ROOT, "wt-pse-code_amd")]     # (WTPSE_PKG_DIR: another build of the package, same-box A/B)
import torch  # noqa: E402
from wtpse_hip import ops  # noqa: E402

DEV = "cuda"

# (Cin0, Cin1, Cout, H, k, name) — the distinct conv shapes of one U-Net at 256x256 input
SHAPES = [
    (3, 0, 16, 256, 3, "inc.conv1"), (16, 0, 16, 256, 3, "inc.conv2"), (16, 0, 32, 128, 3, "down1.conv1"),
    (32, 0, 32, 128, 3, "down1.conv2"), (32, 0, 64, 64, 3, "down2.conv1"), (64, 0, 64, 64, 3, "down2.conv2"),
    (64, 0, 128, 32, 3, "down3.conv1"), (128, 0, 128, 32, 3, "down3.conv2"), (128, 0, 256, 16, 3, "down4.conv1"),
    (256, 0, 256, 16, 3, "down4.conv2"), (256, 0, 128, 32, 1, "up1.conv2"), (128, 128, 256, 32, 3, "up1.conv3"),
    (256, 0, 128, 32, 3, "up2.conv1"), (128, 0, 64, 64, 1, "up2.conv2"), (64, 64, 128, 64, 3, "up2.conv3"),
    (128, 0, 64, 64, 3, "up3.conv1"), (64, 0, 32, 128, 1, "up3.conv2"), (32, 32, 64, 128, 3, "up3.conv3"),
    (64, 0, 32, 128, 3, "up4.conv1"), (32, 0, 16, 256, 1, "up4.conv2"), (16, 16, 32, 256, 3, "up4.conv3"),
    (32, 0, 32, 256, 1, "mu.0"), (32, 0, 8, 256, 1, "mu.2"), (8, 0, 1, 256, 1, "outc"),
]


def timeit(fn, reps):
    fn(); fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2], ts[0]


def pack(w):
    co, ci, k, _ = w.shape
    t = k * k
    wf = ((ci + 3) & ~3) * t * ((co + 15) & ~15)
    wd = ((co + 3) & ~3) * t * ((ci + 15) & ~15)
    packed = torch.empty(wf + wd, device=DEV)
    desc = torch.tensor([0, co, ci, t, 0, wf, 0, 0], dtype=torch.int32, device=DEV)
    ops.lib().call("wtpse_pack_conv_weights", w.reshape(-1).contiguous().data_ptr(), desc.data_ptr(), 1, packed.data_ptr(), ops.stream_ptr())
    return packed, wf


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--only", default="")
    ap.add_argument("--no-pro", action="store_true", help="time forward / weight gradient without the BatchNorm prologue")
    ap.add_argument("--filter", default="")
    a = ap.parse_args()
    B = a.batch
    tot = {"fwd": 0.0, "dgrad": 0.0, "wgrad": 0.0}
    flops_tot = 0.0
    layers = a.only in ("", "fwd", "dgrad", "wgrad")      # the other selections time no convolution layer
    for c0, c1, co, H, k, name in (SHAPES if layers else []):
        if a.filter and a.filter not in name:
            continue
        x0 = torch.randn(B, c0, H, H, device=DEV)
        x1 = torch.randn(B, c1, H, H, device=DEV) if c1 else None
        w = torch.randn(co, c0 + c1, k, k, device=DEV) * 0.05
        bias = torch.zeros(co, device=DEV)
        packed, wd_off = pack(w)
        dy = torch.randn(B, co, H, H, device=DEV)
        # as inside a step: the inputs of all but the first layer carry the BatchNorm-apply + ReLU prologue
        # (forward and weight gradient; the data gradient reads a plain dY).  --no-pro times the bare kernels.
        use_pro = not a.no_pro and c0 > 4
        pro0 = torch.rand(c0, 2, device=DEV) if use_pro else None
        pro1 = torch.rand(c1, 2, device=DEV) if (use_pro and c1) else None
        relu = 3 if use_pro else 0
        flops = 2.0 * (c0 + c1) * co * k * k * H * H * B
        flops_tot += flops
        line = "%-12s %3d+%-3d->%-3d k%d @%3d  " % (name, c0, c1, co, k, H)
        if a.only in ("", "fwd"):
            med, mn = timeit(lambda: ops.conv_fwd(x0, x1, packed.data_ptr(), bias, co, k, pro0, relu, want_stats=True, pro1=pro1), a.reps)
            tot["fwd"] += med
            line += "fwd %7.1f us %5.1f TF | " % (med, flops / med / 1e6)
        if a.only in ("", "dgrad") and c0 + c1 > 4:
            med, mn = timeit(lambda: ops.conv_fwd(dy, None, packed.data_ptr() + 4 * wd_off, None, c0 + c1, k,
                                                  split=(c0 if c1 else None)), a.reps)
            tot["dgrad"] += med
            line += "dgrad %7.1f us %5.1f TF | " % (med, flops / med / 1e6)
        if a.only in ("", "wgrad"):
            dw = torch.empty_like(w)
            med, mn = timeit(lambda: ops.conv_wgrad(dy, x0, x1, k, dw, None, pro0, relu, False, pro1), a.reps)
            tot["wgrad"] += med
            line += "wgrad %7.1f us %5.1f TF" % (med, flops / med / 1e6)
        print(line, flush=True)
    if layers:
        print("sum over listed shapes: fwd %.0f us, dgrad %.0f us, wgrad %.0f us; %.1f GFLOP per pass" %
              (tot["fwd"], tot["dgrad"], tot["wgrad"], flops_tot / 1e9))
    if a.only in ("", "wt"):
        z = torch.randn(B, 16, 256, 256, device=DEV)
        nbytes = z.numel() * 4.0
        st = ops.wt_loss_fwd(z, 3, B // 3, 0.0)
        med, mn = timeit(lambda: ops.wt_loss_fwd(z, 3, B // 3, 0.0), a.reps)
        print("wt_loss_fwd  [%d,16,256,256]  %7.1f us  %6.0f GB/s" % (B, med, nbytes / med / 1e3))
        dz = torch.empty_like(z)
        med, mn = timeit(lambda: ops.wt_loss_bwd(st, dz, False), a.reps)
        print("wt_loss_bwd  [%d,16,256,256]  %7.1f us  %6.0f GB/s" % (B, med, 2 * nbytes / med / 1e3))
    if a.only in ("", "head"):
        x = torch.randn(B, 32, 256, 256, device=DEV)
        pro = torch.rand(32, 2, device=DEV)
        w1, b1 = torch.randn(32, 32, 1, 1, device=DEV) * 0.2, torch.randn(32, device=DEV)
        w2, b2 = torch.randn(8, 32, 1, 1, device=DEV) * 0.2, torch.randn(8, device=DEV)
        w3, b3 = torch.randn(1, 8, 1, 1, device=DEV) * 0.2, torch.randn(1, device=DEV)
        px = x.numel() / 32
        med, _ = timeit(lambda: ops.head_fwd(x, pro, True, w1, b1, w2, b2, w3, b3, True, want_h1=True), a.reps)
        print("head_fwd 32-32-8-1 +tape [%d,32,256,256]  %7.1f us  %6.0f GB/s (73 floats/px)" % (B, med, 73 * 4 * px / med / 1e3))
        med, _ = timeit(lambda: ops.head_fwd(x, pro, True, w1, b1, w2, b2, w3, b3, False), a.reps)
        print("head_fwd 32-32-8-1 no tape                 %7.1f us  %6.0f GB/s (33 floats/px)" % (med, 33 * 4 * px / med / 1e3))
        y, h1, h2 = ops.head_fwd(x, pro, True, w1, b1, w2, b2, w3, b3, True, want_h1=True)
        dy = torch.randn_like(y)
        dpar = torch.zeros(1320 + 9, device=DEV)
        xam, dyam = ops.act_bound(pro, ops.amax_of(x)), ops.amax_of(dy)
        med, _ = timeit(lambda: ops.head_fwd(x, pro, True, w1, b1, w2, b2, w3, b3, True, x_amax=xam), a.reps)
        print("head_fwd 32-32-8-1 as the step calls it     %7.1f us  %6.0f GB/s (h2 tape only under x2h: 41 floats/px)" % (med, 41 * 4 * px / med / 1e3))
        med, _ = timeit(lambda: ops.head_bwd(dy, x, pro, True, h1, h2, w1, w2, w3, dpar, b1=b1, x_amax=xam, dy_amax=dyam), a.reps)
        print("head_bwd 32-32-8-1                         %7.1f us  %6.0f GB/s (x2h: no h1 read, 73 floats/px; fp32: 105)" % (med, 73 * 4 * px / med / 1e3))
    if a.only in ("", "bn"):
        for (C, H) in ((16, 256), (32, 256), (32, 128), (64, 64), (256, 16)):
            y = torch.randn(B, C, H, H, device=DEV)
            dz = torch.randn_like(y)
            ss = torch.rand(C, 2, device=DEV); gamma = torch.rand(C, device=DEV) + 0.5
            mean = torch.zeros(C, device=DEV); invstd = torch.ones(C, device=DEV)
            dg = torch.zeros(C, device=DEV); db = torch.zeros(C, device=DEV)
            med, _ = timeit(lambda: ops.bn_bwd(dz, y, ss, True, gamma, mean, invstd, dg, db), a.reps)
            print("bn_bwd (reduce+finalize+apply) [%d,%d,%d,%d]  %7.1f us  %6.0f GB/s (5 passes)" % (B, C, H, H, med, 5 * y.numel() * 4.0 / med / 1e3))
    if a.only in ("", "bn"):
        # the apply pass alone (dy = k1 g + k2 y + k3: what remains of a BatchNorm backward whose reductions rode in the producing data
        # gradient), with the amax table of dy it fills under the x2h arithmetic
        for C, H in ((32, 256), (64, 128), (128, 64), (256, 32)):
            g = torch.randn(B, C, H, H, device=DEV)
            y = torch.randn(B, C, H, H, device=DEV)
            coef = torch.rand(C, 3, device=DEV)
            med, _ = timeit(lambda: ops.bn_bwd_apply_coef(g, y, coef), a.reps)
            print("bn_bwd_apply_coef [%d,%d,%d,%d]  %7.1f us  %6.0f GB/s (3 passes; x3_terms %d)" % (B, C, H, H, med, 3 * y.numel() * 4.0 / med / 1e3, ops.x3_terms()))
    if a.only in ("", "pw"):
        y = torch.randn(B, 16, 256, 256, device=DEV)
        ss = torch.rand(16, 2, device=DEV)
        nbytes = y.numel() * 4.0
        med, _ = timeit(lambda: ops.affine_act(y, ss, True), a.reps)
        print("affine_act   [%d,16,256,256]  %7.1f us  %6.0f GB/s" % (B, med, 2 * nbytes / med / 1e3))
        med, _ = timeit(lambda: ops.maxpool2_fwd(y), a.reps)
        print("maxpool2_fwd [%d,16,256,256]  %7.1f us  %6.0f GB/s" % (B, med, 1.25 * nbytes / med / 1e3))
        x = torch.randn(B, 32, 128, 128, device=DEV)
        med, _ = timeit(lambda: ops.upsample2x_fwd(x), a.reps)
        print("upsample_fwd [%d,32,128,128]  %7.1f us  %6.0f GB/s" % (B, med, 5 * x.numel() * 4.0 / med / 1e3))
        du = torch.randn(B, 32, 256, 256, device=DEV)
        med, _ = timeit(lambda: ops.upsample2x_bwd(du), a.reps)
        print("upsample_bwd [%d,32,256,256]  %7.1f us  %6.0f GB/s" % (B, med, 1.25 * du.numel() * 4.0 / med / 1e3))
        dp = torch.randn(B, 16, 128, 128, device=DEV)
        med, _ = timeit(lambda: ops.maxpool2_bwd(y, dp, None, False, None, True), a.reps)
        print("maxpool2_bwd [%d,16,256,256]  %7.1f us  %6.0f GB/s" % (B, med, 2.25 * nbytes / med / 1e3))


if __name__ == "__main__":
    main()
