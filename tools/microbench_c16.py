#!/usr/bin/env python3
"""The 16-channel 3x3 layers (inc.conv2, DeepWT, the teacher's inc) at 256x256 / 512x512: fp32-input MFMA (conv.hip MODE 0)
against the x3 arithmetic with register-resident weights (conv.hip MODE 3).  HIP events, interleaved in one process.

    python tools/microbench_c16.py [B]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "wt-pse-code_amd"), os.path.join(ROOT, "tools")]
import torch  # noqa: E402
from wtpse_hip import ops  # noqa: E402
from microbench import timeit, pack, DEV  # noqa: E402


def pack_x16(w):
    n = ops.X16_SIZE
    packed = torch.zeros(2 * n, dtype=torch.int16, device=DEV)
    desc = torch.tensor([0, w.shape[0], w.shape[1], 9, 0, n, 0, 0], dtype=torch.int32, device=DEV)
    ops.lib().call("wtpse_pack_conv16_x3", w.reshape(-1).contiguous().data_ptr(), desc.data_ptr(), 1, packed.data_ptr(), ops.stream_ptr())
    return packed, n


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    for H, b in ((256, B), (512, max(B // 4, 1))):
        x = torch.randn(b, 16, H, H, device=DEV)
        w = torch.randn(16, 16, 3, 3, device=DEV) * 0.05
        bias = torch.zeros(16, device=DEV)
        pro = torch.rand(16, 2, device=DEV)
        yref = torch.randn(b, 16, H, H, device=DEV)
        ss = torch.rand(16, 2, device=DEV)
        mean = torch.rand(16, device=DEV)
        pk, wd = pack(w)
        px, n = pack_x16(w)
        byts = 4.0 * 32 * H * H * b
        flops = 2.0 * 16 * 16 * 9 * H * H * b
        rows = [
            ("forward: prologue, bias, BatchNorm partials",
             lambda: ops.conv_fwd(x, None, pk.data_ptr(), bias, 16, 3, pro, 1, want_stats=True),
             lambda: ops.conv16_x3(x, px.data_ptr(), bias, 16, pro, 1, want_stats=True), byts),
            ("forward: ReLU on load, bias, ReLU out",
             lambda: ops.conv_fwd(x, None, pk.data_ptr(), bias, 16, 3, None, 1, relu_out=True),
             lambda: ops.conv16_x3(x, px.data_ptr(), bias, 16, None, 1, relu_out=True), byts),
            ("forward + Gram partials",
             lambda: ops.conv_fwd_gram(x, pk.data_ptr(), bias),
             lambda: ops.conv16_x3(x, px.data_ptr(), bias, 16, want_gram=True), byts),
            ("data gradient + ReLU mask",
             lambda: ops.conv_fwd(x, None, pk.data_ptr() + 4 * wd, None, 16, 3, mask_ref=yref),
             lambda: ops.conv16_x3(x, px.data_ptr() + 2 * n, None, 16, mask_ref=yref), byts * 1.5),
            ("data gradient + BatchNorm-backward epilogue",
             lambda: ops.dgrad_bnb(x, pk.data_ptr() + 4 * wd, 0, 16, 3, yref, ss, mean, True),
             lambda: ops.conv16_x3(x, px.data_ptr() + 2 * n, None, 16, bnb=(yref, ss, mean, True)), byts * 1.5),
        ]
        print("16 -> 16, 3x3, %dx%d, B=%d  (%.0f MB in+out, %.2f GFLOP)" % (H, H, b, byts / 1e6, flops / 1e9))
        for name, f32, f16, by in rows:
            t0, _ = timeit(f32, 10)
            t1, m1 = timeit(f16, 10)
            print("  %-46s fp32-MFMA %7.1f us %5.2f TB/s | x3 16x16x32 %7.1f us (min %6.1f) %5.2f TB/s %5.1f TF x3-eq  (%.2fx)"
                  % (name, t0, by / t0 / 1e6, t1, m1, by / t1 / 1e6, flops / t1 / 1e6, t0 / t1), flush=True)


if __name__ == "__main__":
    main()
