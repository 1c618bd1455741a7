#!/usr/bin/env python3
"""What each piece of conv_x3r_k's main loop costs: the kernel with pieces left out (template parameter ABL, compiled only with
-DWTPSE_PROBE into tools/probe/_build/ — the product library carries none of it), on layers with 2 / 4 / 8 / 16 chunks per workgroup
(prologue + epilogue = the intercept of time over chunks).

    python tools/probe/x3r_abl.py            # build here (CPU container: the .so travels with the snapshot)
    gpurun -- python tools/probe/x3r_abl.py  # measure
ABL bits: 1 conversion VALU, 2 LDS stores, 4 weight-fragment loads, 8 input-fragment reads, 16 chunk barrier, 32 input tile loads,
64 weight loads issued but not consumed, 128 weight loads spread one per MFMA group, 256 weight fragments a whole chunk ahead (ring of 9
tap slots instead of 3), 512 every 32x32x16 MFMA issued as two 16x16x32 on the same registers (garbage results: clock / cycles only)."""
import ctypes
import os
import subprocess
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "wt-pse-code_amd", "wtpse_hip", "csrc")
OUT = os.path.join(HERE, "_build", "libx3r_abl.so")
sys.path[:0] = [ROOT, os.path.join(ROOT, "wt-pse-code_amd")]


def build():
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DWTPSE_PROBE", "-DWTPSE_SRC_HASH=\"probe\"",
                           "-I", CSRC] + [os.path.join(CSRC, f) for f in ("conv_x3.hip", "conv.hip", "bn.hip")] + ["-o", OUT])


def main():
    if not os.path.isfile(OUT) or os.environ.get("REBUILD"):
        build()
    if not torch.cuda.is_available():
        print("built", OUT)
        return
    torch.cuda.init()
    dll = ctypes.CDLL(OUT)
    vp = ctypes.c_void_p
    dev = torch.device("cuda:0")
    from wtpse_hip import ops
    B = 32
    names = {0: "complete", 1: "-convert", 2: "-lds stores", 3: "-convert -stores", 4: "-A loads", 8: "-B reads", 16: "-barrier", 32: "-X loads",
             35: "-all X work", 39: "-X work -A loads", 47: "-X work -A -B", 63: "MFMA only", 64: "A loads issued, not consumed",
             128: "A loads spread 1/group", 99: "-X work, A not consumed", 163: "-X work, A spread", 256: "weight fragments a whole chunk ahead (ring of 9 tap slots)", 319: "ring of 9, MFMA only",
             260: "ring 3 -A loads", 291: "ring 3 -X work", 512: "complete, MFMAs issued as 2 x 16x16x32 (garbage results)",
             575: "MFMA only, 2 x 16x16x32"}
    for cin, cout, hw in ((32, 64, 128), (64, 64, 128), (128, 64, 128), (256, 64, 128)):
        x = torch.randn(B, cin, hw, hw, device=dev)
        w = torch.randn(cout, cin, 3, 3, device=dev) * 0.05
        xf = ops.x3_packed_size(cout, cin, 9)
        packed = torch.zeros(xf, dtype=torch.int16, device=dev)
        desc = torch.tensor([0, cout, cin, 9, 0, -1, 0, 0], dtype=torch.int32, device=dev)
        assert dll.wtpse_pack_conv_weights_x3(vp(w.data_ptr()), vp(desc.data_ptr()), 1, vp(packed.data_ptr()), None) == 0
        y = torch.empty(B, cout, hw, hw, device=dev)

        def run():
            return dll.wtpse_conv_fwd_x3(vp(x.data_ptr()), cin, None, 0, vp(packed.data_ptr()), None, None, None, 0, vp(y.data_ptr()), None, cout,
                                         None, B, hw, hw, cout, 3, 0, None, None)
        line = []
        for stag in ():
            dll.wtpse_probe_x3r_abl(0); dll.wtpse_x3r_enable(1); dll.wtpse_probe_x3r_stagger(stag)
            for _ in range(20):
                run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(100):
                run()
            e1.record()
            torch.cuda.synchronize()
            line.append("stagger %dk cycles %.1f" % (stag, e0.elapsed_time(e1) * 10))
        dll.wtpse_probe_x3r_stagger(0)
        for abl in (0, 63, 512, 575, 256, 319, -1):
            if abl >= 0:
                dll.wtpse_probe_x3r_abl(abl)
                dll.wtpse_x3r_enable(1)
            else:
                dll.wtpse_probe_x3r_abl(0)
                dll.wtpse_x3r_enable(0)
            for _ in range(20):
                assert run() == 0
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            nrep = 100
            for _ in range(nrep):
                run()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / nrep
            clk = ""
            if abl >= 0:
                nwg = B * (hw // 32) * (hw // 8) * (cout // 64)
                buf = torch.zeros(nwg * 8, dtype=torch.int64, device=dev)
                dll.wtpse_probe_x3r_clock(vp(buf.data_ptr()))
                run()
                torch.cuda.synchronize()
                dll.wtpse_probe_x3r_clock(None)
                st = buf.cpu().numpy().reshape(nwg, 8)
                st = st[st[:, 0] != 0]
                ghz = (st[:, 2] - st[:, 0]) / np.maximum(st[:, 3] - st[:, 1], 1) * 0.1
                cyc = np.median(st[:, 2] - st[:, 0]) / (cin // 16)
                # real time (10 ns ticks): kernel start -> loop, loop, loop end -> exit; and the launch's span
                pro, loop, epi = (np.median(st[:, 1] - st[:, 5]), np.median(st[:, 3] - st[:, 1]), np.median(st[:, 7] - st[:, 3]))
                span = st[:, 7].max() - st[:, 5].min()
                clk = " [%.2f GHz, %.0f cycles per chunk; per workgroup: prologue %.1f us, loop %.1f us, epilogue %.1f us; launch span %.1f us]" % (
                    np.median(ghz), cyc, pro / 100.0, loop / 100.0, epi / 100.0, span / 100.0)
            line.append("%s %.1f%s" % (names.get(abl, "round-3 kernel"), us, clk))
        fl = 2.0 * B * hw * hw * cin * cout * 9
        print("%3d->%d @%d (%d chunks, %.1f us at 419.5 TF): " % (cin, cout, hw, cin // 16, fl / 419.5e6) + " | ".join(line), flush=True)


if __name__ == "__main__":
    main()
