#!/usr/bin/env python3
"""Where the x3 weight-gradient kernel's time goes: the kernel is rebuilt with parts of its loader (or its MFMAs) compiled
out (results are then wrong; only the time matters) and timed on one layer shape.
    python tools/probe/wgrad_abl.py build          # here (hipcc cross-compiles), all variants into tools/probe/_build/
    gpurun -- python tools/probe/wgrad_abl.py [Cin Cout H B]"""
import ctypes
import os
import subprocess
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "wt-pse-code_amd", "wtpse_hip", "csrc")
VARIANTS = ["", "-DEXP_W_NOYLOAD", "-DEXP_W_NOXLOAD", "-DEXP_W_NOYLOAD -DEXP_W_NOXLOAD", "-DEXP_W_NOSPLIT", "-DEXP_W_NOSTORE",
            "-DEXP_W_NOYLOAD -DEXP_W_NOXLOAD -DEXP_W_NOSPLIT -DEXP_W_NOSTORE", "-DEXP_W_NOMFMA"]


def lib(defs):
    return os.path.join(HERE, "_build", "libwgrad_%s.so" % ("".join(c for c in defs if c.isalnum()) or "full"))


def build():
    os.makedirs(os.path.join(HERE, "_build"), exist_ok=True)
    for d in VARIANTS:
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-I", CSRC] + d.split() +
                              [os.path.join(CSRC, "conv_x3.hip"), os.path.join(CSRC, "conv.hip"), "-o", lib(d)])
        print("built", lib(d))


def main():
    if sys.argv[1:2] == ["build"]:
        return build()
    cin, cout, hw, B = (int(v) for v in sys.argv[1:5]) if len(sys.argv) > 4 else (64, 64, 128, 32)
    torch.cuda.init()
    dev = torch.device("cuda:0")
    vp = ctypes.c_void_p
    x = torch.randn(B, cin, hw, hw, device=dev)
    dy = torch.randn(B, cout, hw, hw, device=dev)
    dw = torch.zeros(cout, cin, 3, 3, device=dev)
    flops = 2.0 * B * hw * hw * cin * cout * 9
    for d in VARIANTS:
        dll = ctypes.CDLL(lib(d))
        ks = dll.wtpse_wgrad_x3_ksplit(B, hw, hw, cin, cout)
        slab = torch.zeros(ks * cout * cin * 9, device=dev)

        def run():
            return dll.wtpse_conv_wgrad_x3(vp(dy.data_ptr()), vp(x.data_ptr()), cin, None, 0, None, None, 0, vp(slab.data_ptr()), ks,
                                           vp(dw.data_ptr()), 0, B, hw, hw, cout, 3, None)
        for _ in range(3):
            assert run() == 0
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            run()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 50
        print("%-75s %7.1f us  %6.1f TFLOP/s" % (d or "complete kernel (+ slab fold)", us, flops / us * 1e-6))


if __name__ == "__main__":
    main()
