"""Builds and binds oracle/wt_loss_ref.c (TEST INFRASTRUCTURE): the plain-C, double-precision restatement of the WT loss."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(_HERE, "wt_loss_ref.c")
OUT_DIR = os.path.join(_HERE, "_build")
OUT = os.path.join(OUT_DIR, "libwt_loss_ref.so")


def build():
    os.makedirs(OUT_DIR, exist_ok=True)
    if not os.path.isfile(OUT) or os.path.getmtime(OUT) < os.path.getmtime(SRC):
        subprocess.run(["gcc", "-O2", "-fPIC", "-shared", "-o", OUT, SRC, "-lm"], check=True)
    return OUT


def wt_loss(z, domains, per_domain, margin=0.0, eps=1e-5):
    """z: numpy/torch [B,16,H,W] fp32 -> (ins_offdiag, ins_diag, domain, gram[B,16,16]) in float64."""
    lib = ctypes.CDLL(build())
    z = np.ascontiguousarray(np.asarray(z, dtype=np.float32))
    B, C, H, W = z.shape
    assert C == 16
    out = np.zeros(3, dtype=np.float64)
    gram = np.zeros((B, 256), dtype=np.float64)
    lib.wt_loss_ref.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_double, ctypes.c_int,
                                ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    rc = lib.wt_loss_ref(z.ctypes.data, B, H * W, float(eps), float(margin), int(domains), int(per_domain), out.ctypes.data,
                         gram.ctypes.data)
    assert rc == 0
    return out[0], out[1], out[2], gram.reshape(B, 16, 16)
