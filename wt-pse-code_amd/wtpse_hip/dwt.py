"""Standalone 2-D DWT micro-benchmark (csrc/dwt.hip).  NOT part of WT-PSE and not imported by the drop-in modules: the
reference has no wavelet transform; specification and checker: oracle/dwt_cpu.py (parity unpinned)."""
import torch

from . import ops

WAVELETS = {"haar": 0, "db2": 1}


def _tmp(x):
    B, C, H, W = x.shape
    return ops.workspace("dwt_tmp", 2 * B * C * (H // 2) * (W // 2), x.device)


def dwt2(x, wavelet="haar", levels=1):
    """x [B,C,H,W] fp32 on the GPU -> Mallat-layout coefficients, same shape."""
    ops._chk(x, "x")
    B, C, H, W = x.shape
    out = torch.empty_like(x)
    ops.lib().call("wtpse_dwt2_fwd", x.data_ptr(), out.data_ptr(), _tmp(x).data_ptr(), B * C, H, W, WAVELETS[wavelet], int(levels),
                   ops.stream_ptr())
    return out


def idwt2(coef, wavelet="haar", levels=1):
    ops._chk(coef, "coef")
    B, C, H, W = coef.shape
    out = torch.empty_like(coef)
    ops.lib().call("wtpse_dwt2_inv", coef.data_ptr(), out.data_ptr(), _tmp(coef).data_ptr(), B * C, H, W, WAVELETS[wavelet],
                   int(levels), ops.stream_ptr())
    return out
