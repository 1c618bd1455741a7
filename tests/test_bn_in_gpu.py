"""The second half of a BatchNorm backward formed by the consumer of dy as it loads (nn.BN_IN): every fused consumer against the
stand-alone apply pass (wtpse_bn_bwd_apply_coef) followed by the plain consumer — the same expression per element, so bit-exact."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "wt-pse-code_amd"), os.path.join(ROOT, "tests")]

pytestmark = pytest.mark.gpu


def _operands(B, C, H, W, seed):
    from test_kernels_gpu import rnd, DEV
    g = rnd(B, C, H, W, seed=seed).to(DEV)
    g = g * (rnd(B, C, H, W, seed=seed + 1).to(DEV) > 0)            # masked, as a data-gradient epilogue leaves it
    y = rnd(B, C, H, W, seed=seed + 2).to(DEV)
    coef = torch.stack([rnd(C, seed=seed + 3) * 0.3 + 1.0, rnd(C, seed=seed + 4) * 0.05, rnd(C, seed=seed + 5) * 0.02], 1).contiguous().to(DEV)
    return g, y, coef


@pytest.mark.parametrize("shape", [(2, 16, 16, 32), (3, 5, 8, 8), (2, 32, 64, 64), (1, 16, 256, 256)])
def test_upsample_bwd_bn_equals_apply_then_upsample_bwd(shape):
    from test_kernels_gpu import ops
    o = ops()
    g, y, coef = _operands(*shape, seed=7)
    ref = o.upsample2x_bwd(o.bn_bwd_apply_coef(g, y, coef))
    got = o.upsample2x_bwd_bn(g, y, coef)
    assert torch.equal(ref, got)


# B, K (channels of g = the layer's output channels), rows (the data gradient's output channels = the layer's inputs), H, W, split
X3_IN_CASES = [
    (32, 64, 64, 64, 64, None),       # full 256-pixel tiles, one 64-row block
    (32, 128, 128, 32, 32, None),     # 128-pixel tiles (x3_half), two blocks
    (8, 32, 64, 64, 64, None),        # half tiling at a small batch, two 16-channel chunks
    (64, 128, 256, 16, 16, None),     # 16-wide tiles
    (32, 64, 128, 32, 32, 64),        # split outputs (a concat layer's two inputs)
    (6, 80, 64, 72, 96, None),        # ragged tiles and a ragged last chunk
]


def _pack_dgrad(w):
    from test_conv_x3_gpu import pack_x3
    packed, _, xd = pack_x3(w)
    return packed, packed.data_ptr() + 2 * xd


@pytest.mark.parametrize("case", X3_IN_CASES)
def test_dgrad_x3_in_equals_apply_then_dgrad(case):
    """conv_x3r_k's second loader (dY = k1 g + k2 y + k3 formed on load) against the apply pass + the plain data gradient: the same
    fp32 values reach the bf16 split, so outputs, masked gradients, statistics partials and folded coefficients are bitwise equal."""
    from test_kernels_gpu import ops, rnd, DEV
    o = ops()
    B, K, rows, H, W, split = case
    assert o.x3_bnin_supported(B, H, W, rows), "the case must run conv_x3r_k's 64-row blocks"
    g, y, coef = _operands(B, K, H, W, seed=21)
    w = rnd(K, rows, 3, 3, seed=29, scale=0.2)            # the layer's weight [Cout = K][Cin = rows]
    packed, wptr = _pack_dgrad(w)
    dy = o.bn_bwd_apply_coef(g, y, coef)
    ref = o.conv_fwd_x3(dy, None, wptr, None, rows, 3, None, 0, False, False, split, None)[:2]
    got = o.dgrad_x3_in(g, y, coef, wptr, rows, split)
    for a_, b_ in zip(got, ref):
        assert (a_ is None) == (b_ is None)
        if a_ is not None:
            assert torch.equal(a_, b_), float((a_ - b_).abs().max())
    # with the BatchNorm-backward epilogue of the layer below + its coefficient fold
    c_bn = rows if split is None else split
    bn_y = rnd(B, c_bn, H, W, seed=31).to(DEV)
    bn_ss = torch.stack([rnd(c_bn, seed=32) * 0.3 + 1.0, rnd(c_bn, seed=33) * 0.2], 1).contiguous().to(DEV)
    bn_mean = (rnd(c_bn, seed=34) * 0.1).to(DEV)
    gamma = (rnd(c_bn, seed=35) * 0.2 + 1.0).to(DEV)
    invstd = (rnd(c_bn, seed=36).abs() + 0.5).to(DEV)

    def tail():
        return (gamma, invstd, torch.zeros(c_bn, device=DEV), torch.zeros(c_bn, device=DEV))
    t_ref, t_got = tail(), tail()
    r0, r1, rst, rcoef = o.dgrad_bnb(dy, wptr, 1, rows, 3, bn_y, bn_ss, bn_mean, True, split, False, t_ref)
    g0, g1, gst, gcoef = o.dgrad_bnb(g, wptr, 1, rows, 3, bn_y, bn_ss, bn_mean, True, split, False, t_got, (y, coef))
    torch.cuda.synchronize()
    assert torch.equal(g0, r0) and torch.equal(gst, rst) and torch.equal(gcoef, rcoef)
    assert (g1 is None and r1 is None) or torch.equal(g1, r1)
    assert torch.equal(t_got[2], t_ref[2]) and torch.equal(t_got[3], t_ref[3])


def test_fused_bn_backward_equals_apply_pass_in_the_network():
    """One train-mode update() + backward of WT_PSE at the benchmark's geometry (B = 32, 256x256: the launches that take the fused
    path) with the BatchNorm-backward apply formed on load by both consumers (nn.BN_IN_X3) and with the stand-alone apply pass: every
    parameter gradient bitwise equal — it is the same arithmetic, only never written to HBM in between."""
    import torch.nn.functional as F
    from test_parity_gpu import build_nets
    from oracle.inputs import make_inputs, make_noise
    from wtpse_hip import nn as hnn
    B, pb, H = 32, 10, 256
    img, od, _ = make_inputs(41, B, H, H)
    eps = make_noise(42, (B, 1, H, H))
    grads = []
    calls = {True: 0, False: 0}
    orig = hnn.ops.conv_wgrad_r_bn
    for fused in (True, False):
        def counting(*a, _fused=fused, **k):
            calls[_fused] += 1
            return orig(*a, **k)
        main, _, _, _ = build_nets(pb)
        main.train(); main.zero_grad(); main.set_noise([eps])
        was = hnn.BN_IN_X3
        hnn.BN_IN_X3 = fused
        hnn.ops.conv_wgrad_r_bn = counting
        try:
            out, _, _, ins, dom = main.update(img.cuda(), od.cuda(), two_stage_inputs=img.cuda(), two_step=True)
            loss = F.binary_cross_entropy(torch.sigmoid(out), od.cuda()) + ins + dom
            loss.backward()
            torch.cuda.synchronize()
        finally:
            hnn.BN_IN_X3 = was
            hnn.ops.conv_wgrad_r_bn = orig
        grads.append({k: p.grad.detach().clone() for k, p in main.named_parameters() if p.grad is not None})
    assert calls[True] >= 20 and calls[False] == 0, calls          # the fused path was taken (both U-Nets' MFMA-bound layers) / not taken
    assert len(grads[0]) == len(grads[1]) > 100
    bad = [k for k in grads[0] if not torch.equal(grads[0][k], grads[1][k])]
    assert not bad, "%d of %d gradients differ, e.g. %s" % (len(bad), len(grads[0]), bad[:3])
