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
