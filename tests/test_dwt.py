"""The standalone 2-D DWT micro-benchmark (NOT part of WT-PSE, parity unpinned: the reference has no
wavelet transform).  CPU: the self-defined specification oracle/dwt_cpu.py has the properties it claims (perfect
reconstruction, orthonormality, vanishing moments).  GPU: csrc/dwt.hip against that specification, plus the size-independent
properties at BASELINE.json's shapes (configs[4]: 512x512, 4 levels)."""
import numpy as np
import pytest
import torch

from oracle import dwt_cpu as D


@pytest.mark.parametrize("wavelet", ["haar", "db2"])
def test_specification_properties(wavelet):
    x = np.random.RandomState(0).randn(2, 3, 32, 64)
    for lv in (1, 2, 4):
        c = D.dwt2(x, wavelet, lv)
        assert np.abs(D.idwt2(c, wavelet, lv) - x).max() < 1e-12                     # perfect reconstruction
        assert abs(np.sum(c ** 2) - np.sum(x ** 2)) < 1e-9 * np.sum(x ** 2)          # orthonormal: energy preserved
    # vanishing moments: constants (both) and linear ramps (db2) give zero detail coefficients away from the periodic seam
    const = D.lift_fwd(np.ones(64), wavelet, -1)
    assert np.abs(const[32:]).max() < 1e-12
    if wavelet == "db2":
        ramp = D.lift_fwd(np.arange(64.0), wavelet, -1)
        assert np.abs(ramp[32 + 2:64 - 2]).max() < 1e-10
    # a product of 1-D transforms: the 2-D level equals rows-then-columns
    y = D.lift_fwd(D.lift_fwd(x, wavelet, -1), wavelet, -2)
    assert np.abs(y - D.dwt2(x, wavelet, 1)).max() < 1e-12


@pytest.mark.gpu
@pytest.mark.parametrize("wavelet", ["haar", "db2"])
@pytest.mark.parametrize("shape,levels", [((2, 3, 32, 64), 1), ((1, 2, 64, 32), 3), ((2, 2, 128, 256), 4), ((1, 1, 16, 16), 4),
                                          ((1, 2, 48, 80), 2),
                                          # square 256 / 512 planes: the streaming / fused analysis kernels (rows owned by a wave,
                                          # the levels after the first in LDS)
                                          ((2, 3, 256, 256), 1), ((3, 2, 256, 256), 3), ((1, 2, 256, 256), 7),
                                          ((1, 2, 512, 512), 1), ((2, 1, 512, 512), 2), ((1, 3, 512, 512), 4)])
def test_gpu_dwt_vs_specification(wavelet, shape, levels):
    from wtpse_hip import dwt
    g = torch.Generator().manual_seed(3)
    x = torch.randn(*shape, generator=g)
    ref = D.dwt2(x.numpy(), wavelet, levels)
    c = dwt.dwt2(x.cuda(), wavelet, levels)
    err = np.abs(c.cpu().numpy() - ref).max()
    assert err < 2e-5 * max(1.0, np.abs(ref).max()), err
    y = dwt.idwt2(torch.from_numpy(ref).float().cuda(), wavelet, levels)
    assert float((y.cpu() - x).abs().max()) < 2e-5 * max(1.0, float(x.abs().max()))


@pytest.mark.gpu
@pytest.mark.parametrize("wavelet", ["haar", "db2"])
@pytest.mark.parametrize("shape,levels", [((32, 16, 256, 256), 3), ((16, 16, 512, 512), 4)])
def test_gpu_dwt_full_size_properties(wavelet, shape, levels):
    """configs[2] / configs[4] shapes: round trip, energy preservation, linearity."""
    from wtpse_hip import dwt
    g = torch.Generator(device="cuda").manual_seed(4)
    x = torch.randn(*shape, generator=g, device="cuda")
    c = dwt.dwt2(x, wavelet, levels)
    y = dwt.idwt2(c, wavelet, levels)
    assert float((y - x).abs().max()) < 5e-5
    ex, ec = float(x.double().pow(2).sum()), float(c.double().pow(2).sum())
    assert abs(ex - ec) < 1e-5 * ex
    x2 = torch.randn(*shape, generator=g, device="cuda")
    lin = dwt.dwt2((x + 0.5 * x2).contiguous(), wavelet, levels) - (c + 0.5 * dwt.dwt2(x2, wavelet, levels))
    assert float(lin.abs().max()) < 5e-5
