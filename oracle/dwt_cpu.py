"""TEST INFRASTRUCTURE ONLY — self-defined CPU definition of a 2-D discrete wavelet transform (numpy).

**Parity unpinned, and NOT part of WT-PSE.**  The reference (tonyckc/WT-PSE-code) contains no wavelet transform: its "WT" is
the whitening transform.  BASELINE.json's wording nevertheless names a "2-D DWT analysis/synthesis filter
bank (Haar/Db lifting)" and a "4-level DWT" stress configuration; SURVEY.md §8f-4 lists it as the last "next" row, to be
built only as a standalone HBM-bandwidth micro-benchmark with a self-defined specification.  This file IS that
specification; csrc/dwt.hip is checked against it.  It is never wired into update()/predict().

Specification
  * separable, orthonormal, periodic extension; per level: horizontal (along W) pass, then vertical (along H) pass
  * wavelets by lifting, n = 0..N/2-1, e[n] = x[2n], o[n] = x[2n+1], indices mod N/2:
      haar: d1 = o - e;  s1 = e + d1/2;  s = sqrt2*s1;  d = d1/sqrt2                    ( s=(e+o)/sqrt2, d=(o-e)/sqrt2 )
      db2 : d1[n] = o[n] - sqrt3*e[n]
            s1[n] = e[n] + (sqrt3/4)*d1[n] + ((sqrt3-2)/4)*d1[n+1]
            d2[n] = d1[n] + s1[n-1]
            s = ((sqrt3+1)/sqrt2)*s1;  d = ((sqrt3-1)/sqrt2)*d2
    (Daubechies 4-tap: orthonormal, two vanishing moments — checked in tests/test_dwt.py)
  * Mallat layout, same shape as the input: after a level on an h x w region, [0:h/2, 0:w/2] = LL (low/low),
    [0:h/2, w/2:w] = high horizontal / low vertical, [h/2:h, 0:w/2] = low horizontal / high vertical, [h/2:h, w/2:w] = HH;
    the next level transforms LL in place.  H and W must be divisible by 2^levels.
"""
import numpy as np

R3 = np.sqrt(3.0)
A, B = R3 / 4.0, (R3 - 2.0) / 4.0
C1, C2 = (R3 + 1.0) / np.sqrt(2.0), (R3 - 1.0) / np.sqrt(2.0)


def lift_fwd(x, wavelet, axis):
    x = np.moveaxis(x, axis, -1)
    e, o = x[..., 0::2], x[..., 1::2]
    if wavelet == "haar":
        s, d = (e + o) / np.sqrt(2.0), (o - e) / np.sqrt(2.0)
    elif wavelet == "db2":
        d1 = o - R3 * e
        s1 = e + A * d1 + B * np.roll(d1, -1, axis=-1)
        d2 = d1 + np.roll(s1, 1, axis=-1)
        s, d = C1 * s1, C2 * d2
    else:
        raise ValueError(wavelet)
    return np.moveaxis(np.concatenate([s, d], axis=-1), -1, axis)


def lift_inv(y, wavelet, axis):
    y = np.moveaxis(y, axis, -1)
    n = y.shape[-1] // 2
    s, d = y[..., :n], y[..., n:]
    if wavelet == "haar":
        e, o = (s - d) / np.sqrt(2.0), (s + d) / np.sqrt(2.0)
    elif wavelet == "db2":
        s1, d2 = s / C1, d / C2
        d1 = d2 - np.roll(s1, 1, axis=-1)
        e = s1 - A * d1 - B * np.roll(d1, -1, axis=-1)
        o = d1 + R3 * e
    else:
        raise ValueError(wavelet)
    out = np.empty_like(y)
    out[..., 0::2], out[..., 1::2] = e, o
    return np.moveaxis(out, -1, axis)


def dwt2(x, wavelet="haar", levels=1):
    """x [..., H, W] -> Mallat-layout coefficients of the same shape (float64 arithmetic)."""
    out = np.array(x, dtype=np.float64, copy=True)
    H, W = out.shape[-2:]
    assert H % (1 << levels) == 0 and W % (1 << levels) == 0
    h, w = H, W
    for _ in range(levels):
        reg = lift_fwd(out[..., :h, :w], wavelet, -1)
        out[..., :h, :w] = lift_fwd(reg, wavelet, -2)
        h, w = h // 2, w // 2
    return out


def idwt2(c, wavelet="haar", levels=1):
    out = np.array(c, dtype=np.float64, copy=True)
    H, W = out.shape[-2:]
    for lv in reversed(range(levels)):
        h, w = H >> lv, W >> lv
        reg = lift_inv(out[..., :h, :w], wavelet, -2)
        out[..., :h, :w] = lift_inv(reg, wavelet, -1)
    return out
