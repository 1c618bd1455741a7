"""TEST INFRASTRUCTURE: random-projection fingerprints of gradient tensors, so that the distance of a gradient computed at test time
from an fp64 oracle gradient computed OFFLINE (minutes of CPU, tens of GB) can be measured without committing the oracle tensor.

For a tensor g (n elements) and K Rademacher vectors r_k (entries +-1, generated from a seed with numpy's RandomState — stable across
machines and versions):  s_k = r_k . g.  For any h:  (1/K) sum_k (r_k . h - s_k)^2  is an unbiased estimate of |h - g|^2 with relative
standard deviation sqrt(2/K) (K = 128: 12.5 % on the square, 6 % on the norm).  Tensors with at most SMALL elements are stored whole
(exact distance)."""
import numpy as np
import torch

K = 128
SMALL = 4096
CHUNK = 1 << 16


def signs(seed, k, n0, n1):
    """Rows k of the +-1 matrix restricted to columns [n0, n1) — generated per column chunk so that any consumer can stream."""
    rs = np.random.RandomState((seed * 1000003 + n0 // CHUNK) % (2 ** 31 - 1))
    return rs.randint(0, 2, size=(k, n1 - n0), dtype=np.int8).astype(np.float64) * 2.0 - 1.0


def project(t, seed, k=K):
    """-> float64 [k]: the projections of the flattened tensor `t` (torch, any device / float dtype)."""
    g = t.detach().reshape(-1).double().cpu()
    n = g.numel()
    out = np.zeros(k, np.float64)
    for n0 in range(0, n, CHUNK):
        n1 = min(n, n0 + CHUNK)
        out += signs(seed, k, n0, n1) @ g[n0:n1].numpy()
    return out


def fingerprint(g64, seed):
    """What the fixture keeps of an fp64 oracle gradient: (n, |g|^2, the tensor itself if small else its K projections)."""
    g = g64.detach().reshape(-1).double().cpu()
    n = g.numel()
    return {"n": n, "norm2": float((g * g).sum()), "data": g.numpy().copy() if n <= SMALL else project(g, seed)}


def distance2(h, fp, seed):
    """|h - g|^2 for the oracle gradient g behind fingerprint `fp` (exact for small tensors, estimated otherwise)."""
    hh = h.detach().reshape(-1).double().cpu()
    assert hh.numel() == fp["n"], (hh.numel(), fp["n"])
    if fp["n"] <= SMALL:
        d = hh.numpy() - fp["data"]
        return float((d * d).sum())
    d = project(hh, seed) - fp["data"]
    return float((d * d).mean())
