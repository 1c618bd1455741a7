"""TEST INFRASTRUCTURE (never imported by the product path): brute-force surface distances on small binary masks, an
independent restatement of what medpy 0.5.2's `metric.binary.asd / hd95` compute (reference call sites: Trainer.py:226-239;
medpy is absent from /root/reference and from this image: **parity unpinned**, see wtpse_hip/validate.py).

Definitions used (medpy's published algorithm): a surface pixel is an object pixel with at least one of its four edge
neighbours outside the object (pixels beyond the image count as outside: scipy's binary_erosion, border_value 0); the distance
of a surface pixel to the other mask is the Euclidean distance to that mask's nearest surface pixel.  Pure Python loops: small
cases only."""
import math


def surface(mask):
    h, w = len(mask), len(mask[0])
    out = []
    for y in range(h):
        for x in range(w):
            if not mask[y][x]:
                continue
            for dy, dx in ((-1, 0), (1, 0), (0, -1), (0, 1)):
                yy, xx = y + dy, x + dx
                if yy < 0 or yy >= h or xx < 0 or xx >= w or not mask[yy][xx]:
                    out.append((y, x))
                    break
    return out


def surface_distances(result, reference):
    sr, sf = surface(result), surface(reference)
    if not sr or not sf:
        raise RuntimeError("empty mask")
    return [min(math.hypot(y - v, x - u) for v, u in sf) for y, x in sr]


def asd(result, reference):
    d = surface_distances(result, reference)
    return sum(d) / len(d)


def hd95(result, reference):
    """numpy.percentile(..., 95) with linear interpolation, written out"""
    d = sorted(surface_distances(result, reference) + surface_distances(reference, result))
    pos = 0.95 * (len(d) - 1)
    lo = int(math.floor(pos))
    hi = min(lo + 1, len(d) - 1)
    return d[lo] + (d[hi] - d[lo]) * (pos - lo)
