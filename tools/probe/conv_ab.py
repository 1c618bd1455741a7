#!/usr/bin/env python3
"""Bitwise A/B of wtpse_conv_fwd between two builds of csrc/conv.hip (e.g. HEAD vs working tree) on random inputs.

    python tools/probe/conv_ab.py --build-rev HEAD        # in the build container: csrc/conv.hip of that git revision
                                                          # -> tools/probe/_build/libconv_old.so (travels with gpurun)
    gpurun -- python tools/probe/conv_ab.py tools/probe/_build/libconv_old.so wt-pse-code_amd/wtpse_hip/libwtpse_hip.so

Every fast-path rewrite of the forward kernel in round 1 was accepted only when this printed "bitwise identical"
(outputs and BatchNorm partials) against the previous commit."""
import ctypes
import sys

import torch

if len(sys.argv) > 2 and sys.argv[1] == "--build-rev":
    import os
    import subprocess
    import tempfile
    here = os.path.dirname(os.path.abspath(__file__))
    root = os.path.dirname(os.path.dirname(here))
    tmp = tempfile.mkdtemp()
    for f in ("conv.hip", "common.h"):
        open(os.path.join(tmp, f), "wb").write(subprocess.check_output(
            ["git", "-C", root, "show", "%s:wt-pse-code_amd/wtpse_hip/csrc/%s" % (sys.argv[2], f)]))
    os.makedirs(os.path.join(here, "_build"), exist_ok=True)
    out = os.path.join(here, "_build", "libconv_old.so")
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-I", tmp,
                           os.path.join(tmp, "conv.hip"), "-o", out])
    print("built", out, "from", sys.argv[2])
    sys.exit(0)

vp = ctypes.c_void_p
dev = torch.device("cuda:0")
torch.cuda.init()
libs = [ctypes.CDLL(p) for p in sys.argv[1:3]]


def run(lib, case, seed):
    B, C0, C1, Co, H, W, k, bias, stats, relu, mask, pro, split = case
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g).to(dev)
    x0, x1 = r(B, C0, H, W), (r(B, C1, H, W) if C1 else None)
    w = r(Co, C0 + C1, k, k) * 0.2
    bv = r(Co) if bias else None
    cinp, coutp = (C0 + C1 + 3) & ~3, (Co + 15) & ~15
    packed = torch.zeros(cinp * k * k * coutp + 64, device=dev)
    desc = torch.tensor([0, Co, C0 + C1, k * k, 0, -1, 0, 0], dtype=torch.int32, device=dev)
    assert lib.wtpse_pack_conv_weights(vp(w.data_ptr()), vp(desc.data_ptr()), 1, vp(packed.data_ptr()), None) == 0
    csplit = split if split else Co
    y0 = torch.full((B, csplit, H, W), 7.0, device=dev)
    y1 = torch.full((B, Co - csplit, H, W), 7.0, device=dev) if split else None
    nblk = lib.wtpse_conv_stats_blocks(B, H, W)
    st = torch.full((nblk, Co, 2), 7.0, device=dev) if stats else None
    mk = r(B, Co, H, W) if mask else None
    p0 = torch.stack([r(C0) * 0.5 + 1, r(C0)], 1).contiguous() if pro else None
    p1 = torch.stack([r(C1) * 0.5 + 1, r(C1)], 1).contiguous() if (pro and C1) else None
    P = lambda t: vp(t.data_ptr()) if t is not None else None
    rc = lib.wtpse_conv_fwd(P(x0), C0, P(x1), C1, P(packed), P(bv), P(p0), P(p1), 1 if pro else 0, P(y0), P(y1), csplit, P(st),
                            B, H, W, Co, k, int(relu), P(mk), None)
    assert rc == 0, rc
    torch.cuda.synchronize()
    return [t.clone() for t in (y0, y1, st) if t is not None]


CASES = [
    # B C0 C1 Co H W k bias stats relu mask pro split
    (2, 16, 0, 16, 64, 64, 3, 1, 1, 0, 0, 0, 0), (2, 3, 0, 16, 64, 64, 3, 1, 1, 0, 0, 0, 0), (6, 16, 0, 32, 32, 32, 3, 1, 1, 0, 0, 1, 0),
    (6, 32, 0, 32, 32, 32, 3, 1, 1, 0, 0, 1, 0), (6, 64, 0, 64, 16, 16, 3, 1, 1, 0, 0, 1, 0), (6, 128, 0, 256, 4, 4, 3, 1, 1, 0, 0, 1, 0),
    (6, 256, 0, 256, 4, 4, 3, 1, 1, 0, 0, 1, 0), (6, 256, 0, 128, 8, 8, 1, 1, 1, 0, 0, 1, 0), (6, 128, 128, 256, 8, 8, 3, 1, 1, 0, 0, 1, 0),
    (6, 16, 16, 32, 64, 64, 3, 1, 1, 0, 0, 1, 0), (6, 32, 0, 32, 64, 64, 1, 1, 0, 1, 0, 1, 0), (6, 32, 0, 8, 64, 64, 1, 1, 0, 1, 0, 0, 0),
    (6, 8, 0, 1, 64, 64, 1, 1, 0, 0, 0, 0, 0), (6, 32, 0, 32, 64, 64, 3, 0, 0, 0, 0, 0, 16), (6, 256, 0, 256, 8, 8, 3, 0, 0, 0, 0, 0, 128),
    (6, 8, 0, 32, 64, 64, 1, 0, 0, 0, 1, 0, 0), (6, 16, 0, 16, 64, 64, 3, 0, 0, 0, 1, 0, 0), (6, 1, 0, 8, 64, 64, 1, 0, 0, 0, 1, 0, 0),
    (6, 16, 0, 16, 64, 64, 3, 1, 0, 1, 0, 0, 0), (2, 40, 0, 96, 12, 20, 3, 1, 1, 0, 0, 0, 0), (6, 32, 0, 16, 64, 64, 1, 1, 1, 0, 0, 1, 0),
]
bad = 0
for i, c in enumerate(CASES):
    a, b = run(libs[0], c, 100 + i), run(libs[1], c, 100 + i)
    for j, (u, v) in enumerate(zip(a, b)):
        same = torch.equal(u, v)
        if not same:
            bad += 1
            d = (u - v).abs()
            print("case", c, "output", j, "DIFFERS: max", float(d.max()), "n", int((d > 0).sum()), "of", d.numel(), "scale", float(u.abs().max()))
            if d.dim() == 3:
                print("   per column (sum, sumsq):", (d > 0).sum((0, 1)).tolist(), " channels hit:", (d > 0).any(0).any(1).nonzero().flatten().tolist()[:40])
print("bitwise identical" if not bad else "%d outputs differ" % bad)
