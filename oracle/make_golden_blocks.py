#!/usr/bin/env python3
"""Regenerate the four BLOCK fixtures of tests/golden/blocks.npz (ConvD / ConvU, first and inner: SURVEY.md a-1 / a-2) by running the
REFERENCE's own modules on CPU (build container only; TEST INFRASTRUCTURE — the same recipe as oracle/make_golden.py, whose other
entries of the file — DeepWT, attention — are kept as they are), with one addition (round 5): **the fixture must not sit on a kink**.

A block test compares gradients element by element at 1e-3; a ReLU unit whose BatchNorm output is within fp32 rounding of zero
(round 4's `convu_first` fixture had one at 3.9e-7) makes that comparison a coin flip between ANY two correct fp32 implementations:
whichever side the unit falls on switches the gradient of its 3x3 footprint on or off (measured: the x2h arithmetic, whose
convolution error against fp64 is that of the fp32-input MFMA, flipped it and failed 218 elements, all inside that footprint —
profiles/r05_x2h_parity.md).  So the input seeds are searched (offset t = 0, 1, 2, ... added as 1000 t) until every ReLU'd BatchNorm
output of the reference run has |z| > KINK_MARGIN and every 2x2 max-pool window a winner by more than KINK_MARGIN; t is stored in the
fixture (`<block>.seed_t`) and read back by tests/test_parity_gpu.py::test_blocks_vs_golden."""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from oracle import ref_import  # noqa: E402
from oracle.filler import fill_state_dict  # noqa: E402
from oracle.inputs import make_noise  # noqa: E402
from oracle.make_golden import OUT, SEED_W, np_, pack  # noqa: E402

KINK_MARGIN = 2e-5


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    _, alg, _ = ref_import.load()
    path = os.path.join(OUT, "blocks.npz")
    fx = dict(np.load(path))
    B, H = 4, 16
    blocks = [
        ("convd_first", lambda: alg.ConvD(3, 16, "bn", first=True), (B, 3, H, H), None),
        ("convd", lambda: alg.ConvD(16, 32, "bn"), (B, 16, H, H), None),
        ("convu_first", lambda: alg.ConvU(64, "bn", first=True), (B, 64, H // 2, H // 2), (B, 32, H, H)),
        ("convu", lambda: alg.ConvU(32, "bn"), (B, 64, H // 2, H // 2), (B, 16, H, H)),
    ]
    for bi, (name, ctor, xs, ps) in enumerate(blocks):
        for t in range(64):
            mod = ctor()
            fill_state_dict(mod, SEED_W + 20 + bi)
            mod.train()
            margins = []
            relu_bns = [n for n in ("bn1", "bn2", "bn3") if hasattr(mod, n) and not (n == "bn1" and isinstance(mod, alg.ConvD))]
            hooks = [getattr(mod, n).register_forward_hook(lambda m, i, o: margins.append(float(o.detach().abs().min()))) for n in relu_bns]
            x = make_noise(300 + bi + 1000 * t, xs).requires_grad_(True)
            args = [x]
            if ps is not None:
                prev = make_noise(400 + bi + 1000 * t, ps).requires_grad_(True)
                args.append(prev)
            if isinstance(mod, alg.ConvD) and not mod.first:      # the max-pool in front: winner of every window by more than the margin
                w = F.unfold(x.detach(), 2, stride=2).view(xs[0], xs[1], 4, -1).sort(2, descending=True).values
                margins.append(float((w[:, :, 0] - w[:, :, 1]).min()))
            y = mod(*args)
            for h in hooks:
                h.remove()
            assert len(margins) >= len(relu_bns)
            if min(margins) > KINK_MARGIN:
                break
        else:
            raise RuntimeError("no kink-free seed found for " + name)
        wgt = make_noise(500 + bi + 1000 * t, y.shape)
        (y * wgt).sum().backward()
        d = dict(y=np_(y), dx=np_(x.grad), seed_t=np.array(t), kink_margin=np.array(min(margins)))
        if ps is not None:
            d["dprev"] = np_(prev.grad)
        for n, p in mod.named_parameters():
            d["g." + n] = np_(p.grad)
        for n, b in mod.named_buffers():
            d["buf." + n] = np_(b)
        mod.eval()
        with torch.no_grad():
            d["y_eval"] = np_(mod(*[a.detach() for a in args]))
        for k in [k for k in fx if k.startswith(name + ".")]:
            del fx[k]
        fx.update(pack(name + ".", d))
        print("%-12s seed offset t = %d, smallest |ReLU pre-activation| / max-pool lead %.2e" % (name, t, min(margins)))
    np.savez_compressed(path, **fx)
    print("blocks.npz", len(fx))


if __name__ == "__main__":
    main()
