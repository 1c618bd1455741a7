#!/usr/bin/env python3
"""ONE bounded experiment on the round-3 'v_pk_fma_f32 finding' (DESIGN.md): the BatchNorm-backward epilogue of commit 6b61c00 in four
forms, 300 launches each on identical operands, counting launches whose masked gradient differs from the first launch's.
    v0  as it failed: the two decisions of a register fused by the compiler into v_pk_fma_f32 ... op_sel_hi:[1,0,0] (scale / shift broadcast)
    v1  the shipped guard (decision kept scalar)
    v2  v0's packed FMA, but every output store moved BEHIND all decisions (no store between the y loads and the FMAs that use them)
    v3  an explicit packed FMA on real register pairs (opaque copies of scale / shift): v_pk_fma_f32 WITHOUT operand-select modifiers
    v4  (round 5) ONE broadcast operand (scale a real pair, shift broadcast by op_sel): the form that ships in wgrad_r_k / head_bwd_k
The variants are built in the build container from the historical sources (tools/probe/_build/libpk_v*.so; recipe in DESIGN.md)."""
import ctypes
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "wt-pse-code_amd"), os.path.join(ROOT, "tests")]
vp = ctypes.c_void_p


def main():
    torch.cuda.init()
    dev = "cuda"
    g = torch.Generator().manual_seed(31)
    B, C, H, W = 20, 64, 32, 64
    y = torch.randn(B, C, H, W, generator=g).to(dev)
    du = torch.randn(B, C, H, W, generator=g).to(dev)
    w = (torch.randn(C, C, 3, 3, generator=g) * 0.2).to(dev)
    ss = torch.stack([torch.randn(C, generator=g) * 0.2 + 1.0, torch.randn(C, generator=g) * 0.3], 1).contiguous().to(dev)
    mean = (torch.randn(C, generator=g) * 0.1).to(dev)
    act = y.double() * ss[:, 0].double().view(1, -1, 1, 1) + ss[:, 1].double().view(1, -1, 1, 1)
    for v in (0, 1, 2, 3, 4):
        path = os.path.join(HERE, "_build", "libpk_v%d.so" % v)
        if not os.path.isfile(path):
            print("missing", path)
            continue
        dll = ctypes.CDLL(path)
        xf = ((C + 15) & ~15) * ((C + 31) & ~31) * 9 * 3
        packed = torch.zeros(2 * xf, dtype=torch.int16, device=dev)
        desc = torch.tensor([0, C, C, 9, 0, xf, 0, 0], dtype=torch.int32, device=dev)
        assert dll.wtpse_pack_conv_weights_x3(vp(w.data_ptr()), vp(desc.data_ptr()), 1, vp(packed.data_ptr()), None) == 0
        nblk = dll.wtpse_conv_x3_stats_blocks(B, H, W, C, 3)
        ref = None
        bad_launches, bad_elems, lanes = 0, 0, {}
        wrong_vs_fwd = 0
        for it in range(300):
            junk = torch.empty((1 << 22) + 4096 * it, device=dev).fill_(float(it))
            out = torch.empty(B, C, H, W, device=dev)
            stats = torch.empty(nblk, C, 2, device=dev)
            rc = dll.wtpse_dgrad_x3_bnb(vp(du.data_ptr()), C, vp(packed.data_ptr() + 2 * xf), vp(out.data_ptr()), None, C, vp(y.data_ptr()),
                                        vp(ss.data_ptr()), vp(mean.data_ptr()), 1, 0, C, vp(stats.data_ptr()), B, H, W, C, 3, None)
            assert rc == 0, rc
            torch.cuda.synchronize()
            if ref is None:
                ref = out.clone()
            d = out != ref
            n = int(d.sum())
            # against the forward pass's decision (sign of the exact fp64 value = sign of the single-rounding fmaf)
            wrong = ((out != 0) & ~(act > 0)) | ((out == 0) & (act > 0) & (ref != 0) & False)
            wrong_vs_fwd += int(wrong.sum())
            if n:
                bad_launches += 1
                bad_elems += n
                idx = d.nonzero()
                for b, c, yy, xx in idx.tolist()[:64]:
                    # lane of the element in the 32x32 accumulator layout of conv_x3_k<3,2,5>: pixel x & 31, channel bit 2 -> upper half
                    lane = (xx & 31) + 32 * ((c >> 2) & 1)
                    lanes[lane // 16] = lanes.get(lane // 16, 0) + 1
            del junk
        print("v%d: %d of 300 launches differ from the first (%d elements in all; by lane quarter 0-15/16-31/32-47/48-63: %s); elements kept "
              "although the forward decision was <= 0, summed over the launches: %d" % (v, bad_launches, bad_elems,
                                                                                       [lanes.get(q, 0) for q in range(4)], wrong_vs_fwd), flush=True)


if __name__ == "__main__":
    main()
