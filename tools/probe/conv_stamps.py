#!/usr/bin/env python3
"""Phase timeline of the forward conv kernel (dominant shape 64->64 3x3 @128x128, B=32) from in-kernel s_memtime stamps.

Builds csrc/conv.hip with -DWTPSE_STAMPS into tools/probe/_build/ (git-ignored; the product library never carries the
stamps), runs the shape once and prints, for the workgroups that shared a few CUs, when each phase of each input-channel
chunk started.  Used to find where the matrix pipe idles ("where the conv kernel's time goes").
    gpurun -- python tools/probe/conv_stamps.py [Cin Cout H B]
"""
import ctypes
import os
import subprocess
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "wt-pse-code_amd", "wtpse_hip", "csrc")
OUT = os.path.join(HERE, "_build", "libconv_stamps.so")


def build():
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DWTPSE_STAMPS",
                           os.path.join(CSRC, "conv.hip"), "-o", OUT])


def report(st, nch, what):
    st = st[st[:, 0] != 0]
    hw_id = st[:, 1] & 0xFFFFFFFF
    xcc = (st[:, 1] >> 32) & 0xF
    key = ((xcc * 8 + ((hw_id >> 13) & 0x7)) * 2 + ((hw_id >> 12) & 0x1)) * 16 + ((hw_id >> 8) & 0xF)
    t0 = st[:, 0].min()
    print("%s: workgroups %d, distinct CUs %d" % (what, len(st), len(np.unique(key))))
    dur = {"load+stash": [], "barrier": [], "mfma": [], "epilogue": [], "wg": []}
    for r in st:
        n = min(nch, int(r[62])) if r[62] else nch
        for c in range(n):
            dur["load+stash"].append(r[3 + 4 * c] - r[2 + 4 * c])
            dur["barrier"].append(r[4 + 4 * c] - r[3 + 4 * c])
            dur["mfma"].append(r[5 + 4 * c] - r[4 + 4 * c])
        dur["epilogue"].append(r[61] - r[60])
        dur["wg"].append(r[61] - r[0])
    for k, v in dur.items():
        v = np.array(v)
        print("%-11s mean %7.0f  p10 %7.0f  p50 %7.0f  p90 %7.0f cycles" % (k, v.mean(), *np.percentile(v, [10, 50, 90])))
    for kk in np.unique(key)[:1]:
        rows = st[key == kk]
        rows = rows[np.argsort(rows[:, 0])]
        print("CU key %d: %d workgroups, span %d" % (kk, len(rows), rows[:, 61].max() - rows[:, 0].min()))
        for r in rows[:8]:
            n = min(nch, int(r[62])) if r[62] else nch
            segs = " ".join("%d:%d/%d/%d" % (c, r[3 + 4 * c] - r[2 + 4 * c], r[4 + 4 * c] - r[3 + 4 * c], r[5 + 4 * c] - r[4 + 4 * c])
                            for c in range(min(n, 8)))
            print("  start %7d end %7d tiles %d | load+stash/barrier/mfma %s | epi %d" %
                  (r[0] - rows[:, 0].min(), r[61] - rows[:, 0].min(), r[62], segs, r[61] - r[60]))


def wgrad(dll, cin, cout, hw, B):
    dev = torch.device("cuda:0")
    vp = ctypes.c_void_p
    x = torch.randn(B, cin, hw, hw, device=dev)
    dy = torch.randn(B, cout, hw, hw, device=dev)
    ks = dll.wtpse_wgrad_ksplit(B, hw, hw, cin, cout)
    slab = torch.zeros(ks * cout * cin * 9, device=dev)
    dbs = torch.zeros(ks * cout, device=dev)
    dw = torch.zeros(cout * cin * 9, device=dev)
    db = torch.zeros(cout, device=dev)
    MB = 32 if cout > 16 else 16
    cg = min(cin, MB)
    nwg = ((cout + MB - 1) // MB) * ((cin + cg - 1) // cg) * ks
    stamps = torch.zeros(nwg * 64, dtype=torch.int64, device=dev)

    def run():
        return dll.wtpse_conv_wgrad(vp(dy.data_ptr()), vp(x.data_ptr()), cin, None, 0, None, None, 0, vp(slab.data_ptr()),
                                    vp(dbs.data_ptr()), ks, vp(dw.data_ptr()), vp(db.data_ptr()), 0, B, hw, hw, cout, 3, None)
    for _ in range(3):
        assert run() == 0
    torch.cuda.synchronize()
    assert dll.wtpse_probe_set_stamps(vp(stamps.data_ptr())) == 0
    assert run() == 0
    torch.cuda.synchronize()
    assert dll.wtpse_probe_set_stamps(None) == 0
    print("ksplit %d" % ks)
    st = stamps.cpu().numpy().reshape(-1, 64)
    st = st[st[:, 0] != 0]
    print("wgrad %d->%d @%d B=%d: %d workgroups; per tile (first 8 tiles of every workgroup), cycles:" % (cin, cout, hw, B, len(st)))
    names = ["issue loads half 0", "wait+stash half 0", "issue loads half 1", "wait+stash half 1", "barrier", "mfma loop"]
    cols = [[] for _ in names]
    for r in st:
        for t in range(min(8, int(r[62]))):
            for q in range(6):
                cols[q].append(r[3 + 7 * t + q] - r[2 + 7 * t + q])
    for n, v in zip(names, cols):
        v = np.array(v)
        print("  %-20s mean %7.0f  p10 %7.0f  p50 %7.0f  p90 %7.0f" % (n, v.mean(), *np.percentile(v, [10, 50, 90])))
    hw_id = st[:, 1] & 0xFFFFFFFF
    key = ((((st[:, 1] >> 32) & 0xF) * 8 + ((hw_id >> 13) & 0x7)) * 2 + ((hw_id >> 12) & 0x1)) * 16 + ((hw_id >> 8) & 0xF)
    rows = st[key == np.unique(key)[0]]
    t0 = rows[:, 0].min()
    print("  timeline of the workgroups on one CU (cycles from the first start; L = load phase start, M = MFMA loop start, E = MFMA loop end):")
    for r in rows[np.argsort(rows[:, 0])]:
        print("   wave slot %d:" % (r[1] & 0xF), " ".join("L%d M%d E%d |" % (r[2 + 7 * t] - t0, r[7 + 7 * t] - t0, r[8 + 7 * t] - t0) for t in range(min(8, int(r[62])))))
    tot = np.array([r[61] - r[0] for r in st])
    print("  workgroup life mean %.0f, epilogue mean %.0f, tiles per workgroup %d" % (tot.mean(), np.mean([r[61] - r[60] for r in st]), st[0][62]))


def main():
    args = [a for a in sys.argv[1:] if a not in ("wgrad", "k1")]
    cin, cout, hw, B = [int(v) for v in (args[:4] + ["64", "64", "128", "32"][len(args):])]
    ks = 1 if "k1" in sys.argv else 3
    if "wgrad" in sys.argv:
        if not os.path.isfile(OUT):
            build()
        torch.cuda.init()
        return wgrad(ctypes.CDLL(OUT), cin, cout, hw, B)
    if not os.path.isfile(OUT):
        build()
    torch.cuda.init()
    dll = ctypes.CDLL(OUT)
    dev = torch.device("cuda:0")
    x = torch.randn(B, cin, hw, hw, device=dev)
    w = torch.randn(cout, cin, ks, ks, device=dev) * 0.05
    bias = torch.randn(cout, device=dev)
    cinp, coutp = (cin + 3) & ~3, (cout + 15) & ~15
    packed = torch.zeros(cinp * ks * ks * coutp + 64, device=dev)
    desc = torch.tensor([0, cout, cin, ks * ks, 0, -1, 0, 0], dtype=torch.int32, device=dev)
    vp = ctypes.c_void_p
    assert dll.wtpse_pack_conv_weights(vp(w.data_ptr()), vp(desc.data_ptr()), 1, vp(packed.data_ptr()), None) == 0
    y = torch.empty(B, cout, hw, hw, device=dev)
    nblk = dll.wtpse_conv_stats_blocks(B, hw, hw)
    stats = torch.zeros(nblk * cout * 2, device=dev)
    ngrid = nblk * ((coutp + 15) // 16)           # upper bound on workgroups (16-cout blocks)
    stamps = torch.zeros(ngrid * 64, dtype=torch.int64, device=dev)

    def run():
        return dll.wtpse_conv_fwd(vp(x.data_ptr()), cin, None, 0, vp(packed.data_ptr()), vp(bias.data_ptr()), None, None, 0,
                                  vp(y.data_ptr()), None, cout, vp(stats.data_ptr()) if ks == 3 else None, B, hw, hw, cout, ks, 0, None, None)
    for _ in range(3):
        assert run() == 0
    torch.cuda.synchronize()
    assert dll.wtpse_probe_set_stamps(vp(stamps.data_ptr())) == 0
    assert run() == 0
    torch.cuda.synchronize()
    st = stamps.cpu().numpy().reshape(-1, 64)
    st = st[st[:, 0] != 0]
    hw_id = st[:, 1] & 0xFFFFFFFF
    xcc = (st[:, 1] >> 32) & 0xF
    cu = (hw_id >> 8) & 0xF
    sh = (hw_id >> 12) & 0x1
    se = (hw_id >> 13) & 0x7
    key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    t0 = st[:, 0].min()
    print("workgroups %d, distinct CUs %d, kernel span %d cycles" % (len(st), len(np.unique(key)), st[:, 61].max() - t0))
    nch = (cinp + 7) // 8 if cout > 16 else (cinp + 15) // 16
    dur = {"load+stash": [], "barrier": [], "mfma": [], "epilogue": [], "wg": []}
    for r in st:
        for c in range(nch):
            dur["load+stash"].append(r[3 + 4 * c] - r[2 + 4 * c])
            dur["barrier"].append(r[4 + 4 * c] - r[3 + 4 * c])
            dur["mfma"].append(r[5 + 4 * c] - r[4 + 4 * c])
        dur["epilogue"].append(r[61] - r[60])
        dur["wg"].append(r[61] - r[0])
    for k, v in dur.items():
        v = np.array(v)
        print("%-11s mean %7.0f  p10 %7.0f  p50 %7.0f  p90 %7.0f cycles" % (k, v.mean(), *np.percentile(v, [10, 50, 90])))
    # timeline of one CU per XCD 0/1
    for kk in np.unique(key)[:2]:
        rows = st[key == kk]
        rows = rows[np.argsort(rows[:, 0])]
        print("CU key %d: %d workgroups" % (kk, len(rows)))
        for r in rows:
            segs = " ".join("%d:%d/%d/%d" % (c, r[3 + 4 * c] - r[2 + 4 * c], r[4 + 4 * c] - r[3 + 4 * c], r[5 + 4 * c] - r[4 + 4 * c])
                            for c in range(nch))
            print("  start %7d end %7d simd %d | chunk:load+stash/barrier/mfma %s | epi %d" %
                  (r[0] - t0, r[61] - t0, (r[1] >> 4) & 3, segs, r[61] - r[60]))


if __name__ == "__main__":
    main()
