#!/usr/bin/env python3
"""Phase timeline of the x3 forward / data-gradient kernel from in-kernel s_memtime stamps (thread 0 of every workgroup).

Builds csrc/conv_x3.hip + conv.hip with -DWTPSE_STAMPS into tools/probe/_build/ (git-ignored; the product library never
carries the stamps) and prints, per workgroup: MFMA rows, weight-stash + barrier gaps, input-stash phases, epilogue; and for
one CU the interleaving of its resident workgroups.
    gpurun -- python tools/probe/x3_stamps.py [Cin Cout H B]
Stamp slots (per chunk c < 7, base 3 + 7c): +0/+2/+4 row 0/1/2 MFMAs issued, +1/+3/+(5->) past the row's barrier,
+5 input stash starts (last row), +6 input stash done; 0 start, 2 prologue done, 60 epilogue start, 61 end."""
import ctypes
import os
import subprocess
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "wt-pse-code_amd", "wtpse_hip", "csrc")
STAMPS = os.environ.get("NOSTAMPS", "0") == "0"          # NOSTAMPS=1: time the kernel (with PROBE_DEFS) without any stamp code
OUT = os.path.join(HERE, "_build", "libx3_%s%s.so" % ("stamps" if STAMPS else "plain",
                                                      "".join(c for c in os.environ.get("PROBE_DEFS", "") if c.isalnum())))
sys.path[:0] = [ROOT, os.path.join(ROOT, "wt-pse-code_amd")]


def build():
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-I", CSRC] + (["-DWTPSE_STAMPS"] if STAMPS else []) +
                          os.environ.get("PROBE_DEFS", "").split() + [
                           os.path.join(CSRC, "conv_x3.hip"), os.path.join(CSRC, "conv.hip"), "-o", OUT])


def main():
    cin, cout, hw, B = (int(v) for v in sys.argv[1:5]) if len(sys.argv) > 4 else (64, 64, 128, 32)
    if not os.path.isfile(OUT) or os.environ.get("REBUILD"):
        build()
    if not torch.cuda.is_available():
        print("built", OUT)
        return
    torch.cuda.init()
    dll = ctypes.CDLL(OUT)
    vp = ctypes.c_void_p
    dev = torch.device("cuda:0")
    from wtpse_hip import ops
    zeros = os.environ.get("ZEROS", "0") != "0"          # all-zero operands: the clock the chip holds without data toggling
    x = torch.zeros(B, cin, hw, hw, device=dev) if zeros else torch.randn(B, cin, hw, hw, device=dev)
    w = torch.zeros(cout, cin, 3, 3, device=dev) if zeros else torch.randn(cout, cin, 3, 3, device=dev) * 0.05
    xf = ops.x3_packed_size(cout, cin, 9)
    packed = torch.zeros(xf, dtype=torch.int16, device=dev)
    desc = torch.tensor([0, cout, cin, 9, 0, -1, 0, 0], dtype=torch.int32, device=dev)
    assert dll.wtpse_pack_conv_weights_x3(vp(w.data_ptr()), vp(desc.data_ptr()), 1, vp(packed.data_ptr()), None) == 0
    y = torch.empty(B, cout, hw, hw, device=dev)
    tiles = B * ((hw + 31) // 32) * ((hw + 7) // 8)
    mt = 2 if (cout % 64 == 0 and tiles * (cout // 64) >= 512) else 1
    nwg = tiles * ((cout + 32 * mt - 1) // (32 * mt))
    stamps = torch.zeros(nwg * 64, dtype=torch.int64, device=dev)

    want_stats, want_pro = os.environ.get("STATS", "0") != "0", os.environ.get("PRO", "0") != "0"     # as the forward pass runs it
    nrows = dll.wtpse_conv_x3_stats_blocks(B, hw, hw, cout, 3)
    stats_t = torch.zeros(nrows * cout * 2, device=dev) if want_stats else None
    pro_t = torch.stack([torch.rand(cin, device=dev) + 0.5, torch.randn(cin, device=dev) * 0.1], 1).contiguous() if want_pro else None

    def run():
        return dll.wtpse_conv_fwd_x3(vp(x.data_ptr()), cin, None, 0, vp(packed.data_ptr()), None, vp(pro_t.data_ptr()) if want_pro else None,
                                     None, 1 if want_pro else 0, vp(y.data_ptr()), None, cout, vp(stats_t.data_ptr()) if want_stats else None,
                                     B, hw, hw, cout, 3, 0, None, None)
    for _ in range(3):
        assert run() == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    nrep = int(os.environ.get("REPS", "200"))            # long enough for the clock to settle under the load
    for _ in range(nrep):
        run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / nrep
    print("%s operands, %d back-to-back launches: %.1f us, %.1f TFLOP/s fp32-equivalent = %.0f TFLOP/s of bf16 MFMA  [%s]" % (
        "zero" if zeros else "random", nrep, us, 2.0 * B * hw * hw * cin * cout * 9 / us * 1e-6,
        12.0 * B * hw * hw * cin * cout * 9 / us * 1e-6, os.environ.get("PROBE_DEFS", "") + (" +stats" if want_stats else "") + (" +prologue" if want_pro else "")))
    if not STAMPS:
        return
    assert dll.wtpse_probe_set_stamps_x3(vp(stamps.data_ptr())) == 0
    assert run() == 0
    torch.cuda.synchronize()
    st = stamps.cpu().numpy().reshape(nwg, 64)
    st = st[st[:, 0] != 0]
    nch = int(st[0, 62])
    hw_id = st[:, 1] & 0xFFFFFFFF
    xcc = (st[:, 1] >> 32) & 0xF
    key = ((xcc * 8 + ((hw_id >> 13) & 0x7)) * 2 + ((hw_id >> 12) & 0x1)) * 16 + ((hw_id >> 8) & 0xF)
    print("x3 forward %d->%d @%d B=%d: workgroups %d, chunks %d, MT %d, distinct CUs %d" % (cin, cout, hw, B, len(st), nch, mt, len(np.unique(key))))
    clk = (st[:, 61] - st[:, 0]) / np.maximum(st[:, 59] - st[:, 58], 1) * 0.1
    print("in-kernel clock (s_memtime / s_memrealtime over each workgroup): median %.2f GHz, p10 %.2f, p90 %.2f" % (
        np.median(clk), *np.percentile(clk, [10, 90])))
    mf_cycles = nch * 3 * (36 * mt) * 32                  # matrix-pipe cycles of one wave (one wave of a workgroup per SIMD)
    busy = [np.sum(key == k) * mf_cycles / (st[key == k][:, 61].max() - st[key == k][:, 0].min()) for k in np.unique(key)]
    print("matrix pipe busy per CU (MFMA cycles of its workgroups / its span): median %.0f %%" % (100 * np.median(busy)))
    n = min(nch, 7)
    ph = {"prologue": st[:, 2] - st[:, 0], "epilogue": st[:, 61] - st[:, 60], "workgroup": st[:, 61] - st[:, 0]}
    rows, gaps, xst = [], [], []
    for c in range(n):
        b = 3 + 7 * c
        prev = st[:, 2] if c == 0 else st[:, 3 + 7 * (c - 1) + 5]
        rows += [st[:, b] - prev, st[:, b + 2] - st[:, b + 1], st[:, b + 4] - st[:, b + 3]]
        gaps += [st[:, b + 1] - st[:, b], st[:, b + 3] - st[:, b + 2]]
        if c + 1 < nch:
            xst.append(st[:, b + 6] - st[:, b + 5])
            gaps.append(st[:, b + 5] - st[:, b + 4])          # last row issued -> every wave out of the chunk
    ph["mfma row (72 or 36 MFMAs)"] = np.concatenate(rows)
    ph["row gap (stash W + barrier)"] = np.concatenate(gaps)
    if xst:
        ph["input stash"] = np.concatenate(xst)
    for k, v in ph.items():
        print("%-28s mean %7.0f  p10 %7.0f  p50 %7.0f  p90 %7.0f cycles" % (k, v.mean(), *np.percentile(v, [10, 50, 90])))
    kk = np.unique(key)[0]
    r = st[key == kk]
    r = r[np.argsort(r[:, 0])]
    t0 = r[:, 0].min()
    print("one CU: %d workgroups over %d cycles" % (len(r), r[:, 61].max() - t0))
    for q in r[:6]:
        segs = []
        for c in range(n):
            b = 3 + 7 * c
            segs.append("c%d M%d g%d M%d g%d M%d x%d" % (c, q[b] - (q[2] if c == 0 else q[3 + 7 * (c - 1) + 5]), q[b + 1] - q[b], q[b + 2] - q[b + 1],
                                                        q[b + 3] - q[b + 2], q[b + 4] - q[b + 3], (q[b + 6] - q[b + 4]) if c + 1 < nch else 0))
        print("  start %7d end %7d | pro %d | %s | epi %d" % (q[0] - t0, q[61] - t0, q[2] - q[0], " ; ".join(segs), q[61] - q[60]))


if __name__ == "__main__":
    main()
