#!/usr/bin/env python3
"""Throughput of the device-side input pipeline against the reference's host path (Pillow, one thread per sample).

    gpurun -- python tools/bench_input_pipeline.py [--batch 32] [--in-size 800]
"""
import argparse
import os
import random
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "wt-pse-code_amd")]
from wtpse_hip.input_pipeline import DeviceInputPipeline, draw  # noqa: E402


def pillow_path(img, od, d, S):
    """What the reference does per sample (custom_transforms.py), with Pillow itself."""
    from PIL import Image
    nw, nh, x1, y1 = d
    im, m = Image.fromarray(img).resize((S, S)), Image.fromarray(od).resize((S, S))
    if (nw, nh) != (S, S):
        im, m = im.resize((nw, nh), Image.BILINEAR), m.resize((nw, nh), Image.NEAREST)
    im, m = im.crop((x1, y1, x1 + S, y1 + S)), m.crop((x1, y1, x1 + S, y1 + S))
    a = np.array(im).astype(np.float32)
    a /= 127.5
    a -= 1.0
    mm = np.array(m)
    return a.transpose(2, 0, 1), (mm <= 200).astype(np.float32)[None], (mm <= 50).astype(np.float32)[None]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--in-size", type=int, default=800)
    ap.add_argument("--reps", type=int, default=10)
    a = ap.parse_args()
    S, B, H = 256, a.batch, a.in_size
    rs = np.random.RandomState(0)
    imgs = [rs.randint(0, 256, (H, H, 3)).astype(np.uint8) for _ in range(B)]
    ods = [rs.choice(np.array([0, 128, 255], np.uint8), (H, H)) for _ in range(B)]
    rng = random.Random(1)
    draws = [draw(rng, S) for _ in range(B)]
    pipe = DeviceInputPipeline(S, "cuda")
    out = pipe(imgs, ods, draws)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.reps):
        out = pipe(imgs, ods, draws)
    torch.cuda.synchronize()
    t_all = (time.perf_counter() - t0) / a.reps
    dimgs = [torch.from_numpy(x).cuda() for x in imgs]
    dods = [torch.from_numpy(x).cuda() for x in ods]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.reps):
        out = pipe(dimgs, dods, draws)
    torch.cuda.synchronize()
    t_dev = (time.perf_counter() - t0) / a.reps
    t0 = time.perf_counter()
    ref = [pillow_path(imgs[i], ods[i], draws[i], S) for i in range(B)]
    t_cpu = time.perf_counter() - t0
    same = all(np.array_equal(out[0][i].cpu().numpy(), ref[i][0]) and np.array_equal(out[1][i].cpu().numpy(), ref[i][1])
               for i in range(B))
    print("batch %d of %dx%d uint8 samples -> [%d,3,256,256] fp32" % (B, H, H, B))
    print("  device pipeline, samples already in HBM : %7.2f ms  (%8.0f images/s)" % (1e3 * t_dev, B / t_dev))
    print("  device pipeline incl. host->device copy : %7.2f ms  (%8.0f images/s)" % (1e3 * t_all, B / t_all))
    print("  reference path (Pillow, one host thread): %7.2f ms  (%8.0f images/s)" % (1e3 * t_cpu, B / t_cpu))
    print("  bit-identical to the Pillow path: %s" % same)
    assert same


if __name__ == "__main__":
    main()
