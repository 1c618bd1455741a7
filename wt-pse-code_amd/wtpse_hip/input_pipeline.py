"""Device-side training input pipeline: the MI355X counterpart of the reference's per-sample host transforms
Compose([Resize(256), RandomScaleCrop(256), Normalize_tf(), ToTensor()]) (train.py:58-62; custom_transforms.py:375-391,
330-354,139-176,455-499,581-599) and of get_multi_batch's stack + .cuda() (Trainer.py:45-55).

The decoded uint8 samples are copied to the GPU as they are; resampling, cropping, normalisation and the mask thresholds run
there (csrc/pipeline.hip) and produce the [N,3,S,S] / [N,1,S,S] fp32 batch the training step takes.  The result equals the
reference's bit for bit (tests/test_input_pipeline_gpu.py, fixtures generated from the reference's own classes).

What stays on the host: PNG decoding and the random draws.  The reference draws from Python's `random` module
(`seed = random.random()`, two `random.uniform(1, 1.5)`, two `random.randint` for the crop); `draw()` makes the same draws
in the same order from any `random.Random`-like generator, so a seeded run crops exactly like the reference.

The coefficient tables are Pillow's (src/libImaging/Resample.c precompute_coeffs + normalize_coeffs_8bpc; Geometry.c
ImagingScaleAffine for NEAREST), computed here with the same double-precision operations in the same order.
"""
import numpy as np
import torch

from . import ops

PRECISION_BITS = 22
_FILTER_SUPPORT = {"bilinear": 1.0, "bicubic": 2.0}


def _filter(name, x):
    x = np.abs(x)
    if name == "bilinear":
        return np.where(x < 1.0, 1.0 - x, 0.0)
    a = -0.5
    return np.where(x < 1.0, ((a + 2.0) * x - (a + 3.0)) * x * x + 1,
                    np.where(x < 2.0, (((x - 5) * x + 8) * x - 4) * a, 0.0))


def resample_table(in_size, out_size, filt, first=0, count=None):
    """Pillow's coefficients for resizing an axis in_size -> out_size, for output positions [first, first+count).
    -> (bounds [count,2] int32 (first source index, taps), kk [count,ksize] int32, ksize)."""
    count = out_size - first if count is None else count
    scale = float(in_size) / out_size
    filterscale = scale if scale >= 1.0 else 1.0
    support = _FILTER_SUPPORT[filt] * filterscale
    ksize = int(np.ceil(support)) * 2 + 1
    ss = 1.0 / filterscale
    xx = np.arange(first, first + count, dtype=np.float64)
    center = (xx + 0.5) * scale
    xmin = np.maximum((center - support + 0.5).astype(np.int64), 0)          # C (int) cast: truncation
    xmax = np.minimum((center + support + 0.5).astype(np.int64), in_size)
    n = xmax - xmin
    x = np.arange(ksize, dtype=np.int64)[None, :]
    live = x < n[:, None]
    w = np.where(live, _filter(filt, (x + xmin[:, None] - center[:, None] + 0.5) * ss), 0.0)
    ww = np.zeros(count, np.float64)
    for i in range(ksize):                                                   # sequential sum, as the C loop
        ww = ww + w[:, i]
    k = np.where(ww[:, None] != 0.0, w / np.where(ww == 0.0, 1.0, ww)[:, None], w)
    p = k * float(1 << PRECISION_BITS)
    kk = np.where(k < 0, (-0.5 + p).astype(np.int64), (0.5 + p).astype(np.int64)).astype(np.int32)
    kk = np.where(live, kk, 0).astype(np.int32)
    bounds = np.stack([xmin, n], 1).astype(np.int32)
    return bounds, kk, ksize


def nearest_table(in_size, out_size, first=0, count=None):
    """Source index of output positions [first, first+count) of Image.resize(..., NEAREST) (accumulated `xo += a0`)."""
    count = out_size - first if count is None else count
    a0 = float(in_size) / out_size
    steps = np.full(out_size, a0, np.float64)
    steps[0] = a0 * 0.5
    xo = np.add.accumulate(steps)                                            # sequential adds, as the C loop
    idx = np.clip(xo.astype(np.int64), 0, in_size - 1)
    return idx[first:first + count].astype(np.int32)


def draw(rng, size=256):
    """The draws of RandomScaleCrop + RandomCrop (custom_transforms.py:342-346,167-168) from `rng`, in their order.
    -> (scaled width, scaled height, crop x1, crop y1); (size, size, 0, 0) when the sample is not scaled."""
    seed = rng.random()
    nw = nh = size
    if seed > 0.5:
        nw = int(rng.uniform(1, 1.5) * size)
        nh = int(rng.uniform(1, 1.5) * size)
    if nw == size and nh == size:
        return nw, nh, 0, 0
    return nw, nh, rng.randint(0, nw - size), rng.randint(0, nh - size)


def _dev_u8(a, device):
    t = a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a))
    if t.dtype != torch.uint8:
        raise ValueError("input samples must be uint8 (decoded images), got %s" % t.dtype)
    return t.to(device, non_blocking=True).contiguous()


def _dev_i32(a, device):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.int32)).to(device, non_blocking=True)


class DeviceInputPipeline:
    """batch = pipeline(images, disc_masks, draws): images [H,W,3] uint8 and disc masks [H,W] uint8 per sample (numpy or
    torch, host or device; sizes may differ between samples), draws = [draw(rng, size) per sample]."""

    def __init__(self, size=256, device="cuda"):
        self.size = int(size)
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("the device-side input pipeline runs on the GPU only (no CPU fallback)")
        self._resize_tables = {}

    def _resize_table(self, in_size):
        t = self._resize_tables.get(in_size)
        if t is None:
            b, k, ks = resample_table(in_size, self.size, "bicubic")
            t = self._resize_tables[in_size] = (_dev_i32(b, self.device), _dev_i32(k, self.device), ks)
        return t

    def _resample(self, src, bounds, kk, tab, ksize, L, vertical):
        N, H, W, C = src.shape
        out = torch.empty((N, L, W, C) if vertical else (N, H, L, C), dtype=torch.uint8, device=self.device)
        ops.lib().call("wtpse_resample_u8", src.data_ptr(), out.data_ptr(), bounds.data_ptr(), kk.data_ptr(),
                       0 if tab is None else tab.data_ptr(), ksize, N, H, W, C, L, int(vertical), ops.stream_ptr())
        return out

    def __call__(self, images, disc_masks, draws):
        S, dev = self.size, self.device
        N = len(images)
        assert len(disc_masks) == N and len(draws) == N
        # ---- Resize(S): bicubic, horizontal pass then vertical pass, batched over samples of one input size
        img1 = torch.empty((N, S, S, 3), dtype=torch.uint8, device=dev)
        od1 = torch.empty((N, S, S, 1), dtype=torch.uint8, device=dev)
        groups = {}
        for i, im in enumerate(images):
            groups.setdefault(tuple(im.shape[:2]), []).append(i)
        for (H, W), idx in groups.items():
            im = torch.stack([_dev_u8(images[i], dev) for i in idx])
            md = torch.stack([_dev_u8(disc_masks[i], dev) for i in idx]).unsqueeze(-1)
            sel = torch.tensor(idx, device=dev)
            for src, dst in ((im, img1), (md, od1)):
                t = src
                if W != S:
                    b, k, ks = self._resize_table(W)
                    t = self._resample(t, b, k, None, ks, S, False)
                if H != S:
                    b, k, ks = self._resize_table(H)
                    t = self._resample(t, b, k, None, ks, S, True)
                dst[sel] = t
        # ---- RandomScaleCrop(S): bilinear up-scale to (nw, nh) and crop, only for the S columns / rows that survive;
        # unscaled samples get the identity table.  The disc mask's NEAREST resize + crop is an index gather.
        hb, hk, vb, vk, xi, yi = [], [], [], [], [], []
        for nw, nh, x1, y1 in draws:
            b, k, ks = resample_table(S, nw, "bilinear", x1, S)
            hb.append(b); hk.append(k)
            b, k, ks = resample_table(S, nh, "bilinear", y1, S)
            vb.append(b); vk.append(k)
            xi.append(nearest_table(S, nw, x1, S))
            yi.append(nearest_table(S, nh, y1, S))
        tab = torch.arange(N, dtype=torch.int32, device=dev)
        t = self._resample(img1, _dev_i32(np.stack(hb), dev), _dev_i32(np.stack(hk), dev), tab, 3, S, False)
        img2 = self._resample(t, _dev_i32(np.stack(vb), dev), _dev_i32(np.stack(vk), dev), tab, 3, S, True)
        image = torch.empty((N, 3, S, S), dtype=torch.float32, device=dev)
        od = torch.empty((N, 1, S, S), dtype=torch.float32, device=dev)
        oc = torch.empty((N, 1, S, S), dtype=torch.float32, device=dev)
        xidx, yidx = _dev_i32(np.stack(xi), dev), _dev_i32(np.stack(yi), dev)
        ops.lib().call("wtpse_input_finish", img2.data_ptr(), od1.data_ptr(), xidx.data_ptr(), yidx.data_ptr(), image.data_ptr(),
                       od.data_ptr(), oc.data_ptr(), N, S, ops.stream_ptr())
        return image, od, oc
