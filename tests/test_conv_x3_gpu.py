"""The split-bf16 ("x3") convolution — fp32 operands as three bf16 terms, six bf16 MFMAs per product, fp32 accumulation
(csrc/conv_x3.hip) — against stock PyTorch fp32/fp64 convolutions on the host, at the SAME tolerance as the fp32-MFMA kernel
(tests/test_kernels_gpu.py::test_conv_forward), and against the fp64 result with the fp32-MFMA kernel as the yardstick."""
import pytest
import torch
import torch.nn.functional as F

from test_kernels_gpu import close, rnd, ops, pack, DEV

pytestmark = pytest.mark.gpu


def pack_x3(w):
    """OIHW -> (packed uint16 buffer, xf_off, xd_off) through wtpse_pack_conv_weights_x3."""
    o = ops()
    co, ci, k, _ = w.shape
    t = k * k
    xf = o.x3_packed_size(co, ci, t)
    xd = o.x3_packed_size(ci, co, t)
    flat = w.reshape(-1).contiguous().to(DEV)
    packed = torch.full((xf + xd,), 0x7FC0, dtype=torch.int16, device=DEV)      # bf16 NaN everywhere
    desc = torch.tensor([0, co, ci, t, 0, xf, 0, 0], dtype=torch.int32, device=DEV)
    o.lib().call("wtpse_pack_conv_weights_x3", flat.data_ptr(), desc.data_ptr(), 1, packed.data_ptr(), o.stream_ptr())
    return packed, 0, xf


X3_CASES = [
    # B, C0, C1, Cout, H, W, k
    (2, 16, 0, 32, 16, 16, 3),     # MT 1, 16x16 tile, one chunk
    (2, 32, 0, 64, 8, 8, 3),       # image smaller than a tile
    (1, 64, 64, 128, 16, 16, 3),   # concat, several cout blocks
    (2, 16, 16, 32, 24, 48, 3),    # concat, ragged 8x32 tiles
    (2, 256, 0, 128, 4, 4, 1),     # 1x1, multi-chunk
    (1, 128, 0, 256, 2, 2, 3),     # deepest level of the 32x32 test network
    (2, 40, 0, 96, 12, 20, 3),     # non-power-of-two channels: ragged chunk (40 -> 48) and 3 cout blocks
    (2, 3, 0, 32, 20, 40, 3),      # 3-channel input padded to one chunk
    (20, 32, 32, 64, 32, 64, 3),   # enough tiles for the 64-cout (MT 2) variant
]


@pytest.mark.parametrize("case", X3_CASES)
def test_conv_x3_forward(case):
    o = ops()
    B, C0, C1, Co, H, W, k = case
    x0 = rnd(B, C0, H, W, seed=1)
    x1 = rnd(B, C1, H, W, seed=2) if C1 else None
    w = rnd(Co, C0 + C1, k, k, seed=3, scale=0.2)
    b = rnd(Co, seed=4)
    xin = torch.cat([x0, x1], 1) if C1 else x0
    ref = F.conv2d(xin, w, b, padding=k // 2)
    packed, xf, _ = pack_x3(w)
    y, _, stats = o.conv_fwd_x3(x0.to(DEV), x1.to(DEV) if C1 else None, packed.data_ptr() + 2 * xf, b.to(DEV), Co, k,
                                want_stats=True)
    close(y, ref, what="conv x3")
    s = stats.double().sum(0).cpu()
    close(s[:, 0], ref.double().sum((0, 2, 3)), rtol=1e-4, atol=1e-3, what="stat sum")
    close(s[:, 1], (ref.double() ** 2).sum((0, 2, 3)), rtol=1e-4, atol=1e-3, what="stat sumsq")
    yr, _, _ = o.conv_fwd_x3(x0.to(DEV), x1.to(DEV) if C1 else None, packed.data_ptr() + 2 * xf, None, Co, k, relu_out=True)
    close(yr, F.relu(ref - b.view(1, -1, 1, 1)), what="conv x3 relu nobias")
    # accuracy against fp64, with the fp32-MFMA kernel as the yardstick
    ref64 = F.conv2d(xin.double(), w.double(), b.double(), padding=k // 2)
    pk, wf, _ = pack(w)
    y32, _, _ = o.conv_fwd(x0.to(DEV), x1.to(DEV) if C1 else None, pk.data_ptr() + 4 * wf, b.to(DEV), Co, k)
    e3 = float((y.cpu().double() - ref64).norm() / ref64.norm())
    e32 = float((y32.cpu().double() - ref64).norm() / ref64.norm())
    print("relative L2 error vs fp64: x3 %.2e, fp32 MFMA %.2e" % (e3, e32))
    assert e3 <= 2.0 * e32 + 2e-8, (e3, e32)


@pytest.mark.parametrize("case", [(2, 16, 16, 32, 16, 32, 3), (1, 32, 32, 64, 8, 16, 1), (20, 32, 32, 64, 32, 64, 3)])
def test_conv_x3_prologue(case):
    o = ops()
    B, C0, C1, Co, H, W, k = case
    x0 = rnd(B, C0, H, W, seed=5)
    x1 = rnd(B, C1, H, W, seed=6) if C1 else None
    pro = torch.stack([rnd(C0 + C1, seed=7) * 0.5 + 1.0, rnd(C0 + C1, seed=8)], 1).contiguous()
    w = rnd(Co, C0 + C1, k, k, seed=9, scale=0.2)
    xin = torch.cat([x0, x1], 1) if C1 else x0
    act = xin * pro[:, 0].view(1, -1, 1, 1) + pro[:, 1].view(1, -1, 1, 1)
    act = torch.cat([F.relu(act[:, :C0]), act[:, C0:]], 1)
    ref = F.conv2d(act, w, None, padding=k // 2)
    packed, xf, _ = pack_x3(w)
    y, _, _ = o.conv_fwd_x3(x0.to(DEV), x1.to(DEV) if C1 else None, packed.data_ptr() + 2 * xf, None, Co, k,
                            pro0=pro[:C0].contiguous().to(DEV), pro1=(pro[C0:].contiguous().to(DEV) if C1 else None), pro_relu=1)
    close(y, ref, what="prologue x3")


@pytest.mark.parametrize("case", [(2, 32, 0, 32, 16, 16, 3), (1, 64, 64, 128, 16, 16, 3), (2, 32, 32, 64, 24, 48, 3),
                                  (4, 32, 32, 64, 16, 16, 3),
                                  (2, 256, 0, 128, 4, 4, 1), (2, 40, 0, 96, 12, 20, 3)])
def test_conv_x3_dgrad(case):
    """Data gradient = the same kernel on dY with the transposed, tap-flipped weights; split output (gradient of a concat)
    and the fused ReLU mask."""
    o = ops()
    B, C0, C1, Co, H, W, k = case
    x0 = rnd(B, C0, H, W, seed=11).requires_grad_(True)
    x1 = rnd(B, C1, H, W, seed=12).requires_grad_(True) if C1 else None
    w = rnd(Co, C0 + C1, k, k, seed=13, scale=0.2)
    dy = rnd(B, Co, H, W, seed=15)
    y = F.conv2d(torch.cat([x0, x1], 1) if C1 else x0, w, None, padding=k // 2)
    y.backward(dy)
    packed, _, xd = pack_x3(w)
    d0, d1, _ = o.conv_fwd_x3(dy.to(DEV), None, packed.data_ptr() + 2 * xd, None, C0 + C1, k, split=(C0 if C1 else None))
    close(d0, x0.grad, what="dgrad0 x3")
    if C1:
        close(d1, x1.grad, what="dgrad1 x3")
    else:
        ref_act = rnd(B, C0, H, W, seed=16)
        dm, _, _ = o.conv_fwd_x3(dy.to(DEV), None, packed.data_ptr() + 2 * xd, None, C0, k, mask_ref=ref_act.to(DEV))
        close(dm, x0.grad * (ref_act > 0).float(), what="dgrad x3 + relu mask")


@pytest.mark.parametrize("case", [
    (2, 32, 0, 32, 16, 32, 3),     # 32x32 block (the 4 waves split the pixels), one 128-pixel tile per 4 rows
    (3, 32, 0, 32, 20, 40, 3),     # ragged tiles
    (2, 64, 0, 64, 16, 16, 3),     # 64x64 block (waves = quadrants), 16-wide tiles
    (2, 32, 32, 64, 24, 48, 3),    # concat, quadrants, ragged 2x32 tiles
    (2, 64, 0, 32, 12, 36, 3),     # Cout 32: 32x32 blocks over a 64-channel input
    (1, 128, 128, 256, 8, 8, 3),   # many blocks, image smaller than a tile
    (4, 96, 0, 64, 32, 32, 3),     # 96 input channels: 32x32 blocks
    (20, 32, 32, 64, 32, 64, 3),   # several tiles per workgroup
])
def test_conv_x3_wgrad(case):
    o = ops()
    B, C0, C1, Co, H, W, k = case
    x0 = rnd(B, C0, H, W, seed=11)
    x1 = rnd(B, C1, H, W, seed=12) if C1 else None
    w = rnd(Co, C0 + C1, k, k, seed=13, scale=0.2).requires_grad_(True)
    dy = rnd(B, Co, H, W, seed=15)
    pro = torch.stack([rnd(C0 + C1, seed=7) * 0.5 + 1.0, rnd(C0 + C1, seed=8)], 1).contiguous()
    xin = torch.cat([x0, x1], 1) if C1 else x0
    act = F.relu(xin * pro[:, 0].view(1, -1, 1, 1) + pro[:, 1].view(1, -1, 1, 1))
    F.conv2d(act.double(), w.double(), None, padding=k // 2).backward(dy.double())
    ref64 = w.grad.double()
    assert o.wgrad_x3_supported(C0 + C1, Co, k, C0 if C1 else 8)
    dw = torch.full_like(w.detach(), float("nan")).to(DEV)
    args = (dy.to(DEV), x0.to(DEV), x1.to(DEV) if C1 else None, k)
    kw = dict(pro0=pro[:C0].contiguous().to(DEV), pro_relu=3, pro1=(pro[C0:].contiguous().to(DEV) if C1 else None))
    o.conv_wgrad_x3(*args, dw, **kw)
    scale = float(ref64.abs().max())
    close(dw, ref64, rtol=2e-4, atol=2e-5 * max(scale, 1.0), what="wgrad x3")
    dw32 = torch.empty_like(dw)
    o.conv_wgrad(*args, dw32, None, kw["pro0"], 3, False, kw["pro1"])
    e3 = float((dw.cpu().double() - ref64).norm() / ref64.norm())
    e32 = float((dw32.cpu().double() - ref64).norm() / ref64.norm())
    print("relative L2 error vs fp64: x3 %.2e, fp32 MFMA %.2e" % (e3, e32))
    assert e3 <= 2.0 * e32 + 2e-8, (e3, e32)
    o.conv_wgrad_x3(*args, dw, accumulate=True, **kw)
    close(dw, 2 * ref64, rtol=2e-4, atol=4e-5 * max(scale, 1.0), what="wgrad x3 accumulate")


@pytest.mark.parametrize("case", [
    # B, C0, C1, Cout, H, W, bias
    (2, 16, 0, 16, 8, 32, True),      # one 16x16 block, one strip, bias gradient (the DeepWT layers)
    (3, 16, 0, 16, 20, 64, True),     # two strips: the pixels left / right of a strip come from the neighbouring one
    (2, 32, 0, 32, 16, 32, False),    # 32x32 block per wave
    (2, 16, 16, 32, 12, 64, False),   # concat whose halves are one 16-channel fragment each (up4.conv3)
    (2, 64, 0, 32, 9, 32, False),     # two cin blocks, odd height
    (1, 128, 128, 256, 5, 32, False), # many (cout, cin) pairs, fewer rows than the ring is deep
    (2, 16, 0, 32, 7, 96, False),     # 16 -> 32 (down1.conv1): 2 x 1 fragments
    (2, 48, 0, 16, 33, 32, True),     # 1 x 1 fragments, three cin blocks, bias
    (24, 32, 32, 64, 32, 64, False),  # several units per wave
    (40, 16, 0, 16, 64, 64, True),    # row segments (more waves than columns) and several units per wave
])
def test_conv_wgrad_r(case):
    """The register-resident x3 weight gradient (csrc/wgrad_r.hip) against the fp64 gradient of stock conv2d, at the tolerance
    of the LDS-based x3 kernel, with the fp32-MFMA kernel as the yardstick, plus the bias gradient and accumulation."""
    o = ops()
    B, C0, C1, Co, H, W, bias = case
    k = 3
    x0 = rnd(B, C0, H, W, seed=11)
    x1 = rnd(B, C1, H, W, seed=12) if C1 else None
    w = rnd(Co, C0 + C1, k, k, seed=13, scale=0.2).requires_grad_(True)
    bb = rnd(Co, seed=14).requires_grad_(True)
    dy = rnd(B, Co, H, W, seed=15)
    pro = torch.stack([rnd(C0 + C1, seed=7) * 0.5 + 1.0, rnd(C0 + C1, seed=8)], 1).contiguous()
    xin = torch.cat([x0, x1], 1) if C1 else x0
    act = F.relu(xin * pro[:, 0].view(1, -1, 1, 1) + pro[:, 1].view(1, -1, 1, 1))
    F.conv2d(act.double(), w.double(), bb.double(), padding=1).backward(dy.double())
    ref64, refb = w.grad.double(), bb.grad.double()
    assert o.wgrad_r_supported(C0 + C1, Co, k, C0 if C1 else 16, W)
    dw = torch.full_like(w.detach(), float("nan")).to(DEV)
    db = torch.full((Co,), float("nan"), device=DEV) if bias else None
    args = (dy.to(DEV), x0.to(DEV), x1.to(DEV) if C1 else None)
    kw = dict(pro0=pro[:C0].contiguous().to(DEV), pro_relu=3, pro1=(pro[C0:].contiguous().to(DEV) if C1 else None))
    o.conv_wgrad_r(*args, dw, db, **kw)
    scale = float(ref64.abs().max())
    close(dw, ref64, rtol=2e-4, atol=2e-5 * max(scale, 1.0), what="wgrad r")
    if bias:
        close(db, refb, rtol=1e-4, atol=1e-5 * max(float(refb.abs().max()), 1.0), what="dbias r")
    dw32 = torch.empty_like(dw)
    o.conv_wgrad(args[0], args[1], args[2], k, dw32, None, kw["pro0"], 3, False, kw["pro1"])
    e3 = float((dw.cpu().double() - ref64).norm() / ref64.norm())
    e32 = float((dw32.cpu().double() - ref64).norm() / ref64.norm())
    print("relative L2 error vs fp64: wgrad_r %.2e, fp32 MFMA %.2e" % (e3, e32))
    assert e3 <= 2.0 * e32 + 2e-8, (e3, e32)
    o.conv_wgrad_r(*args, dw, db, accumulate=True, **kw)
    close(dw, 2 * ref64, rtol=2e-4, atol=4e-5 * max(scale, 1.0), what="wgrad r accumulate")
    # without a prologue (a dense input), no bias
    F.conv2d(xin.double(), w.double(), None, padding=1)
    w.grad = None
    F.conv2d(xin.double(), w.double(), None, padding=1).backward(dy.double())
    o.conv_wgrad_r(*args, dw, None)
    close(dw, w.grad.double(), rtol=2e-4, atol=2e-5 * max(float(w.grad.abs().max()), 1.0), what="wgrad r, no prologue")
