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
    (20, 32, 32, 64, 32, 64, 3),   # 64 output channels on few tiles: 32-channel blocks (the 64-channel blocks need 512 workgroups)
    (32, 64, 0, 64, 64, 64, 3),    # 512 tiles of 256 pixels x one 64-channel block: the step's default kernel conv_x3r_k<2,1,4,5>
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
    (4, 32, 0, 32, 16, 16, False),    # W = 16: two images side by side per 32-pixel step
    (5, 64, 64, 128, 16, 16, False),  # ... an odd batch (the last pair is a single image), concat
    (3, 32, 0, 64, 7, 16, False),     # ... odd height and batch
    (32, 128, 0, 256, 16, 16, False), # ... down4.conv1 at the benchmark's batch
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
    if W != 16:      # (the two-images-per-step form of the 16-wide maps takes a materialised dY only)
        # dY as the un-applied second half of a BatchNorm backward: dY = k1 * g + k2 * y + k3 formed on load
        gg, yy = rnd(B, Co, H, W, seed=17), rnd(B, Co, H, W, seed=18)
        coef = torch.stack([rnd(Co, seed=19) * 0.3 + 1.0, rnd(Co, seed=20) * 0.2, rnd(Co, seed=21) * 0.1], 1).contiguous()
        dyl = coef[:, 0].view(1, -1, 1, 1) * gg + coef[:, 1].view(1, -1, 1, 1) * yy + coef[:, 2].view(1, -1, 1, 1)
        w.grad = None
        F.conv2d(act.double(), w.double(), None, padding=1).backward(dyl.double())
        o.conv_wgrad_r_bn(gg.to(DEV), yy.to(DEV), coef.to(DEV), args[1], args[2], dw, **kw)
        close(dw, w.grad.double(), rtol=2e-4, atol=2e-5 * max(float(w.grad.abs().max()), 1.0), what="wgrad r, BatchNorm-apply on load")
        dwp = torch.empty_like(dw)
        o.conv_wgrad_r(dyl.to(DEV), args[1], args[2], dwp, None, **kw)
        close(dw, dwp, rtol=2e-5, atol=2e-6 * max(float(w.grad.abs().max()), 1.0), what="fused vs materialised dY")
    # without a prologue (a dense input), no bias
    F.conv2d(xin.double(), w.double(), None, padding=1)
    w.grad = None
    F.conv2d(xin.double(), w.double(), None, padding=1).backward(dy.double())
    o.conv_wgrad_r(*args, dw, None)
    close(dw, w.grad.double(), rtol=2e-4, atol=2e-5 * max(float(w.grad.abs().max()), 1.0), what="wgrad r, no prologue")


@pytest.mark.parametrize("case", [
    # B, Cl (channels of the BatchNorm'd tensor), Cother (other side of the split, 0: none), bn_second, Cn (next conv's outputs), H, W, k, relu, x3
    (3, 16, 0, False, 16, 24, 40, 3, True, False),     # 16-channel fp32 path (inc), ragged tiles
    (2, 32, 0, False, 32, 16, 32, 3, True, False),     # fp32 32-wide path
    (2, 32, 0, False, 32, 16, 32, 3, False, True),     # x3, no activation (ConvD.conv1)
    (4, 64, 0, False, 64, 16, 16, 3, True, True),      # x3, 16-wide tiles
    (20, 64, 0, False, 64, 32, 64, 3, True, True),     # x3, 64-cout blocks
    (2, 32, 32, True, 64, 12, 36, 3, True, True),      # ConvU.conv3: gradient of a concat, the BatchNorm'd tensor is the second half
    (2, 16, 16, False, 16, 20, 20, 1, True, False),    # teacher fusion (1x1): the first half
    (2, 128, 0, False, 64, 8, 8, 1, True, True),       # ConvU.conv2 (1x1) on the x3 path
])
def test_dgrad_bnb(case):
    """Data gradient with the BatchNorm-backward reductions in its epilogue + wtpse_bn_bwd_from_stats, against autograd (fp64)
    through conv -> BatchNorm(train) [-> ReLU] -> [cat] -> conv, and against the stand-alone path (dgrad, then bn_bwd)."""
    o = ops()
    B, Cl, Co, bn_second, Cn, H, W, k, relu, x3 = case
    y = rnd(B, Cl, H, W, seed=31).double().requires_grad_(True)            # raw conv output of the layer below
    other = rnd(B, Co, H, W, seed=32).double().requires_grad_(True) if Co else None
    gamma = (rnd(Cl, seed=33) * 0.2 + 1).double().requires_grad_(True)
    beta = (rnd(Cl, seed=34) * 0.2).double().requires_grad_(True)
    w = rnd(Cn, Cl + Co, k, k, seed=35, scale=0.2)
    du = rnd(B, Cn, H, W, seed=36)
    z = F.batch_norm(y, None, None, gamma, beta, True, 0.1, 1e-5)
    on_kink = z.detach().abs() < 2e-6 if relu else torch.zeros_like(z, dtype=torch.bool)
    if relu:
        z = F.relu(z)
    zin = z if other is None else (torch.cat([other, z], 1) if bn_second else torch.cat([z, other], 1))
    F.conv2d(zin, w.double(), None, padding=k // 2).backward(du.double())
    # forward-side quantities as the engine has them
    yd = y.detach().float().to(DEV)
    mean = yd.double().mean((0, 2, 3))
    var = yd.double().var((0, 2, 3), unbiased=False)
    invstd = (1.0 / torch.sqrt(var + 1e-5))
    g_d, b_d = gamma.detach().float().to(DEV), beta.detach().float().to(DEV)
    ss = torch.stack([g_d.double() * invstd, b_d.double() - mean * g_d.double() * invstd], 1).float().contiguous()
    mean_f, invstd_f = mean.float().contiguous(), invstd.float().contiguous()
    if x3:
        packed, _, xd = pack_x3(w)
        wptr = packed.data_ptr() + 2 * xd
    else:
        packed, _, wd = pack(w)
        wptr = packed.data_ptr() + 4 * wd
    split = None if not Co else (Co if bn_second else Cl)
    g0, g1, stats, _ = o.dgrad_bnb(du.to(DEV), wptr, x3, Cl + Co, k, yd, ss, mean_f, relu, split, bn_second)
    g = g1 if (Co and bn_second) else g0
    dg, dbt = torch.empty(Cl, device=DEV), torch.empty(Cl, device=DEV)
    dy = o.bn_bwd_from_stats(g, yd, stats, g_d, mean_f, invstd_f, dg, dbt)
    # the same launch folding its partials itself (last-arriver tickets): same masked gradient bit for bit, coefficients and
    # dgamma / dbeta equal to the stand-alone fold up to the order of the fp64 sums
    dg_t, dbt_t = torch.full((Cl,), 7.0, device=DEV), torch.full((Cl,), 7.0, device=DEV)
    h0, h1, stats_t, coef = o.dgrad_bnb(du.to(DEV), wptr, x3, Cl + Co, k, yd, ss, mean_f, relu, split, bn_second,
                                        tail=(g_d, invstd_f.to(DEV), dg_t, dbt_t))
    h = h1 if (Co and bn_second) else h0
    assert torch.equal(h, g) and torch.equal(stats_t, stats)
    dy_t = o.bn_bwd_apply_coef(h, yd, coef)
    close(dg_t, dg, rtol=1e-6, atol=1e-6 * float(dg.abs().max()) + 1e-9, what="tail dgamma")
    close(dbt_t, dbt, rtol=1e-6, atol=1e-6 * float(dbt.abs().max()) + 1e-9, what="tail dbeta")
    close(dy_t, dy, rtol=1e-5, atol=1e-6 * float(dy.abs().max()), what="tail dy")
    torch.cuda.synchronize()
    assert all(int(t[0].abs().sum()) == 0 for t in o._TICKETS.values()), "tickets must be left at zero"
    kz = lambda t: torch.where(on_kink.to(t.device), torch.zeros_like(t), t)
    assert int(on_kink.sum()) <= max(2, on_kink.numel() // 100000)
    sc = float(y.grad.abs().max())
    close(kz(dy), kz(y.grad.float()), rtol=1e-3, atol=2e-4 * sc, what="dy")
    close(dg, gamma.grad, rtol=1e-3, atol=2e-4 * float(gamma.grad.abs().max()) + 1e-5, what="dgamma")
    close(dbt, beta.grad, rtol=1e-3, atol=2e-4 * float(beta.grad.abs().max()) + 1e-5, what="dbeta")
    if Co:
        close(g0 if bn_second else g1, other.grad, rtol=1e-3, atol=2e-4 * float(other.grad.abs().max()), what="other half")
    # the stand-alone path: plain data gradient, then the three-kernel BatchNorm backward
    if x3:
        d0, d1, _ = o.conv_fwd_x3(du.to(DEV), None, wptr, None, Cl + Co, k, split=split)
    else:
        d0, d1, _ = o.conv_fwd(du.to(DEV), None, wptr, None, Cl + Co, k, split=split)
    dz = d1 if (Co and bn_second) else d0
    dg2, dbt2 = torch.empty(Cl, device=DEV), torch.empty(Cl, device=DEV)
    dy2 = o.bn_bwd(dz, yd, ss, relu, g_d, mean_f, invstd_f, dg2, dbt2)
    close(dy, dy2, rtol=1e-4, atol=2e-5 * sc, what="fused vs stand-alone dy")
    close(dg, dg2, rtol=1e-4, atol=1e-4 * float(dg2.abs().max()) + 1e-6, what="fused vs stand-alone dgamma")
    close(dbt, dbt2, rtol=1e-4, atol=1e-4 * float(dbt2.abs().max()) + 1e-6, what="fused vs stand-alone dbeta")


# ---- the 16-channel 3x3 layers in the x3 arithmetic (csrc/conv.hip MODE 3) ---------------------------------------------
def pack_x16(w):
    """OIHW (<= 16 x <= 16 x 3 x 3) -> (buffer, forward offset, data-gradient offset) through wtpse_pack_conv16_x3."""
    o = ops()
    co, ci = w.shape[:2]
    n = o.X16_SIZE
    flat = w.reshape(-1).contiguous().to(DEV)
    packed = torch.full((2 * n,), 0x7FC0, dtype=torch.int16, device=DEV)
    desc = torch.tensor([0, co, ci, 9, 0, n, 0, 0], dtype=torch.int32, device=DEV)
    o.lib().call("wtpse_pack_conv16_x3", flat.data_ptr(), desc.data_ptr(), 1, packed.data_ptr(), o.stream_ptr())
    return packed, 0, n


X16_CASES = [
    # B, Cin, Cout, H, W
    (2, 16, 16, 16, 16),
    (3, 16, 16, 24, 40),      # ragged tiles
    (2, 8, 16, 8, 8),         # image smaller than a tile, half the reduction channels
    (2, 16, 5, 32, 64),       # few output channels
    (1, 3, 16, 20, 12),       # 3-channel input
    (2, 16, 16, 64, 128),     # several tiles per image in both directions
]


@pytest.mark.parametrize("case", X16_CASES)
def test_conv16_x3_forward(case):
    o = ops()
    B, Ci, Co, H, W = case
    x = rnd(B, Ci, H, W, seed=41)
    w = rnd(Co, Ci, 3, 3, seed=42, scale=0.2)
    b = rnd(Co, seed=43)
    ref = F.conv2d(x, w, b, padding=1)
    packed, xf, xd = pack_x16(w)
    y, stats, gram = o.conv16_x3(x.to(DEV), packed.data_ptr() + 2 * xf, b.to(DEV), Co, want_stats=True, want_gram=(Co == 16))
    close(y, ref, what="conv16 x3")
    s = stats.double().sum(0).cpu()
    close(s[:, 0], ref.double().sum((0, 2, 3)), rtol=1e-4, atol=1e-3, what="stat sum")
    close(s[:, 1], (ref.double() ** 2).sum((0, 2, 3)), rtol=1e-4, atol=1e-3, what="stat sumsq")
    if gram is not None:
        gp, S = gram
        G = gp.double().view(B, S, 16, 16).sum(1).cpu()
        f = ref.double().reshape(B, 16, -1)
        close(G, f @ f.transpose(1, 2), rtol=1e-4, atol=1e-3, what="gram")
        # the same numbers as the fp32-MFMA kernel's Gram epilogue gives on this output
        pk, wf, _ = pack(w)
        y0, (gp0, S0) = o.conv_fwd_gram(x.to(DEV), pk.data_ptr() + 4 * wf, b.to(DEV))
        assert S0 == S
        close(gp.double().view(B, S, 256).sum(1), gp0.double().view(B, S, 256).sum(1), rtol=1e-4, atol=1e-3, what="gram vs fp32 path")
    yr, _, _ = o.conv16_x3(x.to(DEV), packed.data_ptr() + 2 * xf, None, Co, relu_out=True)
    close(yr, F.relu(ref - b.view(1, -1, 1, 1)), what="conv16 x3 relu nobias")
    # prologue: per-channel affine + ReLU on load, zero padding AFTER the prologue
    pro = torch.stack([rnd(Ci, seed=44) * 0.5 + 1.0, rnd(Ci, seed=45)], 1).contiguous()
    act = F.relu(x * pro[:, 0].view(1, -1, 1, 1) + pro[:, 1].view(1, -1, 1, 1))
    yp, _, _ = o.conv16_x3(x.to(DEV), packed.data_ptr() + 2 * xf, b.to(DEV), Co, pro.to(DEV), 1)
    close(yp, F.conv2d(act, w, b, padding=1), what="conv16 x3 prologue")
    yq, _, _ = o.conv16_x3(x.to(DEV), packed.data_ptr() + 2 * xf, None, Co, None, 1)          # ReLU-on-load only
    close(yq, F.conv2d(F.relu(x), w, None, padding=1), what="conv16 x3 relu on load")
    # data gradient direction + ReLU mask
    du = rnd(B, Co, H, W, seed=46)
    dref = F.conv_transpose2d(du, w, padding=1)
    d, _, _ = o.conv16_x3(du.to(DEV), packed.data_ptr() + 2 * xd, None, Ci)
    close(d, dref, what="dgrad16 x3")
    mref = rnd(B, Ci, H, W, seed=47)
    dm, _, _ = o.conv16_x3(du.to(DEV), packed.data_ptr() + 2 * xd, None, Ci, mask_ref=mref.to(DEV))
    close(dm, dref * (mref > 0), what="dgrad16 x3 masked")
    # accuracy against fp64, with the fp32-MFMA kernel as the yardstick
    ref64 = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    pk, wf, _ = pack(w)
    y32, _, _ = o.conv_fwd(x.to(DEV), None, pk.data_ptr() + 4 * wf, b.to(DEV), Co, 3)
    e3 = float((y.cpu().double() - ref64).norm() / ref64.norm())
    e32 = float((y32.cpu().double() - ref64).norm() / ref64.norm())
    print("relative L2 error vs fp64: x3 %.2e, fp32 MFMA %.2e" % (e3, e32))
    assert e3 <= 2.0 * e32 + 2e-8, (e3, e32)


@pytest.mark.parametrize("case", [(2, 16, 16, 16, 32, True), (3, 16, 8, 24, 40, True), (2, 12, 16, 32, 32, False)])
def test_conv16_x3_bnb(case):
    """The BatchNorm-backward epilogue on the 16-channel path, against the same epilogue of the fp32-input kernel and against
    the stand-alone BatchNorm backward."""
    o = ops()
    B, Cl, Cn, H, W, relu = case
    yd = rnd(B, Cl, H, W, seed=51).to(DEV)                      # raw conv output of the layer below
    w = rnd(Cn, Cl, 3, 3, seed=52, scale=0.2)
    du = rnd(B, Cn, H, W, seed=53).to(DEV)
    mean = yd.double().mean((0, 2, 3))
    invstd = 1.0 / torch.sqrt(yd.double().var((0, 2, 3), unbiased=False) + 1e-5)
    g_d = (rnd(Cl, seed=54) * 0.2 + 1).to(DEV)
    b_d = (rnd(Cl, seed=55) * 0.2).to(DEV)
    ss = torch.stack([g_d.double() * invstd, b_d.double() - mean * g_d.double() * invstd], 1).float().contiguous()
    mean_f, invstd_f = mean.float().contiguous(), invstd.float().contiguous()
    packed, _, xd = pack_x16(w)
    g, stats, _ = o.conv16_x3(du, packed.data_ptr() + 2 * xd, None, Cl, bnb=(yd, ss, mean_f, relu))
    dg, dbt = torch.empty(Cl, device=DEV), torch.empty(Cl, device=DEV)
    dy = o.bn_bwd_from_stats(g, yd, stats, g_d, mean_f, invstd_f, dg, dbt)
    pk, _, wd = pack(w)
    dz, _, _ = o.conv_fwd(du, None, pk.data_ptr() + 4 * wd, None, Cl, 3)
    dg2, dbt2 = torch.empty(Cl, device=DEV), torch.empty(Cl, device=DEV)
    dy2 = o.bn_bwd(dz, yd, ss, relu, g_d, mean_f, invstd_f, dg2, dbt2)
    sc = float(dy2.abs().max())
    close(dy, dy2, rtol=1e-4, atol=2e-5 * sc, what="fused vs stand-alone dy")
    close(dg, dg2, rtol=1e-4, atol=1e-4 * float(dg2.abs().max()) + 1e-6, what="dgamma")
    close(dbt, dbt2, rtol=1e-4, atol=1e-4 * float(dbt2.abs().max()) + 1e-6, what="dbeta")


@pytest.mark.parametrize("case", [
    # layout, B, C0, C1, Cout, H, W, k
    (1, 20, 32, 32, 64, 32, 64, 3),     # x3, 64-channel row blocks, 160 tiles -> 3 ticket groups
    (1, 2, 16, 0, 32, 16, 16, 3),       # x3, one group
    (1, 3, 64, 0, 96, 8, 16, 1),        # x3 1x1, three 32-channel blocks
    (0, 3, 3, 0, 16, 24, 40, 3),        # fp32 16-channel path
    (0, 4, 8, 0, 40, 16, 32, 1),        # fp32 ragged channels
    (2, 20, 16, 0, 16, 64, 64, 3),      # 16-channel x3 fragments, 320 tiles -> 5 groups
    (1, 33, 16, 0, 32, 128, 128, 3),    # 2112 workgroups: beyond WTPSE_TAIL_MAX_WGS the call runs the stand-alone finalize itself
])
def test_conv_fwd_bnf(case):
    """A convolution that finishes its own BatchNorm statistics (last-arriver tickets) against conv + wtpse_bn_finalize: same output
    bit for bit, same scale/shift, mean, invstd and running statistics up to the order of the fp64 sums; repeatable; tickets
    back at zero."""
    o = ops()
    layout, B, C0, C1, Co, H, W, k = case
    x0 = rnd(B, C0, H, W, seed=61).to(DEV)
    x1 = rnd(B, C1, H, W, seed=62).to(DEV) if C1 else None
    w = rnd(Co, C0 + C1, k, k, seed=63, scale=0.2)
    bias = rnd(Co, seed=64).to(DEV)
    pro0 = torch.stack([rnd(C0, seed=65) * 0.5 + 1.0, rnd(C0, seed=66)], 1).contiguous().to(DEV)
    pro1 = torch.stack([rnd(C1, seed=67) * 0.5 + 1.0, rnd(C1, seed=68)], 1).contiguous().to(DEV) if C1 else None
    gamma, beta = (rnd(Co, seed=69) * 0.2 + 1).to(DEV), (rnd(Co, seed=70) * 0.2).to(DEV)
    if layout == 1:
        packed, xf, _ = pack_x3(w)
        wptr = packed.data_ptr() + 2 * xf
        y_ref, _, stats = o.conv_fwd_x3(x0, x1, wptr, bias, Co, k, pro0, 3, want_stats=True, pro1=pro1)
    elif layout == 2:
        packed, xf, _ = pack_x16(w)
        wptr = packed.data_ptr() + 2 * xf
        y_ref, stats, _ = o.conv16_x3(x0, wptr, bias, Co, pro0, 1, want_stats=True)
    else:
        packed, wf, _ = pack(w)
        wptr = packed.data_ptr() + 4 * wf
        y_ref, _, stats = o.conv_fwd(x0, x1, wptr, bias, Co, k, pro0, 3, want_stats=True, pro1=pro1)
    rm0, rv0 = rnd(Co, seed=71).to(DEV), (rnd(Co, seed=72).abs() + 0.5).to(DEV)
    rm, rv, nbt = rm0.clone(), rv0.clone(), torch.zeros(1, dtype=torch.int64, device=DEV)
    ss_ref, mean_ref, invstd_ref = o.bn_finalize(stats, B * H * W, gamma, beta, rm, rv, nbt)
    got = None
    for it in range(20):
        rm2, rv2, nbt2 = rm0.clone(), rv0.clone(), torch.zeros(1, dtype=torch.int64, device=DEV)
        y, ss, mean, invstd = o.conv_fwd_bnf(x0, x1, wptr, layout, bias, Co, k, pro0, 3 if layout != 2 else 1, pro1, gamma, beta,
                                             rm2, rv2, nbt2)
        cur = (y, ss, mean, invstd, rm2, rv2)
        if got is None:
            got = tuple(t.clone() for t in cur)
            assert torch.equal(y, y_ref)
            close(ss, ss_ref, rtol=2e-6, atol=2e-6, what="scale/shift")
            close(mean, mean_ref, rtol=2e-6, atol=1e-7, what="mean")
            close(invstd, invstd_ref, rtol=2e-6, atol=1e-7, what="invstd")
            close(rm2, rm, rtol=2e-6, atol=1e-7, what="running mean")
            close(rv2, rv, rtol=2e-6, atol=1e-7, what="running var")
            assert int(nbt2) == 1
        else:
            for a, b in zip(cur, got):
                assert torch.equal(a, b), "launch %d differs" % it
    torch.cuda.synchronize()
    assert all(int(t[0].abs().sum()) == 0 for t in o._TICKETS.values()), "tickets must be left at zero"


X3R_CASES = [
    # B, C0, C1, Cout, H, W: every tile shape of conv_x3r_k (2 x 2 waves / 1 x 4 waves, 32- and 16-wide tiles, 128-pixel tiles)
    (20, 32, 32, 64, 32, 64),      # WM 2, 32-wide tiles, concat, two chunks per input
    (20, 64, 0, 64, 32, 32),       # WM 2, four chunks
    (36, 64, 0, 128, 16, 16),      # WM 2, 16-wide tiles
    (2, 16, 16, 32, 24, 48),       # WM 1, ragged 8x32 tiles
    (2, 40, 0, 96, 12, 20),        # WM 1, ragged chunk, three cout blocks
    (2, 128, 0, 64, 16, 16),       # WM 1, 128-pixel tiles (few workgroups on a 16-wide map)
    (3, 16, 0, 32, 40, 72),        # a single chunk (the prefetch runs past the end)
    (32, 64, 0, 64, 64, 64),       # x3_mt2: 512 tiles x one 64-channel block — conv_x3r_k<2,1,4,5> (2 x 2 waves) against conv_x3_k<3,2,5> (1 x 4)
    (64, 32, 32, 128, 32, 16),     # x3_mt2 on 16-wide tiles: conv_x3r_k<2,1,4,4>
]


# cases in which a launch runs 64-channel blocks as 2 x 2 waves on one side of the comparison and another split of the tile on the
# other (conv_x3_k: 1 x 4 waves; or the 32-channel blocks on 128-pixel tiles): the statistics partials are summed over other groups
# of pixels there — equal to fp32 summation order, not bitwise
X3R_WM2_CASES = {(32, 64, 0, 64, 64, 64), (64, 32, 32, 128, 32, 16)}


def _x3_case_runner(case):
    """-> (ops, run): run() launches every kind of 3x3 x3 convolution on the case's operands and returns all outputs."""
    o = ops()
    B, C0, C1, Co, H, W = case
    dev = DEV
    x0 = rnd(B, C0, H, W, seed=31).to(dev)
    x1 = rnd(B, C1, H, W, seed=32).to(dev) if C1 else None
    w = rnd(Co, C0 + C1, 3, 3, seed=33, scale=0.2)
    b = rnd(Co, seed=34).to(dev)
    pro = torch.stack([rnd(C0 + C1, seed=35) * 0.5 + 1.0, rnd(C0 + C1, seed=36)], 1).contiguous().to(dev)
    pro0, pro1 = pro[:C0].contiguous(), (pro[C0:].contiguous() if C1 else None)
    dy = rnd(B, Co, H, W, seed=37).to(dev)
    ref_act = rnd(B, C0 + C1, H, W, seed=38).to(dev)
    bn_y = rnd(B, C0, H, W, seed=39).to(dev)
    bn_ss = torch.stack([rnd(C0, seed=40) * 0.3 + 1.0, rnd(C0, seed=41) * 0.2], 1).contiguous().to(dev)
    bn_mean = (rnd(C0, seed=42) * 0.1).to(dev)
    packed, xf, xd = pack_x3(w)

    def run():
        outs = []
        y, _, st = o.conv_fwd_x3(x0, x1, packed.data_ptr() + 2 * xf, b, Co, 3, pro0, 1, want_stats=True, pro1=pro1)
        outs += [y, st]
        outs.append(o.conv_fwd_x3(x0, x1, packed.data_ptr() + 2 * xf, None, Co, 3, relu_out=True)[0])
        d0, d1, _ = o.conv_fwd_x3(dy, None, packed.data_ptr() + 2 * xd, None, C0 + C1, 3, split=(C0 if C1 else None))
        outs += [d0] + ([d1] if C1 else [])
        if not C1:
            outs.append(o.conv_fwd_x3(dy, None, packed.data_ptr() + 2 * xd, None, C0, 3, mask_ref=ref_act)[0])
        if C0 % 16 == 0:
            g0, g1, st, _ = o.dgrad_bnb(dy, packed.data_ptr() + 2 * xd, 1, C0 + C1, 3, bn_y, bn_ss, bn_mean, True,
                                        split=(C0 if C1 else None))
            outs += [g0, st] + ([g1] if C1 else [])
        torch.cuda.synchronize()
        return outs

    return o, run


@pytest.mark.parametrize("case", X3R_CASES)
def test_x3r_equals_x3(case):
    """conv_x3r_k (weights fed from registers, double-buffered input tile; the default for 3x3) against conv_x3_k (weights staged in
    LDS): same products in the same order on every accumulator, so every output — result, BatchNorm partials, masked data gradient,
    BatchNorm-backward epilogue — must be BITWISE equal.  All other x3 tests run the default kernel against stock PyTorch."""
    o, run = _x3_case_runner(case)
    assert o.lib().query("wtpse_x3r_enable", 2) == 1, "conv_x3r_k on the 64-channel blocks must be the default"
    try:
        new = run()                              # mode 2: conv_x3r_k on every 3x3 launch, also the shapes it is not the default for
        o.lib().query("wtpse_x3r_enable", 0)
        old = run()
    finally:
        o.lib().query("wtpse_x3r_enable", 1)
    # 64-channel blocks: the two kernels split a tile between their waves differently (2 x 2 waves of 32 channels x 128 pixels against
    # 1 x 4 of 64 x 64), so a channel's BatchNorm partial is summed over other groups of pixels — convolution outputs and masked
    # gradients stay bitwise equal, the statistics partials agree to fp32 summation order (stated and tested explicitly)
    _assert_same_outputs(new, old, stats_exact=case not in X3R_WM2_CASES)


def _assert_same_outputs(new, old, stats_exact=True):
    """Bitwise, except statistics partials (the 3-d outputs) that come in a different number of rows (a launch that took the 128-pixel
    tiling of x3_half on one side only) or, with stats_exact = False, from a different split of the tile between the waves: their
    column sums must agree to fp32 summation order."""
    for i, (a_, b_) in enumerate(zip(new, old)):
        if a_.shape == b_.shape and (stats_exact or a_.dim() != 3):
            assert torch.equal(a_, b_), "output %d differs: max |d| = %g" % (i, float((a_ - b_).abs().max()))
        else:
            assert a_.dim() == 3 and a_.shape[1:] == b_.shape[1:], (a_.shape, b_.shape)
            sa, sb = a_.double().sum(0), b_.double().sum(0)
            scale = b_.double().abs().sum(0) + 1e-30
            assert float(((sa - sb).abs() / scale).max()) < 1e-5, "statistics %d: column sums differ" % i


@pytest.mark.parametrize("case", X3R_CASES + [(32, 64, 64, 128, 64, 64), (32, 32, 0, 32, 128, 128)])
def test_xcd_order_equals_dispatch_order(case):
    """The XCD-aware workgroup order (ConvX3Args::xcd_tiles: each XCD a contiguous range of tiles, a tile's output-channel blocks back
    to back) against plain dispatch order: the same workgroups compute the same things, the BatchNorm tails fold the same partials
    in the same order — every output bitwise equal.  The two large cases are launches of the benchmark's step (8192 / 4096 tiles)."""
    o, run = _x3_case_runner(case)
    assert o.lib().query("wtpse_x3_xcd", 0) == 1, "the XCD-aware order must be the default"
    try:
        plain = run()
        o.lib().query("wtpse_x3_xcd", 1)
        xcd = run()
    finally:
        o.lib().query("wtpse_x3_xcd", 1)
    for i, (a_, b_) in enumerate(zip(xcd, plain)):
        assert torch.equal(a_, b_), "output %d differs: max |d| = %g" % (i, float((a_ - b_).abs().max()))


X3_HALF_CASES = [
    # mid-sized launches: too few 256-pixel tiles for 64-channel blocks, output channels a multiple of 64 -> conv_x3r_k<2,1,2> on 128-pixel tiles
    (32, 64, 0, 128, 32, 32),      # 32-wide tiles (8 x ... 4 rows), two output-channel blocks
    (32, 32, 32, 128, 32, 32),     # concat
    (64, 32, 0, 256, 16, 16),      # 16-wide tiles, four output-channel blocks
    (16, 48, 0, 64, 40, 56),       # ragged tiles
]


@pytest.mark.parametrize("case", X3_HALF_CASES)
def test_x3_half_tiling_equals_32_channel_blocks(case):
    """The 64-channel blocks on 128-pixel tiles (x3_half, conv_x3.hip) against what these launches ran on before (32-channel blocks of
    conv_x3_k, WTPSE_X3R=0): every accumulator sees the same products in the same order — outputs and masked gradients BITWISE equal;
    the statistics partials come in a different number of rows (other tiles), their column sums agree to fp32 summation order."""
    o, run = _x3_case_runner(case)
    B, C0, C1, Co, H, W = case
    n_half = o.lib().query("wtpse_conv_x3_stats_blocks", B, H, W, Co, 3)
    try:
        o.lib().query("wtpse_x3r_enable", 0)
        n_old = o.lib().query("wtpse_conv_x3_stats_blocks", B, H, W, Co, 3)
        old = run()
    finally:
        o.lib().query("wtpse_x3r_enable", 1)
    assert n_half == 2 * n_old or (H % 8 or W % 16), "the case must take the 128-pixel tiling (%d vs %d rows)" % (n_half, n_old)
    assert n_half > n_old
    new = run()
    assert any(a_.shape != b_.shape for a_, b_ in zip(new, old)), "the statistics partials must come in more rows"
    _assert_same_outputs(new, old)


# ---------------------------------------------------------------- x2h: the range side of the two-fp16-term arithmetic
def _x2h_only():
    o = ops()
    if o.x3_terms() != 2:
        pytest.skip("x2h is not the active arithmetic (WTPSE_X3_TERMS)")
    return o


@pytest.mark.parametrize("wscale,gscale", [(0.2, 1.0), (3e-6, 1.0), (4e3, 1.0), (0.2, 1e-8), (0.2, 3e5), (1e-4, 1e-7)])
def test_x2h_scales_follow_the_data(wscale, gscale):
    """fp16 has five exponent bits; x2h keeps fp32 accuracy over fp32's range by scaling every operand tensor with a power of two:
    the weights from the layer's largest magnitude (pack kernel), a GRADIENT operand from its amax table.  Data gradients and weight
    gradients with weights from 3e-6 to 4e3 and gradients from 1e-8 (a mean loss over millions of pixels) to 3e5, against fp64: the
    relative L2 error must stay at the level of the O(1) case (<= 4e-7), whatever the magnitudes."""
    o = _x2h_only()
    B, Ci, Co, H, W = 4, 64, 64, 32, 32
    w = rnd(Co, Ci, 3, 3, seed=71, scale=wscale)
    dy = rnd(B, Co, H, W, seed=72) * gscale
    x = rnd(B, Ci, H, W, seed=73)
    packed, _, xd = pack_x3(w)
    dyd = dy.to(DEV)
    am = o.amax_of(dyd)
    d, _, _ = o.conv_fwd_x3(dyd, None, packed.data_ptr() + 2 * xd, None, Ci, 3, in_amax=am)
    ref = F.conv_transpose2d(dy.double(), w.double(), padding=1)
    e = float((d.cpu().double() - ref).norm() / ref.norm())
    assert e <= 4e-7, ("dgrad", wscale, gscale, e)
    dw = torch.empty(Co, Ci, 3, 3, device=DEV)
    o.conv_wgrad_r(dyd, x.to(DEV), None, dw, dy_amax=am)
    refw = torch.nn.grad.conv2d_weight(x.double(), (Co, Ci, 3, 3), dy.double(), padding=1)
    ew = float((dw.cpu().double() - refw).norm() / refw.norm())
    assert ew <= 4e-7, ("wgrad", wscale, gscale, ew)
    # the table is what the tensor's largest magnitude is: float bits, the maximum over the 64 shards
    got = float(am.view(torch.float32).max())
    assert got == float(dy.abs().max()), (got, float(dy.abs().max()))


ACT_SCALES = [1e-6, 1e-3, 1.0, 30.0, 1e4]


def amax_value(tab):
    """The value an amax table holds: float bits, the maximum over the 64 shards (one word per 64-byte line)."""
    return float(tab[::16].view(torch.float32).max())


@pytest.mark.parametrize("ascale", ACT_SCALES)
def test_x2h_activation_scales_follow_the_data(ascale):
    """Round 6: a FORWARD activation is scaled by the power of two that brings a bound of its largest
    magnitude — an amax table, as for gradients — into [2^14, 2^15) instead of the fixed 2^2 of round 5 (which kept fp32 accuracy only
    for 2^-5 <= |x| < 2^14).  Forward convolution (64-channel and 32-channel blocks, concat with one table per half), the 16-channel
    kernel and the weight gradient's X operand with activations from 1e-6 to 1e4, against fp64: the relative L2 error stays at the
    level of the O(1) case (<= 4e-7) at every scale, and nothing overflows."""
    o = _x2h_only()
    B, C0, C1, Co, H, W = 4, 32, 32, 64, 32, 32
    w = rnd(Co, C0 + C1, 3, 3, seed=171, scale=0.2)
    x0 = rnd(B, C0, H, W, seed=172) * ascale
    x1 = rnd(B, C1, H, W, seed=173) * (ascale * 0.01)          # the second half of the concat a hundred times smaller
    packed, xf, _ = pack_x3(w)
    x0d, x1d = x0.to(DEV), x1.to(DEV)
    a0, a1 = o.amax_of(x0d), o.amax_of(x1d)
    assert amax_value(a0) == float(x0.abs().max())
    out_amax = o.fwd_amax_table(x0d.device)
    y, _, _ = o.conv_fwd_x3(x0d, x1d, packed.data_ptr() + 2 * xf, None, Co, 3, in_amax=a0, in_amax1=a1, out_amax=out_amax)
    ref = F.conv2d(torch.cat([x0, x1], 1).double(), w.double(), padding=1)
    e = float((y.cpu().double() - ref).norm() / ref.norm())
    assert bool(torch.isfinite(y).all()) and e <= 4e-7, ("conv_fwd_x3", ascale, e)
    # the producer's epilogue left the amax of what it stored
    assert amax_value(out_amax) == float(y.abs().max())
    assert y.wt_amax is out_amax
    # a prologue in front: the bound is of the tensor AS LOADED (relu(x * 3 - 1 * ascale)), from wtpse_act_bound
    pro = torch.stack([torch.full((C0,), 3.0), torch.full((C0,), -1.0 * ascale)], 1).contiguous().to(DEV)
    ab = o.act_bound(pro, a0)
    want = 3.0 * float(x0.abs().max()) + ascale
    assert abs(amax_value(ab) - want) <= 1e-6 * want
    w1 = rnd(32, C0, 3, 3, seed=174, scale=0.2)                 # 32-channel blocks (conv_x3_k)
    p1, xf1, _ = pack_x3(w1)
    y1, _, _ = o.conv_fwd_x3(x0d, None, p1.data_ptr() + 2 * xf1, None, 32, 3, pro0=pro, pro_relu=1, in_amax=ab)
    act = F.relu(x0.double() * 3.0 - ascale)
    ref1 = F.conv2d(act, w1.double(), padding=1)
    e1 = float((y1.cpu().double() - ref1).norm() / ref1.norm())
    assert bool(torch.isfinite(y1).all()) and e1 <= 4e-7, ("conv_fwd_x3 + prologue", ascale, e1)
    # weight gradient: X scaled from its bound, dY from its amax
    dy = rnd(B, 32, H, W, seed=175) * 1e-5
    dyd = dy.to(DEV)
    dw = torch.empty(32, C0, 3, 3, device=DEV)
    o.conv_wgrad_r(dyd, x0d, None, dw, pro0=pro, pro_relu=1, dy_amax=o.amax_of(dyd), x_amax0=ab)
    refw = torch.nn.grad.conv2d_weight(act, (32, C0, 3, 3), dy.double(), padding=1)
    ew = float((dw.cpu().double() - refw).norm() / refw.norm())
    assert bool(torch.isfinite(dw).all()) and ew <= 4e-7, ("wgrad", ascale, ew)
    # the 16-channel kernel (conv_fwd_k MODE 4)
    w16 = rnd(16, 16, 3, 3, seed=176, scale=0.3)
    x16 = rnd(B, 16, H, W, seed=177) * ascale
    p16, f16, _ = pack_x16(w16)
    y16, _, _ = o.conv16_x3(x16.to(DEV), p16.data_ptr() + 2 * f16, None, 16, in_amax=o.amax_of(x16.to(DEV)))
    ref16 = F.conv2d(x16.double(), w16.double(), padding=1)
    e16 = float((y16.cpu().double() - ref16).norm() / ref16.norm())
    assert bool(torch.isfinite(y16).all()) and e16 <= 4e-7, ("conv16_x3", ascale, e16)


@pytest.mark.parametrize("gscale", [1e-3, 1.0, 30.0, 1e4])
def test_x2h_train_mode_batchnorm_bound(gscale):
    """The bound a train-mode BatchNorm's output travels with needs no look at the data: |gamma| sqrt(N - 1) + |beta| per channel
    (Samuelson's inequality), left in the activation's amax table by the statistics' fold — in the launch (wtpse_conv_fwd_bnf, up to
    8192 workgroups) or by wtpse_bn_finalize.  conv -> BatchNorm(gamma = gscale x O(1)) -> ReLU -> conv through the engine's own block
    schedule (nn.convbn_fwd) against fp64 at gamma scales from 1e-3 to 1e4: the second convolution's output stays within the fp32
    noise of the first (<= 1e-6 relative L2, the same at every scale), finite everywhere; the table holds the bound."""
    o = _x2h_only()
    import math
    from wtpse_hip import nn as E
    B, C, H, W = 4, 64, 32, 32

    class Two(E.HipNet):
        def __init__(self):
            super().__init__()
            self.conv1, self.bn1 = E.ConvP(C, C, 3), E.BNP(C)
            self.conv2, self.bn2 = E.ConvP(C, C, 3), E.BNP(C)
            self._finish_init()
    net = Two().to(DEV)
    g = torch.Generator().manual_seed(181)
    with torch.no_grad():
        net.bn1.weight.copy_(((torch.rand(C, generator=g) * 0.4 + 0.8) * gscale).to(DEV))
        net.bn1.bias.copy_(((torch.rand(C, generator=g) - 0.5) * 0.2 * gscale).to(DEV))
    net.train()
    net.ensure_ready(repack=True)
    x = rnd(B, C, H, W, seed=182)
    with o.fwd_scope(torch.device(DEV)):
        a1, _ = E.convbn_fwd(net.conv1, net.bn1, x.to(DEV), None, True, True, want_tape=False)
        a2, _ = E.convbn_fwd(net.conv2, net.bn2, a1, None, True, True, want_tape=False)
    assert a1.amax is not None
    N = B * H * W
    want = float((net.bn1.weight.abs() * math.sqrt(N - 1) + net.bn1.bias.abs()).max())
    got = amax_value(a1.amax)
    assert abs(got - want) <= 1e-5 * want, (got, want)
    sd = {k: v.detach().cpu().double() for k, v in net.state_dict().items()}
    y1 = F.conv2d(x.double(), sd["conv1.weight"], sd["conv1.bias"], padding=1)
    z1 = F.relu(F.batch_norm(y1, None, None, sd["bn1.weight"], sd["bn1.bias"], True, 0.1, 1e-5))
    assert float(z1.abs().max()) <= want                     # it IS a bound
    y2 = F.conv2d(z1, sd["conv2.weight"], sd["conv2.bias"], padding=1)
    e = float((a2.t.cpu().double() - y2).norm() / y2.norm())
    assert bool(torch.isfinite(a2.t).all()) and e <= 1e-6, (gscale, e)


def test_nonfinite_forward_operands_are_stored_as_nan():
    """what round 5 read as 'the fp16 MFMA returns -inf for some NaN operands' was the epilogue's output clamp — v_max_f32
    returns the OTHER operand for a NaN, so a NaN accumulator was stored as -inf (no output ReLU) or 0 (output ReLU) in EVERY
    arithmetic.  Forward launches now clamp with a compare + select that keeps the NaN, as torch's conv / ReLU do (the reference's
    `isnan` checks — the mu scrub, shape_networks.py:490 — expect to find it).  NaN and inf inputs; 64- and 32-channel blocks with and
    without an output ReLU, the 16-channel kernel and the fp32-input MFMA kernel."""
    o = ops()
    B, Ci, H, W = 2, 32, 16, 32
    for Co, bad, relu in ((64, float("nan"), False), (32, float("nan"), True), (64, float("inf"), False)):
        w = rnd(Co, Ci, 3, 3, seed=191, scale=0.2)
        packed, xf, _ = pack_x3(w)
        x = rnd(B, Ci, H, W, seed=192)
        x[1, 2, 3, 4] = bad
        xd = x.to(DEV)
        y, _, _ = o.conv_fwd_x3(xd, None, packed.data_ptr() + 2 * xf, None, Co, 3, relu_out=relu, in_amax=o.amax_of(xd))
        assert bool(torch.isnan(y[1, :, 2:5, 3:6]).all()) or (bad == float("inf") and o.x3_terms() != 2), (Co, bad, y[1, :4, 2:5, 3:6])
        assert not bool(torch.isfinite(y[1, :, 2:5, 3:6]).any()) and bool(torch.isfinite(y[0]).all())
        pk, wf, _ = pack(w)
        y32, _, _ = o.conv_fwd(xd, None, pk.data_ptr() + 4 * wf, None, Co, 3, relu_out=relu)
        assert not bool(torch.isfinite(y32[1, :, 2:5, 3:6]).any()) and bool(torch.isfinite(y32[0]).all())
        if bad != bad:
            assert bool(torch.isnan(y32[1, :, 2:5, 3:6]).all())
    w16 = rnd(16, 16, 3, 3, seed=193, scale=0.3)
    x16 = rnd(B, 16, H, W, seed=194)
    x16[0, 5, 8, 9] = float("nan")
    p16, f16, _ = pack_x16(w16)
    y16, _, _ = o.conv16_x3(x16.to(DEV), p16.data_ptr() + 2 * f16, None, 16, relu_out=True, in_amax=o.amax_of(x16.to(DEV)))
    assert bool(torch.isnan(y16[0, :, 7:10, 8:11]).all()) and bool(torch.isfinite(y16[1]).all())


def test_x2h_out_of_range_activation_is_loud():
    """WITHOUT a bound table (a bare C-ABI caller: the engine always passes one, test_x2h_activation_scales_follow_the_data) a forward
    activation is scaled by the fixed 2^2 of round 5 (full precision for 2^-5 <= |x| < 2^14).  A value beyond 65504 / 4 overflows its fp16
    term: the outputs it feeds are NON-FINITE — loud, the caller's NaN check fires — never a wrong finite number; every output that does
    not see the outlier is as accurate as without it.  Values far below 2^-5 lose relative, not absolute, precision (absolute error
    <= 2^-27 per element and unit weight)."""
    o = _x2h_only()
    B, Ci, Co, H, W = 2, 32, 32, 16, 32
    w = rnd(Co, Ci, 3, 3, seed=81, scale=0.2)
    x = rnd(B, Ci, H, W, seed=82)
    x[0, 3, 8, 16] = 1e7                      # one outlier far beyond the format's range
    packed, xf, _ = pack_x3(w)
    y, _, _ = o.conv_fwd_x3(x.to(DEV), None, packed.data_ptr() + 2 * xf, None, Co, 3)
    y = y.cpu()
    ref = F.conv2d(x.double(), w.double(), padding=1)
    clean = torch.ones_like(ref, dtype=torch.bool)
    clean[0, :, 7:10, 15:18] = False          # the 3x3 footprint of the outlier
    assert bool(torch.isfinite(y[clean]).all()) and not bool(torch.isfinite(y[~clean]).any())
    e = float((y.double() - ref)[clean].norm() / ref[clean].norm())
    assert e <= 4e-7, e
    # tiny activations: absolute error bounded by 2^-27 x sum |w| per output
    xt = rnd(B, Ci, H, W, seed=83) * 1e-5
    yt, _, _ = o.conv_fwd_x3(xt.to(DEV), None, packed.data_ptr() + 2 * xf, None, Co, 3)
    reft = F.conv2d(xt.double(), w.double(), padding=1)
    bound = 2.0 ** -27 * float(w.abs().sum((1, 2, 3)).max())
    assert float((yt.cpu().double() - reft).abs().max()) <= bound, (float((yt.cpu().double() - reft).abs().max()), bound)


def test_x2h_zero_and_nonfinite_gradients():
    """amax = 0 (an all-zero gradient) and a non-finite amax fall back to scale 1: zeros stay zeros, a NaN stays non-finite in every output
    it feeds (and only there), as it would in fp32."""
    o = _x2h_only()
    B, Ci, Co, H, W = 2, 32, 32, 16, 32
    w = rnd(Co, Ci, 3, 3, seed=91, scale=0.2)
    packed, _, xd = pack_x3(w)
    z = torch.zeros(B, Co, H, W, device=DEV)
    d, _, _ = o.conv_fwd_x3(z, None, packed.data_ptr() + 2 * xd, None, Ci, 3, in_amax=o.amax_of(z))
    assert float(d.abs().max()) == 0.0
    n = rnd(B, Co, H, W, seed=92).to(DEV)
    n[1, 2, 3, 4] = float("nan")
    d, _, _ = o.conv_fwd_x3(n, None, packed.data_ptr() + 2 * xd, None, Ci, 3, in_amax=o.amax_of(n))
    # (non-finite, not necessarily NaN: a data gradient's epilogue clamps with v_max_f32, which turns a NaN into its -inf bound — loud either way)
    assert not bool(torch.isfinite(d[1, :, 2:5, 3:6]).any()) and bool(torch.isfinite(d[0]).all())
