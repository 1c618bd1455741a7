"""The `bf16` mode of BASELINE.json configs[1] (seg-net only, bf16): wtpse_x3_terms(1) — every fp32 operand of the MFMA-bound
convolutions (forward, data gradient, weight gradient of the layers with more than 16 output channels) rounded to ONE bf16 term
(round to nearest even), one bf16 MFMA product per multiply, fp32 accumulation.

This mode is OUTSIDE the 1e-4 parity bar by construction (8 significant bits per operand).  What is checked, and at what tolerance:
  * kernels: against stock PyTorch convolutions in fp64 on operands rounded to bf16 the same way — what is left is the fp32
    accumulation order: rtol 2e-5 of the result's scale;
  * network (the seg-only update / predict of configs[1]): logits against the CPU oracle evaluated with bf16-rounded operands in
    exactly those layers (same rounding points: after the fused BatchNorm+ReLU prologue).  Two bf16 evaluations of a 25-layer
    network do not agree to 1e-6: a rounding that falls the other way is a 4e-3 step.  The yardstick is measured in the test — the
    oracle's own bf16 evaluation under fp64 instead of fp32 accumulation, same operands: ~1e-3 in eval mode, ~1e-1 in train mode
    (batch statistics renormalise every layer) — and the HIP path must be within 3x of it (eval mode: also within 5e-3 absolute).
The fp32 workloads never run in this mode: the switch is process-global and restored by every test."""
import pytest
import torch
import torch.nn.functional as F

from test_kernels_gpu import rnd, ops, DEV
from test_conv_x3_gpu import pack_x3

pytestmark = pytest.mark.gpu


def bf16r(t):
    return t.to(torch.bfloat16).to(t.dtype)


@pytest.fixture
def bf16_mode():
    o = ops()
    was = o.lib().query("wtpse_x3_terms", 1)
    assert was in (2, 3), "the library must default to an fp32-accuracy arithmetic (x2h or x3)"
    yield o
    o.lib().query("wtpse_x3_terms", was)


def rel(a, b):
    return float((a.double().cpu() - b.double()).abs().max() / b.double().abs().max())


@pytest.mark.parametrize("case", [(2, 32, 32, 64, 24, 48, 3), (20, 32, 32, 64, 32, 64, 3), (2, 16, 0, 32, 16, 32, 3), (2, 256, 0, 128, 8, 8, 1),
                                  (3, 64, 0, 64, 16, 16, 3)])
def test_bf16_conv_forward_dgrad(bf16_mode, case):
    o = bf16_mode
    B, C0, C1, Co, H, W, k = case
    x0 = rnd(B, C0, H, W, seed=1)
    x1 = rnd(B, C1, H, W, seed=2) if C1 else None
    w = rnd(Co, C0 + C1, k, k, seed=3, scale=0.2)
    b = rnd(Co, seed=4)
    pro = torch.stack([rnd(C0 + C1, seed=7) * 0.5 + 1.0, rnd(C0 + C1, seed=8)], 1).contiguous()
    xin = torch.cat([x0, x1], 1) if C1 else x0
    act = F.relu(xin * pro[:, 0].view(1, -1, 1, 1) + pro[:, 1].view(1, -1, 1, 1))            # the prologue, in fp32 as the kernel does
    act32 = torch.addcmul(pro[:, 1].view(1, -1, 1, 1), xin, pro[:, 0].view(1, -1, 1, 1)).clamp_min(0)
    ref = F.conv2d(bf16r(act32).double(), bf16r(w).double(), b.double(), padding=k // 2)
    packed, xf, xd = pack_x3(w)
    y, _, stats = o.conv_fwd_x3(x0.to(DEV), x1.to(DEV) if C1 else None, packed.data_ptr() + 2 * xf, b.to(DEV), Co, k,
                                pro0=pro[:C0].contiguous().to(DEV), pro1=(pro[C0:].contiguous().to(DEV) if C1 else None), pro_relu=3,
                                want_stats=True)
    # (an activation within one fp32 ulp of a bf16 rounding boundary may round the other way after the fused multiply-add: allow a
    # handful of such elements their bf16 step)
    d = (y.double().cpu() - ref).abs() / ref.abs().max()
    assert float(d.median()) < 2e-6 and float((d > 2e-5).float().mean()) < 2e-3, (float(d.max()), float(d.median()))
    assert float(d.max()) < 3e-3
    s = stats.double().sum(0).cpu()
    assert rel(s[:, 0], y.double().cpu().sum((0, 2, 3))) < 1e-5
    # data gradient: dX = conv_transpose(bf16(dY), bf16(W))
    dy = rnd(B, Co, H, W, seed=15)
    refd = F.conv_transpose2d(bf16r(dy).double(), bf16r(w).double(), None, padding=k // 2)
    d0, d1, _ = o.conv_fwd_x3(dy.to(DEV), None, packed.data_ptr() + 2 * xd, None, C0 + C1, k, split=(C0 if C1 else None))
    got = torch.cat([d0, d1], 1) if C1 else d0
    assert rel(got, refd) < 2e-5
    # and it IS a different arithmetic: the fp32-accuracy result is ~1e-3 away
    o.lib().query("wtpse_x3_terms", 3)
    y3, _, _ = o.conv_fwd_x3(x0.to(DEV), x1.to(DEV) if C1 else None, packed.data_ptr() + 2 * xf, b.to(DEV), Co, k,
                             pro0=pro[:C0].contiguous().to(DEV), pro1=(pro[C0:].contiguous().to(DEV) if C1 else None), pro_relu=3)
    o.lib().query("wtpse_x3_terms", 1)
    assert 1e-4 < rel(y, y3.double().cpu()) < 3e-2


@pytest.mark.parametrize("case", [(2, 32, 0, 32, 16, 32), (3, 32, 32, 64, 24, 64), (2, 64, 0, 32, 12, 32), (4, 128, 0, 64, 32, 32)])
def test_bf16_wgrad(bf16_mode, case):
    o = bf16_mode
    B, C0, C1, Co, H, W = case
    x0 = rnd(B, C0, H, W, seed=21)
    x1 = rnd(B, C1, H, W, seed=22) if C1 else None
    dy = rnd(B, Co, H, W, seed=23)
    xin = torch.cat([x0, x1], 1) if C1 else x0
    xr = bf16r(xin).double().requires_grad_(False)
    w = torch.zeros(Co, C0 + C1, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv2d(xr, w, None, padding=1).backward(bf16r(dy).double())
    dw = torch.empty(Co, C0 + C1, 3, 3, device=DEV)
    assert o.wgrad_r_supported(C0 + C1, Co, 3, C0 if C1 else 16, W)
    o.conv_wgrad_r(dy.to(DEV), x0.to(DEV), x1.to(DEV) if C1 else None, dw)
    assert rel(dw, w.grad) < 2e-5


def test_bf16_seg_net_vs_bf16_oracle(bf16_mode):
    """configs[1]: WT_PSE with whitening = shape_prior = False (plain U-Net), update() in train mode and predict() in eval mode."""
    import oracle.wtpse_cpu as O
    from oracle.inputs import make_inputs
    from test_parity_gpu import build_nets, HP
    from wtpse_hip import nn as E
    hp = dict(HP, whitening=False, shape_prior=False)
    B, pb, H = 6, 2, 256         # the benchmark's resolution: 16x16 deepest maps (at 64x64 the train-mode BatchNorm of 4x4 maps turns
    img, od, _ = make_inputs(77, B, H, H)          # one flipped bf16 rounding into a 4 % logit change: measured 1.1e-1 there)
    main, _, _, _ = build_nets(pb, full=False)
    sd = {k: v.detach().cpu().clone() for k, v in main.state_dict().items()}
    # ConvU in the reference's written order (upsample -> 1x1 conv) for this comparison: with bf16 operands the schedule's exact
    # identity conv(up(x)) = up(conv(x)) holds only up to WHERE the operands are rounded, and the oracle rounds where the reference
    # graph has its convolutions
    swapped, E.CONVU_CONV_FIRST = E.CONVU_CONV_FIRST, False
    try:
        main.train()
        with torch.no_grad():
            out = main.update(img.to(DEV), od.to(DEV))[0].cpu()
        main.eval()
        with torch.no_grad():
            pred = main.predict(None, img.to(DEV))[0].cpu()
    finally:
        E.CONVU_CONV_FIRST = swapped

    # the oracle with bf16-rounded operands in the layers the x3 kernels run (nn.x3_eligible: > 16 output channels, 16..256 input
    # channels, 3x3 or >= 64 input channels); rounding AFTER the activation that feeds the convolution, as the kernels' loaders do
    real = F.conv2d

    def conv_bf16(x, w, b=None, stride=1, padding=0, *a, **kw):
        if E.x3_eligible(w.shape[1], w.shape[0], w.shape[2]):
            x, w = bf16r(x), bf16r(w)
        return real(x, w, b, stride, padding, *a, **kw)

    def run(fn):
        O.F.conv2d = fn
        try:
            with torch.no_grad():
                o_up = O.wt_pse_update(dict(sd), hp, img, od, img, False, None, 3, pb)[0]
                sd_eval = {k: v.clone() for k, v in sd.items()}
                o_pr = O.wt_pse_predict(sd_eval, None, hp, img, False)[0]
            return o_up, o_pr
        finally:
            O.F.conv2d = real
    def conv_bf16_acc64(x, w, b=None, stride=1, padding=0, *a, **kw):      # the same bf16 operands, accumulated in fp64
        if E.x3_eligible(w.shape[1], w.shape[0], w.shape[2]):
            return real(bf16r(x).double(), bf16r(w).double(), None if b is None else b.double(), stride, padding, *a, **kw).float()
        return real(x, w, b, stride, padding, *a, **kw)
    up16, pr16 = run(conv_bf16)
    up64, pr64 = run(conv_bf16_acc64)
    up32, pr32 = run(real)
    # yardstick: how far the ORACLE's own bf16 evaluation moves when nothing but the accumulation precision changes (same operands,
    # same rounding points).  Eval mode ~1e-3; train mode ~1e-1: batch statistics renormalise every layer, an accumulation-order
    # difference of 1e-7 flips a bf16 rounding here and there, and 25 layers of that decorrelate two evaluations almost fully
    y_up, y_pr = float((up16 - up64).abs().max()), float((pr16 - pr64).abs().max())
    e_up, e_pr = float((out - up16).abs().max()), float((pred - pr16).abs().max())
    f_up, f_pr = float((out - up32).abs().max()), float((pred - pr32).abs().max())
    print("bf16 mode, logits: vs the bf16-operand oracle %.2e (update) %.2e (predict); the oracle's own bf16 evaluation under another "
          "accumulation precision %.2e / %.2e; vs the fp32 oracle %.2e / %.2e; logit scale %.2f"
          % (e_up, e_pr, y_up, y_pr, f_up, f_pr, float(up32.abs().max())))
    # tolerance: 3x that yardstick (+1e-3), and — eval mode, a smooth function of the operands — 5e-3 absolute
    assert e_pr < 5e-3 and e_pr <= 3 * y_pr + 1e-3, (e_pr, y_pr)
    assert e_up <= 3 * y_up + 1e-3 and e_up < f_up, (e_up, y_up, f_up)
    assert f_up > 1e-4 or f_pr > 1e-4, "the bf16 mode cannot be within the fp32 parity bar: is it active?"


def test_bf16_block_backward_vs_bf16_oracle(bf16_mode):
    """bf16-mode GRADIENTS through chained layers, not kernel by kernel.  A ConvD block (max-pool, three conv +
    train-mode BatchNorm layers, 32 -> 64 channels: every convolution of it runs with one bf16 term per operand in all three
    directions — forward, data gradient, weight gradient on wgrad_r_k<2,2,...,TERMS 1>) forward + backward through the engine's own
    schedule (nn.convd_fwd / convd_bwd: fused statistics, prologues, BatchNorm-backward epilogues), against the oracle's ConvD
    (oracle/wtpse_cpu.py: algorithms.py:897-917) whose convolutions round their operands to bf16 where the kernels do — x and W
    in the forward, dY and W in the data gradient, dY and x in the weight gradient — and accumulate in fp64.
    Yardstick, as for the logits above: the distance of the SAME bf16-operand oracle accumulated in fp32 (what changes is only the
    accumulation: one flipped bf16 rounding per few thousand elements, renormalised by three BatchNorms); the HIP gradients must sit
    within 3x that distance + 1e-3 of each tensor's scale.  (Whole-network train-mode gradients decorrelate almost fully between two
    bf16 evaluations — measured 1e-1 on the logits above — and eval-mode BatchNorm has no backward in the reference's training
    path: three layers is the depth at which the comparison still measures the kernels.)"""
    import oracle.wtpse_cpu as O
    from oracle.filler import fill_state_dict
    from oracle.inputs import make_noise
    from wtpse_hip import nn as E
    B, Ci, Co, H = 4, 32, 64, 64

    class Holder(E.HipNet):
        def __init__(self, blk):
            super().__init__()
            self.blk = blk
            self._finish_init()

    h = Holder(E.ConvDBlock(Ci, Co)).to(DEV)
    fill_state_dict(h.blk, 4242)
    h.ensure_ready(repack=True)
    x = make_noise(901, (B, Ci, H, H))
    dz = make_noise(902, (B, Co, H // 2, H // 2))
    y, tape = E.convd_fwd(h.blk, x.to(DEV), True)
    yd = y.dense()
    h.begin_backward()
    dx = E.convd_bwd(h.blk, tape, dz.to(DEV), None, need_dx=True)
    h.end_backward()
    torch.cuda.synchronize()
    got = {"y": yd.cpu(), "dx": dx.cpu()}
    for k, p in h.blk.named_parameters():
        if not (k.startswith("conv") and k.endswith(".bias")):          # biases in front of a train-mode BatchNorm: gradient 0 by construction
            got["g." + k] = p.grad.detach().cpu()

    real = F.conv2d

    def make_conv(acc):
        class ConvBf16(torch.autograd.Function):
            @staticmethod
            def forward(ctx, xx, w, b, pad):
                xr, wr = bf16r(xx), bf16r(w)
                ctx.save_for_backward(xr, wr)
                ctx.pad, ctx.xs, ctx.ws = pad, xx.shape, w.shape
                return real(xr.to(acc), wr.to(acc), None if b is None else b.to(acc), padding=pad).to(xx.dtype)

            @staticmethod
            def backward(ctx, g):
                xr, wr = ctx.saved_tensors
                gr = bf16r(g)
                gx = torch.nn.grad.conv2d_input(ctx.xs, wr.to(acc), gr.to(acc), padding=ctx.pad).to(g.dtype)
                gw = torch.nn.grad.conv2d_weight(xr.to(acc), ctx.ws, gr.to(acc), padding=ctx.pad).to(g.dtype)
                return gx, gw, g.sum((0, 2, 3)), None

        def conv(xx, w, b=None, stride=1, padding=0, *a, **kw):
            assert E.x3_eligible(w.shape[1], w.shape[0], w.shape[2]) and E.x3_eligible(w.shape[0], w.shape[1], w.shape[2])
            return ConvBf16.apply(xx, w, b, padding)
        return conv

    def oracle(acc):
        sd = {"b." + k: v.detach().cpu().clone() for k, v in h.blk.state_dict().items()}
        # the HIP forward above already updated the running statistics: the oracle's own update of its copies is irrelevant here
        leaves = {k: v.requires_grad_(True) for k, v in sd.items() if v.is_floating_point() and not O.is_buffer(k)}
        sd.update(leaves)
        xx = x.clone().requires_grad_(True)
        O.F.conv2d = make_conv(acc)
        try:
            z = O.conv_d(sd, "b.", xx, False, True)
            z.backward(dz)
        finally:
            O.F.conv2d = real
        out = {"y": z.detach(), "dx": xx.grad}
        for k, v in leaves.items():
            kk = k[2:]
            if not (kk.startswith("conv") and kk.endswith(".bias")):
                out["g." + kk] = v.grad
        return out

    o64, o32 = oracle(torch.float64), oracle(torch.float32)
    worst = 0.0
    for k in sorted(got):
        sc = float(o64[k].double().norm()) + 1e-30
        e = float((got[k].double() - o64[k].double()).norm()) / sc
        yard = float((o32[k].double() - o64[k].double()).norm()) / sc
        worst = max(worst, e / (3 * yard + 1e-3))
        print("bf16 block %-22s HIP vs bf16-operand oracle (fp64 accumulation) %.2e of the tensor's norm; the oracle's own fp32 accumulation %.2e"
              % (k, e, yard))
        assert e <= 3 * yard + 1e-3, (k, e, yard)
    # and it IS the bf16 arithmetic: the same block under the fp32-accuracy arithmetic is ~1e-3..1e-2 away
    assert worst > 0.0
