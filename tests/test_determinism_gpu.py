"""Race detector for the training step: no kernel uses floating-point atomics and every reduction folds its partials in a
fixed order, so two runs from the same state must agree bit for bit — also with the weight gradients on their side stream
(wtpse_hip/nn.py::_wgrad_side).  A difference means a missing stream dependency or an unsynchronised scratch buffer."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "wt-pse-code_amd")]


@pytest.mark.gpu
def test_two_runs_bitwise_identical():
    import bench
    from wtpse_hip.step import TrainStep
    from wtpse_hip.synth import make_batch, default_hparams
    dev = torch.device("cuda:0")
    hp = default_hparams(True)
    B = 6

    def run():
        torch.manual_seed(0)
        nets = bench.build_nets(hp, B // 3, dev)
        for n in nets:
            n.seed_noise(1234)
        ts = TrainStep(*nets, hp, dp=None)
        for k in range(3):
            image, od, oc = make_batch(B, 64, 64, dev, seed=10 + k)
            res = ts.step(image, od, oc)
        torch.cuda.synchronize()
        return [n.flat_params().clone() for n in nets], {k: float(v) for k, v in res.items()}

    a, la = run()
    b, lb = run()
    assert all(v == v for v in la.values()), la
    assert la == lb
    for x, y in zip(a, b):
        assert torch.equal(x, y)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["plan", True])
def test_captured_step_equals_eager_step(mode):
    """The step replayed from native launch plans (TrainStep(graph="plan")) or from HIP graphs (graph=True) against the same
    step launched eagerly: same kernels, same order of every reduction, the Adam step number and the Philox position read
    from device memory in both — the parameters of all four networks, their BatchNorm buffers and the losses must agree
    bit for bit after 4 steps on changing batches (fresh sampling noise and bias corrections on every replay)."""
    import bench
    from wtpse_hip import ops
    from wtpse_hip.step import TrainStep
    from wtpse_hip.synth import make_batch, default_hparams
    dev = torch.device("cuda:0")
    hp = default_hparams(True)
    B = 6

    def run(graph):
        torch.manual_seed(0)
        nets = bench.build_nets(hp, B // 3, dev)
        for n in nets:
            n.seed_noise(1234)
        ts = TrainStep(*nets, hp, dp=None, graph=graph)
        losses = []
        for k in range(4):
            image, od, oc = make_batch(B, 64, 64, dev, seed=10 + k)
            res = ts.step(image, od, oc)
            losses.append({k2: float(v) for k2, v in res.items()})
        torch.cuda.synchronize()
        assert (ts._graphs is not None) == bool(graph)
        if graph == "plan":
            assert sum(ops.lib().raw("wtpse_plan_size")(p) for _, _, p in ts._graphs) > 1000
        bufs = [torch.cat([b.detach().reshape(-1).double() for b in n.buffers()]) for n in nets]
        return [n.flat_params().clone() for n in nets], bufs, losses, [o.t for o in ts.opt.values()]

    pe, be, le, te = run(False)
    pg, bg, lg, tg = run(mode)
    assert te == tg == [4, 4, 4, 4]
    assert all(v == v for d in lg for v in d.values()), lg
    assert le == lg, (le, lg)
    for x, y in zip(pe + be, pg + bg):
        assert torch.equal(x, y)


@pytest.mark.gpu
def test_eager_work_between_plan_replays():
    """VERDICT r04 (ticket ring): launches recorded into a plan keep their ticket slices for as long as the plan lives; eager work
    between two replays — an eval-mode predict() here, plus enough eager training launches on a second set of networks to take the
    eager ring's pointer once round — must not disturb them.  The recorded step's networks after 3 replays interleaved with that
    eager work == the same 3 steps launched eagerly with the same interleaved work, bit for bit; and the recording's tickets come
    from its own buffer (ops.ticket_scope), not from the ring."""
    import bench
    from wtpse_hip import ops
    from wtpse_hip.step import TrainStep
    from wtpse_hip.synth import make_batch, default_hparams
    dev = torch.device("cuda:0")
    hp = default_hparams(True)
    B = 6

    def run(graph):
        torch.manual_seed(0)
        nets = bench.build_nets(hp, B // 3, dev)
        other = bench.build_nets(hp, B // 3, dev)
        for n in nets + other:
            n.seed_noise(4321)
        ts = TrainStep(*nets, hp, dp=None, graph=graph)
        side = TrainStep(*other, hp, dp=None, graph=False)
        ring0 = ops._TICKETS[dev][1] if dev in ops._TICKETS else 0
        preds = []
        for k in range(3):
            image, od, oc = make_batch(B, 64, 64, dev, seed=20 + k)
            ts.step(image, od, oc)
            # eager work between the replays: an eval-mode prediction on the recorded step's own networks ...
            nets[0].eval(); nets[1].eval()
            with torch.no_grad():
                preds.append(nets[0].predict(nets[1], image)[0].clone())
            nets[0].train(); nets[1].train()
            # ... and an eager training step of other networks (hundreds of ticket-taking launches)
            side.step(image, od, oc)
        torch.cuda.synchronize()
        if graph:
            assert ts._ticket_scope.buf.data_ptr() != ops._TICKETS[dev][0].data_ptr()
            assert not bool(ts._ticket_scope.buf.any()), "a recorded launch left its tickets non-zero"
        assert not bool(ops._TICKETS[dev][0].any()), "an eager launch left its tickets non-zero"
        return [n.flat_params().clone() for n in nets], preds, ring0

    pe, qe, _ = run(False)
    pp, qp, _ = run("plan")
    for x, y in zip(pe + qe, pp + qp):
        assert torch.equal(x, y)


@pytest.mark.gpu
def test_plan_refuses_to_replay_under_other_switches():
    """the tiling of the x3 launches — hence the size of their statistics / tail buffers — and the format of the
    packed weights depend on run-time switches (wtpse_x3r_enable, wtpse_x3_terms); a recorded plan replays the entry points, which
    re-read them.  A plan therefore remembers wtpse_tuning_state() and wtpse_plan_replay refuses (status -2, nothing launched) when it
    differs; with the switches back, the same plan replays again and the step continues where the eager sequence would be."""
    import bench
    from wtpse_hip import ops
    from wtpse_hip.lib import WtpseError
    from wtpse_hip.step import TrainStep
    from wtpse_hip.synth import make_batch, default_hparams
    dev = torch.device("cuda:0")
    hp = default_hparams(True)
    B = 6
    torch.manual_seed(0)
    nets = bench.build_nets(hp, B // 3, dev)
    ts = TrainStep(*nets, hp, dp=None, graph="plan")
    image, od, oc = make_batch(B, 64, 64, dev, seed=30)
    ts.step(image, od, oc)
    torch.cuda.synchronize()
    before = [n.flat_params().clone() for n in nets]
    L = ops.lib()
    state = L.query("wtpse_tuning_state")
    for name, other in (("wtpse_x3r_enable", 0), ("wtpse_x3_terms", 3 if ops.x3_terms() != 3 else 2)):
        was = L.query(name, other)
        try:
            assert L.query("wtpse_tuning_state") != state
            with pytest.raises(WtpseError, match="refused"):
                ts.step(image, od, oc)
        finally:
            L.query(name, was)
        torch.cuda.synchronize()
        assert L.query("wtpse_tuning_state") == state
        for x, y in zip(before, (n.flat_params() for n in nets)):
            assert torch.equal(x, y), "a refused replay must not have launched anything"
    res = ts.step(image, od, oc)                 # the same plan, switches restored
    torch.cuda.synchronize()
    assert all(float(v) == float(v) for v in res.values())
    assert any(not torch.equal(x, n.flat_params()) for x, n in zip(before, nets))


def _pack_x16(o, w):
    """OIHW 16x16x3x3 -> (buffer, forward offset, data-gradient offset) through wtpse_pack_conv16_x3 (csrc/conv.hip MODE 3)."""
    from test_kernels_gpu import DEV
    packed = torch.zeros(2 * o.X16_SIZE, dtype=torch.int16, device=DEV)
    desc = torch.tensor([0, w.shape[0], w.shape[1], 9, 0, o.X16_SIZE, 0, 0], dtype=torch.int32, device=DEV)
    o.lib().call("wtpse_pack_conv16_x3", w.reshape(-1).contiguous().to(DEV).data_ptr(), desc.data_ptr(), 1, packed.data_ptr(), o.stream_ptr())
    return packed, 0, o.X16_SIZE


@pytest.mark.gpu
@pytest.mark.parametrize("layout,C,H,W,k,B", [
    (1, 64, 32, 64, 3, 20), (1, 32, 16, 32, 3, 8), (1, 128, 8, 8, 1, 8), (0, 16, 32, 64, 3, 20), (0, 32, 16, 32, 3, 8),
    (2, 16, 32, 64, 3, 20),       # the 16-channel x3 kernel (csrc/conv.hip MODE 3): the sibling epilogue of the round-3 finding
    (2, 16, 256, 256, 3, 8),      # ... on the maps it runs on in a step (2048 workgroups)
    (1, 64, 128, 128, 3, 32),     # the benchmark's batch: 2048 workgroups (rounds 3-5: the largest launch that folds its own statistics)
    (1, 32, 256, 256, 3, 32),     # 8192 workgroups: since round 6 the largest launch that folds its own statistics (two levels: 128 groups)
    (1, 32, 256, 512, 3, 32),     # 16384 workgroups: beyond the threshold, the stand-alone finalize kernel behind the launch
])
def test_dgrad_bnb_repeatable(layout, C, H, W, k, B):
    """The data gradient with the BatchNorm-backward epilogue, repeated on the same operands with the allocator's blocks
    dirtied in between: masked gradient and partials must come out bit-identical every time.  (Round 3 found a form of this
    epilogue — the two ReLU decisions of a register fused into one v_pk_fma_f32 — whose masks were wrong for a few lanes in
    ~10 % of the launches; this is the regression test, for all three kernels that carry the epilogue and for the step's largest
    launches.)"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_kernels_gpu import rnd, ops, pack, DEV
    from test_conv_x3_gpu import pack_x3
    o = ops()
    big = B * H * W >= (1 << 20)
    y = rnd(B, C, H, W, seed=41).to(DEV)
    du = rnd(B, C, H, W, seed=42).to(DEV)
    w = rnd(C, C, k, k, seed=43, scale=0.2)
    ss = torch.stack([rnd(C, seed=44) * 0.2 + 1.0, rnd(C, seed=45) * 0.3], 1).contiguous().to(DEV)
    mean = (rnd(C, seed=46) * 0.1).to(DEV)
    x3 = layout
    if layout == 1:
        packed, _, xd = pack_x3(w)
        wptr = packed.data_ptr() + 2 * xd
    elif layout == 2:
        packed, _, xd = _pack_x16(o, w)
        wptr = packed.data_ptr() + 2 * xd
    else:
        packed, _, wd = pack(w)
        wptr = packed.data_ptr() + 4 * wd
    g0, _, st0, _ = o.dgrad_bnb(du, wptr, x3, C, k, y, ss, mean, True)
    g0, st0 = g0.clone(), st0.clone()
    # the masks against the forward pass's own decision fmaf(y, scale, shift) > 0: in fp64 the product is exact and the sum keeps
    # its sign, so the sign of the fp64 value IS the sign of the single-rounding fmaf (an unfused fp32 y * scale + shift is not: it
    # differed on 1 element of 67 M here)
    act = y.double() * ss[:, 0].double().view(1, -1, 1, 1) + ss[:, 1].double().view(1, -1, 1, 1)
    assert int(((g0 != 0) & ~(act > 0)).sum()) == 0, "a masked-out element carries a gradient"
    del act
    for it in range(40 if big else 150):
        junk = torch.empty((1 << 22) + 4096 * it, device=DEV).fill_(float(it))
        g, _, st, _ = o.dgrad_bnb(du, wptr, x3, C, k, y, ss, mean, True)
        assert torch.equal(g, g0), "launch %d: %d masked-gradient elements differ" % (it, int((g != g0).sum()))
        assert torch.equal(st, st0), "launch %d: partials differ" % it
        del junk
    # the launch that folds its own partials (last-arriver tickets, csrc/common.h: bnb_tail): whoever arrives last, the
    # coefficients and dgamma / dbeta are the same bits, and the tickets are back at zero
    gamma = (rnd(C, seed=47) * 0.2 + 1.0).to(DEV)
    invstd = (rnd(C, seed=48).abs() + 0.5).to(DEV)
    ref = None
    for it in range(30 if big else 100):
        junk = torch.empty((1 << 22) + 4096 * it, device=DEV).fill_(float(it))
        dg, db = torch.empty(C, device=DEV), torch.empty(C, device=DEV)
        g, _, st, coef = o.dgrad_bnb(du, wptr, x3, C, k, y, ss, mean, True, tail=(gamma, invstd, dg, db))
        cur = (coef.clone(), dg.clone(), db.clone())
        if ref is None:
            ref = cur
            assert torch.equal(g, g0) and torch.equal(st, st0)
        for a, b, what in zip(cur, ref, ("coef", "dgamma", "dbeta")):
            assert torch.equal(a, b), "launch %d: %s differs" % (it, what)
        del junk
    assert all(int(t[0].abs().sum()) == 0 for t in o._TICKETS.values()), "tickets must be left at zero"


@pytest.mark.gpu
@pytest.mark.parametrize("Cin,Cout,H,W,bias,B,reps", [(16, 16, 64, 64, True, 12, 100), (32, 32, 32, 64, False, 12, 100), (64, 32, 16, 32, False, 12, 100),
                                                        # the step's own launch of wgrad_r_k<2,2,PRO>: B=32, 64 -> 64 @128x128, prologue on
                                                        (64, 64, 128, 128, False, 32, 30)])
def test_wgrad_r_repeatable(Cin, Cout, H, W, bias, B, reps):
    """The register-resident weight gradient: repeated launches, bit-identical gradients (fixed-order folds, no atomics).  Its
    BatchNorm+ReLU prologue is a packed FMA with ONE operand-select broadcast (tests/test_isa_checks.py); the last case is the
    launch a training step makes of it, at the benchmark's batch."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_kernels_gpu import rnd, ops, DEV
    o = ops()
    x = rnd(B, Cin, H, W, seed=51).to(DEV)
    dy = rnd(B, Cout, H, W, seed=52).to(DEV)
    pro = torch.stack([rnd(Cin, seed=53) * 0.5 + 1.0, rnd(Cin, seed=54)], 1).contiguous().to(DEV)
    dw0 = torch.empty(Cout, Cin, 3, 3, device=DEV)
    db0 = torch.empty(Cout, device=DEV) if bias else None
    o.conv_wgrad_r(dy, x, None, dw0, db0, pro0=pro, pro_relu=1)
    for it in range(reps):
        junk = torch.empty((1 << 22) + 4096 * it, device=DEV).fill_(float(it))
        dw = torch.empty_like(dw0)
        db = torch.empty_like(db0) if bias else None
        o.conv_wgrad_r(dy, x, None, dw, db, pro0=pro, pro_relu=1)
        assert torch.equal(dw, dw0), "launch %d: %d weight-gradient elements differ" % (it, int((dw != dw0).sum()))
        if bias:
            assert torch.equal(db, db0), "launch %d: bias gradient differs" % it
        del junk


@pytest.mark.gpu
@pytest.mark.parametrize("B,C,H,W", [(8, 32, 32, 64), (32, 16, 256, 256)])
def test_maxpool_bwd_bnb_repeatable(B, C, H, W):
    """The max-pool backward that also applies the ReLU mask of the conv + BatchNorm layer below and forms its BatchNorm-backward
    reductions (csrc/pointwise.hip: the third carrier of the ReLU decision `fmaf(y, scale, shift) > 0`): repeated launches on the
    same operands give the same bits, and the mask equals the forward pass's decision."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_kernels_gpu import rnd, ops, DEV
    o = ops()
    x = rnd(B, C, H, W, seed=61).to(DEV)
    dout = rnd(B, C, H // 2, W // 2, seed=62).to(DEV)
    pro = torch.stack([rnd(C, seed=63) * 0.2 + 1.0, rnd(C, seed=64) * 0.3], 1).contiguous().to(DEV)
    mean = (rnd(C, seed=65) * 0.1).to(DEV)
    r = o.maxpool2_bwd_bnb(x, dout, None, pro, True, mean)
    assert r is not None
    g0, st0 = r[0].clone(), r[1].clone()
    act = x.double() * pro[:, 0].double().view(1, -1, 1, 1) + pro[:, 1].double().view(1, -1, 1, 1)     # sign of fmaf(x, scale, shift)
    assert int(((g0 != 0) & ~(act > 0)).sum()) == 0
    for it in range(60):
        junk = torch.empty((1 << 22) + 4096 * it, device=DEV).fill_(float(it))
        g, st = o.maxpool2_bwd_bnb(x, dout, None, pro, True, mean)
        assert torch.equal(g, g0), "launch %d: %d elements differ" % (it, int((g != g0).sum()))
        assert torch.equal(st, st0), "launch %d: partials differ" % it
        del junk
