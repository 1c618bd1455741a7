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
@pytest.mark.parametrize("x3,C,H,W,k", [(True, 64, 32, 64, 3), (True, 32, 16, 32, 3), (True, 128, 8, 8, 1), (False, 16, 32, 64, 3),
                                        (False, 32, 16, 32, 3)])
def test_dgrad_bnb_repeatable(x3, C, H, W, k):
    """The data gradient with the BatchNorm-backward epilogue, 150 launches on the same operands with the allocator's blocks
    dirtied in between: masked gradient and partials must come out bit-identical every time.  (Round 3 found a form of this
    epilogue — the two ReLU decisions of a register fused into one v_pk_fma_f32 — whose masks were wrong for a few lanes in
    ~10 % of the launches; this is the regression test.)"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_kernels_gpu import rnd, ops, pack, DEV
    from test_conv_x3_gpu import pack_x3
    o = ops()
    B = 20 if H * W >= 2048 else 8
    y = rnd(B, C, H, W, seed=41).to(DEV)
    du = rnd(B, C, H, W, seed=42).to(DEV)
    w = rnd(C, C, k, k, seed=43, scale=0.2)
    ss = torch.stack([rnd(C, seed=44) * 0.2 + 1.0, rnd(C, seed=45) * 0.3], 1).contiguous().to(DEV)
    mean = (rnd(C, seed=46) * 0.1).to(DEV)
    if x3:
        packed, _, xd = pack_x3(w)
        wptr = packed.data_ptr() + 2 * xd
    else:
        packed, _, wd = pack(w)
        wptr = packed.data_ptr() + 4 * wd
    g0, _, st0, _ = o.dgrad_bnb(du, wptr, x3, C, k, y, ss, mean, True)
    g0, st0 = g0.clone(), st0.clone()
    for it in range(150):
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
    for it in range(100):
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
@pytest.mark.parametrize("Cin,Cout,H,W,bias", [(16, 16, 64, 64, True), (32, 32, 32, 64, False), (64, 32, 16, 32, False)])
def test_wgrad_r_repeatable(Cin, Cout, H, W, bias):
    """The register-resident weight gradient: 100 launches, bit-identical gradients (fixed-order folds, no atomics)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_kernels_gpu import rnd, ops, DEV
    o = ops()
    B = 12
    x = rnd(B, Cin, H, W, seed=51).to(DEV)
    dy = rnd(B, Cout, H, W, seed=52).to(DEV)
    pro = torch.stack([rnd(Cin, seed=53) * 0.5 + 1.0, rnd(Cin, seed=54)], 1).contiguous().to(DEV)
    dw0 = torch.empty(Cout, Cin, 3, 3, device=DEV)
    db0 = torch.empty(Cout, device=DEV) if bias else None
    o.conv_wgrad_r(dy, x, None, dw0, db0, pro0=pro, pro_relu=1)
    for it in range(100):
        junk = torch.empty((1 << 22) + 4096 * it, device=DEV).fill_(float(it))
        dw = torch.empty_like(dw0)
        db = torch.empty_like(db0) if bias else None
        o.conv_wgrad_r(dy, x, None, dw, db, pro0=pro, pro_relu=1)
        assert torch.equal(dw, dw0), "launch %d: %d weight-gradient elements differ" % (it, int((dw != dw0).sum()))
        if bias:
            assert torch.equal(db, db0), "launch %d: bias gradient differs" % it
        del junk
