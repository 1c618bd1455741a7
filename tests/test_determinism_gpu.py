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
