"""Data-parallel training on the GPU (-m gpu).

* exact mode, two ranks sharing the one MI355X of the test box (gloo transport): gradients, BatchNorm statistics, loss
  values and sampling noise must reproduce the single-process run on the global batch;
* the RCCL leg (backend "nccl"): world_size 1 on the test box's GPU (all-reduce AVG, broadcast and a training step in
  both modes go through RCCL), world_size 2 when the box has two GPUs (skipped otherwise).

Gradient check, calibrated as in test_parity_gpu.assert_calibrated: the yardstick is the distance of the CPU oracle's
own fp32 run from its fp64 run on the global batch; the data-parallel gradients must be within 10x of that."""
import numpy as np
import pytest
import torch

from test_dp_host import run_workers

pytestmark = pytest.mark.gpu
DEV = "cuda"
B_G, PB_G, H_G = 12, 4, 64


def _rel_to(module, flat, ref):
    """Relative L2 distance of the flat gradient vector `flat` (module's parameter order) to the {name: grad} dict `ref`,
    over the tensors the reference reaches (pre-BatchNorm conv biases excluded: SURVEY.md Appendix A)."""
    from test_parity_gpu import is_prebn_bias
    num = den = 0.0
    for k, p in module.named_parameters():
        if is_prebn_bias(k) or k not in ref:
            continue
        off = module.param_offset(p)
        g = flat[off:off + p.numel()].double().view(p.shape)
        num += float((g - ref[k]).pow(2).sum())
        den += float(ref[k].pow(2).sum())
    return (num / max(den, 1e-60)) ** 0.5


def _single_device_reference():
    from wtpse_hip import ops
    from oracle.inputs import make_inputs, make_noise
    from test_parity_gpu import build_nets
    img, od, _ = make_inputs(600, B_G, H_G, H_G)
    eps = make_noise(700, (B_G, 1, H_G, H_G))
    main, shape, _, _ = build_nets(PB_G)
    sd_before = ({k: v.detach().cpu().clone() for k, v in main.state_dict().items()},
                 {k: v.detach().cpu().clone() for k, v in shape.state_dict().items()})
    for n in (main, shape):
        n.train()
        n.ensure_ready(repack=True)
    x, m = img.to(DEV), od.to(DEV)
    main.set_noise([eps])
    r, tape = main._forward_update(x, m, x, want_tape=True)
    out, _, scal = r
    main._backward_update(tape, ops.bce_sigmoid_bwd(out, m), None, None)
    g_main = main.flat_grads().cpu().clone()
    sd_mid = {k: v.detach().cpu().clone() for k, v in main.state_dict().items()}     # BN running stats advanced by call A
    s2, tape2 = shape._forward_update(main, x, m, want_tape=True)
    shape._backward_update(tape2, None, None, None, None)
    g_shape = shape.flat_grads().cpu().clone()
    return main, shape, out.cpu(), scal.cpu(), s2.cpu(), g_main, g_shape, sd_before, sd_mid


# The data-parallel gradients may be this many times as far from the fp64 oracle as the reference's own fp32 CPU run (same yardstick and
# bound as test_parity_gpu.assert_calibrated at the benchmark's resolution; measured ratios are printed)
CAL_DP = 3.0


def _check_exact(res, world):
    from oracle import wtpse_cpu as O
    from oracle.inputs import make_inputs, make_noise
    from test_parity_gpu import HP, oracle_grads, build_nets
    main, shape, out, scal, s2, g_main, g_shape, sd_before, sd_mid = _single_device_reference()
    img, od, _ = make_inputs(600, B_G, H_G, H_G)
    eps = make_noise(700, (B_G, 1, H_G, H_G))

    def loss_a(sd):
        dt = sd["outc.0.weight"].dtype
        o, _, _, i2, d2 = O.wt_pse_update(sd, HP, img.to(dt), od.to(dt), img.to(dt), True, eps.to(dt), 3, PB_G)
        return O.seg_loss_od(o, od.to(dt)) + i2 + d2

    def loss_b(sds, sdm):
        dt = sds["mu_prior.0.weight"].dtype
        r = O.shape_update(sds, sdm, HP, img.to(dt), od.to(dt), img.to(dt), True, eps.to(dt), eps.to(dt), PB_G)
        return r[0] + r[1] + r[4]
    a32 = oracle_grads(loss_a, [sd_before[0]], torch.float32)[0]
    a64 = oracle_grads(loss_a, [sd_before[0]], torch.float64)[0]
    b32 = oracle_grads(loss_b, [sd_before[1], sd_mid], torch.float32)[0]
    b64 = oracle_grads(loss_b, [sd_before[1], sd_mid], torch.float64)[0]
    # the data-parallel backward differentiates the global-batch loss: same gradients as the oracle's on the global batch.
    # (_backward_update's default weights are w_ins = w_dom = 1 and the BCE-of-sigmoid gradient, i.e. loss_a / loss_b.)
    yard_a = _rel_to(main, torch.cat([a32[k].reshape(-1) if k in a32 else torch.zeros(p.numel(), dtype=torch.float64)
                                      for k, p in main.named_parameters()]), a64)
    yard_b = _rel_to(shape, torch.cat([b32[k].reshape(-1) if k in b32 else torch.zeros(p.numel(), dtype=torch.float64)
                                       for k, p in shape.named_parameters()]), b64)
    one_a, one_b = _rel_to(main, g_main, a64), _rel_to(shape, g_shape, b64)
    print("relative L2 distance to the fp64 oracle — CPU fp32: A %.2e B %.2e; 1 GPU: A %.2e B %.2e" % (yard_a, yard_b, one_a, one_b))
    for rk in res:
        da, db = _rel_to(main, rk["g_main"], a64), _rel_to(shape, rk["g_shape"], b64)
        print("  %d ranks, rank rows %s...: A %.2e B %.2e" % (world, rk["rows"][:3], da, db))
        assert da <= CAL_DP * yard_a + 2e-4, (da, yard_a)
        assert db <= CAL_DP * yard_b + 2e-4, (db, yard_b)
        assert torch.allclose(rk["out"], out[rk["rows"]], atol=1e-4)
        assert torch.allclose(rk["scal_main"], scal, rtol=1e-3, atol=1e-6)
        if "scal_shape" in rk:
            assert torch.allclose(rk["scal_shape"][1:], s2[1:], rtol=1e-3, atol=1e-6)
    for rk in res[1:]:
        assert torch.equal(res[0]["g_main"], rk["g_main"])      # after the all-reduce every rank holds the same gradient
        assert torch.equal(res[0]["g_shape"], rk["g_shape"])
    return main


def test_exact_mode_matches_single_device(tmp_path):
    res = run_workers("gpu", 2, tmp_path, extra=(B_G, PB_G, H_G), timeout=900)
    main = _check_exact(res, 2)
    main.seed_noise(99)
    nz = main.next_noise((B_G, 1, H_G, H_G)).cpu()
    for rk in res:
        assert torch.equal(rk["noise"], nz[rk["rows"]])
    # call B's teacher forward advanced the BatchNorm buffers once more on both sides
    for rk in res:
        for k, v in main.named_buffers():
            assert torch.allclose(rk["bufs"][k].float(), v.cpu().float(), rtol=1e-4, atol=1e-5), k


def test_overlapped_ddp_step_world2(tmp_path):
    """The DEFAULT multi-GPU path of bench.py (eager launches, throughput mode, gradient all-reduces started inside the backward)
    with two ranks sharing the test box's GPU over gloo: bitwise the parameters of the non-overlapped exchange, also when the step is
    captured as HIP graphs (the in-backward announcements must then stay out of the capture: ADVICE r03), and identical on both
    ranks.  RCCL itself still needs two GPUs (test_rccl_world2); this pins the host logic, the stream order and the capture guard."""
    res = run_workers("ddp_overlap", 2, tmp_path, extra=(B_G, PB_G, 32), timeout=900)
    for rk in res:
        assert rk["overlap_equals_single"] and rk["graph_equals_single"], (rk["overlap_equals_single"], rk["graph_equals_single"])
        assert rk["pieces_overlap"] >= 8, rk["pieces_overlap"]          # the announcements really happened (2 steps x 4 backwards)
        for tag in ("overlap", "single", "graph"):
            assert all(np.isfinite(v) for v in rk["losses_" + tag].values()), rk["losses_" + tag]
    for a, b in zip(res[0]["params"], res[1]["params"]):
        assert torch.equal(a, b)


def _check_steps(res):
    for rk in res:
        assert rk["mean"] and rk["sum"] and rk["bcast"], {k: rk[k] for k in ("mean", "sum", "bcast")}
        for tag in ("exact", "ddp", "plan"):
            assert all(np.isfinite(v) for v in rk["losses_" + tag].values()), rk["losses_" + tag]
        assert rk["plan_equals_eager"] and rk["plan_segments"] == 5, (rk["plan_equals_eager"], rk["plan_segments"])
    for rk in res[1:]:
        for tag in ("exact", "ddp"):          # same averaged gradient + same Adam -> identical parameters on every rank
            for a, b in zip(res[0]["params_" + tag], rk["params_" + tag]):
                assert torch.equal(a, b), tag


def test_rccl_world1(tmp_path):
    """backend "nccl" = RCCL with one rank on the box's GPU: dist.init_process_group(device_id=...), ReduceOp.AVG,
    broadcast_params and a TrainStep in both modes all execute through RCCL; with one rank the exact-mode result must
    equal the single-device run."""
    res = run_workers("nccl", 1, tmp_path, extra=(B_G, PB_G, H_G), timeout=900)
    _check_steps(res)
    _check_exact(res, 1)


def test_rccl_world2(tmp_path):
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (RCCL refuses two ranks on one device)")
    res = run_workers("nccl", 2, tmp_path, extra=(B_G, PB_G, H_G), timeout=900)
    _check_steps(res)
    _check_exact(res, 2)
