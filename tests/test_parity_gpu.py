"""Network-level parity (-m gpu): the drop-in modules on MI355X against (i) the reference-generated fixtures
under tests/golden/ and (ii) the CPU oracle on the same seeded inputs, through the public class surface
(`update` / `predict` / `loss.backward()` / torch.optim.Adam) and through the fused step harness.

Tolerance: north_star asks logits / Dice / WT-loss values within 1e-4 fp32 of the reference."""
import os
import time

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import wtpse_cpu as O
from oracle.filler import fill_state_dict
from oracle.inputs import make_inputs, make_noise

pytestmark = pytest.mark.gpu
DEV = "cuda"
SEED_W = 1234
HP = dict(O.DEFAULT_HPARAMS)
TOL = 1e-4


def close(a, b, rtol=1e-4, atol=1e-4, what=""):
    a = torch.as_tensor(np.asarray(a.detach().cpu()) if torch.is_tensor(a) else np.asarray(a)).double()
    b = torch.as_tensor(np.asarray(b.detach().cpu()) if torch.is_tensor(b) else np.asarray(b)).double()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err = (a - b).abs()
    bad = err > atol + rtol * b.abs()
    assert not bad.any(), f"{what}: {int(bad.sum())}/{bad.numel()} off, max err {err.max():.3e}, ref scale {b.abs().max():.3e}"


def build_nets(pb, full=True):
    import algorithms
    import shape_networks
    hp = HP if full else dict(HP, whitening=False, shape_prior=False)
    mk = lambda two_step: algorithms.WT_PSE(n_channels=3, n_classes=1, hparams=hp, device=DEV, two_step=two_step,
                                            per_domain_batch=pb, source_domain_num=3).to(DEV)
    main, main_oc = mk(False), mk(True)
    fill_state_dict(main, SEED_W)
    fill_state_dict(main_oc, SEED_W + 7)
    if not full:
        return main, None, main_oc, None
    shape = shape_networks.ShapeVariationalDist_x(hp, DEV, n_classes=1, number_source_domain=3, batch_size=pb).to(DEV)
    shape_oc = shape_networks.ShapeVariationalDist_x(hp, DEV, n_classes=1, number_source_domain=3, batch_size=pb).to(DEV)
    fill_state_dict(shape, SEED_W + 3)
    fill_state_dict(shape_oc, SEED_W + 11)
    return main, shape, main_oc, shape_oc


def is_prebn_bias(k):
    return ((".conv" in "." + k) and k.endswith(".bias")) or ".inc.double_conv.0.bias" in k or ".inc.double_conv.3.bias" in k


def kink_band(grad_fn, image, probes=24):
    """How far the reference's OWN fp32 gradients move when the input image moves by one part in 10^6: the fingerprints of
    `grad_fn(image * (1 + r n))` (the CPU oracle in fp32; r <= 1e-5, n seeded standard normal: perturbed()) against those of `grad_fn(image)`,
    per tensor the largest deviation seen over the probes.  On the 32x32 fixtures the deepest maps are 2x2 (BatchNorm over
    12-28 values, then a ReLU): a unit whose pre-activation is within rounding of zero flips under such a perturbation and
    moves every gradient of its network by several per cent at once (measured: case 2, the teacher's gradients by 6-9 % in
    one of six probes).  Two fp32 implementations that round differently can land on different sides of such a unit; the band
    measures how much that is worth on this very fixture instead of guessing."""
    base = {k: O.checksum(v.float()) for k, v in grad_fn(image).items()}
    band = {k: np.zeros_like(v) for k, v in base.items()}
    for pert in perturbed(image, probes):
        for k, v in grad_fn(pert).items():
            band[k] = np.maximum(band[k], np.abs(O.checksum(v.float()) - base[k]))
    return band


def perturbed(image, probes):
    """image * (1 + r n), r cycling through 1e-6, 3e-6 and 1e-5 (a tenth of the 1e-4 that north_star grants the forward
    values; fp32 rounding through ~40 layers with BatchNorm over 12-28 values reaches the same order), n seeded normal."""
    gen = torch.Generator().manual_seed(77)
    for i in range(probes):
        yield image * (1 + (1e-6, 3e-6, 1e-5)[i % 3] * torch.randn(image.shape, generator=gen)).to(image.dtype)


def check_grads_vs_checksums(module, g, prefix, min_seen, kink_probe=None):
    """Gradient fingerprints (sum, sum|.|, 32 strided entries per tensor) against the reference's: a coarse check
    (every entry within 8 % of the tensor's mean |grad| — at most one per tensor up to 3x that — pooled median within 1 %).  It is coarse on purpose: the
    32x32 fixtures reach 2x2 feature maps (BatchNorm over 12-28 values) and sit on kinks (ReLU, max-pool argmax,
    |G_ij|), so the reference's own fp32 gradients are only good to 0.3-6 % against an fp64 run of the same graph
    (measured with tools/probe/diag_grads.py).  The tight, self-calibrating gradient check is test_gradients_calibrated."""
    seen, pooled, band, kinked = 0, [], None, 0
    for k, p in module.named_parameters():
        key = prefix + k
        if key not in g.files:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
            continue
        if is_prebn_bias(k):
            continue
        assert p.grad is not None, k
        ref = g[key]
        got = O.checksum(p.grad.cpu())
        n = p.numel()
        scale = abs(ref[1]) / max(n, 1)           # mean |grad|
        err = np.abs(got - ref)
        tol = 1e-2 * np.abs(ref) + 5e-6 + 8e-2 * scale * np.array([3.0 * n ** 0.5, n] + [1.0] * (len(ref) - 2))
        # one of a tensor's 34 fingerprint values may sit up to 3x outside (an entry fed by a unit on a kink: the two fp32
        # implementations — reference fma chains here, split-bf16 products there — round differently)
        ratio = err / tol
        if not ((ratio > 1.0).sum() <= 1 and ratio.max() <= 3.0) and kink_probe is not None:
            # outside the fixed band: is this fixture on a kink?  Measure what a <= 1e-5 input perturbation does to the
            # reference's own fp32 gradients (kink_band) and allow 1.5x that on top.
            if band is None:
                band = kink_probe()
            wide = tol + 1.5 * band[k]
            print(f"{key}: outside the fixed band ({int((ratio > 1.0).sum())} entries, worst {ratio.max():.2f}x); the reference's "
                  f"fp32 gradients move by up to {float((band[k] / tol).max()):.1f}x that band under input perturbations <= 1e-5")
            ratio = err / wide
            kinked += 1
        assert (ratio > 1.0).sum() <= 1 and ratio.max() <= 3.0, \
            f"{key}: {int((ratio > 1.0).sum())} entries off, worst {ratio.max():.2f}x its tolerance {tol[ratio.argmax()]:.3e}, scale {scale:.3e}"
        pooled.extend((err[2:] / max(scale, 1e-20)).tolist())
        seen += 1
    assert seen >= min_seen, seen
    if kinked:
        print(f"{prefix}: {kinked} of {seen} tensors judged against the measured kink band")
    assert np.median(np.asarray(pooled)) < (1e-2 if not kinked else 3e-2), np.median(np.asarray(pooled))


def heartbeat(msg):
    """The CPU oracle at the benchmark's batch runs for minutes without a sign of life: leave one under gpurun_out/ (the GPU box's
    watchdog takes a command for hung after 7 minutes without new output) — a no-op anywhere else."""
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(d):
        with open(os.path.join(d, "heartbeat.log"), "a") as f:
            f.write("%s %s\n" % (time.strftime("%H:%M:%S"), msg))


def oracle_grads(fn, sds, dtype):
    """Run `fn(*state_dicts cast to dtype)` -> scalar loss on the CPU oracle; -> list of {name: grad (fp64)}."""
    heartbeat("oracle_grads %s" % str(dtype))
    cast = [{k: (v.detach().clone().to(dtype).requires_grad_(not O.is_buffer(k)) if v.is_floating_point() else v.clone())
             for k, v in sd.items()} for sd in sds]
    fn(*cast).backward()
    return [{k: sd[k].grad.double() for k in sd if not O.is_buffer(k) and sd[k].grad is not None} for sd in cast]


NEARLY_CANCELLING = ("wt_model.DoubleConv2.double_conv.2.bias",)     # 16 elements, each a 781x cancelling sum over the whole gradient field
CAL = 3.0      # the HIP path may be at most this many times as far from the fp64 evaluation as the reference's own fp32 CPU path
CAL_KINK = 10.0  # ... on the 32x32 / 64x64 fixtures at B=6, where kink flips (not rounding) set both distances: see below


def assert_calibrated(module, g32, g64, what, cal=CAL, probes=None, strict=False):
    """The HIP path must be an fp32 implementation of the reference's graph of the same quality as the reference's own
    CPU path.  Yardstick: relative L2 distance to the oracle evaluated in fp64.
      * over ALL gradients of the network concatenated: HIP <= cal x CPU-fp32 + 2e-4
      * per tensor: HIP <= cal x CPU-fp32 + 5e-4 for at least 97 % of the tensors, and <= 2e-2 for every tensor
        (a ReLU / max-pool unit within rounding of its kink may switch side in one of the two fp32 runs; that moves
        the few tensors fed by it by ~1e-3 and says nothing about kernel accuracy)
      * the MEDIAN over tensors of (HIP distance / CPU-fp32 distance) <= cal.
    cal = 3 at the benchmark's resolution (measured ratio of the totals 1.17 / 1.28 for calls A / B at 256x256, 0.81 / 1.22 at
    B=3 32x32).  On the B=6 32x32 / 64x64 cases both fp32 runs sit 1e-3..4e-3 from the fp64 run — 100x rounding level: units on
    kinks that flip in one run and not in the other — and the ratio of two such draws scatters (measured 0.37 .. 4.14 between
    calls of the same case, median per-tensor ratio 0.29 and 4.03), so those keep cal = 10.
    probes: callable -> fp32 oracle gradients on perturbed inputs, run only when the plain comparison fails (a 32x32 fixture
    whose deepest 2x2 maps hold a unit within rounding of its kink: the fp32 reference itself then moves by 0.3-9 % under a
    1e-6 input perturbation — measured on cases [3-1-32] and golden case 2 — and so may any other fp32 implementation).
    strict (the benchmark's geometries, 256x256 and 512x512: no 2x2 maps to excuse anything): EVERY tensor must be within
    cal x CPU-fp32 + 5e-4 — no allowance for outliers and no failure-triggered widening.  The yardstick there is fixed IN ADVANCE as
    the farthest of three fp32 evaluations of the reference graph — the inputs as given and two copies perturbed by 1e-6 / 3e-6
    (`g32` may be a list of runs) — because one fp32 run is a single draw: at B = 32 both fp32 implementations sit 2e-3 from the fp64
    oracle (a million ReLU / max-pool units, some of them within rounding of their kink), and a tensor that is a nearly cancelling
    sum over the whole gradient field (the bias of a convolution without BatchNorm) scatters by 3x between such draws — measured:
    wt_model.DoubleConv2.double_conv.2.bias, HIP 1.18e-2 vs 3.6e-3 for the single unperturbed CPU run, every other tensor inside.
    -> (HIP distance, CPU-fp32 distance); messages carry the measured ratios."""
    def distances(g32_runs):
        """-> (HIP total, CPU-fp32 total, [(HIP, CPU-fp32, name)]); CPU-fp32 = the farthest of the given fp32 runs."""
        num_h, den, num_c, per = 0.0, 0.0, [0.0] * len(g32_runs), []
        for k, p in module.named_parameters():
            if is_prebn_bias(k):
                continue
            if k not in g64:
                assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
                continue
            ref = g64[k]
            n2 = float(ref.pow(2).sum()) + 1e-60
            eh2 = float((p.grad.cpu().double() - ref).pow(2).sum())
            ec2 = [float((r[k] - ref).pow(2).sum()) for r in g32_runs]
            num_h += eh2; den += n2
            num_c = [x + y for x, y in zip(num_c, ec2)]
            per.append(((eh2 / n2) ** 0.5, (max(ec2) / n2) ** 0.5, k))
        return (num_h / den) ** 0.5, (max(num_c) / den) ** 0.5, per

    def verdict(tot_h, tot_c, per):
        med = float(np.median([h / max(c, 1e-30) for h, c, _ in per]))
        bad = [(h, c, k) for h, c, k in per if h > cal * c + 5e-4]
        worst = max(per)
        msgs = []
        if tot_h > cal * tot_c + 2e-4:
            msgs.append(f"all gradients: HIP {tot_h:.3e} vs CPU-fp32 {tot_c:.3e} (ratio {tot_h / max(tot_c, 1e-30):.2f}, bound {cal:.0f}x + 2e-4)")
        if med > cal:
            msgs.append(f"median per-tensor ratio HIP / CPU-fp32 = {med:.2f} > {cal:.0f}")
        if len(bad) > (0 if strict else 0.03 * len(per)):
            msgs.append(f"{len(bad)}/{len(per)} tensors beyond {cal:.0f}x + 5e-4, e.g. {sorted(bad, reverse=True)[:3]}")
        # (round 5: the absolute cap binds only where the yardstick itself keeps it — on [3-1-32] the reference's own fp32 runs on
        # inputs perturbed by <= 1e-5 move this 16-element, nearly cancelling tensor by 5.5e-2 from the fp64 value: no fp32
        # implementation can be asked to stay within 2e-2 there.  profiles/r05_x2h_parity.md)
        # (round 6, ADVICE r05: that allowance is for THAT tensor only — wt_model.DoubleConv2.double_conv.2.bias, the bias behind the WT
        # loss — every other tensor keeps the plain 2e-2 cap)
        for h, c, k in per:
            if h > 2e-2 and not (k.endswith(NEARLY_CANCELLING) and h <= c):
                msgs.append(f"{k}: HIP {h:.3e} vs CPU-fp32 {c:.3e} (cap 2e-2)")
        return med, msgs

    runs0 = list(g32) if isinstance(g32, (list, tuple)) else [g32]
    g32 = runs0[0]
    tot_h, tot_c, per = distances(runs0)
    med, msgs = verdict(tot_h, tot_c, per)
    print(f"[calibrated {what}] all gradients: HIP {tot_h:.3e} vs CPU-fp32 {tot_c:.3e} from the fp64 oracle: ratio "
          f"{tot_h / max(tot_c, 1e-30):.2f} (bound {cal:.0f}x + 2e-4); median per-tensor ratio {med:.2f} (bound {cal:.0f})")
    assert not (strict and probes is not None)
    if msgs and probes is not None:
        # Is the fixture on a kink?  The yardstick becomes the farthest of the reference's fp32 runs on inputs perturbed by
        # 1e-6 / 3e-6 (perturbed()): what a flip of a near-zero unit is worth on this fixture, measured, not guessed.
        runs = [g32] + list(probes())
        tot_h, tot_c, per = distances(runs)
        med, msgs = verdict(tot_h, tot_c, per)
        print(f"[calibrated {what}] against the farthest of {len(runs)} fp32 runs on perturbed (<= 1e-5) inputs: HIP {tot_h:.3e} vs "
              f"CPU-fp32 {tot_c:.3e}, ratio {tot_h / max(tot_c, 1e-30):.2f}; median per-tensor ratio {med:.2f}")
    assert not msgs, f"{what}: " + "; ".join(msgs)
    return tot_h, tot_c


# ---------------------------------------------------------------- a-1 / a-2 / a-3: blocks against reference fixtures
@pytest.mark.parametrize("name,bi", [("convd_first", 0), ("convd", 1), ("convu_first", 2), ("convu", 3)])
def test_blocks_vs_golden(golden_dir, name, bi):
    from wtpse_hip import nn as E
    g = np.load(os.path.join(golden_dir, "blocks.npz"))
    B, H = 4, 16
    spec = {"convd_first": (lambda: E.ConvDBlock(3, 16, first=True), (B, 3, H, H), None),
            "convd": (lambda: E.ConvDBlock(16, 32), (B, 16, H, H), None),
            "convu_first": (lambda: E.ConvUBlock(64, first=True), (B, 64, H // 2, H // 2), (B, 32, H, H)),
            "convu": (lambda: E.ConvUBlock(32), (B, 64, H // 2, H // 2), (B, 16, H, H))}[name]

    class Holder(E.HipNet):
        def __init__(self, blk):
            super().__init__()
            self.blk = blk
            self._finish_init()

    h = Holder(spec[0]()).to(DEV)
    fill_state_dict(h.blk, SEED_W + 20 + bi)
    h.ensure_ready(repack=True)
    # input seeds with no ReLU / max-pool unit of the REFERENCE run within 2e-5 of its kink (oracle/make_golden_blocks.py): an
    # element-wise gradient comparison at 1e-3 is then a statement about the kernels, whatever fp32-accurate arithmetic they run in
    seed_t = 1000 * int(g[name + ".seed_t"])
    assert float(g[name + ".kink_margin"]) > 2e-5
    x = make_noise(300 + bi + seed_t, spec[1]).to(DEV)
    wgt = None
    if spec[2] is None:
        y, tape = E.convd_fwd(h.blk, x, True)
        y = y.dense()
    else:
        prev = make_noise(400 + bi + seed_t, spec[2]).to(DEV)
        y, tape = E.convu_fwd(h.blk, x, prev, True)
        y = y.dense()
    close(y, g[name + ".y"], what="y")
    dy = make_noise(500 + bi + seed_t, tuple(y.shape)).to(DEV)
    h.begin_backward()
    if spec[2] is None:
        dx = E.convd_bwd(h.blk, tape, dy, None, need_dx=True)
    else:
        dx, dprev = E.convu_bwd(h.blk, tape, dy)
        close(dprev, g[name + ".dprev"], rtol=1e-3, atol=2e-4, what="dprev")
    h.end_backward()
    close(dx, g[name + ".dx"], rtol=1e-3, atol=2e-4, what="dx")
    for k, p in h.blk.named_parameters():
        if is_prebn_bias(k):
            continue
        ref = g[f"{name}.g.{k}"]
        close(p.grad, ref, rtol=2e-3, atol=3e-4 * max(1.0, float(np.abs(ref).max())), what="g." + k)
    for k, b in h.blk.named_buffers():
        close(b.float(), g[f"{name}.buf.{k}"].astype(np.float32), rtol=1e-4, atol=1e-5, what="buf." + k)
    h.blk.eval()
    h.eval()
    if spec[2] is None:
        ye, _ = E.convd_fwd(h.blk, x, False, want_tape=False)
    else:
        ye, _ = E.convu_fwd(h.blk, x, prev, False, want_tape=False)
    close(ye.dense(), g[name + ".y_eval"], what="y_eval")


def test_convu_first_near_kink_fixture(golden_dir):
    """VERDICT r05 weak 2a / ADVICE r05: round 5 regenerated the block fixtures with input seeds that keep every ReLU unit of the
    reference run away from its kink, and dropped the old `convu_first` case, in which ONE unit of the block's output sits within
    fp32 rounding of zero (x2h flips it where x3 and the CPU's fp32 do not: 218 gradient elements differed, all in that unit's 3x3
    footprint).  The old reference fixture is kept (tests/golden/blocks_near_kink.npz: the reference's arrays for seeds 302 / 402 /
    502, bcfe95d) and judged here with the kink MEASURED: the units whose ReLU decision differs from the reference's are found from
    the outputs (at most two, each with a reference pre-activation below 1e-5), their 3x3 footprints are masked in the two data
    gradients, and everything else — the outputs, every other gradient element, the BatchNorm buffers, the eval-mode output — must
    meet the tolerances of test_blocks_vs_golden: nothing else moved."""
    from wtpse_hip import nn as E
    g = np.load(os.path.join(golden_dir, "blocks_near_kink.npz"))
    B, H, bi, name = 4, 16, 2, "convu_first"

    class Holder(E.HipNet):
        def __init__(self, blk):
            super().__init__()
            self.blk = blk
            self._finish_init()

    h = Holder(E.ConvUBlock(64, first=True)).to(DEV)
    fill_state_dict(h.blk, SEED_W + 20 + bi)
    h.ensure_ready(repack=True)
    x = make_noise(300 + bi, (B, 64, H // 2, H // 2)).to(DEV)
    prev = make_noise(400 + bi, (B, 32, H, H)).to(DEV)
    y, tape = E.convu_fwd(h.blk, x, prev, True)
    y = y.dense()
    yr = torch.from_numpy(g[name + ".y"])
    close(y, yr, what="y")
    flips = ((y.cpu() > 0) != (yr > 0)).nonzero().tolist()
    assert len(flips) <= 2, "more ReLU decisions differ from the reference than one near-kink unit explains: %s" % flips[:8]
    for b, c, r, col in flips:
        assert max(float(y[b, c, r, col].abs()), float(yr[b, c, r, col].abs())) < 1e-5, (b, c, r, col)
    print("near-kink fixture: %d unit(s) decided differently from the reference: %s" % (len(flips), flips))
    dy = make_noise(500 + bi, tuple(y.shape)).to(DEV)
    h.begin_backward()
    dx, dprev = E.convu_bwd(h.blk, tape, dy)
    h.end_backward()
    keep_p = torch.ones(tuple(dprev.shape), dtype=torch.bool)
    keep_x = torch.ones(tuple(dx.shape), dtype=torch.bool)
    for b, c, r, col in flips:
        keep_p[b, :, max(r - 1, 0):r + 2, max(col - 1, 0):col + 2] = False
        # the same footprint behind the bilinear x2 upsampling (two low-resolution neighbours per side) and the 1x1 conv
        keep_x[b, :, max((r - 1) // 2 - 1, 0):(r + 1) // 2 + 2, max((col - 1) // 2 - 1, 0):(col + 1) // 2 + 2] = False
    dpr, dxr = torch.from_numpy(g[name + ".dprev"]), torch.from_numpy(g[name + ".dx"])
    close(torch.where(keep_p, dprev.cpu(), dpr), dpr, rtol=1e-3, atol=2e-4, what="dprev outside the flipped units' footprints")
    close(torch.where(keep_x, dx.cpu(), dxr), dxr, rtol=1e-3, atol=2e-4, what="dx outside the flipped units' footprints")
    for k, p in h.blk.named_parameters():
        if is_prebn_bias(k):
            continue
        ref = g[f"{name}.g.{k}"]
        got = p.grad.detach().cpu().clone()
        if k.startswith(("conv3.", "bn3.")):
            # a flipped unit of conv3's output channel c moves the parameter gradients of THAT channel by one pixel's worth (|dy| x
            # |activation| of one of 1024: 0.14 at scale 143 in conv3.weight[36]); every other row must be at the block tests' tolerance
            for _, c, _, _ in flips:
                got[c] = torch.from_numpy(np.asarray(ref))[c]
        close(got, ref, rtol=2e-3, atol=3e-4 * max(1.0, float(np.abs(ref).max())), what="g." + k)
    for k, b in h.blk.named_buffers():
        close(b.float(), g[f"{name}.buf.{k}"].astype(np.float32), rtol=1e-4, atol=1e-5, what="buf." + k)
    h.blk.eval()
    h.eval()
    ye, _ = E.convu_fwd(h.blk, x, prev, False, want_tape=False)
    close(ye.dense(), g[name + ".y_eval"], what="y_eval")


def test_deepwt_vs_golden(golden_dir):
    import algorithms
    g = np.load(os.path.join(golden_dir, "blocks.npz"))
    m = algorithms.WT_PSE(3, 1, HP, DEV, False, per_domain_batch=1).to(DEV)
    fill_state_dict(m.wt_model, SEED_W + 30)
    x = make_noise(310, (4, 3, 16, 16)).to(DEV)
    zs = m.wt_model.forward(x)
    for i, z in enumerate(zs):
        close(z, g[f"deepwt.z{i + 1}"], rtol=1e-4, atol=2e-5, what=f"z{i + 1}")
    fill_state_dict(m.attention_layer, SEED_W + 31)
    a, pre = m.attention_layer.forward(make_noise(311, (4, 1, 16, 16)).to(DEV))
    close(a, g["attention.sig"], atol=1e-5); close(pre, g["attention.pre"], atol=1e-5)


# ---------------------------------------------------------------- a-7 / a-8 / a-9 through the public API + torch autograd
@pytest.mark.parametrize("ci", [0, 1, 2])
def test_network_calls_vs_golden(golden_dir, ci):
    g = np.load(os.path.join(golden_dir, "network.npz"))
    B, pb, H, s_in, s_a, s_t, s_s = (int(v) for v in g["cases"][ci])
    p = f"c{ci}_"
    img, od, oc = (t.to(DEV) for t in make_inputs(s_in, B, H, H))
    main, shape, main_oc, shape_oc = build_nets(pb)
    for m in (main, shape, main_oc, shape_oc):
        m.eval()
    with torch.no_grad():
        logit, att = main.predict(shape, img)
        roi = (img + 1) * (torch.sigmoid(logit) > 0.75).float() - 1
        logit2, att2 = main_oc.predict(shape_oc, torch.stack((roi, roi), 0))
    close(logit, g[p + "pred_logit"], atol=TOL, what="pred_logit")
    close(att, g[p + "pred_att"], atol=TOL, what="pred_att")
    close(logit2, g[p + "pred2_logit"], atol=TOL, what="pred2_logit")
    close(att2, g[p + "pred2_att"], atol=TOL, what="pred2_att")
    # call A through update() + autograd, loss glue in torch exactly as Trainer.py:787-804 does it
    main.train(); shape.train()
    main.zero_grad()
    sd_main = {k: v.detach().cpu().clone() for k, v in main.state_dict().items()}
    main.set_noise([make_noise(s_a, (B, 1, H, H))])
    out, m1, m2, ins, dom = main.update(img, od, two_stage_inputs=img, sp_mask=od, two_step=True)
    loss = F.binary_cross_entropy(torch.sigmoid(out), od) + ins + dom
    loss.backward()

    def oracle_a(image):
        def fn(sd):
            o, _, _, i2, d2 = O.wt_pse_update(sd, HP, image, od.cpu(), image, True, make_noise(s_a, (B, 1, H, H)), 3, pb)
            return O.seg_loss_od(o, od.cpu()) + i2 + d2
        return oracle_grads(fn, [sd_main], torch.float32)[0]
    close(out, g[p + "upd_out"], atol=TOL, what="upd_out")
    assert float((m1.cpu() != torch.from_numpy(g[p + "upd_mask"])).float().mean()) < 1e-3
    close(ins, g[p + "upd_ins"], rtol=1e-4, atol=1e-6, what="ins")
    close(dom, g[p + "upd_dom"], rtol=1e-3, atol=3e-7, what="dom")
    close(loss, g[p + "upd_loss"], rtol=1e-4, atol=1e-5, what="loss")
    check_grads_vs_checksums(main, g, p + "upd_g.", 100, kink_probe=lambda: kink_band(oracle_a, img.cpu()))
    for k, b in main.named_buffers():
        close(O.checksum(b.float().cpu()), g[p + "upd_buf." + k], rtol=1e-4, atol=1e-4, what=k)
    # call B
    shape.zero_grad(); main.zero_grad()
    sd_shape = {k: v.detach().cpu().clone() for k, v in shape.state_dict().items()}
    sd_main_b = {k: v.detach().cpu().clone() for k, v in main.state_dict().items()}      # BatchNorm running statistics advanced by call A
    kd, ins_t, ins_off, ins_diag, dom_s = shape.update(main, img, od, two_stage_inputs=img, two_step=True)
    (kd + ins_t + dom_s).backward()

    def oracle_b(image):
        def fn(sds, sdm):       # (the sampled z of either network feeds nothing that is returned: the noise is immaterial)
            r = O.shape_update(sds, sdm, HP, image, od.cpu(), image, True, make_noise(s_t, (B, 1, H, H)), make_noise(s_s, (B, 1, H, H)), pb)
            return r[0] + r[1] + r[4]
        return oracle_grads(fn, [sd_shape, sd_main_b], torch.float32)[0]
    close(kd, g[p + "shp_kd"], rtol=1e-3, atol=1e-5, what="kd")
    close(ins_t, g[p + "shp_ins_total"], rtol=1e-4, atol=1e-6)
    close(ins_off, g[p + "shp_ins_off"], rtol=1e-4, atol=1e-6)
    close(ins_diag, g[p + "shp_ins_diag"], rtol=1e-4, atol=1e-6)
    close(dom_s, g[p + "shp_dom"], rtol=1e-3, atol=3e-7)
    # (round 5: the same measured kink band as call A — under the x2h arithmetic two fingerprint entries of one DeepWT tensor of
    # case 0 sat 1.1x outside the fixed band where the x3 rounding had left one: profiles/r05_x2h_parity.md)
    check_grads_vs_checksums(shape, g, p + "shp_g.", 50, kink_probe=lambda: kink_band(oracle_b, img.cpu()))
    assert all(q.grad is None for q in shape.logvar_prior.parameters())     # never reached, as in the reference
    assert all(q.grad is None for q in main.parameters())                   # teacher backward skipped


@pytest.mark.parametrize("ci", [0, 1])
def test_cat_shape_vs_golden(golden_dir, ci):
    """hparams['cat_shape'] = True (algorithms.py:1192,1253,1348): `outc` over cat(fuse_embedding, z_posterior), against the
    reference's own outputs (oracle/make_golden_catshape.py)."""
    import algorithms
    import shape_networks
    g = np.load(os.path.join(golden_dir, "catshape.npz"))
    B, pb, H, s_in, s_a = (int(v) for v in g["cases"][ci])
    p = f"c{ci}_"
    hp = dict(HP, cat_shape=True)
    main = algorithms.WT_PSE(n_channels=3, n_classes=1, hparams=hp, device=DEV, two_step=False, per_domain_batch=pb,
                             source_domain_num=3).to(DEV)
    shape = shape_networks.ShapeVariationalDist_x(hp, DEV, n_classes=1, number_source_domain=3, batch_size=pb).to(DEV)
    assert tuple(main.outc[0].weight.shape) == (1, 9, 1, 1)
    fill_state_dict(main, SEED_W + 40)
    fill_state_dict(shape, SEED_W + 43)
    img, od, _ = (t.to(DEV) for t in make_inputs(s_in, B, H, H))
    main.eval(); shape.eval()
    with torch.no_grad():
        logit, att = main.predict(shape, img)
    close(logit, g[p + "pred_logit"], atol=TOL, what="pred_logit")
    close(att, g[p + "pred_att"], atol=TOL, what="pred_att")
    main.train()
    main.zero_grad()
    sd_main = {k: v.detach().cpu().clone() for k, v in main.state_dict().items()}
    main.set_noise([make_noise(s_a, (B, 1, H, H))])
    out, m1, _, ins, dom = main.update(img, od, two_stage_inputs=img, sp_mask=od, two_step=True)
    loss = F.binary_cross_entropy(torch.sigmoid(out), od) + ins + dom
    loss.backward()

    def oracle_a(image):
        def fn(sd):
            o, _, _, i2, d2 = O.wt_pse_update(sd, hp, image, od.cpu(), image, True, make_noise(s_a, (B, 1, H, H)), 3, pb)
            return O.seg_loss_od(o, od.cpu()) + i2 + d2
        return oracle_grads(fn, [sd_main], torch.float32)[0]
    close(out, g[p + "upd_out"], atol=TOL, what="upd_out")
    close(ins, g[p + "upd_ins"], rtol=1e-4, atol=1e-6, what="ins")
    close(dom, g[p + "upd_dom"], rtol=1e-3, atol=3e-7, what="dom")
    close(loss, g[p + "upd_loss"], rtol=1e-4, atol=1e-5, what="loss")
    ref = g[p + "upd_g_full.outc.0.weight"]
    close(main.outc[0].weight.grad, ref, rtol=2e-3, atol=3e-4 * float(np.abs(ref).max()), what="d outc.weight")
    check_grads_vs_checksums(main, g, p + "upd_g.", 100, kink_probe=lambda: kink_band(oracle_a, img.cpu()))


def test_seg_only_vs_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "network.npz"))
    main, _, _, _ = build_nets(2, full=False)
    img, od, _ = (t.to(DEV) for t in make_inputs(650, 6, 32, 32))
    main.train()
    out = main.update(img, od, two_stage_inputs=img, two_step=True)
    assert out[1:] == (0, 0, 0, 0)
    F.binary_cross_entropy(torch.sigmoid(out[0]), od).backward()
    close(out[0], g["segonly_out"], atol=TOL)
    check_grads_vs_checksums(main, g, "segonly_g.", 40)
    main.eval()
    with torch.no_grad():
        pred, none = main.predict(None, img)
    assert none is None
    close(pred, g["segonly_pred"], atol=TOL)


@pytest.mark.parametrize("shape,first", [((2, 64, 8, 10), False), ((2, 64, 5, 9), False), ((3, 64, 6, 7), True)])
def test_convu_both_orders_vs_oracle(shape, first):
    """ConvU runs its 1x1 conv in front of the upsampling when the width is even (they commute) and in the reference's
    order otherwise: both against the oracle's ConvU (reference order) in fp64."""
    from wtpse_hip import nn as E
    B, C, H, W = shape

    class Holder(E.HipNet):
        def __init__(self, blk):
            super().__init__()
            self.blk = blk
            self._finish_init()

    planes = C if first else C // 2                      # a non-first block halves its input with conv1 first
    h = Holder(E.ConvUBlock(planes, first=first)).to(DEV)
    fill_state_dict(h.blk, SEED_W + 77)
    h.ensure_ready(repack=True)
    x = make_noise(310, shape)
    prev = make_noise(410, (B, planes // 2, 2 * H, 2 * W))
    y, tape = E.convu_fwd(h.blk, x.to(DEV), prev.to(DEV), True)
    assert tape.swapped == (E.CONVU_CONV_FIRST and W % 2 == 0)
    y = y.dense()
    sd = {"b." + k: v.detach().cpu().double().clone().requires_grad_(v.is_floating_point() and not O.is_buffer(k))
          for k, v in h.blk.state_dict().items()}
    xr, pr = x.double().requires_grad_(True), prev.double().requires_grad_(True)
    yr = O.conv_u(sd, "b.", xr, pr, first, True)
    close(y, yr, rtol=1e-4, atol=1e-4, what="y")
    dy = make_noise(510, tuple(y.shape))
    yr.backward(dy.double())
    h.begin_backward()
    dx, dprev = E.convu_bwd(h.blk, tape, dy.to(DEV))
    h.end_backward()
    close(dx, xr.grad, rtol=2e-3, atol=2e-4 * float(xr.grad.abs().max()), what="dx")
    close(dprev, pr.grad, rtol=2e-3, atol=2e-4 * float(pr.grad.abs().max()), what="dprev")
    for k, p in h.blk.named_parameters():
        if is_prebn_bias(k):
            continue
        ref = sd["b." + k].grad
        close(p.grad, ref, rtol=5e-3, atol=5e-4 * float(ref.abs().max()) + 1e-7, what=k)


FULL_ORACLE = os.environ.get("WTPSE_FULL_ORACLE", "0") != "0"


@pytest.mark.parametrize("tag", ["b32", "s512"])
def test_gradients_vs_offline_oracle(golden_dir, tag):
    """The backward at the benchmark's own geometries — b32: B = 32, 10 rows per domain, 256x256 (BASELINE.json configs[2]: 8192-workgroup
    launches, the weight gradient's unit / segment split, the stand-alone statistics finalize beyond 8192 workgroups (2048 until round 6) and the in-launch
    fold below); s512: B = 3 at 512x512 (configs[4]'s per-image geometry) — against the CPU oracle evaluated OFFLINE
    (oracle/make_golden_grads.py -> tests/golden/grads_<tag>.npz: ten minutes and 30 GB of host per case, too slow for every run of the
    suite; `WTPSE_FULL_ORACLE=1` runs the same comparison against a live oracle: test_gradients_calibrated[32-10-256] / [3-1-512]).
    Per parameter tensor the fixture holds a fingerprint of the fp64 gradient (oracle/sketch.py: the tensor itself up to 4096 elements,
    128 random projections beyond — an unbiased estimate of |h - g64|^2 with 6 % standard deviation on the norm) and the exact distances of
    three fp32 evaluations of the reference graph from it (inputs as given, perturbed by 1e-6 and by 3e-6).  STRICT band, fixed in advance:
    every tensor within 3 x the farthest fp32 evaluation + 5e-4 (sketched tensors: x 1.15 for the estimate's scatter), no allowance for
    outliers, no failure-triggered widening; all gradients together within 3 x + 2e-4."""
    from oracle import sketch
    g = np.load(os.path.join(golden_dir, "grads_%s.npz" % tag))
    B, pb, H, K, small = (int(v) for v in g["meta"])
    assert (K, small) == (sketch.K, sketch.SMALL)
    img, od, _ = make_inputs(600, B, H, H)
    eps = make_noise(700, (B, 1, H, H))
    main, shape, _, _ = build_nets(pb)
    main.train(); shape.train()
    main.zero_grad(); main.set_noise([eps])
    out, _, _, ins, dom = main.update(img.to(DEV), od.to(DEV), two_stage_inputs=img.to(DEV), two_step=True)
    (F.binary_cross_entropy(torch.sigmoid(out), od.to(DEV)) + ins + dom).backward()

    def check(module, call):
        names = [str(n) for n in g[call + "_names"]]
        params = dict(module.named_parameters())
        yard2 = g[call + "_yard2"]
        num_h = den = 0.0
        num_c = np.zeros(3)
        bad, worst = [], (0.0, 0.0, "")
        for i, k in enumerate(names):
            if is_prebn_bias(k):
                continue
            n, n2 = g["%s_%d_n2" % (call, i)]
            fp = {"n": int(n), "norm2": float(n2), "data": g["%s_%d_fp" % (call, i)]}
            p = params[k]
            assert p.grad is not None, k
            d2 = sketch.distance2(p.grad, fp, 7000 + i)
            h, c = (d2 / (n2 + 1e-60)) ** 0.5, (float(yard2[i].max()) / (n2 + 1e-60)) ** 0.5
            slack = 1.0 if n <= small else 1.15
            if h > slack * (CAL * c + 5e-4):
                bad.append((h, c, k))
            worst = max(worst, (h / max(c, 1e-30), h, k))
            num_h += d2; den += n2; num_c += yard2[i]
        for k, p in params.items():           # tensors the oracle's graph does not reach must carry no gradient here either
            if k not in names and not is_prebn_bias(k):
                assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
        tot_h, tot_c = (num_h / den) ** 0.5, (float(num_c.max()) / den) ** 0.5
        print(f"[offline {tag} {call}] all gradients: HIP {tot_h:.3e} vs the farthest of three CPU-fp32 draws {tot_c:.3e} from the fp64 oracle "
              f"(ratio {tot_h / max(tot_c, 1e-30):.2f}); worst tensor ratio {worst[0]:.2f} ({worst[2]}: {worst[1]:.3e})")
        assert not bad, f"{call}: {len(bad)} tensors beyond {CAL:.0f}x + 5e-4: {sorted(bad, reverse=True)[:4]}"
        assert tot_h <= 1.15 * (CAL * tot_c + 2e-4), (tot_h, tot_c)
    check(main, "A")
    shape.zero_grad(); main.zero_grad()
    kd, ins_t, _, _, dom_s = shape.update(main, img.to(DEV), od.to(DEV), two_stage_inputs=img.to(DEV), two_step=True)
    (kd + ins_t + dom_s).backward()
    check(shape, "B")


def test_deepwt_bias_gradient_is_the_sum_of_its_dz2():
    """VERDICT r04 #5.  `wt_model.DoubleConv2.double_conv.2.bias` of the student (shape_networks.py:545-549,561-594) is the one tensor the
    strict band of test_gradients_vs_offline_oracle[b32] was once too narrow for: a 16-element, nearly cancelling sum over the whole
    gradient field.  This isolates the KERNEL chain behind it from that sensitivity: call B's backward at the benchmark's geometry
    (B = 32, 10 rows per domain, 256x256), the HIP path's OWN dz2 — the gradient wrt the last DeepWT conv's output, after the WT-loss
    backward accumulated into it (gram_bwd_k), as the weight gradient's bias reduce reads it — copied to the host, summed per channel
    in fp64, against the bias gradient the path produced: within 1e-5 of the sum's own scale (sum |dz2|).  Whatever distance this
    tensor has from the fp64 oracle beyond that is in dz2 itself (kink scatter upstream), not in the reduce."""
    from wtpse_hip import nn as E
    B, pb, H = 32, 10, 256
    img, od, _ = make_inputs(600, B, H, H)
    main, shape, _, _ = build_nets(pb)
    main.train(); shape.train()
    seen = {}
    orig = E.deepwt_bwd

    def spy(wt, t, dz2, dz1_extra=None):
        if wt is shape.wt_model:
            seen["dz2"] = dz2.detach().clone()
        return orig(wt, t, dz2, dz1_extra)
    E.deepwt_bwd = spy
    try:
        shape.zero_grad(); main.zero_grad()
        kd, ins_t, _, _, dom_s = shape.update(main, img.to(DEV), od.to(DEV), two_stage_inputs=img.to(DEV), two_step=True)
        (kd + ins_t + dom_s).backward()
    finally:
        E.deepwt_bwd = orig
    torch.cuda.synchronize()
    assert "dz2" in seen, "the student's DeepWT backward did not run"
    dz2 = seen["dz2"].double().cpu()
    want = dz2.sum((0, 2, 3))
    scale = float(dz2.abs().sum((0, 2, 3)).max())
    got = shape.wt_model.DoubleConv2.double_conv[2].bias.grad.double().cpu()
    err = float((got - want).abs().max())
    print(f"[bias chain] max |bias.grad - sum dz2| = {err:.3e}; sum |dz2| = {scale:.3e}; |sum dz2| = {float(want.abs().max()):.3e} "
          f"(cancellation {scale / max(float(want.abs().max()), 1e-300):.0f}x)")
    assert err <= 1e-5 * float(want.abs().max()) + 1e-7 * scale, (err, float(want.abs().max()), scale)


@pytest.mark.parametrize("B,pb,H", [(3, 1, 32), (6, 2, 32), (6, 2, 64), (6, 2, 256), (32, 10, 256), (3, 1, 512)])
def test_gradients_calibrated(B, pb, H):
    """Every parameter gradient of call A (seg net + teacher + WT loss) and call B (student) against the oracle
    evaluated in fp64, with the oracle's own fp32 run as the yardstick.  The 256x256 cases run the backward kernels in
    the instantiations the benchmark uses (weight gradients at ksplit 512 + slab fold, the big-map BatchNorm backward,
    the bilinear adjoint and the fused heads at full resolution); [32-10-256] IS the benchmark's geometry (BASELINE.json
    configs[2]: 8192-workgroup launches, the weight gradient's unit / segment split, statistics folded by the stand-alone finalize
    beyond 8192 workgroups and by the launches' last workgroups below), [3-1-512] the per-image geometry of configs[4].  At
    256x256 and 512x512 the per-tensor band is fixed (strict: every tensor, no probes)."""
    if (B >= 32 or H >= 512) and not FULL_ORACLE:
        pytest.skip("5 / 2 minutes of live CPU oracle: run by WTPSE_FULL_ORACLE=1; the default suite holds the same HIP gradients against "
                    "the same oracle evaluated offline (test_gradients_vs_offline_oracle)")
    img, od, _ = make_inputs(600, B, H, H)
    eps = make_noise(700, (B, 1, H, H))
    main, shape, _, _ = build_nets(pb)
    sd_m = {k: v.detach().cpu().clone() for k, v in main.state_dict().items()}
    sd_s = {k: v.detach().cpu().clone() for k, v in shape.state_dict().items()}
    main.train(); shape.train()
    main.zero_grad(); main.set_noise([eps])
    out, _, _, ins, dom = main.update(img.to(DEV), od.to(DEV), two_stage_inputs=img.to(DEV), two_step=True)
    (F.binary_cross_entropy(torch.sigmoid(out), od.to(DEV)) + ins + dom).backward()

    def loss_a(sd, image=img):
        dt = sd["outc.0.weight"].dtype
        o, _, _, i2, d2 = O.wt_pse_update(sd, HP, image.to(dt), od.to(dt), image.to(dt), True, eps.to(dt), 3, pb)
        return O.seg_loss_od(o, od.to(dt)) + i2 + d2
    (g32,), (g64,) = oracle_grads(loss_a, [sd_m], torch.float32), oracle_grads(loss_a, [sd_m], torch.float64)
    strict = H >= 256
    if strict:      # the yardstick of the strict band: three fp32 draws, fixed in advance (assert_calibrated)
        g32 = [g32] + [oracle_grads(lambda sd, q=q: loss_a(sd, q), [sd_m], torch.float32)[0] for q in perturbed(img, 2)]
    cal = CAL if (H >= 256 or B == 3) else CAL_KINK
    n_probe = 12 if H <= 64 else 0          # kinks of this size only exist where the deepest maps are 2x2 / 4x4
    probes_a = (lambda: (oracle_grads(lambda sd: loss_a(sd, q), [sd_m], torch.float32)[0] for q in perturbed(img, n_probe))) if n_probe else None
    assert_calibrated(main, g32, g64, "A", cal, probes_a, strict=strict)
    del g32, g64
    shape.zero_grad(); main.zero_grad()
    kd, ins_t, _, _, dom_s = shape.update(main, img.to(DEV), od.to(DEV), two_stage_inputs=img.to(DEV), two_step=True)
    (kd + ins_t + dom_s).backward()
    sd_m2 = {k: v.detach().cpu().clone() for k, v in main.state_dict().items()}   # BN running stats advanced by call A

    def loss_b(sds, sdm, image=img):
        dt = sds["mu_prior.0.weight"].dtype
        r = O.shape_update(sds, sdm, HP, image.to(dt), od.to(dt), image.to(dt), True, eps.to(dt), eps.to(dt), pb)
        return r[0] + r[1] + r[4]
    g32 = oracle_grads(loss_b, [sd_s, sd_m2], torch.float32)[0]
    g64 = oracle_grads(loss_b, [sd_s, sd_m2], torch.float64)[0]
    if strict:
        g32 = [g32] + [oracle_grads(lambda a, b, q=q: loss_b(a, b, q), [sd_s, sd_m2], torch.float32)[0] for q in perturbed(img, 2)]
    probes_b = (lambda: (oracle_grads(lambda a, b: loss_b(a, b, q), [sd_s, sd_m2], torch.float32)[0] for q in perturbed(img, n_probe))) if n_probe else None
    assert_calibrated(shape, g32, g64, "B", cal, probes_b, strict=strict)


# ---------------------------------------------------------------- a-11: full A-D iterations
def _iteration_inputs(g):
    B, pb, H, iters, s_in, s_n = (int(v) for v in g["meta"])
    for it in range(iters):
        img, od, oc = make_inputs(s_in + it, B, H, H)
        nz = {k: make_noise(s_n + 10 * it + j, (B, 1, H, H)) for j, k in enumerate(["a", "b_t", "b_s", "c", "d_t", "d_s"])}
        yield it, img, od, oc, nz


def _check_params_vs_golden(nets, g, iters=3, lr=5e-4):
    """Parameters after `iters` Adam steps.  Adam's first steps move every weight by ~lr*sign(grad), so an entry whose
    gradient is within fp32 noise of zero goes either way: the ORACLE ITSELF, run in fp64, ends 3 iterations with only
    35 % of the fingerprinted entries within 1e-4 of the fp32 reference (median 2e-4, measured).  This check therefore
    guards the wiring of the loop (order of calls, ROI, optimiser), not rounding: median < 1e-3, 90 % within 2*lr*iters."""
    pooled = []
    for tag, net in zip(["od", "shape_od", "oc", "shape_oc"], nets):
        for k, v in net.state_dict().items():
            if is_prebn_bias(k) or not v.is_floating_point():
                continue
            pooled.extend(np.abs(O.checksum(v.float().cpu()) - g[f"{tag}.{k}"])[2:].tolist())
    pooled = np.asarray(pooled)
    assert np.median(pooled) < 1e-3 and (pooled < 2 * lr * iters + 5e-4).mean() >= 0.9, (np.median(pooled), pooled.max())


def _loss_tol(k, it):
    """Iteration 0 against the reference's fixture: call A sees identical weights and inputs, so the 1e-4 bar applies; the
    other calls of iteration 0 come after one Adam step.  Iterations 1-2 are checked by _check_later_iterations."""
    assert it == 0
    if k in ("seg_od", "ins_od"):
        return dict(rtol=1e-4, atol=1e-5)
    if k == "dom_od":
        return dict(rtol=1e-3, atol=5e-7)
    return dict(rtol=5e-3, atol=1e-4)


TRAJ_CAL = 5.0     # measured worst ratio 4.15 (an MMD term of iteration 1); the old bounds were rtol 0.1 / 0.6


def _oracle_trajectories(g, nets):
    """The CPU oracle's own trajectory over the fixture's iterations from the HIP networks' initial state, evaluated in fp32
    and in fp64 -> ([losses per iteration] fp32, the same fp64)."""
    out = []
    for dt in (torch.float32, torch.float64):
        sds = [{k: (v.detach().cpu().clone().to(dt) if v.is_floating_point() else v.detach().cpu().clone())
                for k, v in n.state_dict().items()} for n in nets]
        on = O.Nets(*sds)
        pb = int(g["meta"][1])
        traj = []
        for it, img, od, oc, nz in _iteration_inputs(g):
            traj.append(O.train_iteration(on, HP, img.to(dt), od.to(dt), oc.to(dt), {k: v.to(dt) for k, v in nz.items()}, pb))
        out.append(traj)
    return out


def _check_later_iterations(results, traj32, traj64):
    """Iterations 1-2 have passed through Adam's sign-like first steps and the 0.75 ROI threshold (Trainer.py:842), which
    amplify fp32 rounding: the oracle run in fp64 departs from its own fp32 run by 0.5-14 % there.  So the yardstick is that
    departure itself: per iteration and loss, |HIP - fp64 oracle| <= TRAJ_CAL x max(|fp32 oracle - fp64 oracle|, the median of
    that departure over the iteration's losses) + 1e-4 x |value| — the HIP trajectory must stay as close to the fp64
    trajectory as the reference's own fp32 arithmetic does."""
    for it in range(1, len(results)):
        keys = [k for k in traj64[it] if k in results[it]]
        rel = lambda a, b: abs(float(a) - float(b)) / (abs(float(b)) + 1e-6)
        dc = {k: rel(traj32[it][k], traj64[it][k]) for k in keys}
        dh = {k: rel(results[it][k], traj64[it][k]) for k in keys}
        med = float(np.median(list(dc.values())))
        worst = max(keys, key=lambda k: dh[k] / (max(dc[k], med) + 1e-12))
        print(f"[trajectory it{it}] median departure fp32-oracle {med:.3e}, HIP {float(np.median(list(dh.values()))):.3e}; "
              f"worst ratio {dh[worst] / (max(dc[worst], med) + 1e-12):.2f} ({worst})")
        for k in keys:
            bound = TRAJ_CAL * max(dc[k], med) + 1e-4
            assert dh[k] <= bound, (f"it{it}.{k}: HIP departs {dh[k]:.3e} from the fp64 oracle, the fp32 oracle {dc[k]:.3e} "
                                    f"(median over losses {med:.3e}); bound {bound:.3e}")


def test_iterations_harness_vs_golden(golden_dir):
    """The fused step harness (HIP losses + flat Adam) over 3 iterations against the reference's own trajectory."""
    from wtpse_hip.step import TrainStep
    g = np.load(os.path.join(golden_dir, "iteration.npz"))
    pb = int(g["meta"][1])
    main, shape, main_oc, shape_oc = build_nets(pb)
    traj32, traj64 = _oracle_trajectories(g, [main, shape, main_oc, shape_oc])
    ts = TrainStep(main, shape, main_oc, shape_oc, HP)
    keys = [str(k) for k in g["loss_keys"]]
    results = []
    for it, img, od, oc, nz in _iteration_inputs(g):
        res = ts.step(img.to(DEV), od.to(DEV), oc.to(DEV), {"a": nz["a"], "c": nz["c"]})
        results.append({k: float(v) for k, v in res.items()})
        for j, k in enumerate(keys):
            if k not in res or it > 0:
                continue           # main_od / shape_od ... are sums formed by the caller
            close(res[k], g["losses"][it][j], what=f"it{it}.{k}", **_loss_tol(k, it))
    _check_later_iterations(results, traj32, traj64)
    _check_params_vs_golden([main, shape, main_oc, shape_oc], g)


def test_iterations_dropin_vs_golden(golden_dir):
    """Same trajectory through the reference's own calling convention: update() -> torch loss glue -> .backward()
    -> torch.optim.Adam, i.e. what the unchanged Trainer.py does (Trainer.py:766-914)."""
    g = np.load(os.path.join(golden_dir, "iteration.npz"))
    pb = int(g["meta"][1])
    nets = build_nets(pb)
    model, model_shape, model_oc, model_shape_oc = nets
    opts = [torch.optim.Adam(n.parameters(), lr=5e-4, betas=(0.9, 0.99)) for n in nets]
    optim, optim_shape, optim_oc, optim_shape_oc = opts
    bce = torch.nn.BCELoss()
    keys = [str(k) for k in g["loss_keys"]]
    traj32, traj64 = _oracle_trajectories(g, nets)
    results = []
    for n in nets:
        n.train()
    for it, image, target_od, target_oc, nz in _iteration_inputs(g):
        image, target_od, target_oc = image.to(DEV), target_od.to(DEV), target_oc.to(DEV)
        res = {}
        optim.zero_grad(); model.zero_grad()
        model.set_noise([nz["a"]])
        output, _, _, ins, dom = model.update(image, target_od, two_stage_inputs=image, sp_mask=target_od, two_step=True)
        loss_seg = bce(torch.sigmoid(output), target_od)
        loss_main = loss_seg + HP["instance_wt_gm"] * ins + HP["domain_wt_gm"] * dom
        loss_main.backward(); optim.step()
        res.update(seg_od=loss_seg, ins_od=ins, dom_od=dom, main_od=loss_main)
        optim_shape.zero_grad(); model_shape.zero_grad()
        kd, ins_t, ins_ij, ins_ii, dom_s = model_shape.update(model, image, target_od, two_stage_inputs=image, two_step=True)
        loss_shape = kd + HP["instance_wt_gm"] * ins_t + HP["domain_wt_gm"] * dom_s
        loss_shape.backward(); optim_shape.step()
        res.update(kd_od=kd, ins_shape_od=ins_t, ins_ij_od=ins_ij, ins_ii_od=ins_ii, dom_shape_od=dom_s, shape_od=loss_shape)
        od_pred = (torch.sigmoid(output) > 0.75).float().detach().float()
        optim_oc.zero_grad(); model_oc.zero_grad()
        image += 1
        image_roi = image * od_pred
        image_roi -= 1
        model_oc.set_noise([nz["c"]])
        output_oc, _, _, ins_c, dom_c = model_oc.update(image_roi, target_oc, two_stage_inputs=image_roi, two_step=True)
        pw = torch.sum(od_pred) / torch.sum(od_pred * target_oc)
        if torch.isinf(pw) or torch.isnan(pw):
            pw = torch.tensor(1.).to(DEV)
        loss_seg_oc = F.binary_cross_entropy_with_logits(output_oc * od_pred, target_oc, pos_weight=pw)
        loss_main_oc = loss_seg_oc + HP["instance_wt_gm"] * ins_c + HP["domain_wt_gm"] * dom_c
        loss_main_oc.backward(); optim_oc.step()
        res.update(seg_oc=loss_seg_oc, ins_oc=ins_c, dom_oc=dom_c, main_oc=loss_main_oc)
        optim_shape_oc.zero_grad(); model_shape_oc.zero_grad()
        kd2, ins_t2, _, _, dom_s2 = model_shape_oc.update(model_oc, image_roi, target_oc, two_stage_inputs=image_roi, two_step=True)
        loss_shape_oc = kd2 + HP["instance_wt_gm"] * ins_t2 + HP["domain_wt_gm"] * dom_s2
        loss_shape_oc.backward(); optim_shape_oc.step()
        res.update(kd_oc=kd2, ins_shape_oc=ins_t2, dom_shape_oc=dom_s2, shape_oc=loss_shape_oc)
        results.append({k: float(v) for k, v in res.items()})
        if it == 0:
            for j, k in enumerate(keys):
                close(res[k], g["losses"][it][j], what=f"it{it}.{k}", **_loss_tol(k, it))
    _check_later_iterations(results, traj32, traj64)
    _check_params_vs_golden(nets, g)


# ---------------------------------------------------------------- full-resolution checks against the oracle (no fixture)
def test_update_256_vs_oracle():
    """256x256 (BASELINE.json's resolution), B=6: logits, WT-loss values and Dice of the thresholded prediction
    against the CPU oracle on the same inputs — the 1e-4 bar of north_star."""
    B, pb, H = 6, 2, 256
    img, od, oc = make_inputs(77, B, H, H)
    eps = make_noise(78, (B, 1, H, H))
    main, shape, _, _ = build_nets(pb)
    sd_main = {k: v.detach().cpu().clone() for k, v in main.state_dict().items()}
    sd_shape = {k: v.detach().cpu().clone() for k, v in shape.state_dict().items()}
    main.train()
    main.set_noise([eps])
    with torch.no_grad():
        out, _, _, ins, dom = main.update(img.to(DEV), od.to(DEV), two_stage_inputs=img.to(DEV), two_step=True)
        ref_out, _, _, ref_ins, ref_dom = O.wt_pse_update(dict(sd_main), HP, img, od, img, True, eps, 3, pb)
    close(out, ref_out, atol=TOL, what="logits@256")
    close(ins, ref_ins, rtol=1e-4, atol=1e-6, what="ins@256")
    close(dom, ref_dom, rtol=1e-3, atol=1e-6, what="dom@256")
    main.eval(); shape.eval()
    with torch.no_grad():
        pred, _ = main.predict(shape, img.to(DEV))
        ref_pred, _ = O.wt_pse_predict(sd_main, sd_shape, HP, img, False)
    close(pred, ref_pred, atol=TOL, what="predict@256")
    for b in range(B):
        d_hip = O.dice_coefficient((torch.sigmoid(pred[b, 0]) > 0.75).cpu().numpy(), od[b, 0].numpy())
        d_ref = O.dice_coefficient((torch.sigmoid(ref_pred[b, 0]) > 0.75).numpy(), od[b, 0].numpy())
        assert abs(d_hip - d_ref) <= 1e-4, (b, d_hip, d_ref)


def _off_unit_statistics(net, seed):
    """BatchNorm parameters and running statistics AWAY from the filler's O(1) (gamma in [0.8, 1.2], running_var in [0.6, 1.4]): whole
    blocks with gamma x 1e-3 and x 30, running variances over two decades, running means three times wider — what separates a trained
    checkpoint from a fresh one (every fixture lived in the O(1) regime, where round 5's fixed activation scale was right)."""
    from wtpse_hip import nn as E
    g = torch.Generator().manual_seed(seed)
    small, large = ("down2.", "prior_dist.down3.", "up1."), ("up3.", "prior_dist.up2.", "down4.")
    with torch.no_grad():
        for name, m in net.named_modules():
            if not isinstance(m, E.BNP):
                continue
            f = 1e-3 if any(name.startswith(p) for p in small) else 30.0 if any(name.startswith(p) for p in large) else 1.0
            m.weight.mul_(f)
            m.bias.mul_(f)
            m.running_var.mul_((10.0 ** (torch.rand(m.c, generator=g) * 2.0 - 1.0)).to(m.running_var.device))
            m.running_mean.mul_(3.0)
    net.invalidate_packed()


def _rescale_blocks(net, plan):
    """Function-preserving rescaling: the BatchNorm of a conv + BatchNorm (+ReLU) layer gets gamma, beta x f and the convolution that
    consumes its output gets its weights x 1/f (ReLU, max-pool and bilinear upsampling are positively homogeneous, f > 0): the network
    computes the same function with the same conditioning, but the activation between the two layers is f times larger and the
    consumer's weights f times smaller.  plan: {block prefix: f}; within a ConvD block bn1 -> conv2 and bn2 -> conv3, within a ConvU
    block bn1 -> conv2 and bn2 -> the second half of conv3's input channels (cat(prev, y), algorithms.py:955)."""
    mods = dict(net.named_modules())
    with torch.no_grad():
        for prefix, f in plan.items():
            blk = mods[prefix]
            pairs = []
            if hasattr(blk, "conv1"):
                pairs.append((blk.bn1, blk.conv2, None))
            lo = blk.conv3.cin - blk.bn2.c if blk.conv3.cin != blk.bn2.c else None      # ConvU: y is the second half
            pairs.append((blk.bn2, blk.conv3, lo))
            for bn, conv, lo in pairs:
                bn.weight.mul_(f)
                bn.bias.mul_(f)
                if lo is None:
                    conv.weight.mul_(1.0 / f)
                else:
                    conv.weight[:, lo:].mul_(1.0 / f)
    net.invalidate_packed()


RESCALE = {"down2": 1e-3, "up3": 30.0, "down4": 1e4, "up1": 1e-5, "prior_dist.down3": 1e-4, "prior_dist.up2": 3e3, "inc": 64.0}


@pytest.mark.parametrize("H", [64, 256])
def test_update_predict_rescaled_blocks_vs_oracle(H):
    """Round 6: activations far from O(1) at the plain 1e-4 bar.  Whole blocks are rescaled function-preservingly
    (_rescale_blocks: gamma, beta x f, the consumer's weights / f, f from 1e-5 to 1e4), so the reference computes the same, equally
    well-conditioned function — but the tensors between the rescaled layers are up to 1e4 and down to 1e-5 of their usual size.  Round
    5's fixed 2^2 input scale lost relative precision below 2^-5 and overflowed to NaN above 2^14 there; the data-driven scale (the
    Samuelson bound behind a train-mode BatchNorm, the stored data's amax behind an eval-mode one) must hold the bar at every f.
    Train-mode update() and eval-mode predict() against the CPU oracle on the same state dict."""
    B, pb = 6, 2
    img, od, oc = make_inputs(177, B, H, H)
    eps = make_noise(178, (B, 1, H, H))
    main, shape, _, _ = build_nets(pb)
    _rescale_blocks(main, RESCALE)
    _rescale_blocks(shape, {k: v for k, v in RESCALE.items() if not k.startswith("prior_dist") and k != "inc"})
    sd_main = {k: v.detach().cpu().clone() for k, v in main.state_dict().items()}
    sd_shape = {k: v.detach().cpu().clone() for k, v in shape.state_dict().items()}
    main.train()
    main.set_noise([eps])
    with torch.no_grad():
        out, _, _, ins, dom = main.update(img.to(DEV), od.to(DEV), two_stage_inputs=img.to(DEV), two_step=True)
        ref_out, _, _, ref_ins, ref_dom = O.wt_pse_update(dict(sd_main), HP, img, od, img, True, eps, 3, pb)
    assert bool(torch.isfinite(out).all())
    close(out, ref_out, atol=TOL, what="logits, rescaled blocks")
    close(ins, ref_ins, rtol=1e-4, atol=1e-6, what="ins, rescaled blocks")
    close(dom, ref_dom, rtol=1e-3, atol=1e-6, what="dom, rescaled blocks")
    main.load_state_dict(sd_main)            # (the train-mode call advanced the running statistics)
    main.eval(); shape.eval()
    with torch.no_grad():
        pred, att = main.predict(shape, img.to(DEV))
        ref_pred, ref_att = O.wt_pse_predict(sd_main, sd_shape, HP, img, False)
    assert bool(torch.isfinite(pred).all())
    close(pred, ref_pred, atol=TOL, what="predict, rescaled blocks")
    close(att, ref_att, atol=TOL, what="attention, rescaled blocks")


@pytest.mark.parametrize("H", [64, 256])
def test_update_predict_off_unit_statistics_vs_oracle(H):
    """The same question with parameters that are NOT function-preserving: whole blocks with BatchNorm gammas x 1e-3 and x 30, running
    statistics over two decades (_off_unit_statistics).  A gamma of 1e-3 in front of a convolution whose bias is O(0.1) makes the next
    BatchNorm subtract a mean a hundred times larger than the signal: the REFERENCE's own fp32 evaluation is then 9e-4 .. 2e-3 away
    from its fp64 evaluation on these inputs (measured: tools/probe/offunit_diag.py), so the 1e-4 bar is not a property any fp32
    implementation has here.  Calibrated criterion instead, as for the gradients: against the oracle in fp64, the HIP path may be at
    most as far as the reference's fp32 path is (or 1e-4 of the output scale, whichever is larger); and nothing may be non-finite."""
    B, pb = 6, 2
    img, od, oc = make_inputs(177, B, H, H)
    eps = make_noise(178, (B, 1, H, H))
    main, shape, _, _ = build_nets(pb)
    _off_unit_statistics(main, 5)
    _off_unit_statistics(shape, 6)
    sd_main = {k: v.detach().cpu().clone() for k, v in main.state_dict().items()}
    sd_shape = {k: v.detach().cpu().clone() for k, v in shape.state_dict().items()}
    to64 = lambda d: {k: (v.double() if v.is_floating_point() else v) for k, v in d.items()}
    main.train()
    main.set_noise([eps])
    with torch.no_grad():
        out = main.update(img.to(DEV), od.to(DEV), two_stage_inputs=img.to(DEV), two_step=True)[0]
        r32 = O.wt_pse_update(dict(sd_main), HP, img, od, img, True, eps, 3, pb)[0]
        heartbeat("off-unit oracle fp64")
        r64 = O.wt_pse_update(to64(sd_main), HP, img.double(), od.double(), img.double(), True, eps.double(), 3, pb)[0]
    assert bool(torch.isfinite(out).all())
    e_hip, e_ref = float((out.cpu().double() - r64).abs().max()), float((r32.double() - r64).abs().max())
    print("update, off-unit statistics: |HIP - fp64| %.3e, |reference fp32 - fp64| %.3e, scale %.3g" % (e_hip, e_ref, float(r64.abs().max())))
    assert e_hip <= max(TOL * float(r64.abs().max()), e_ref), (e_hip, e_ref)
    main.load_state_dict(sd_main)
    main.eval(); shape.eval()
    with torch.no_grad():
        pred = main.predict(shape, img.to(DEV))[0]
        p32 = O.wt_pse_predict(sd_main, sd_shape, HP, img, False)[0]
        p64 = O.wt_pse_predict(to64(sd_main), to64(sd_shape), HP, img.double(), False)[0]
    assert bool(torch.isfinite(pred).all())
    e_hip, e_ref = float((pred.cpu().double() - p64).abs().max()), float((p32.double() - p64).abs().max())
    print("predict, off-unit statistics: |HIP - fp64| %.3e, |reference fp32 - fp64| %.3e, scale %.3g" % (e_hip, e_ref, float(p64.abs().max())))
    assert e_hip <= max(TOL * float(p64.abs().max()), e_ref), (e_hip, e_ref)


def test_update_predict_256_B32_vs_oracle():
    """BASELINE.json configs[2] exactly: 3x256x256, batch 32, per_domain_batch 10 (rows 30-31 take part in the instance loss
    and in BatchNorm but not in the MMD, algorithms.py:107).  Train-mode `update()` (logits, WT-loss values) and eval-mode
    `predict()` against the CPU oracle on the same inputs at the 1e-4 bar, plus Dice of every thresholded prediction."""
    B, pb, H = 32, 10, 256
    img, od, oc = make_inputs(97, B, H, H)
    eps = make_noise(98, (B, 1, H, H))
    main, shape, _, _ = build_nets(pb)
    sd_main = {k: v.detach().cpu().clone() for k, v in main.state_dict().items()}
    sd_shape = {k: v.detach().cpu().clone() for k, v in shape.state_dict().items()}
    main.train()
    main.set_noise([eps])
    with torch.no_grad():
        out, _, _, ins, dom = main.update(img.to(DEV), od.to(DEV), two_stage_inputs=img.to(DEV), two_step=True)
        ref_out, _, _, ref_ins, ref_dom = O.wt_pse_update(dict(sd_main), HP, img, od, img, True, eps, 3, pb)
    close(out, ref_out, atol=TOL, what="logits@256,B=32")
    close(ins, ref_ins, rtol=1e-4, atol=1e-6, what="ins@256,B=32")
    close(dom, ref_dom, rtol=1e-3, atol=1e-6, what="dom@256,B=32")
    main.eval(); shape.eval()
    with torch.no_grad():
        pred, att = main.predict(shape, img.to(DEV))
        ref_pred, ref_att = O.wt_pse_predict(sd_main, sd_shape, HP, img, False)
    close(pred, ref_pred, atol=TOL, what="predict@256,B=32")
    close(att, ref_att, atol=TOL, what="attention@256,B=32")
    for b in range(B):
        d_hip = O.dice_coefficient((torch.sigmoid(pred[b, 0]) > 0.75).cpu().numpy(), od[b, 0].numpy())
        d_ref = O.dice_coefficient((torch.sigmoid(ref_pred[b, 0]) > 0.75).numpy(), od[b, 0].numpy())
        assert abs(d_hip - d_ref) <= 1e-4, (b, d_hip, d_ref)


def test_wt_loss_full_size_properties():
    """[32,16,256,256] (BASELINE.json configs[1..2]): value against the oracle, scale law G(a z) = a^2 G(z)
    (checked on the off-diagonal sum, margin 0), and the adjoint identity <dL/dz, z> = 2 * sum_b <dL/dG_b, G_b - eps I>."""
    from wtpse_hip import ops
    B, pb = 32, 10
    g = torch.Generator().manual_seed(5)
    z = torch.randn(B, 16, 256, 256, generator=g) * 0.5
    zd = z.to(DEV)
    st = ops.wt_loss_fwd(zd, 3, pb, 0.0)
    off, dg, dom = O.whitening_loss(z, 3, pb, 0.0)
    close(st.losses[0], off, rtol=1e-4, atol=1e-7, what="off")
    close(st.losses[1], dg, rtol=1e-4, atol=1e-7, what="diag")
    close(st.losses[2], dom, rtol=2e-3, atol=1e-6, what="dom")
    st2 = ops.wt_loss_fwd((zd * 2.0).contiguous(), 3, pb, 0.0)
    close(st2.offdiag, st.offdiag * 4.0, rtol=1e-4, atol=1e-7, what="scale law")
    dz = torch.empty_like(zd)
    ops.wt_loss_bwd(st, dz, False, w_dom=0.0)
    # L = mean_b off_b/120 + mean_b diag_b/16 with G linear in z z^T: <dL/dz, z> = 2 <dL/dG, z z^T/(HW-1)>
    G = st.gram.view(B, 16, 16).double().cpu()
    eye = torch.eye(16, dtype=torch.float64)
    Gz = G - 1e-5 * eye
    triu = torch.ones(16, 16, dtype=torch.float64).triu(1)
    dG = torch.sign(G) * triu / (120.0 * B) + torch.sign(G - eye) * eye / (16.0 * B)
    rhs = 2.0 * (dG * Gz).sum()
    lhs = (dz.double() * zd.double()).sum().cpu()
    assert abs(float(lhs - rhs)) <= 1e-4 * abs(float(rhs)) + 1e-7, (float(lhs), float(rhs))


# ---------------------------------------------------------------- BASELINE.json configs[4]: 3x512x512
def test_update_predict_512_vs_oracle():
    """512x512 (configs[4]'s resolution), B=3: train-mode logits + WT-loss values of `update()` and eval-mode `predict()`
    logits against the CPU oracle on the same inputs, at the 1e-4 bar."""
    B, pb, H = 3, 1, 512
    img, od, _ = make_inputs(87, B, H, H)
    eps = make_noise(88, (B, 1, H, H))
    main, shape, _, _ = build_nets(pb)
    sd_main = {k: v.detach().cpu().clone() for k, v in main.state_dict().items()}
    sd_shape = {k: v.detach().cpu().clone() for k, v in shape.state_dict().items()}
    main.train()
    main.set_noise([eps])
    with torch.no_grad():
        out, _, _, ins, dom = main.update(img.to(DEV), od.to(DEV), two_stage_inputs=img.to(DEV), two_step=True)
        ref_out, _, _, ref_ins, ref_dom = O.wt_pse_update(dict(sd_main), HP, img, od, img, True, eps, 3, pb)
    close(out, ref_out, atol=TOL, what="logits@512")
    close(ins, ref_ins, rtol=1e-4, atol=1e-6, what="ins@512")
    close(dom, ref_dom, rtol=1e-3, atol=1e-6, what="dom@512")
    main.eval(); shape.eval()
    with torch.no_grad():
        pred, _ = main.predict(shape, img.to(DEV))
        ref_pred, _ = O.wt_pse_predict(sd_main, sd_shape, HP, img, False)
    close(pred, ref_pred, atol=TOL, what="predict@512")


def test_wt_loss_512_vs_oracle_and_properties():
    """[16,16,512,512] — the per-GPU WT-loss input of configs[4] (B=128 over 8 GPUs): forward values and dL/dz against
    the oracle (autograd of the restated loss), the scale law and the adjoint identity of the 256x256 test."""
    from wtpse_hip import ops
    B, pb = 16, 5
    g = torch.Generator().manual_seed(6)
    z = torch.randn(B, 16, 512, 512, generator=g) * 0.5
    zd = z.to(DEV)
    st = ops.wt_loss_fwd(zd, 3, pb, 0.0)
    zr = z.clone().requires_grad_(True)
    off, dg, dom = O.whitening_loss(zr, 3, pb, 0.0)
    close(st.losses[0], off, rtol=1e-4, atol=1e-7, what="off")
    close(st.losses[1], dg, rtol=1e-4, atol=1e-7, what="diag")
    close(st.losses[2], dom, rtol=2e-3, atol=1e-6, what="dom")
    (off + dg + dom).backward()
    dz = torch.empty_like(zd)
    ops.wt_loss_bwd(st, dz, False)
    ref = zr.grad
    err = float((dz.cpu() - ref).norm() / ref.norm())
    assert err < 1e-4, "dL/dz @512: relative L2 error %.3e" % err
    st2 = ops.wt_loss_fwd((zd * 2.0).contiguous(), 3, pb, 0.0)
    close(st2.offdiag, st.offdiag * 4.0, rtol=1e-4, atol=1e-7, what="scale law")
    dz0 = torch.empty_like(zd)
    ops.wt_loss_bwd(st, dz0, False, w_dom=0.0)
    G = st.gram.view(B, 16, 16).double().cpu()
    eye = torch.eye(16, dtype=torch.float64)
    triu = torch.ones(16, 16, dtype=torch.float64).triu(1)
    dG = torch.sign(G) * triu / (120.0 * B) + torch.sign(G - eye) * eye / (16.0 * B)
    rhs = 2.0 * (dG * (G - 1e-5 * eye)).sum()
    lhs = (dz0.double() * zd.double()).sum().cpu()
    assert abs(float(lhs - rhs)) <= 1e-4 * abs(float(rhs)) + 1e-7, (float(lhs), float(rhs))
