#!/usr/bin/env python3
"""WT-PSE training throughput on MI355X (BASELINE.json metric: training images/sec at 256x256; WT-loss GB/s).

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W
    python bench.py --gpus N ...        (WORLD_SIZE unset: starts those N ranks itself as a child `torch.distributed.run`,
                                         before this process touches a GPU, and relays rank 0's line)

One "step" = one full iteration of the reference's hot loop (Trainer.py:766-914): calls A-D, four backward passes,
four Adam steps over the four networks, on one synthetic batch resident in HBM.  Workload = BASELINE.json configs[2]
(full WT-PSE, 3x256x256, batch 32 per GPU; weak scaling: the global batch is 32*N, configs[3] at N = 8).
Prints ONE JSON line on rank 0.
"""
import argparse
import glob
import json
import os
import sys
import time

# dmabuf IPC for RCCL / cross-process device memory: HSA reads this when the runtime initialises, i.e. at the first
# torch.cuda call, so it is set before torch is imported
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.environ.get("WTPSE_PKG_DIR") or os.path.join(ROOT, "wt-pse-code_amd")     # (WTPSE_PKG_DIR: same-box A/B against another build of the package)
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F32_PEAK_TF = 157.3     # MI355X_MICROARCH.md: fp32-input MFMA = fp32 vector peak
# the x3 kernels form one fp32 product from several 16-bit MFMA products: their bound is the dense bf16 / fp16 MFMA peak (16x the
# fp32-input rate, MI355X_MICROARCH.md "Matrix cores": ~2.5 PFLOP/s; the two formats run at the same rate) / the number of products
# — 6 in the x3 arithmetic (three bf16 terms per operand, rounds 2-4), 3 in x2h (two fp16 terms, the default since round 5), 1 in
# the bf16 mode.  Set in main() from the library's setting (wtpse_x3_terms).
X3_PRODUCTS = {3: 6, 2: 3, 1: 1}
X3_NAME = {3: "x3 (3 bf16 terms per fp32 operand, 6 bf16 MFMA products per multiply, fp32 accumulation)",
           2: "x2h (2 fp16 terms per fp32 operand — power-of-two operand scaling —, 3 fp16 MFMA products per multiply, fp32 accumulation)",
           1: "bf16 mode (operands rounded to one bf16 term, one MFMA product, fp32 accumulation)"}
MFMA_X3_PEAK_TF = 16.0 * 157.3 / 3.0
X3_TERMS = 2
GFLOP_PER_IMAGE = 224.5      # SURVEY.md §8d / BASELINE.md §5: necessary conv FLOPs of one full iteration at 256x256


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32, help="images per GPU")
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--workload", choices=["full", "seg"], default="full",
                    help="full = configs[2] (seg + shape nets + WT loss); seg = configs[1] (seg-net only)")
    ap.add_argument("--dtype", choices=["f32", "bf16"], default="f32",
                    help="f32 (default): fp32 results — the MFMA-bound layers in the library's current fp32-accurate arithmetic (x2h: two fp16 "
                         "terms per operand, three products; WTPSE_X3_TERMS=3: x3, three bf16 terms, six products); bf16: BASELINE.json configs[1]'s stated dtype — those layers with ONE bf16 term per operand and one "
                         "MFMA product, fp32 accumulation (wtpse_x3_terms(1)); outside the 1e-4 parity bar, reported as its own line only")
    ap.add_argument("--bn-sync", type=int, default=0, help="1: BatchNorm statistics over the global batch (parity mode)")
    ap.add_argument("--launch", choices=["eager", "plan", "graph"], default=None,
                    help="default: plan on one GPU, eager with several (the gradient all-reduces then start inside the backward, "
                         "overlapped with it: collectives cannot sit inside a recorded stretch).  plan: the step is recorded once during set-up and replayed from native code (csrc/plan.hip): "
                         "5 ms instead of 22 ms of host time per step at B=32 (same images/s within noise: 573 vs 577), 319 vs 261 "
                         "images/s at the reference's own B=6 where eager launches are host-bound; eager: one ctypes call per "
                         "launch; graph: hipGraph replay (slower than eager on this runtime, profiles/r02_hipgraph_vs_eager.txt).  "
                         "Exact data-parallel mode always runs eagerly")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-full", action="store_true",
                    help="SURVEY.md 8d protocol in full: 3 warm-up + 10 timed iterations at B=6 and B=30 (minutes of CPU time)")
    ap.add_argument("--cpu-threads", type=int, default=0,
                    help="threads of the CPU baseline; 0 (default) = min(CPUs in the affinity mask, physical cores of the host, the cgroup's "
                         "CPU-time quota) (SURVEY.md 8d: all physical cores the process may use); profiles/r06_cpu_baseline_threads.json: "
                         "16 threads beside 128 on a box with a 16-core quota")
    ap.add_argument("--roi-presteps", type=int, default=70,
                    help="untimed set-up steps that train the optic-disc net so that its prediction (the ROI of calls C/D) is no "
                         "longer empty: a FIXED count (default 70: od_pred then covers ~21 %% of the pixels of the seed-1 batch; "
                         "reported as `roi_presteps`), so that runs and profiles are comparable; -1 = adaptive (until 10-60 %% of "
                         "the pixels are inside, at most 400), 0 = none")
    ap.add_argument("--no-kernel-roofline", action="store_true")
    ap.add_argument("--kernels-only", action="store_true",
                    help="run only the per-kernel roofline launches (used under rocprofv3 --pmc to measure HBM traffic)")
    return ap.parse_args()


def build_nets(hp, pb, dev, seed=1):
    import algorithms
    import shape_networks
    torch.manual_seed(seed)
    mk = lambda ts: algorithms.WT_PSE(3, 1, hp, dev, ts, per_domain_batch=pb, source_domain_num=3).to(dev)
    model_od, model_oc = mk(False), mk(True)
    if not hp["whitening"]:
        return model_od, None, model_oc, None
    mks = lambda: shape_networks.ShapeVariationalDist_x(hp, dev, n_classes=1, number_source_domain=3, batch_size=pb).to(dev)
    return model_od, mks(), model_oc, mks()


def aux_seg_only(B, H, dev, steps, warmup):
    """BASELINE.json configs[1] — "seg-net only (no DWT loss), bf16" — measured after the headline run: call A with whitening and
    shape prior off, backward, Adam (`--workload seg`), with fp32 results (x3 arithmetic) and in the bf16 mode (`--dtype bf16`:
    one bf16 term per operand in the MFMA-bound layers; outside the 1e-4 parity bar by construction)."""
    from wtpse_hip import ops
    from wtpse_hip.step import TrainStep
    from wtpse_hip.synth import make_batch, default_hparams
    hp = default_hparams(False)
    image, target_od, target_oc = make_batch(B, H, H, dev, seed=1)
    out = {"workload": "BASELINE.json configs[1]: segmentation net only (whitening and shape prior off), 3x%dx%d, batch %d" % (H, H, B),
           "step": "call A + backward + Adam (Trainer.py:766-808)"}
    for name, terms in (("f32", ops.x3_terms()), ("bf16", 1)):
        was = ops.lib().query("wtpse_x3_terms", terms)
        try:
            ts = TrainStep(*build_nets(hp, B // 3, dev), hp, dp=None, graph="plan")
            for _ in range(2 + warmup):
                ts.step(image, target_od, target_oc)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                res = ts.step(image, target_od, target_oc)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            assert all(float(v) == float(v) for v in res.values()), "NaN loss in the seg-only run"
            out[name] = {"value": B * steps / dt, "unit": "images/s", "ms_per_step": 1e3 * dt / steps, "steps": steps,
                         "conv_tflops_end_to_end": B * steps / dt * seg_gflop_per_image(H) / 1e3}
            del ts
        finally:
            ops.lib().query("wtpse_x3_terms", was)
    out["bf16"]["dtype_note"] = "one bf16 term per operand, one MFMA product, fp32 accumulation; NOT within the 1e-4 parity bar"
    return out


def time_kernel(fn, reps=20):
    """Average duration (ms) of one launch from HIP events on the stream the kernels run on.  `fn`: a callable, or a LIST of
    callables on different operand sets that the launches rotate through — the HBM-bound kernels are timed on >= 4 sets with a
    footprint > 512 MB, so that no launch finds its input in the 256 MiB Infinity Cache from the launch before (with
    one 134 MB set the 'HBM' rates were cache rates)."""
    fns = fn if isinstance(fn, (list, tuple)) else [fn]
    for f in fns:
        f()
    fns[0]()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        fns[i % len(fns)]()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


ROT = 5        # operand sets of the HBM-bound micro-benchmarks: 5 x 134 MB = 671 MB of inputs alone


# The convolution layers of ONE U-Net of the step at 256x256: (Cin0, Cin1 (virtual concat), Cout, H / 256, k, name)
UNET_LAYERS = [
    (3, 0, 16, 1, 3, "inc.conv1"), (16, 0, 16, 1, 3, "inc.conv2"), (16, 0, 16, 1, 3, "inc.conv3"),
    (16, 0, 32, 2, 3, "down1.conv1"), (32, 0, 32, 2, 3, "down1.conv2"), (32, 0, 32, 2, 3, "down1.conv3"),
    (32, 0, 64, 4, 3, "down2.conv1"), (64, 0, 64, 4, 3, "down2.conv2"), (64, 0, 64, 4, 3, "down2.conv3"),
    (64, 0, 128, 8, 3, "down3.conv1"), (128, 0, 128, 8, 3, "down3.conv2"), (128, 0, 128, 8, 3, "down3.conv3"),
    (128, 0, 256, 16, 3, "down4.conv1"), (256, 0, 256, 16, 3, "down4.conv2"), (256, 0, 256, 16, 3, "down4.conv3"),
    (256, 0, 128, 16, 1, "up1.conv2 (before the upsampling)"), (128, 128, 256, 8, 3, "up1.conv3"),
    (256, 0, 128, 8, 3, "up2.conv1"), (128, 0, 64, 8, 1, "up2.conv2 (before the upsampling)"), (64, 64, 128, 4, 3, "up2.conv3"),
    (128, 0, 64, 4, 3, "up3.conv1"), (64, 0, 32, 4, 1, "up3.conv2 (before the upsampling)"), (32, 32, 64, 2, 3, "up3.conv3"),
    (64, 0, 32, 2, 3, "up4.conv1"), (32, 0, 16, 2, 1, "up4.conv2 (before the upsampling)"), (16, 16, 32, 1, 3, "up4.conv3"),
]


def seg_gflop_per_image(H):
    """Necessary convolution GFLOP per image of one seg-only step (configs[1]: two plain U-Nets — OD and OC — each forward + data
    gradients + weight gradients; the first layer has no data gradient): from the layer table above plus the mu / outc 1x1 heads."""
    fwd = sum(2.0 * (c0 + c1) * co * k * k * (H // d) ** 2 for c0, c1, co, d, k, _ in UNET_LAYERS)
    fwd += 2.0 * (32 * 32 + 32 * 8 + 8 * 1) * H * H
    first = 2.0 * 3 * 16 * 9 * H * H
    return 2.0 * (3.0 * fwd - first) / 1e9


def unet_layer_rooflines(B, H, dev):
    """Every convolution of one U-Net, launched AS THE STEP LAUNCHES IT (round 6): forward with the BatchNorm+ReLU
    prologue on the inputs, bias and the BatchNorm statistics of the output finished in the launch (nn.convbn_fwd; the 1x1 convs in
    front of the upsampling: plain, their statistics come from the upsampling kernel); data gradient with the BatchNorm-backward
    epilogue of the layer below — mask load of its raw output, the two reductions, the coefficient fold (nn._dgrad(below=); the first
    convolution of a ConvD block feeds the max-pool backward: plain) —; weight gradient through nn._wgrad.  HIP events, operands
    rotating through two sets (cold: inside a step no launch finds its operands in the Infinity Cache either).  Rounds 1-5 timed the
    PLAIN launches here (no statistics fold, no BatchNorm-backward epilogue): 16 % / 32 % less work per forward / data-gradient launch
    (tools/probe/instep_gap.py) — that, not clocks or cold operands (3-6 %), was the 19 % between the isolated and the in-step rate of
    the family.  The plain timings stay in the rows (`*_plain_ms`).
    -> per-layer rows and, for the layers on the x3 kernels (the MFMA-bound ones), the FLOP-WEIGHTED rate over ALL of them per
    direction — not the four friendliest layers."""
    from wtpse_hip import nn as E
    from wtpse_hip import ops
    rows = []
    for c0, c1, co, div, k, name in UNET_LAYERS:
        Hc = H // div
        cin = c0 + c1

        class Holder(E.HipNet):
            def __init__(self):
                super().__init__()
                self.conv, self.bn = E.ConvP(cin, co, k), E.BNP(co)
                self.bn_below = E.BNP(c1 if c1 else cin)
                self._finish_init()
        net = Holder().to(dev)
        net.train()
        net.ensure_ready(repack=True)
        layer = net.conv
        pro = c0 > 4
        sets = []
        for i in range(2):
            x0 = torch.randn(B, c0, Hc, Hc, device=dev)
            x1 = torch.randn(B, c1, Hc, Hc, device=dev) if c1 else None
            a0 = E.Act(x0, torch.rand(c0, 2, device=dev) + 0.5, True) if pro else E.Act(x0)
            a1 = (E.Act(x1, torch.rand(c1, 2, device=dev) + 0.5, True) if c1 else None)
            dy = torch.randn(B, co, Hc, Hc, device=dev)
            E.act_amax(a0); E.act_amax(a1)
            if ops.x3_terms() == 2:
                ops.amax_of(dy)
            sets.append((a0, a1, dy))
        # the layer below, as the data gradient's BatchNorm-backward epilogue sees it (the second half of a concat, else all of the input)
        cb = c1 if c1 else cin
        below = E.Tape()
        below.y = torch.randn(B, cb, Hc, Hc, device=dev)
        below.ss = torch.rand(cb, 2, device=dev) + 0.5
        below.mean = torch.randn(cb, device=dev) * 0.1
        below.invstd = torch.rand(cb, device=dev) + 0.5
        below.relu, below.bn = True, net.bn_below
        fl = 2.0 * cin * co * k * k * Hc * Hc * B
        # a conv layer's own roofline is min(MFMA peak, HBM bandwidth x its ideal intensity), i.e. its floor
        # is the LARGER of flop / MFMA peak and (input + output bytes, fp32, once) / HBM peak
        nbytes = 4.0 * B * (cin + co) * Hc * Hc
        r = {"layer": "%s %d%s->%d k%d @%dx%d" % (name, c0, ("+%d" % c1) if c1 else "", co, k, Hc, Hc), "flop": fl, "bytes": nbytes,
             "floor_ms": max(fl / (MFMA_X3_PEAK_TF * 1e9), nbytes / (HBM_PEAK_GBS * 1e6)),
             "floor_bound": "mfma" if fl / (MFMA_X3_PEAK_TF * 1e9) >= nbytes / (HBM_PEAK_GBS * 1e6) else "hbm",
             "fwd_path": "x3" if layer.xf_off >= 0 else ("x3/16" if layer.x16f_off >= 0 else "fp32"),
             "dgrad_path": "x3" if layer.xd_off >= 0 else ("x3/16" if layer.x16d_off >= 0 else "fp32")}
        before_up = "before the upsampling" in name
        dgrad_plain = name.startswith("down") and name.endswith("conv1")     # feeds the max-pool backward
        with ops.fwd_scope(dev):
            r["fwd_plain_ms"] = time_kernel([(lambda s=s_: E._conv(layer, s[0], s[1], False, True)) for s_ in sets])
            if before_up:
                r["fwd_ms"] = time_kernel([(lambda s=s_: E._conv(layer, s[0], None, False, False)) for s_ in sets])
            else:
                r["fwd_ms"] = time_kernel([(lambda s=s_: E.convbn_fwd(layer, net.bn, s[0], s[1], True, True, want_tape=False)) for s_ in sets])
        net.begin_backward()
        if cin > 4:
            split = c0 if c1 else None
            r["dgrad_plain_ms"] = time_kernel([(lambda s=s_: E._dgrad(layer, s[2], split)) for s_ in sets])
            if dgrad_plain:
                r["dgrad_ms"] = r["dgrad_plain_ms"]
            else:
                r["dgrad_ms"] = time_kernel([(lambda s=s_: E._dgrad(layer, s[2], split, below0=None if c1 else below, below1=below if c1 else None))
                                             for s_ in sets])
        r["wgrad_ms"] = time_kernel([(lambda s=s_: E._wgrad(layer, s[2], s[0], s[1], with_bias=False)) for s_ in sets])
        for d in ("fwd", "dgrad", "wgrad", "fwd_plain", "dgrad_plain"):
            if d + "_ms" in r:
                r[d + "_tflops"] = fl / r[d + "_ms"] / 1e9
        rows.append(r)
        del sets, below, net
    out = {"layers": rows}
    for d, pathkey in (("fwd", "fwd_path"), ("dgrad", "dgrad_path"), ("wgrad", "fwd_path")):
        sel = [r for r in rows if r[pathkey] == "x3" and d + "_ms" in r and (d != "wgrad" or "k3" in r["layer"])]
        fl, ms = sum(r["flop"] for r in sel), sum(r[d + "_ms"] for r in sel)
        out[d] = {"tflops": fl / ms / 1e9, "frac": fl / ms / 1e9 / MFMA_X3_PEAK_TF, "layers": len(sel), "ms_sum": ms,
                  "min_tflops": min(r[d + "_tflops"] for r in sel), "min_tflops_3x3": min(r[d + "_tflops"] for r in sel if " k3 " in r["layer"])}
    # mean algorithmic bytes (input + output, fp32) per launch over the 3x3 x3 forward / data-gradient launches of this table PLUS the
    # eight conv3-of-up1..up4 launches of kernel_rooflines(): the launch set `traffic` of the x3 convolution family is averaged over in
    # `--kernels-only` runs (tools/pmc_traffic.py matches conv_x3_k<3,...> / conv_x3r_k; every timed entry makes the same number of launches)
    ent = []
    for r in rows:
        if " k3 " not in r["layer"]:
            continue
        c0, c1, co, div, k, name = next(t for t in UNET_LAYERS if r["layer"].startswith(t[5]))
        nb = 4.0 * B * (c0 + c1 + co) * (H // div) ** 2
        if r["fwd_path"] == "x3":
            ent.append(nb)
        if r["dgrad_path"] == "x3" and "dgrad_ms" in r:
            ent.append(nb)
        if name in ("up1.conv3", "up2.conv3", "up3.conv3", "up4.conv3"):
            ent += [nb, nb]
    out["x3_3x3_mean_bytes"] = sum(ent) / max(len(ent), 1)
    f, g = out["fwd"], out["dgrad"]
    fl = sum(r["flop"] for r in rows if r["fwd_path"] == "x3") + sum(r["flop"] for r in rows if r["dgrad_path"] == "x3" and "dgrad_ms" in r)
    out["fwd_dgrad"] = {"tflops": fl / (f["ms_sum"] + g["ms_sum"]) / 1e9, "frac": fl / (f["ms_sum"] + g["ms_sum"]) / 1e9 / MFMA_X3_PEAK_TF}
    # the same launches WITHOUT the step's fused epilogues (what rounds 1-5 reported as `roofline.achieved`)
    pms = sum(r["fwd_plain_ms"] for r in rows if r["fwd_path"] == "x3") + sum(r["dgrad_plain_ms"] for r in rows if r["dgrad_path"] == "x3" and "dgrad_plain_ms" in r)
    out["fwd_dgrad"]["plain_tflops"] = fl / pms / 1e9
    out["fwd_dgrad"]["plain_frac"] = fl / pms / 1e9 / MFMA_X3_PEAK_TF
    # the per-layer min(MFMA, HBM x intensity) roofline over the same launches: sum of the layers' floors / sum of their measured times
    floor = sum(r["floor_ms"] for r in rows if r["fwd_path"] == "x3") + sum(r["floor_ms"] for r in rows if r["dgrad_path"] == "x3" and "dgrad_ms" in r)
    out["fwd_dgrad"]["per_layer_min_frac"] = floor / (f["ms_sum"] + g["ms_sum"])
    out["fwd_dgrad"]["hbm_bound_layers"] = [r["layer"] for r in rows if r["fwd_path"] == "x3" and r["floor_bound"] == "hbm"]
    return out


FAMILIES = (("x3_conv", ("conv_x3_k", "conv_x3r_k")), ("x3_wgrad", ("wgrad_r_k<2", "wgrad_r_k<1, 2", "wgrad_r_k<2, 1", "conv_wgrad_x3_k", "wgrad_fold4_k")),
            ("bn_backward", ("bn_bwd_",)), ("conv16", ("conv_fwd_k<3, 3", "conv_fwd_k<3, 4", "wgrad_r_k<1, 1")), ("heads", ("head_",)),
            ("conv_fp32", ("conv_fwd_k", "conv_wgrad_k", "wgrad_reduce_k")))


def dominant_kernel_share():
    """The kernel FAMILY with the largest share of the summed kernel time in the newest in-step rocprofv3 summary committed under
    profiles/ (r*_bench_b32_single_stream_kernel_stats.csv: the kernels back to back; falls back to the three-stream profile): the
    `roofline` object is that family's (by family, not by the largest single template instantiation)."""
    import csv
    files = (sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_b32_single_stream_kernel_stats.csv")))
             or sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_b32_kernel_stats.csv"))))
    if not files:
        return None
    # the profile belongs to the library it was measured on: profiles/r<NN>_STAMP.json carries that library's source hash
    # (tools/profile_round.sh); a profile of OTHER kernels than the ones loaded now is named, but no in-step rate is derived from it
    stamp_path = os.path.join(ROOT, "profiles", os.path.basename(files[-1]).split("_")[0] + "_STAMP.json")
    stamp_ok = None
    try:
        from wtpse_hip import build
        stamp_ok = json.load(open(stamp_path)).get("source_hash") == build.source_hash()
    except (OSError, ValueError):
        pass
    with open(files[-1]) as f:
        rows = [r for r in csv.DictReader(f) if not r["Name"].startswith("__amd_rocclr")]
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    fam = {}
    for r in rows:
        for name, pats in FAMILIES:
            if any(p in r["Name"] for p in pats):
                fam[name] = fam.get(name, 0.0) + float(r["TotalDurationNs"])
                break
    top = max(fam, key=fam.get)
    steps = sum(int(r["Calls"]) for r in rows if r["Name"].startswith("adam_k")) / 4.0     # four Adam launches per step
    return {"profile": os.path.relpath(files[-1], ROOT), "family": top, "percent": 100.0 * fam[top] / tot,
            "families_percent": {k: round(100.0 * v / tot, 1) for k, v in sorted(fam.items(), key=lambda kv: -kv[1])},
            "kernel": top, "steps_in_profile": steps, "family_ms_per_step": (fam[top] / steps / 1e6) if steps else None,
            "all_kernels_ms_per_step": (tot / steps / 1e6) if steps else None, "same_library": stamp_ok}


# U-Net passes of one full iteration on the x3 forward / data-gradient family (calls A-D, Trainer.py:766-914; dead teacher
# backward of calls B/D skipped): A and C = main + teacher forward and data gradient (4 each), B and D = teacher + student forward,
# student data gradient (3 each); weight-gradient passes: A/C 2 each, B/D 1 each
X3_FWD_PASSES, X3_DGRAD_PASSES, X3_WGRAD_PASSES = 8, 6, 6


def in_step_fraction(dom, un, B):
    """The leading family's rate INSIDE the step, from the committed back-to-back (single-stream) rocprofv3 summary: the FLOPs one
    step issues on that family (per-layer table of this run: which layers run on the x3 kernels, forward / data gradient counted
    separately) / the family's summed kernel time per step in that profile.  Not measured by this run."""
    if not dom or not dom.get("family_ms_per_step") or not dom.get("same_library"):
        return None
    rows = un["layers"]
    f = sum(r["flop"] for r in rows if r["fwd_path"] == "x3")
    g = sum(r["flop"] for r in rows if r["dgrad_path"] == "x3" and "dgrad_ms" in r)
    w = sum(r["flop"] for r in rows if r["fwd_path"] == "x3" and " k3 " in r["layer"])
    if dom["family"] == "x3_wgrad":
        flop = X3_WGRAD_PASSES * w
    else:
        flop = X3_FWD_PASSES * f + X3_DGRAD_PASSES * g
    tf = flop / dom["family_ms_per_step"] / 1e9
    return {"tflops": tf, "frac": tf / MFMA_X3_PEAK_TF, "gflop_per_step_on_family": flop / 1e9, "batch": B,
            "family_ms_per_step": dom["family_ms_per_step"], "profile": dom["profile"]}


def kernel_rooflines(B, H, dev):
    """Live HIP-event timings of the kernels BASELINE.json names a roofline for, at the benchmark's own shapes, launched
    through the same dispatch the training step uses (wtpse_hip/nn.py: layers with > 16 output channels run the split-bf16
    "x3" kernels of csrc/conv_x3.hip).
    * `x3_conv`: conv_x3_k<3,2,5> — forward (BatchNorm+ReLU prologue on both inputs of the virtual concat, bias, BatchNorm
      partials in the epilogue) and data gradient (split output) of conv3 of up1..up4, 1208 MFLOP/img each (SURVEY.md
      Appendix B): the FLOP-heaviest launches of the kernel that leads the in-step profile.
    * `x3_wgrad`: conv_wgrad_x3_k + the slab fold on the same four layers.
    * `conv`: the fp32-input-MFMA kernel conv_fwd_k<3,2,5> of round 1 on up3.conv3 (kept for comparison; WTPSE_X3=0 path).
    * the WT-loss Gram kernels (compute_whitening_loss forward / backward, 16*H*W*4 bytes/img/pass) against HBM.
    FLOPs are the convolution's 2*Cin*Cout*9*H*W*B throughout (one fp32 multiply-add = 2 FLOP), whatever the kernel
    issues to form them."""
    from wtpse_hip import ops
    from wtpse_hip import nn as E
    out = {}
    wg, fw, dg = [], [], []
    for name, C, Hc in (("up1.conv3", 256, H // 8), ("up2.conv3", 128, H // 4), ("up3.conv3", 64, H // 2), ("up4.conv3", 32, H)):
        class Holder(E.HipNet):
            def __init__(self):
                super().__init__()
                self.conv = E.ConvP(C, C, 3)
                self._finish_init()
        net = Holder().to(dev)
        net.ensure_ready(repack=True)
        layer = net.conv
        x0, x1 = torch.randn(B, C // 2, Hc, Hc, device=dev), torch.randn(B, C // 2, Hc, Hc, device=dev)
        p0, p1 = torch.rand(C // 2, 2, device=dev) + 0.5, torch.rand(C // 2, 2, device=dev) + 0.5
        dy = torch.randn(B, C, Hc, Hc, device=dev)
        a0, a1 = E.Act(x0, p0, True), E.Act(x1, p1, True)
        fl = 2.0 * C * C * 9 * Hc * Hc * B
        tag = "%s %d+%d->%d @%dx%d" % (name, C // 2, C // 2, C, Hc, Hc)
        ms = time_kernel(lambda: E._conv(layer, a0, a1, False, True))
        fw.append({"layer": tag, "ms": ms, "tflops": fl / ms / 1e9, "flop": fl})
        ms = time_kernel(lambda: E._dgrad(layer, dy, C // 2))
        dg.append({"layer": tag, "ms": ms, "tflops": fl / ms / 1e9, "flop": fl})
        net.begin_backward()
        ms = time_kernel(lambda: E._wgrad(layer, dy, a0, a1, with_bias=False))
        wg.append({"layer": tag, "ms": ms, "tflops": fl / ms / 1e9, "flop": fl})
        del x0, x1, dy, net
    x3 = X3_NAME[X3_TERMS] if E.X3 else "fp32-input MFMA (WTPSE_X3=0)"

    def agg(rows, kernel):
        tot_ms, tot_fl = sum(r["ms"] for r in rows), sum(r["flop"] for r in rows)
        return {"kernel": kernel, "ms": tot_ms / len(rows), "tflops": tot_fl / tot_ms / 1e9, "flop_per_launch": tot_fl / len(rows),
                "launches": rows}
    shape = "conv3 of up1..up4 (3x3, C/2+C/2->C, C=256..32 @%d..%d), B=%d" % (H // 8, H, B)
    out["x3_fwd"] = agg(fw, "conv_x3_k forward, %s: %s, BatchNorm+ReLU prologue on both inputs, bias + BatchNorm partials" % (x3, shape))
    out["x3_dgrad"] = agg(dg, "conv_x3_k data gradient, %s: %s, split output" % (x3, shape))
    out["x3_conv"] = agg(fw + dg, "conv_x3_k forward + data gradient, %s: %s" % (x3, shape))
    out["x3_wgrad"] = agg(wg, "wgrad_r_k (register-resident operands, v_mfma_f32_16x16x32_bf16) + slab fold, %s: %s, "
                          "BatchNorm+ReLU prologue on both inputs" % (x3, shape))
    # the 16-channel 256x256 layers (inc.conv2/3, DeepWT, teacher inc): forward / data gradient in the x3 arithmetic on
    # v_mfma_f32_16x16x32_bf16 with register-resident weight fragments (conv_fwd_k<3,3,5>; WTPSE_X16=0: the fp32-input MFMA
    # conv_fwd_k<3,0,5>), weight gradient on wgrad_r_k<1,1> — HBM-bound: 2 x 16 x H x W x 4 bytes per image either way
    C16 = 16
    x16s = [torch.randn(B, C16, H, H, device=dev) for _ in range(ROT)]
    dy16s = [torch.randn(B, C16, H, H, device=dev) for _ in range(ROT)]
    x16, dy16 = x16s[0], dy16s[0]
    p16 = torch.rand(C16, 2, device=dev) + 0.5

    class Holder16(E.HipNet):
        def __init__(self):
            super().__init__()
            self.conv = E.ConvP(C16, C16, 3)
            self._finish_init()
    n16 = Holder16().to(dev)
    n16.ensure_ready(repack=True)
    a16 = E.Act(x16, p16, True)
    a16s = [E.Act(x, p16, True) for x in x16s]
    nb16 = 2.0 * B * C16 * H * H * 4
    ms = time_kernel([(lambda a=a: E._conv(n16.conv, a, None, False, True)) for a in a16s])
    out["c16_fwd"] = {"kernel": "conv_fwd_k<3,%d,5> 16->16 3x3 @%dx%d B=%d (BatchNorm+ReLU prologue, bias, BatchNorm partials)" % ((4 if X3_TERMS == 2 else 3) if E.X16 else 0, H, H, B),
                      "ms": ms, "gbs": nb16 / ms / 1e6, "bytes_per_launch": nb16, "tflops": 2.0 * 16 * 16 * 9 * H * H * B / ms / 1e9}
    n16.begin_backward()
    ms = time_kernel([(lambda a=a, d=d: E._wgrad(n16.conv, d, a, None, with_bias=True)) for a, d in zip(a16s, dy16s)])
    out["c16_wgrad"] = {"kernel": "wgrad_r_k<1,1> + slab fold 16->16 3x3 @%dx%d B=%d (prologue, with bias gradient)" % (H, H, B),
                        "ms": ms, "gbs": nb16 / ms / 1e6, "bytes_per_launch": nb16, "tflops": 2.0 * 16 * 16 * 9 * H * H * B / ms / 1e9}
    C, Hc = 64, H // 2
    x = torch.randn(B, C, Hc, Hc, device=dev)
    w = torch.randn(C, C, 3, 3, device=dev) * 0.05
    packed = torch.empty(C * 9 * C * 2, device=dev)
    desc = torch.tensor([0, C, C, 9, 0, C * 9 * C, 0, 0], dtype=torch.int32, device=dev)
    ops.lib().call("wtpse_pack_conv_weights", w.data_ptr(), desc.data_ptr(), 1, packed.data_ptr(), ops.stream_ptr())
    y = torch.empty(B, C, Hc, Hc, device=dev)
    bias = torch.zeros(C, device=dev)
    stats = torch.empty(ops.lib().query("wtpse_conv_stats_blocks", B, Hc, Hc) * C * 2, device=dev)

    def conv():   # the fp32-input-MFMA kernel, as round 1 launched it: bias + BatchNorm (sum, sum^2) partials in the epilogue
        ops.lib().call("wtpse_conv_fwd", x.data_ptr(), C, 0, 0, packed.data_ptr(), bias.data_ptr(), 0, 0, 0, y.data_ptr(), 0, C,
                       stats.data_ptr(), B, Hc, Hc, C, 3, 0, 0, 0, ops.stream_ptr())
    ms = time_kernel(conv)
    flops = 2.0 * C * C * 9 * Hc * Hc * B
    out["conv"] = {"kernel": "conv_fwd_k<3,2,5> (fp32-input MFMA) 64->64 3x3 @%dx%d B=%d (+bias, BN partials)" % (Hc, Hc, B), "ms": ms,
                   "tflops": flops / ms / 1e9, "flop_per_launch": flops}
    zs = [torch.randn(B, 16, H, H, device=dev) for _ in range(ROT)]
    z = zs[0]
    L = ops.lib()
    S = L.query("wtpse_wt_split", B, H * H, 0)
    partial = torch.empty(B * S * 256, device=dev)
    bufs = [torch.empty(B * 256, device=dev), torch.empty(B * 120, device=dev), torch.empty(B, device=dev),
            torch.empty(B, device=dev), torch.empty(B + 1, dtype=torch.float64, device=dev), torch.empty(B * 120, device=dev),
            torch.empty(3, device=dev)]
    pb = B // 3

    def wt(zz):
        L.call("wtpse_wt_loss_fwd", zz.data_ptr(), B, 16, H * H, 1e-5, 0.0, 3, pb, partial.data_ptr(), *[b.data_ptr() for b in bufs],
               ops.stream_ptr())
    ms = time_kernel([(lambda zz=zz: wt(zz)) for zz in zs])
    nbytes = B * 16 * H * H * 4.0
    out["wt_fwd"] = {"kernel": "wtpse_wt_loss_fwd (gram_partial_k + tail) [%d,16,%d,%d], %d operand sets in rotation" % (B, H, H, ROT), "ms": ms,
                     "gbs": nbytes / ms / 1e6, "bytes_per_launch": nbytes}
    # in a training step the Gram partials come from the epilogue of the DeepWT conv that writes z (wtpse_conv_fwd_gram):
    # the loss then costs the extra epilogue time plus the tail on the partials, and never reads z
    t_plain = time_kernel([(lambda x=x: E._conv(n16.conv, x)) for x in x16s])
    t_gram = time_kernel([(lambda x=x: E._conv_gram(n16.conv, x)) for x in x16s])
    zz, gp = E._conv_gram(n16.conv, x16)
    t_tail = time_kernel(lambda: ops.wt_loss_fwd(zz, 3, pb, 0.0, gram_partial=gp))
    del x16, dy16, x16s, dy16s, a16s, n16
    out["wt_fwd"]["fused_in_step"] = {
        "what": "Gram partials in the epilogue of the conv that writes z (16->16 3x3, the kernel of c16_fwd) + wtpse_wt_loss_fwd_partials",
        "conv_ms": t_plain, "conv_with_gram_epilogue_ms": t_gram, "tail_on_partials_ms": t_tail,
        "loss_cost_ms": (t_gram - t_plain) + t_tail, "hbm_bytes_avoided_per_call": nbytes,
        "equivalent_gbs": nbytes / ((t_gram - t_plain) + t_tail) / 1e6}
    dzs = [torch.empty_like(z) for _ in range(ROT)]
    M = torch.randn(B * 256, device=dev) * 1e-3

    def wtb(zz, dz):
        L.call("wtpse_wt_loss_bwd", zz.data_ptr(), B, 16, H * H, 0.0, 3, pb, bufs[0].data_ptr(), bufs[2].data_ptr(), bufs[3].data_ptr(),
               bufs[5].data_ptr(), 0, 0, 0, 1.0, 1.0, 1.0, M.data_ptr(), dz.data_ptr(), 0, ops.stream_ptr())
    ms = time_kernel([(lambda zz=zz, dz=dz: wtb(zz, dz)) for zz, dz in zip(zs, dzs)])
    out["wt_bwd"] = {"kernel": "wtpse_wt_loss_bwd (gram_bwd_k) [%d,16,%d,%d], %d operand sets in rotation" % (B, H, H, ROT), "ms": ms,
                     "gbs": 2 * nbytes / ms / 1e6, "bytes_per_launch": 2 * nbytes}
    # counter calibration streams (tools/pmc_traffic.py): plain copies of a known size at 16 and 4 bytes per lane, rotating too
    ncal = zs[0].numel()
    for width in (16, 4):
        ms = time_kernel([(lambda a=a, b=b: L.call("wtpse_copy_probe", a.data_ptr(), b.data_ptr(), ncal, width, ops.stream_ptr()))
                          for a, b in zip(zs, dzs)])
        out["copy_w%d" % width] = {"kernel": "copy_w%d_k: %d MB in + %d MB out, %d bytes per lane" % (width, ncal * 4 >> 20, ncal * 4 >> 20, width),
                                   "ms": ms, "gbs": 2.0 * ncal * 4 / ms / 1e6, "bytes_per_launch": 2.0 * ncal * 4}
    del z, zs, dzs, partial
    # The literal 2-D DWT BASELINE.json's wording names: a stand-alone micro-benchmark, NOT part of WT-PSE, parity unpinned
    # (csrc/dwt.hip).  Algorithmic bytes: every level reads and writes its region once.
    from wtpse_hip import dwt
    out["dwt"] = []
    for shp, lv in (((B, 16, H, H), 3), ((max(B // 2, 1), 16, 2 * H, 2 * H), 4)):
        xds = [torch.randn(*shp, device=dev) for _ in range(ROT if shp[2] == H else 3)]
        xd = xds[0]
        nb = 8.0 * xd.numel() * sum(0.25 ** l for l in range(lv))
        for wv in ("haar", "db2"):
            ms = time_kernel([(lambda x=x: dwt.dwt2(x, wv, lv)) for x in xds])
            out["dwt"].append({"kernel": "wtpse_dwt2_fwd %s, %d levels, %s" % (wv, lv, list(shp)), "ms": ms, "gbs": nb / ms / 1e6,
                               "bytes_per_launch": nb, "min_bytes": 8.0 * xd.numel(), "gbs_min_bytes": 8.0 * xd.numel() / ms / 1e6})
        del xd, xds
    out["unet"] = unet_layer_rooflines(B, H, dev)
    return out


def log(msg):
    if int(os.environ.get("RANK", "0")) == 0:
        print("[bench %.1fs] %s" % (time.time() - T0, msg), file=sys.stderr, flush=True)


T0 = time.time()


def physical_cores():
    """(CPU model, physical cores of the host) from /proc/cpuinfo; (None, None) if it cannot be read."""
    model, phys = None, set()
    try:
        pid = cid = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name") and model is None:
                model = line.split(":", 1)[1].strip()
            elif line.startswith("physical id"):
                pid = line.split(":", 1)[1].strip()
            elif line.startswith("core id"):
                cid = line.split(":", 1)[1].strip()
            elif not line.strip():
                if pid is not None and cid is not None:
                    phys.add((pid, cid))
                pid = cid = None
    except OSError:
        pass
    return model, (len(phys) or None)


def cpu_quota():
    """CPU-time quota of this process's cgroup in cores (cgroup v2 cpu.max / v1 cfs_quota_us), or None when there is none."""
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else max(1, int(round(float(q) / float(p))))
    except (OSError, ValueError):
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else max(1, int(round(q / p)))
    except (OSError, ValueError):
        return None


def cpu_baseline(H, full, full_protocol=False, threads=0):
    """The CPU restatement (oracle/, bit-checked against the reference in the build container) timed on this box's host
    cores: full A-D iterations (4 forward + 4 backward + 4 Adam) at B = 6 — what `--batch-size 8` yields
    in the reference — and B = 30, and compute_whitening_loss alone on [32,16,H,H].  Default: a bounded sample
    (1 warm-up + 3 timed at B = 6, one iteration at B = 30; ~40 s); --cpu-baseline-full: 3 warm-up + 10 timed at both."""
    from oracle import wtpse_cpu as O
    from oracle.inputs import make_inputs, make_noise
    from wtpse_hip.synth import default_hparams
    import algorithms
    import shape_networks
    hp = default_hparams(full)
    if not full:
        return None
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    granted = cores
    model, phys = physical_cores()
    quota = cpu_quota()
    # all physical cores — as many as this process may USE.  The affinity mask of a 1-GPU box shows all 256 logical CPUs
    # of the host, but its cgroup grants a CPU-TIME quota of 16 cores: with 128 threads the same iteration took 17.6 s
    # instead of 2.2 s (0.34 vs 2.7 images/s, measured in round 6: profiles/r06_cpu_baseline_threads.json) — the threads queue for the
    # quota.  Threads = min(affinity, physical cores, cgroup quota); all three are reported.
    cores = threads if threads > 0 else max(1, min(granted, phys or granted, quota or granted))
    torch.set_num_threads(cores)

    def host_info():
        """CPU model and PHYSICAL core count of the box (state it next to `cores`)."""
        return {"cpu_model": model, "physical_cores_on_host": phys, "logical_cpus_on_host": os.cpu_count(),
                "cpus_granted_to_this_process": granted, "cgroup_cpu_quota_cores": quota, "threads_used": cores}

    def iteration_rate(B, warm, timed):
        pb = B // 3
        sds = []
        for ctor in (lambda: algorithms.WT_PSE(3, 1, hp, "cpu", False, per_domain_batch=pb),
                     lambda: shape_networks.ShapeVariationalDist_x(hp, "cpu", 1, 3, pb),
                     lambda: algorithms.WT_PSE(3, 1, hp, "cpu", True, per_domain_batch=pb),
                     lambda: shape_networks.ShapeVariationalDist_x(hp, "cpu", 1, 3, pb)):
            m = ctor()                   # only used as a weight container (state_dict); no compute on these modules
            sds.append({k: v.detach().clone() for k, v in m.state_dict().items()})
        nets = O.Nets(*sds)
        img, od, oc = make_inputs(1, B, H, H)
        nz = {k: make_noise(10 + j, (B, 1, H, H)) for j, k in enumerate(["a", "b_t", "b_s", "c", "d_t", "d_s"])}
        ts = []
        for i in range(warm + timed):
            t0 = time.time()
            O.train_iteration(nets, hp, img, od, oc, nz, pb)
            log("cpu baseline: B=%d iteration %d/%d took %.1f s" % (B, i + 1, warm + timed, time.time() - t0))
            if i >= warm:
                ts.append(time.time() - t0)
        ts.sort()
        return B / ts[len(ts) // 2], ts

    r6, t6 = iteration_rate(6, 3 if full_protocol else 1, 10 if full_protocol else 3)
    r30, t30 = iteration_rate(30, 3 if full_protocol else 0, 10 if full_protocol else 1)
    # (B=30 is slower PER IMAGE than B=6 on the CPU: at B=6 a 16-channel 256x256 activation is 25 MB and the step's working set stays
    # in the EPYC's 256 MB+ of L3; at B=30 every tensor is 126 MB and streams from DRAM — profiles/r06_cpu_baseline_full.json has the
    # per-image time at B = 6, 12, 18, 30)
    by_batch = None
    if full_protocol:
        by_batch = {}
        for bb in (12, 18):
            rb, tb = iteration_rate(bb, 1, 3)
            by_batch[str(bb)] = {"images_per_s": rb, "s_per_iteration": tb[len(tb) // 2]}
        by_batch["6"] = {"images_per_s": r6, "s_per_iteration": t6[len(t6) // 2]}
        by_batch["30"] = {"images_per_s": r30, "s_per_iteration": t30[len(t30) // 2]}
    z = torch.randn(32, 16, H, H)
    tw = []
    for i in range(13):
        t0 = time.time()
        with torch.no_grad():
            O.whitening_loss(z, 3, 10, 0.0)
        if i >= 3:
            tw.append(time.time() - t0)
    tw.sort()
    wt_gbs = z.numel() * 4.0 / tw[len(tw) // 2] / 1e9
    return {"value": r6, "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port", "host": host_info(), "by_batch": by_batch,
            "protocol": "SURVEY.md 8d in full: 3 warm-up + 10 timed iterations at B=6 and B=30" if full_protocol else
                        "bounded sample (1 warm-up + 3 timed at B=6, one iteration at B=30); the full protocol: --cpu-baseline-full",
            "sample": "median of %d full A-D iterations (4 fwd + 4 bwd + 4 Adam) after %d warm-up, B=6, 3x%dx%d, torch CPU fp32 "
                      "(%.1f s each)" % (len(t6), 3 if full_protocol else 1, H, H, t6[len(t6) // 2]),
            "b30": {"value": r30, "unit": "images/s", "sample": "median of %d iteration(s), B=30 (%.1f s each)" % (len(t30), t30[len(t30) // 2])},
            "wt_loss_fwd": {"value": wt_gbs, "unit": "GB/s", "sample": "compute_whitening_loss forward on [32,16,%d,%d], median of 10 "
                            "after 3 warm-up (%.1f ms)" % (H, H, 1e3 * tw[len(tw) // 2])}}


def _rnd(v, n=5):
    """floats to n significant digits (the stdout line is for a parser, not for reading noise)"""
    if isinstance(v, float):
        return float("%.*g" % (n, v))
    if isinstance(v, dict):
        return {k: _rnd(x, n) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [_rnd(x, n) for x in v]
    return v


COMPACT_LIMIT = 8192


def compact_line(line):
    """The ONE stdout line of the contract, kept small enough for the driver's parser (round 4's line was 31 KB and
    `BENCH_r04.json.parsed` came back null): the contract's keys, `roofline` (dominant kernel family: live isolated-launch rate,
    the in-step rate from the committed profile, the whole step, PMC traffic against algorithmic bytes), the two WT-loss lines with
    their rate against this box's own streaming copy, `cpu_baseline`.  Per-layer tables and every other roofline object go to the
    detail file / stderr (`emit`)."""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roi_presteps", "host_enqueue_ms_per_step", "conv_tflops_end_to_end", "gflop_per_image")
    out = {k: line[k] for k in keep if k in line}
    if "degenerate_roi" in line:
        out["degenerate_roi_images_per_s"] = line["degenerate_roi"]["value"]
    r = line.get("roofline")
    copy16 = (line.get("roofline_copy_w16") or {}).get("achieved")
    if r:
        tr = r.get("traffic") or {}
        dom = r.get("dominant_in_profile") or {}
        ins = r.get("in_step") or {}
        out["roofline"] = {
            "bound": r["bound"], "kernel": "conv_x3r_k + conv_x3_k (x3 forward + data gradient family)" if "forward + data" in r["kernel"]
            else "wgrad_r_k (x3 weight gradient family)",
            "achieved": r["achieved"], "peak": r["peak"], "unit": r["unit"], "frac": r["frac"],
            "per_layer_min_frac": r.get("per_layer_min_frac"),
            "achieved_what": "FLOP-weighted over every launch of one U-Net on this family AS THE STEP MAKES THEM (statistics fold / BatchNorm-backward "
                             "epilogue in the launch, cold operands), HIP events, this run; plain_frac: the same launches without the fused epilogues "
                             "(what rounds 1-5 reported)",
            "plain_frac": r.get("plain_frac"),
            "in_step_frac": r.get("in_step_frac"), "in_step_tflops": ins.get("tflops"), "in_step_family_ms_per_step": ins.get("family_ms_per_step"),
            "in_step_source": (dom.get("profile") and ("committed %s (back-to-back kernels), not this run" % dom["profile"] if dom.get("same_library")
                                                       else "withheld: %s was measured on another library than the one loaded" % dom["profile"])),
            "family_percent_of_step_kernel_time": dom.get("percent") if dom.get("same_library") else None,
            "step_frac": r.get("step_frac"), "best_layers_frac": r.get("best_layers_frac"),
            # continuity with rounds 2-4, whose bound was the x3 scheme's 419.5 TFLOP/s (6 products per multiply): the same TFLOP/s against it
            "frac_of_round4_x3_bound_419_5": r["achieved"] / (16.0 * 157.3 / 6.0),
            "traffic": tr.get("hbm_bytes"), "traffic_read": tr.get("hbm_read_bytes"), "traffic_write": tr.get("hbm_write_bytes"),
            "traffic_algorithmic": (r.get("traffic_algorithmic") or {}).get("hbm_bytes"),
            "traffic_source": "committed profiles/pmc_traffic.json (rocprofv3 --pmc, separate passes, copy-kernel calibrated), same library" if tr else r.get("traffic_source"),
            "peak_note": r.get("peak_note")}
    for k in ("roofline_wt_fwd", "roofline_wt_bwd", "roofline_c16_fwd", "roofline_x3_wgrad"):
        w = line.get(k)
        if not w:
            continue
        o = {"bound": w["bound"], "achieved": w["achieved"], "peak": w["peak"], "unit": w["unit"], "frac": w["frac"],
             "ms_per_launch": w.get("ms_per_launch"), "traffic": (w.get("traffic") or {}).get("hbm_bytes")}
        if w["bound"] == "hbm":
            o["algorithmic_bytes"] = w.get("bytes_per_launch")
            o["frac_of_copy"] = (w["achieved"] / copy16) if copy16 else None
        if "fused_in_step" in w:
            o["fused_in_step_loss_cost_ms"] = w["fused_in_step"]["loss_cost_ms"]
        out[k] = o
    if copy16:
        out["copy_yardstick_gbs"] = copy16
    sh = line.get("step_hbm")
    if sh:
        out["step_hbm"] = {"gb_per_step": sh["gb_per_step"], "floor_ms_at_copy_rate": sh["floor_ms_at_copy_rate"],
                           "floor_frac_of_step": sh["floor_frac_of_step"], "source": sh["source"]}
    s1 = line.get("configs1_seg_only")
    if s1:
        out["configs1_seg_only"] = {k: {"value": s1[k]["value"], "ms_per_step": s1[k]["ms_per_step"]} for k in ("f32", "bf16") if k in s1}
    c = line.get("cpu_baseline")
    if c:
        out["cpu_baseline"] = {"value": c["value"], "unit": c["unit"], "cores": c["cores"], "kind": c["kind"], "sample": c["sample"],
                               "cpu_model": (c.get("host") or {}).get("cpu_model"),
                               "physical_cores_on_host": (c.get("host") or {}).get("physical_cores_on_host"),
                               "cpus_granted": (c.get("host") or {}).get("cpus_granted_to_this_process"),
                               "cgroup_quota_cores": (c.get("host") or {}).get("cgroup_cpu_quota_cores"),
                               "b30_images_per_s": (c.get("b30") or {}).get("value"),
                               "wt_loss_fwd_gbs": (c.get("wt_loss_fwd") or {}).get("value")}
    out["detail"] = line.get("detail_file")
    return _rnd(out)


def emit(line, args):
    """Full record -> a side file (gpurun_out/ on the GPU box, else $TMPDIR) and stderr; compact record -> the ONE stdout line."""
    full = json.dumps(line)
    name = "bench_detail_n%d_%s%s.json" % (line["n_gpus"], args.workload, "" if args.dtype == "f32" else "_" + args.dtype)
    for d in (os.path.join(ROOT, "gpurun_out"), os.environ.get("TMPDIR", "/tmp")):
        try:
            os.makedirs(d, exist_ok=True)
            with open(os.path.join(d, name), "w") as f:
                f.write(full + "\n")
            line["detail_file"] = os.path.relpath(os.path.join(d, name), ROOT) if d.startswith(ROOT) else os.path.join(d, name)
            break
        except OSError:
            continue
    print("[bench detail] " + full, file=sys.stderr, flush=True)
    log("timed region (repeated behind the detail record): %.3f s for %d steps = %.1f %s" %
        (line["ms_per_step"] * line["steps"] / 1e3, line["steps"], line["value"], line["unit"]))
    small = json.dumps(compact_line(line))
    if len(small) >= COMPACT_LIMIT:          # never hand the driver a line it cannot parse: drop the optional objects
        c = compact_line(line)
        for k in ("configs1_seg_only", "roofline_x3_wgrad", "roofline_c16_fwd", "copy_yardstick_gbs", "degenerate_roi_images_per_s"):
            c.pop(k, None)
        small = json.dumps(c)
    assert len(small) < COMPACT_LIMIT, "bench line too long for the driver's parser: %d bytes" % len(small)
    print(small, flush=True)


def step_hbm_traffic(ms_per_step, copy_gbs):
    """HBM bytes ONE training step moves (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over whole steps, calibrated on the copy
    kernels: tools/pmc_step_traffic.py -> profiles/r<NN>_step_traffic.json, same-library stamp as the in-step profile) and what that
    traffic alone costs at this box's own streaming-copy rate: the HBM floor of the step as it is scheduled now."""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_step_traffic.json")))
    if not files:
        return None
    try:
        from wtpse_hip import build
        stamp = json.load(open(os.path.join(ROOT, "profiles", os.path.basename(files[-1]).split("_")[0] + "_STAMP.json")))
        if stamp.get("source_hash") != build.source_hash():
            return None
        t = json.load(open(files[-1]))
    except (OSError, ValueError):
        return None
    gb = t["_total"]["GB_per_step"]
    floor_ms = gb / copy_gbs * 1e3 if copy_gbs else None
    return {"gb_per_step": gb, "floor_ms_at_copy_rate": floor_ms, "floor_frac_of_step": (floor_ms / ms_per_step) if floor_ms else None,
            "by_family_gb": {k: (v["read_MB_per_step"] + v["write_MB_per_step"]) / 1e3 for k, v in t.items() if not k.startswith("_")},
            "source": "committed %s (rocprofv3 --pmc, separate FETCH_SIZE / WRITE_SIZE passes over whole steps), not this run" % os.path.relpath(files[-1], ROOT)}


def rccl_version():
    """Version of the collective library behind the `nccl` backend (RCCL on ROCm), as torch reports it; None if it cannot say."""
    try:
        v = torch.cuda.nccl.version()
        return ".".join(str(x) for x in v) if isinstance(v, (tuple, list)) else str(v)
    except Exception:
        return None


def _x3_on():
    from wtpse_hip import nn as E
    return bool(E.X3)


def measured_traffic():
    """HBM bytes per launch measured with rocprofv3 --pmc (FETCH_SIZE / WRITE_SIZE in separate passes; factors calibrated per access
    width on the copy kernels of the same run: tools/pmc_traffic.py) on `bench.py --kernels-only`, committed under profiles/.  The
    file carries the source hash of the library it was measured on (tools/profile_round.sh): a measurement taken on OTHER kernels
    than the ones loaded now is refused.  -> ({key: traffic}, note)."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if not os.path.isfile(path):
        return {}, "no committed measurement"
    with open(path) as f:
        tr = json.load(f)
    from wtpse_hip import build
    stamp = tr.get("_stamp", {})
    if stamp.get("source_hash") != build.source_hash():
        return {}, ("profiles/pmc_traffic.json was measured on library %s, the loaded library is %s: traffic withheld (re-run "
                    "tools/profile_round.sh)" % (stamp.get("source_hash"), build.source_hash()))
    return tr, ("committed profile profiles/pmc_traffic.json (rocprofv3 --pmc of `bench.py --kernels-only`, git %s, same library "
                "sources as loaded now), not measured by this run" % stamp.get("git_head", "?"))


def self_launch_command(argv, gpus, port=None):
    """`python bench.py --gpus N` started bare (WORLD_SIZE unset), as the driver starts N = 1: the command that runs the N ranks —
    one process per GPU under torch.distributed.run, rendezvous on 127.0.0.1 (the container hostname may not resolve) — as a CHILD
    of this process.  WTPSE_BENCH_LAUNCHER overrides the launcher (tests/test_bench_line_cpu.py substitutes a fake one)."""
    if port is None:
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
    launcher = os.environ.get("WTPSE_BENCH_LAUNCHER")
    head = launcher.split() if launcher else [sys.executable, "-m", "torch.distributed.run"]
    return head + ["--nnodes=1", "--nproc-per-node", str(gpus), "--master-addr", "127.0.0.1", "--master-port", str(port),
                   os.path.abspath(__file__)] + list(argv)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # (this used to die on the assert below.)  Nothing in this process has initialised the GPU yet — `import torch`
        # does not — and nothing will: the ranks are fresh children, never an exec of a process that touched the device; their stdout
        # (rank 0's ONE JSON line) and stderr are this process's own, the exit status is theirs.
        import subprocess
        cmd = self_launch_command(sys.argv[1:], args.gpus)
        log("WORLD_SIZE unset with --gpus %d: starting the ranks as a child process: %s" % (args.gpus, " ".join(cmd)))
        sys.exit(subprocess.run(cmd).returncode)
    if args.bn_sync and args.batch % 3 != 0:
        sys.exit("--bn-sync 1 (exact data-parallel mode) draws the sampling noise of every domain's rows from one global "
                 "Philox stream and needs a per-GPU batch that is a multiple of the 3 source domains: --batch %d is not "
                 "(use e.g. --batch 30 or 33)" % args.batch)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.launch is None:
        # N > 1: eager launches with the gradient exchange overlapped with the backward — at 32 images per GPU the step is GPU-bound
        # (host 22 ms vs GPU 42 ms); a small per-GPU batch is host-bound when launched eagerly and takes the five-stretch launch
        # plan instead (collectives between the stretches: tests/dp_worker.py, plan_equals_eager)
        args.launch = "plan" if (world == 1 or args.batch < 16) else "eager"
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, "--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d (or leave WORLD_SIZE unset: bench.py starts the ranks itself)" % (args.gpus, world, args.gpus)
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs"
    ndev = torch.cuda.device_count()
    backend = os.environ.get("WTPSE_DIST_BACKEND", "nccl")     # "gloo": rehearsal of the N>1 path on a 1-GPU box
    if backend == "nccl":
        assert local_rank < ndev, "rank %d has no GPU (%d visible)" % (local_rank, ndev)
    dev = torch.device("cuda", local_rank % ndev)
    torch.cuda.set_device(dev)
    dp = None
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
        from wtpse_hip.dp import DataParallel
        dp = DataParallel(world, rank, dev, bn_sync=bool(args.bn_sync))
        ranks_seen = dist.get_world_size()
    else:
        ranks_seen = 1

    from wtpse_hip import ops as _ops
    if args.dtype == "bf16":
        _ops.lib().query("wtpse_x3_terms", 1)
    global MFMA_X3_PEAK_TF, X3_TERMS
    X3_TERMS = _ops.x3_terms()
    MFMA_X3_PEAK_TF = 16.0 * 157.3 / X3_PRODUCTS[X3_TERMS]
    if args.kernels_only:
        kr = kernel_rooflines(args.batch, args.size, dev)
        print(json.dumps(kr))
        return
    from wtpse_hip.step import TrainStep
    from wtpse_hip.synth import make_batch, default_hparams
    full = args.workload == "full"
    hp = default_hparams(full)
    B, H = args.batch, args.size
    pb = B // 3                                  # per-domain rows on this rank; the MMD sees 3*pb*world rows
    nets = build_nets(hp, pb, dev)
    ts = TrainStep(*nets, hp, dp=dp, graph={"plan": "plan", "graph": True, "eager": False}[args.launch])
    image, target_od, target_oc = make_batch(B, H, H, dev, seed=1 + rank)

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    def timed(nsteps):
        """barrier + sync, nsteps steps, barrier + sync; -> (seconds = max over ranks, host enqueue seconds, last losses)"""
        barrier()
        c0 = time.process_time()
        t0 = time.perf_counter()
        for _ in range(nsteps):
            res = ts.step(image, target_od, target_oc)
        t_host = time.perf_counter() - t0        # host wall time to enqueue the steps: includes the time the runtime holds the
        timed.cpu = time.process_time() - c0     # thread back when the launch queues are full; .cpu = CPU time actually spent
        barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            import torch.distributed as dist
            tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt = float(tmax.item())
        return dt, t_host, {k: float(v) for k, v in res.items()}

    log("nets built, batch in HBM; priming the per-stream allocator pools%s" % (" and capturing the step" if ts.graph else ""))
    for i in range(2):       # part of set-up, like building the nets: the caching allocator's pools (one per stream) reach
        ts.step(image, target_od, target_oc)    # their steady size after two steps (graph mode: the capture happens here)
    torch.cuda.synchronize()
    # Set-up, continued: at the seed-1 initial weights the optic-disc net predicts an EMPTY disc, so calls C/D would run on a
    # constant ROI (all -1), i.e. half of the step on trivial operands, which the chip clocks higher than real data
    # (MI355X_MICROARCH.md, DVFS).  The degenerate state is timed first (reported as `degenerate_roi`), then the nets train on
    # the synthetic batch until the prediction covers a sensible part of the image, and the contract's W warm-up + K timed
    # steps run in that state.
    def od_fraction():       # the same number on every rank: the set-up loop below must take the same decisions everywhere
        f = ts.last_od_pred.mean().reshape(1).double()
        if world > 1:
            import torch.distributed as dist
            dist.all_reduce(f)
            f /= world
        return float(f)

    degenerate = None
    frac = od_fraction()
    presteps = 0
    if args.roi_presteps != 0 and full and not (0.02 <= frac <= 0.6):
        dt0, _, _ = timed(min(args.steps, 10))
        degenerate = {"value": world * B * min(args.steps, 10) / dt0, "unit": "images/s", "od_pred_fraction": frac,
                      "note": "same step, ROI of calls C/D empty (initial weights): constant operands in half of the step"}
        limit = args.roi_presteps if args.roi_presteps > 0 else 400
        while presteps < limit:
            for _ in range(10):
                ts.step(image, target_od, target_oc)
            presteps += 10
            frac = od_fraction()
            if args.roi_presteps < 0 and 0.1 <= frac <= 0.6:
                break
        log("%d set-up steps: od_pred now covers %.3f of the pixels (target discs %.3f)" % (presteps, frac, float(target_od.mean())))
    log("warmup")
    for i in range(args.warmup):
        ts.step(image, target_od, target_oc)
        torch.cuda.synchronize()
    dt, t_host, losses = timed(args.steps)
    log("timed region: %.3f s for %d steps (host enqueue %.3f s)" % (dt, args.steps, t_host))
    assert all(v == v for v in losses.values()), "NaN loss: %s" % losses

    if rank == 0:
        ips = world * B * args.steps / dt
        line = {
            "metric": ("training images/sec (%dx%d fundus) — " % (H, H)) + ("full WT-PSE iteration" if full else "seg-net only"),
            "value": ips, "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": ("bf16 (layers with > 16 output channels: operands rounded to one bf16 term, one MFMA product, fp32 accumulation; the "
                      "16-channel layers, the 1x1 heads, BatchNorm and Adam in fp32) — NOT within the 1e-4 parity bar"
                      if args.dtype == "bf16" else
                      "f32 (layers with > 16 output channels and the fused 1x1 heads: %s; the 16-channel layers the same on 16x16x32, the 1-/3-channel input layers and small 1x1 convs fp32-input MFMA)" % X3_NAME[X3_TERMS]
                      if _x3_on() else "f32"), "data": "synthetic",
            "config": {"workload": (("BASELINE.json configs[2]: full WT-PSE (seg+shape nets + WT loss)" if H == 256 else
                                     "BASELINE.json configs[4] per-GPU share: full WT-PSE at high resolution") if full else
                                    "BASELINE.json configs[1]: seg-net only") + ", 3x%dx%d, batch %d per GPU" % (H, H, B),
                       "global_batch": B * world, "image": [3, H, H], "parallelism": "dp%d" % world,
                       # what a SCALE record can be checked against: the ranks torch.distributed itself counted, the collective library
                       "n_ranks_seen": ranks_seen, "dist_backend": (backend if world > 1 else None), "rccl_version": rccl_version(),
                       "bn_sync": bool(args.bn_sync), "step": "calls A-D + 4 backward + 4 Adam (Trainer.py:766-914)",
                       "launch": ("native launch plan" if ts.plan else "hipGraph replay") if ts.graph else "eager",
                       "roi": "od_pred covers %.3f of the pixels after %d untimed set-up steps" % (frac, presteps)},
            "roi_presteps": presteps,
            "host_enqueue_ms_per_step": 1e3 * t_host / args.steps,
            "host_cpu_ms_per_step": 1e3 * timed.cpu / args.steps,
            "conv_tflops_end_to_end": (ips * GFLOP_PER_IMAGE * (H * H / 65536.0) / 1e3 if full else ips * seg_gflop_per_image(H) / 1e3),
            "gflop_per_image": GFLOP_PER_IMAGE * (H * H / 65536.0) if full else seg_gflop_per_image(H),
            "losses": losses,
        }
        if degenerate is not None:
            line["degenerate_roi"] = degenerate
        if world == 1 and not args.no_kernel_roofline:
            log("kernel rooflines")
            kr = kernel_rooflines(B, H, dev)
            tr, tr_note = measured_traffic() if (B, H) == (32, 256) else ({}, "traffic is measured at B=32, 256x256 only")
            dom = dominant_kernel_share()
            lead = "x3_wgrad" if (dom and dom["family"] == "x3_wgrad") else "x3_conv"

            def mfma_line(k, traffic_key):
                r = kr[k]
                return {"bound": "mfma", "kernel": r["kernel"], "achieved": r["tflops"], "peak": MFMA_X3_PEAK_TF,
                        "unit": "TFLOP/s", "frac": r["tflops"] / MFMA_X3_PEAK_TF, "traffic": tr.get(traffic_key),
                        "traffic_source": tr_note,
                        "dvfs_ceiling_note": "on random operands the chip holds 1.5-1.9 GHz under bf16 MFMA streams, not 2.4: a bare "
                                             "LDS-fed MFMA loop reaches ~250 TFLOP/s x3-equivalent, a register-fed 16x16x32 loop ~320 "
                                             "(profiles/r03_mfma_peak.txt, tools/probe/mfma_peak.hip)",
                        "frac_of_fp32_mfma_peak": r["tflops"] / MFMA_F32_PEAK_TF,
                        "peak_note": "dense 16-bit MFMA peak (16 x 157.3 = 2516.8) / %d product(s) per fp32 multiply; fp32-input MFMA peak 157.3" % X3_PRODUCTS[X3_TERMS],
                        "ms_per_launch": r["ms"], "flop_per_launch": r["flop_per_launch"], "launches": r.get("launches")}
            if dom:
                dom["source"] = "committed profile %s, not measured by this run" % dom["profile"]
            # Headline roofline: the kernel FAMILY that leads the in-step profile, priced on EVERY layer of a U-Net that runs on it
            # (FLOP-weighted: total FLOP / total time of the launches measured live here), not on its best layers.  `launches` keeps
            # the four conv3 layers of up1..up4 (rounds 1-3's number) for comparison; `step_frac` is the whole step against the bound.
            un = kr["unet"]
            fam = un["wgrad"] if lead == "x3_wgrad" else un["fwd_dgrad"]
            head = mfma_line(lead, lead)
            head.update({"achieved": fam["tflops"], "frac": fam["frac"], "per_layer_min_frac": fam.get("per_layer_min_frac"),
                         "plain_frac": fam.get("plain_frac"), "plain_tflops": fam.get("plain_tflops"),
                         "per_layer_min_frac_note": "SURVEY.md 8d: sum over the same launches of max(flop / MFMA peak, (input + output bytes) / "
                                                    "8 TB/s) / sum of their measured times — the layers whose own bound is HBM, not MFMA: %s"
                                                    % ", ".join(fam.get("hbm_bound_layers", [])) if lead == "x3_conv" else None,
                         "traffic_algorithmic": {"hbm_bytes": un["x3_3x3_mean_bytes"],
                                                 "note": "input + output bytes, mean over the same 3x3 launches `traffic` is averaged over"}
                         if lead == "x3_conv" else None,
                         "achieved_note": "FLOP-weighted over all %d forward + %d data-gradient launches of one U-Net that run on the x3 "
                                          "kernels (3x3 and 1x1, every level: `roofline_unet_layers`), HIP events" % (un["fwd"]["layers"], un["dgrad"]["layers"])
                         if lead == "x3_conv" else "FLOP-weighted over all %d 3x3 weight-gradient launches of one U-Net on the x3 kernels" % un["wgrad"]["layers"],
                         "best_layers_tflops": kr[lead]["tflops"], "best_layers_frac": kr[lead]["tflops"] / MFMA_X3_PEAK_TF,
                         "step_frac": (line["conv_tflops_end_to_end"] / MFMA_X3_PEAK_TF) if line["conv_tflops_end_to_end"] else None,
                         "step_frac_note": "conv_tflops_end_to_end (224.5 GFLOP of necessary convolution work per image x images/s) / the x3 bound: "
                                           "the whole step, every kernel and every gap included"})
            ins = in_step_fraction(dom, un, B) if (B, H) == (32, 256) and full and args.dtype == "f32" else None
            head["in_step_frac"] = ins["frac"] if ins else None
            head["in_step"] = ins
            line["roofline"] = dict(head, dominant_in_profile=dom)
            if (B, H) == (32, 256) and full and args.dtype == "f32":
                line["step_hbm"] = step_hbm_traffic(line["ms_per_step"], kr["copy_w16"]["gbs"])
            line["roofline_unet_layers"] = {"what": "every convolution of one U-Net as the step launches it (B=%d): FLOP-weighted TFLOP/s over the "
                                                    "layers on the x3 kernels, per direction, and the per-layer rows" % B,
                                            "fwd": un["fwd"], "dgrad": un["dgrad"], "wgrad": un["wgrad"], "fwd_dgrad": un["fwd_dgrad"],
                                            "peak": MFMA_X3_PEAK_TF, "layers": un["layers"]}
            for k in ("x3_fwd", "x3_dgrad", "x3_wgrad"):
                line["roofline_" + k] = mfma_line(k, "x3_wgrad" if k == "x3_wgrad" else "x3_conv")
            for k in ("copy_w16", "copy_w4"):
                w = kr[k]
                line["roofline_" + k] = {"bound": "hbm", "kernel": w["kernel"], "achieved": w["gbs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                         "frac": w["gbs"] / HBM_PEAK_GBS, "traffic": tr.get(k), "ms_per_launch": w["ms"],
                                         "bytes_per_launch": w["bytes_per_launch"],
                                         "note": "plain streaming copy: what this box's HBM sustains at this access width (the yardstick "
                                                 "for the HBM-bound kernels; MI355X_MICROARCH.md: ~6.3 TB/s achievable)"}
            c = kr["conv"]
            line["roofline_conv_fwd_fp32"] = {"bound": "mfma", "kernel": c["kernel"], "achieved": c["tflops"], "peak": MFMA_F32_PEAK_TF,
                                              "unit": "TFLOP/s", "frac": c["tflops"] / MFMA_F32_PEAK_TF, "traffic": tr.get("conv"),
                                              "ms_per_launch": c["ms"], "flop_per_launch": c["flop_per_launch"]}
            for k in ("wt_fwd", "wt_bwd"):
                w = kr[k]
                line["roofline_" + k] = {"bound": "hbm", "kernel": w["kernel"], "achieved": w["gbs"], "peak": HBM_PEAK_GBS,
                                         "unit": "GB/s", "frac": w["gbs"] / HBM_PEAK_GBS, "traffic": tr.get(k),
                                         "ms_per_launch": w["ms"], "bytes_per_launch": w["bytes_per_launch"]}
                if "fused_in_step" in w:
                    line["roofline_" + k]["fused_in_step"] = w["fused_in_step"]
            for k in ("c16_fwd", "c16_wgrad"):
                w = kr[k]
                line["roofline_" + k] = {"bound": "hbm", "kernel": w["kernel"], "achieved": w["gbs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                         "frac": w["gbs"] / HBM_PEAK_GBS, "traffic": tr.get(k), "ms_per_launch": w["ms"],
                                         "bytes_per_launch": w["bytes_per_launch"], "tflops": w["tflops"]}
            line["roofline_dwt"] = [{"bound": "hbm", "kernel": w["kernel"], "achieved": w["gbs_min_bytes"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                     "frac": w["gbs_min_bytes"] / HBM_PEAK_GBS, "traffic": tr.get("dwt_" + w["kernel"].split()[1].rstrip(",")),
                                     "ms_per_launch": w["ms"], "bytes_per_launch": w["min_bytes"],
                                     "bytes_note": "bytes_per_launch = 2 x 4 B x elements: the fused kernel keeps the levels after the first in "
                                                   "LDS, so the plane is read once and the coefficients are written once; round 2 priced every "
                                                   "level's read + write of its region (achieved_level_sum), which a fused transform never moves",
                                     "achieved_level_sum": w["gbs"], "bytes_level_sum": w["bytes_per_launch"],
                                     "note": "stand-alone micro-benchmark: the reference has no wavelet transform — not part of WT-PSE, "
                                             "parity unpinned (SURVEY.md 8f-4)"} for w in kr["dwt"]]
        if world == 1 and full and args.dtype == "f32" and not args.no_kernel_roofline and (B, H) == (32, 256):
            # BASELINE.json configs[1] beside the headline (configs[2]): the segmentation net alone, fp32 results and the bf16 mode
            log("configs[1] (seg-net only): fp32 results, bf16 mode")
            line["configs1_seg_only"] = aux_seg_only(B, H, dev, args.steps, args.warmup)
        if world == 1 and not args.no_cpu_baseline:
            log("cpu baseline")
            line["cpu_baseline"] = cpu_baseline(H, full, args.cpu_baseline_full, args.cpu_threads)
            log("done")
        emit(line, args)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
