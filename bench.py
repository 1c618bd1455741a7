#!/usr/bin/env python3
"""WT-PSE training throughput on MI355X (BASELINE.json metric: training images/sec at 256x256; WT-loss GB/s).

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

One "step" = one full iteration of the reference's hot loop (Trainer.py:766-914): calls A-D, four backward passes,
four Adam steps over the four networks, on one synthetic batch resident in HBM.  Workload = BASELINE.json configs[2]
(full WT-PSE, 3x256x256, batch 32 per GPU; weak scaling: the global batch is 32*N, configs[3] at N = 8).
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "wt-pse-code_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F32_PEAK_TF = 157.3     # MI355X_MICROARCH.md: fp32-input MFMA = fp32 vector peak
GFLOP_PER_IMAGE = 224.5      # SURVEY.md §8d / BASELINE.md §5: necessary conv FLOPs of one full iteration at 256x256


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=32, help="images per GPU")
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--workload", choices=["full", "seg"], default="full",
                    help="full = configs[2] (seg + shape nets + WT loss); seg = configs[1] (seg-net only)")
    ap.add_argument("--bn-sync", type=int, default=0, help="1: BatchNorm statistics over the global batch (parity mode)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
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


def time_kernel(fn, reps=20):
    """Average duration (ms) of one launch of `fn` from HIP events on the stream the kernels run on."""
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def kernel_rooflines(B, H, dev):
    """Live HIP-event timings of the kernels BASELINE.json names a roofline for, at the benchmark's own shapes:
    the dominant kernel of a step — conv_fwd_k<3,2,5> (forward + data-gradient of the >= 64-channel 3x3 layers, 23 % of
    a step in profiles/r01_bench_b32_kernel_stats.csv) on one of its FLOP-heaviest launches, up3.conv3's 3x3 64->64 at
    half resolution (1208 MFLOP/img — SURVEY.md Appendix B) — against the fp32 MFMA peak; and the WT-loss Gram kernels
    (compute_whitening_loss forward / backward, 16*H*W*4 bytes/img/pass) against HBM."""
    from wtpse_hip import ops
    out = {}
    C, Hc = 64, H // 2
    x = torch.randn(B, C, Hc, Hc, device=dev)
    w = torch.randn(C, C, 3, 3, device=dev) * 0.05
    packed = torch.empty(C * 9 * C * 2, device=dev)
    desc = torch.tensor([0, C, C, 9, 0, C * 9 * C, 0, 0], dtype=torch.int32, device=dev)
    ops.lib().call("wtpse_pack_conv_weights", w.data_ptr(), desc.data_ptr(), 1, packed.data_ptr(), ops.stream_ptr())
    y = torch.empty(B, C, Hc, Hc, device=dev)
    bias = torch.zeros(C, device=dev)
    stats = torch.empty(ops.lib().query("wtpse_conv_stats_blocks", B, Hc, Hc) * C * 2, device=dev)

    def conv():   # as the training step launches it: bias + BatchNorm (sum, sum^2) partials in the epilogue
        ops.lib().call("wtpse_conv_fwd", x.data_ptr(), C, 0, 0, packed.data_ptr(), bias.data_ptr(), 0, 0, 0, y.data_ptr(), 0, C,
                       stats.data_ptr(), B, Hc, Hc, C, 3, 0, 0, ops.stream_ptr())
    ms = time_kernel(conv)
    flops = 2.0 * C * C * 9 * Hc * Hc * B
    out["conv"] = {"kernel": "conv_fwd_k<3,2,5> 64->64 3x3 @%dx%d B=%d (+bias, BN partials)" % (Hc, Hc, B), "ms": ms,
                   "tflops": flops / ms / 1e9, "flop_per_launch": flops}
    z = torch.randn(B, 16, H, H, device=dev)
    L = ops.lib()
    S = L.query("wtpse_wt_split", B, H * H, 0)
    partial = torch.empty(B * S * 256, device=dev)
    bufs = [torch.empty(B * 256, device=dev), torch.empty(B * 120, device=dev), torch.empty(B, device=dev),
            torch.empty(B, device=dev), torch.empty(B, dtype=torch.float64, device=dev), torch.empty(B * 120, device=dev),
            torch.empty(3, device=dev)]
    pb = B // 3

    def wt():
        L.call("wtpse_wt_loss_fwd", z.data_ptr(), B, 16, H * H, 1e-5, 0.0, 3, pb, partial.data_ptr(), *[b.data_ptr() for b in bufs],
               ops.stream_ptr())
    ms = time_kernel(wt)
    nbytes = B * 16 * H * H * 4.0
    out["wt_fwd"] = {"kernel": "wtpse_wt_loss_fwd (gram_partial_k + 3 small) [%d,16,%d,%d]" % (B, H, H), "ms": ms,
                     "gbs": nbytes / ms / 1e6, "bytes_per_launch": nbytes}
    dz = torch.empty_like(z)
    M = torch.randn(B * 256, device=dev) * 1e-3
    bpi = (H * H + 1023) // 1024

    def wtb():
        L.call("wtpse_wt_loss_bwd", z.data_ptr(), B, 16, H * H, 0.0, 3, pb, bufs[0].data_ptr(), bufs[2].data_ptr(), bufs[3].data_ptr(),
               bufs[5].data_ptr(), 0, 0, 0, 1.0, 1.0, 1.0, M.data_ptr(), dz.data_ptr(), 0, ops.stream_ptr())
    ms = time_kernel(wtb)
    out["wt_bwd"] = {"kernel": "wtpse_wt_loss_bwd (gram_bwd_k) [%d,16,%d,%d]" % (B, H, H), "ms": ms,
                     "gbs": 2 * nbytes / ms / 1e6, "bytes_per_launch": 2 * nbytes}
    return out


def cpu_baseline(H, full):
    """The CPU restatement (oracle/, bit-checked against the reference in the build container) timed on this box's
    host cores on a bounded sample: one full iteration at B = 6 (what `--batch-size 8` yields in the reference)."""
    from oracle import wtpse_cpu as O
    from oracle.inputs import make_inputs, make_noise
    from wtpse_hip.synth import default_hparams
    import algorithms
    import shape_networks
    hp = default_hparams(full)
    B, pb = 6, 2
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 16))       # a 1-GPU box is granted 16 host cores; more threads only oversubscribe them
    torch.set_num_threads(cores)
    sds = []
    for i, ctor in enumerate([lambda: algorithms.WT_PSE(3, 1, hp, "cpu", False, per_domain_batch=pb),
                              lambda: shape_networks.ShapeVariationalDist_x(hp, "cpu", 1, 3, pb) if full else None,
                              lambda: algorithms.WT_PSE(3, 1, hp, "cpu", True, per_domain_batch=pb),
                              lambda: shape_networks.ShapeVariationalDist_x(hp, "cpu", 1, 3, pb) if full else None]):
        m = ctor()                       # only used as a weight container (state_dict); no compute on these modules
        sds.append({k: v.detach().clone() for k, v in m.state_dict().items()} if m is not None else {})
    if not full:
        return None
    nets = O.Nets(*sds)
    img, od, oc = make_inputs(1, B, H, H)
    nz = {k: make_noise(10 + j, (B, 1, H, H)) for j, k in enumerate(["a", "b_t", "b_s", "c", "d_t", "d_s"])}
    t0 = time.time()
    O.train_iteration(nets, hp, img, od, oc, nz, pb)
    dt = time.time() - t0
    return {"value": B / dt, "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "1 full A-D iteration (4 fwd + 4 bwd + 4 Adam), B=6, 3x%dx%d, torch CPU fp32, %.1f s" % (H, H, dt)}


def log(msg):
    if int(os.environ.get("RANK", "0")) == 0:
        print("[bench %.1fs] %s" % (time.time() - T0, msg), file=sys.stderr, flush=True)


T0 = time.time()


def measured_traffic():
    """HBM bytes per launch measured with rocprofv3 --pmc (FETCH_SIZE / WRITE_SIZE in separate passes, gfx950
    correction applied: FETCH_SIZE x2 for wide coalesced streams) on `bench.py --kernels-only`; committed under
    profiles/ by tools/pmc_traffic.py.  Returns {} when no measurement has been committed."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.isfile(path):
        with open(path) as f:
            return json.load(f)
    return {}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d (WORLD_SIZE=%d)" % (args.gpus, world)
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
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
        from wtpse_hip.dp import DataParallel
        dp = DataParallel(world, rank, dev, bn_sync=bool(args.bn_sync))

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
    ts = TrainStep(*nets, hp, dp=dp)
    image, target_od, target_oc = make_batch(B, H, H, dev, seed=1 + rank)

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    log("nets built, batch in HBM; priming the per-stream allocator pools")
    for i in range(2):       # part of set-up, like building the nets: the caching allocator's pools (one per stream) reach
        ts.step(image, target_od, target_oc)    # their steady size after two steps; the W warmup steps below are the contract's
    torch.cuda.synchronize()
    log("warmup")
    for i in range(args.warmup):
        ts.step(image, target_od, target_oc)
        torch.cuda.synchronize()
        log("warmup step %d done" % i)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = ts.step(image, target_od, target_oc)
    t_host = time.perf_counter() - t0            # host time to enqueue the steps (no GPU wait inside a step)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    log("timed region: %.3f s for %d steps (host enqueue %.3f s)" % (dt, args.steps, t_host))
    losses = {k: float(v) for k, v in res.items()}
    assert all(v == v for v in losses.values()), "NaN loss: %s" % losses

    if rank == 0:
        ips = world * B * args.steps / dt
        line = {
            "metric": ("training images/sec (%dx%d fundus) — " % (H, H)) + ("full WT-PSE iteration" if full else "seg-net only"),
            "value": ips, "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": ("BASELINE.json configs[2]: full WT-PSE (seg+shape nets + WT loss)" if full else
                                    "BASELINE.json configs[1]: seg-net only") + ", 3x%dx%d, batch %d per GPU" % (H, H, B),
                       "global_batch": B * world, "image": [3, H, H], "parallelism": "dp%d" % world,
                       "bn_sync": bool(args.bn_sync), "step": "calls A-D + 4 backward + 4 Adam (Trainer.py:766-914)"},
            "host_enqueue_ms_per_step": 1e3 * t_host / args.steps,
            "conv_tflops_end_to_end": ips * GFLOP_PER_IMAGE * (H * H / 65536.0) / 1e3 if full else None,
            "losses": losses,
        }
        if world == 1 and not args.no_kernel_roofline:
            log("kernel rooflines")
            kr = kernel_rooflines(B, H, dev)
            tr = measured_traffic() if (B, H) == (32, 256) else {}
            c = kr["conv"]
            line["roofline"] = {"bound": "mfma", "kernel": c["kernel"], "achieved": c["tflops"], "peak": MFMA_F32_PEAK_TF,
                                "unit": "TFLOP/s", "frac": c["tflops"] / MFMA_F32_PEAK_TF, "traffic": tr.get("conv"),
                                "ms_per_launch": c["ms"], "flop_per_launch": c["flop_per_launch"]}
            for k in ("wt_fwd", "wt_bwd"):
                w = kr[k]
                line["roofline_" + k] = {"bound": "hbm", "kernel": w["kernel"], "achieved": w["gbs"], "peak": HBM_PEAK_GBS,
                                         "unit": "GB/s", "frac": w["gbs"] / HBM_PEAK_GBS, "traffic": tr.get(k),
                                         "ms_per_launch": w["ms"], "bytes_per_launch": w["bytes_per_launch"]}
        if world == 1 and not args.no_cpu_baseline:
            log("cpu baseline")
            line["cpu_baseline"] = cpu_baseline(H, full)
            log("done")
        print(json.dumps(line))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
