"""Worker processes for the data-parallel tests (spawned by test_dp_host.py / test_dp_gpu.py)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "wt-pse-code_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def _init(rank, world, port, backend="gloo", device=None):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # before anything initialises HSA
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)


def host_logic(rank, world, port, out_dir):
    """CPU + gloo: sharding maps, domain-major all-gather, flat-gradient averaging."""
    import torch
    import torch.distributed as dist
    from wtpse_hip.dp import DataParallel, local_rows
    _init(rank, world, port)
    dp = DataParallel(world, rank, torch.device("cpu"), bn_sync=True)
    D, n_g = 3, 4
    n_l = n_g // world
    glob = torch.arange(D * n_g * 5, dtype=torch.float32).reshape(D * n_g, 5)          # row r = [5r .. 5r+4]
    mine = local_rows(n_g, D, world, rank)
    local = torch.cat([glob[mine], torch.full((1, 5), -1.0)])                            # + one unused tail row
    gathered = dp.gather_domain_major(local, n_l, D)
    ok_gather = torch.equal(gathered, glob)
    g = torch.full((1000,), float(rank + 1))
    dp.allreduce_grads(None, g)
    ok_mean = torch.allclose(g, torch.full((1000,), (world + 1) / 2.0))
    s = dp.allreduce_sum(torch.tensor([1.0, 2.0]) * (rank + 1))
    ok_sum = torch.allclose(s, torch.tensor([1.0, 2.0]) * world * (world + 1) / 2)
    torch.save({"gather": ok_gather, "mean": ok_mean, "sum": ok_sum, "rows": mine}, os.path.join(out_dir, "r%d.pt" % rank))
    dist.destroy_process_group()


def host_buckets(rank, world, port, out_dir):
    """CPU + gloo: the bucketed, overlapped gradient exchange (pieces announced out of order, gaps left for the final call)
    equals the single all-reduce of the whole buffer."""
    import torch
    import torch.distributed as dist
    from wtpse_hip.dp import DataParallel
    _init(rank, world, port)
    n = 10007
    base = (torch.arange(n, dtype=torch.float32) % 97.0) * world                 # sums and means stay exact in fp32
    mine = base * (rank + 1)
    want = base * (world + 1) / 2.0
    owner = object()
    res = {}
    for overlap in (True, False):
        dp = DataParallel(world, rank, torch.device("cpu"), bn_sync=False, overlap=overlap)
        g = mine.clone()
        dp.bucket_ready(owner, g, 9000, n)            # the heads, first
        dp.bucket_ready(owner, g, 3000, 7000)         # the decoder
        dp.bucket_ready(owner, g, 7000, 7000)         # an empty range
        assert (id(owner) in dp._pieces) == overlap
        dp.allreduce_grads(owner, g)                  # [0, 3000) and [7000, 9000) are reduced here
        res["overlap" if overlap else "single"] = bool(torch.equal(g, want))
        assert id(owner) not in dp._pieces
    torch.save(res, os.path.join(out_dir, "r%d.pt" % rank))
    dist.destroy_process_group()


def gpu_exact(rank, world, port, out_dir, B_g, pb_g, H):
    """GPU (all ranks share cuda:0, gloo transport): calls A and B in exact mode; dumps the averaged flat gradients."""
    import torch
    import torch.distributed as dist
    from wtpse_hip import ops
    from wtpse_hip.dp import DataParallel, local_rows
    from oracle.inputs import make_inputs, make_noise
    from test_parity_gpu import build_nets, HP
    _init(rank, world, port)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dp = DataParallel(world, rank, dev, bn_sync=True)
    n_l = pb_g // world
    rows = local_rows(pb_g, 3, world, rank)
    img, od, _ = make_inputs(600, B_g, H, H)
    eps = make_noise(700, (B_g, 1, H, H))
    main, shape, _, _ = build_nets(n_l)
    for n in (main, shape):
        n.train()
        n.ensure_ready(repack=True)
        object.__setattr__(n, "_dp", dp)
    x, m = img[rows].to(dev).contiguous(), od[rows].to(dev).contiguous()
    main.set_noise([eps[rows]])
    res, tape = main._forward_update(x, m, x, want_tape=True)
    out, _, scal = res
    d_out = ops.bce_sigmoid_bwd(out, m)
    main._backward_update(tape, d_out, None, None, w_ins=1.0, w_dom=1.0)
    g_main = main.flat_grads().detach().cpu().clone()
    scal_main = scal.detach().cpu().clone()
    s2, tape2 = shape._forward_update(main, x, m, want_tape=True)
    shape._backward_update(tape2, None, None, None, None)
    g_shape = shape.flat_grads().detach().cpu().clone()
    # sampling noise from the global Philox stream (no injection): rows must equal the 1-GPU draw
    main.seed_noise(99)
    nz = main.next_noise((len(rows), 1, H, H)).cpu()
    torch.save({"g_main": g_main, "g_shape": g_shape, "scal_main": scal_main, "scal_shape": s2.detach().cpu(),
                "out": out.detach().cpu(), "rows": rows, "noise": nz,
                "bufs": {k: v.detach().cpu().clone() for k, v in main.named_buffers()}}, os.path.join(out_dir, "r%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


def gpu_ddp_overlap(rank, world, port, out_dir, B_g, pb_g, H):
    """GPU (all ranks share cuda:0, gloo transport), throughput (DDP) mode: one TrainStep with the gradient exchange started INSIDE
    the backward (dp.bucket_ready from an issue stream behind the weight-gradient streams), one with the single exchange between
    backward and Adam, one captured as HIP graphs (graph=True: the collectives must stay between the captured stretches).  All three
    must leave bitwise the same parameters: the exchange adds the same numbers in the same order whatever its timing."""
    import torch
    import torch.distributed as dist
    from wtpse_hip.dp import DataParallel, local_rows
    from wtpse_hip.step import TrainStep
    from oracle.inputs import make_inputs
    from test_parity_gpu import build_nets, HP
    _init(rank, world, port)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    n_l = pb_g // world
    rows = local_rows(pb_g, 3, world, rank)
    img, od, oc = make_inputs(600, B_g, H, H)
    x, m, c = img[rows].to(dev).contiguous(), od[rows].to(dev).contiguous(), oc[rows].to(dev).contiguous()
    res = {}
    params = {}
    for tag, overlap, graph in (("overlap", True, False), ("single", False, False), ("graph", True, True)):
        nets = build_nets(n_l)
        for n in nets:
            n.seed_noise(55)
        dpm = DataParallel(world, rank, dev, bn_sync=False, overlap=overlap)
        ts = TrainStep(*nets, HP, dp=dpm, graph=graph)
        pieces_seen = [0]
        if overlap and not graph:
            orig = dpm.bucket_ready

            def counting(net, gflat, lo, hi, streams=(), _o=orig):
                pieces_seen[0] += 1
                return _o(net, gflat, lo, hi, streams)
            dpm.bucket_ready = counting
        for _ in range(2):
            lo = ts.step(x, m, c)
        torch.cuda.synchronize()
        res["losses_" + tag] = {k: float(v) for k, v in lo.items()}
        res["pieces_" + tag] = pieces_seen[0]
        params[tag] = [n.flat_params().detach().cpu().clone() for n in nets]
        ts.close()
    res["params"] = params["overlap"]
    res["overlap_equals_single"] = all(bool(torch.equal(a, b)) for a, b in zip(params["overlap"], params["single"]))
    res["graph_equals_single"] = all(bool(torch.equal(a, b)) for a, b in zip(params["graph"], params["single"]))
    torch.save(res, os.path.join(out_dir, "r%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


def nccl_rccl(rank, world, port, out_dir, B_g, pb_g, H):
    """RCCL leg (backend "nccl"), one GPU per rank: all-reduce AVG, broadcast_params, the exact-mode calls A/B of
    gpu_exact, and one TrainStep in each mode.  world = 1 runs it on a single GPU (the collectives still go through
    RCCL); world = 2 needs two GPUs."""
    import torch
    import torch.distributed as dist
    from wtpse_hip import ops
    from wtpse_hip.dp import DataParallel, local_rows
    from wtpse_hip.step import TrainStep
    from oracle.filler import fill_state_dict
    from oracle.inputs import make_inputs, make_noise
    from test_parity_gpu import build_nets, HP
    dev = torch.device("cuda", rank % max(torch.cuda.device_count(), 1))
    torch.cuda.set_device(dev)
    _init(rank, world, port, "nccl", dev)
    res = {}
    dp = DataParallel(world, rank, dev, bn_sync=True)
    assert dp._avg_native
    g = torch.full((1000,), float(rank + 1), device=dev)
    dp.allreduce_grads(None, g)
    res["mean"] = bool(torch.allclose(g.cpu(), torch.full((1000,), (world + 1) / 2.0)))
    s = dp.allreduce_sum(torch.tensor([1.0, 2.0], device=dev) * (rank + 1))
    res["sum"] = bool(torch.allclose(s.cpu(), torch.tensor([1.0, 2.0]) * world * (world + 1) / 2))
    n_l = pb_g // world
    rows = local_rows(pb_g, 3, world, rank)
    img, od, oc = make_inputs(600, B_g, H, H)
    eps = make_noise(700, (B_g, 1, H, H))
    # exact mode, calls A and B (as gpu_exact), gradient exchange over RCCL
    main, shape, main_oc, shape_oc = build_nets(n_l)
    if rank > 0:                                  # broadcast_params must overwrite these with rank 0's
        for i, n in enumerate((main, shape, main_oc, shape_oc)):
            fill_state_dict(n, 999 + i)
    for n in (main, shape, main_oc, shape_oc):
        n.train()
        n.ensure_ready(repack=True)
        object.__setattr__(n, "_dp", dp)
    dp.broadcast_params([main, shape, main_oc, shape_oc])
    ref_main, _, _, _ = build_nets(n_l)
    res["bcast"] = bool(torch.equal(main.flat_params().cpu(), ref_main.to(dev).flat_params().cpu()))
    x, m = img[rows].to(dev).contiguous(), od[rows].to(dev).contiguous()
    main.set_noise([eps[rows]])
    r, tape = main._forward_update(x, m, x, want_tape=True)
    out, _, scal = r
    main._backward_update(tape, ops.bce_sigmoid_bwd(out, m), None, None, w_ins=1.0, w_dom=1.0)
    res["g_main"] = main.flat_grads().detach().cpu().clone()
    res["scal_main"] = scal.detach().cpu().clone()
    res["out"] = out.detach().cpu()
    s2, tape2 = shape._forward_update(main, x, m, want_tape=True)
    shape._backward_update(tape2, None, None, None, None)
    res["g_shape"] = shape.flat_grads().detach().cpu().clone()
    res["rows"] = rows
    # one full training step per mode through the harness (Adam on the averaged gradients)
    for mode in (True, False):
        nets = build_nets(n_l)
        dpm = DataParallel(world, rank, dev, bn_sync=mode)
        ts = TrainStep(*nets, HP, dp=dpm)
        lo = ts.step(img[rows].to(dev), od[rows].to(dev), oc[rows].to(dev))
        torch.cuda.synchronize()
        tag = "exact" if mode else "ddp"
        res["losses_" + tag] = {k: float(v) for k, v in lo.items()}
        res["params_" + tag] = [n.flat_params().detach().cpu().clone() for n in nets]
    # the same throughput-mode step replayed from launch plans: the gradient all-reduces run between the recorded stretches
    nets_e, nets_p = build_nets(n_l), build_nets(n_l)
    for nn_ in nets_e + nets_p:
        nn_.seed_noise(77)
    ts_e = TrainStep(*nets_e, HP, dp=DataParallel(world, rank, dev, bn_sync=False))
    ts_p = TrainStep(*nets_p, HP, dp=DataParallel(world, rank, dev, bn_sync=False), graph="plan")
    for _ in range(2):
        ts_e.step(img[rows].to(dev), od[rows].to(dev), oc[rows].to(dev))
        lp = ts_p.step(img[rows].to(dev), od[rows].to(dev), oc[rows].to(dev))
    torch.cuda.synchronize()
    res["plan_equals_eager"] = all(bool(torch.equal(a.flat_params(), b.flat_params())) for a, b in zip(nets_e, nets_p))
    res["plan_segments"] = len(ts_p._graphs)
    res["losses_plan"] = {k: float(v) for k, v in lp.items()}
    torch.save(res, os.path.join(out_dir, "r%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    fn, rank, world, port, out_dir = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
    if fn == "host":
        host_logic(rank, world, port, out_dir)
    elif fn == "buckets":
        host_buckets(rank, world, port, out_dir)
    elif fn == "nccl":
        nccl_rccl(rank, world, port, out_dir, *[int(a) for a in sys.argv[6:9]])
    elif fn == "ddp_overlap":
        gpu_ddp_overlap(rank, world, port, out_dir, *[int(a) for a in sys.argv[6:9]])
    else:
        gpu_exact(rank, world, port, out_dir, *[int(a) for a in sys.argv[6:9]])
