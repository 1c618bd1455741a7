"""Data-parallel host logic on CPU (gloo, world_size 2): row sharding, the domain-major all-gather used by the global
MMD, the flat-gradient average and scalar sums.  No kernels are involved (the C-ABI library is not even loaded)."""
import os
import socket
import subprocess
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def run_workers(fn, world, out_dir, extra=(), timeout=300):
    port = free_port()
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "dp_worker.py"), fn, str(r), str(world), str(port), str(out_dir)]
                              + [str(e) for e in extra]) for r in range(world)]
    codes = [p.wait(timeout=timeout) for p in procs]
    assert codes == [0] * world, codes
    return [torch.load(os.path.join(out_dir, "r%d.pt" % r), weights_only=False) for r in range(world)]


def test_sharding_maps():
    from wtpse_hip.dp import local_rows, gathered_to_global_index
    for world in (1, 2, 4):
        D, n_g = 3, 8
        allrows = sorted(r for g in range(world) for r in local_rows(n_g, D, world, g))
        assert allrows == list(range(D * n_g))                       # a partition of the global batch
        for g in range(world):
            rows = local_rows(n_g, D, world, g)
            n_l = n_g // world
            assert [r // n_g for r in rows] == [d for d in range(D) for _ in range(n_l)]   # local batch is domain-major
        idx = gathered_to_global_index(n_g // world, D, world)
        stacked = [r for g in range(world) for r in local_rows(n_g, D, world, g)]
        assert [stacked[i] for i in idx.tolist()] == list(range(D * n_g))


def test_gloo_world2(tmp_path):
    res = run_workers("host", 2, tmp_path)
    for r in res:
        assert r["gather"] and r["mean"] and r["sum"], r
    assert sorted(res[0]["rows"] + res[1]["rows"]) == list(range(12))


def test_gloo_bucketed_exchange_world2_and_3(tmp_path):
    """The overlapped gradient exchange (ranges announced while the backward is still running, dp.bucket_ready) gives the same
    averages as the single all-reduce, with 2 and with 3 ranks."""
    for world in (2, 3):
        d = tmp_path / ("w%d" % world)
        d.mkdir()
        for r in run_workers("buckets", world, d):
            assert r == {"overlap": True, "single": True}, (world, r)
