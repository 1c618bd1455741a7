"""BUILD CONTAINER / any CPU box (TEST INFRASTRUCTURE): the CPU oracle's backward at the benchmark's geometries, evaluated OFFLINE
-> tests/golden/grads_<tag>.npz, so that the GPU suite can hold the HIP gradients against them without minutes of host time per run.

For call A (WT_PSE.update + BCE + WT terms) and call B (ShapeVariationalDist_x.update) of tests/test_parity_gpu.py::test_gradients_calibrated,
same seeded inputs / weights / noise:  per parameter tensor the fingerprint of the fp64 gradient (oracle/sketch.py) and the EXACT relative
distances |g32_i - g64| / |g64| of three fp32 evaluations (the inputs as given and two copies perturbed by 1e-6 / 3e-6: the yardstick).
    python oracle/make_golden_grads.py b32      # B=32, pb=10, 256x256 (BASELINE.json configs[2]); ~30 GB, ~20 min on 8 cores
    python oracle/make_golden_grads.py s512     # B=3,  pb=1,  512x512 (configs[4]'s per-image geometry)
Only numbers are written; the oracle is this repository's own CPU restatement (pinned to the reference by tests/test_oracle_golden.py)."""
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path[:0] = [ROOT, os.path.join(ROOT, "wt-pse-code_amd"), os.path.join(ROOT, "tests")]
from oracle import wtpse_cpu as O  # noqa: E402
from oracle import sketch  # noqa: E402
from oracle.filler import fill_state_dict  # noqa: E402
from oracle.inputs import make_inputs, make_noise  # noqa: E402

CASES = {"b32": (32, 10, 256), "s512": (3, 1, 512)}
SEED_W = 1234
HP = dict(O.DEFAULT_HPARAMS)


def state_dicts(pb):
    """The seeded weights of test_parity_gpu.build_nets(), on the CPU (the drop-in modules serve as key containers only)."""
    import algorithms
    import shape_networks
    main = algorithms.WT_PSE(n_channels=3, n_classes=1, hparams=HP, device="cpu", two_step=False, per_domain_batch=pb, source_domain_num=3)
    shape = shape_networks.ShapeVariationalDist_x(HP, "cpu", n_classes=1, number_source_domain=3, batch_size=pb)
    fill_state_dict(main, SEED_W)
    fill_state_dict(shape, SEED_W + 3)
    return ({k: v.detach().clone() for k, v in main.state_dict().items()}, {k: v.detach().clone() for k, v in shape.state_dict().items()})


def perturbed(image, probes):           # = tests/test_parity_gpu.perturbed
    gen = torch.Generator().manual_seed(77)
    for i in range(probes):
        yield image * (1 + (1e-6, 3e-6, 1e-5)[i % 3] * torch.randn(image.shape, generator=gen)).to(image.dtype)


def grads(fn, sds, dtype):              # = tests/test_parity_gpu.oracle_grads
    cast = [{k: (v.detach().clone().to(dtype).requires_grad_(not O.is_buffer(k)) if v.is_floating_point() else v.clone())
             for k, v in sd.items()} for sd in sds]
    fn(*cast).backward()
    return {k: cast[0][k].grad.double() for k in cast[0] if not O.is_buffer(k) and cast[0][k].grad is not None}


def main():
    tag = sys.argv[1]
    B, pb, H = CASES[tag]
    torch.set_num_threads(int(os.environ.get("OMP_NUM_THREADS", os.cpu_count() or 8)))
    img, od, _ = make_inputs(600, B, H, H)
    eps = make_noise(700, (B, 1, H, H))
    sd_m, sd_s = state_dicts(pb)

    def loss_a(sd, image=img):
        dt = sd["outc.0.weight"].dtype
        o, _, _, i2, d2 = O.wt_pse_update(sd, HP, image.to(dt), od.to(dt), image.to(dt), True, eps.to(dt), 3, pb)
        return O.seg_loss_od(o, od.to(dt)) + i2 + d2

    def loss_b(sds, sdm, image=img):
        dt = sds["mu_prior.0.weight"].dtype
        r = O.shape_update(sds, sdm, HP, image.to(dt), od.to(dt), image.to(dt), True, eps.to(dt), eps.to(dt), pb)
        return r[0] + r[1] + r[4]
    out = {"meta": np.array([B, pb, H, sketch.K, sketch.SMALL])}
    for call, fn, sds in (("A", loss_a, [sd_m]), ("B", loss_b, [sd_s, sd_m])):
        t0 = time.time()
        g64 = grads(fn, sds, torch.float64)
        print(call, "fp64 %.0f s" % (time.time() - t0), flush=True)
        names = sorted(g64)
        out[call + "_names"] = np.array(names)
        for i, k in enumerate(names):
            fp = sketch.fingerprint(g64[k], 7000 + i)
            out["%s_%d_fp" % (call, i)] = fp["data"]
            out["%s_%d_n2" % (call, i)] = np.array([fp["n"], fp["norm2"]])
        dist = np.zeros((len(names), 3))
        runs = [img] + list(perturbed(img, 2))
        for j, q in enumerate(runs):
            t0 = time.time()
            g32 = grads((lambda *a, q=q: fn(*a, q)), sds, torch.float32)
            for i, k in enumerate(names):
                dist[i, j] = float((g32[k] - g64[k]).pow(2).sum())
            print(call, "fp32 draw %d %.0f s" % (j, time.time() - t0), flush=True)
            del g32
        out[call + "_yard2"] = dist                  # squared distances |g32_j - g64|^2 per tensor
        del g64
    dst = os.path.join(ROOT, "tests", "golden", "grads_%s.npz" % tag)
    np.savez_compressed(dst, **out)
    print("wrote", dst, os.path.getsize(dst), "bytes")


if __name__ == "__main__":
    main()
