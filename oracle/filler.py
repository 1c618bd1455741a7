"""Deterministic, name-keyed weight filler (test infrastructure; see oracle/__init__.py).

The reference's networks are 25 MB (WT_PSE) and 13 MB (shape net) of fp32
weights, far too much to commit as fixtures.  Instead both sides of every parity
check fill their ``state_dict`` with the same values, generated per *key name*
from a seed: the golden generator applies it to the imported reference modules,
the tests apply it to the HIP modules and to the CPU oracle.  Since the drop-in
boundary promises the reference's ``state_dict`` key names, equal
names give equal weights with nothing but a seed in the fixture.

Value ranges are chosen so activations stay O(1) through ~25 conv+BN layers and
BatchNorm running statistics are non-trivial (eval-mode ``predict`` would
otherwise only ever see mean 0 / var 1).
"""
import zlib

import numpy as np
import torch


def _rng(name: str, seed: int) -> np.random.RandomState:
    return np.random.RandomState((zlib.crc32(name.encode()) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFF)


def fill_value(name: str, shape, seed: int) -> np.ndarray:
    """Value of state_dict entry `name` (shape `shape`) under `seed`."""
    r = _rng(name, seed)
    shape = tuple(int(s) for s in shape)
    leaf = name.rsplit(".", 1)[-1]
    if leaf == "num_batches_tracked":
        return np.zeros(shape, dtype=np.int64)
    if leaf == "running_mean":
        return (r.uniform(-0.2, 0.2, size=shape)).astype(np.float32)
    if leaf == "running_var":
        return (r.uniform(0.6, 1.4, size=shape)).astype(np.float32)
    if len(shape) == 4:  # conv weight [Cout, Cin, kh, kw]
        fan_in = shape[1] * shape[2] * shape[3]
        bound = np.sqrt(3.0 / fan_in)
        return r.uniform(-bound, bound, size=shape).astype(np.float32)
    if len(shape) == 1:
        # BatchNorm affine weight lives next to a running_mean; conv bias does not.
        # Both are 1-D, so the caller disambiguates through `is_bn_weight`.
        return r.uniform(-0.1, 0.1, size=shape).astype(np.float32)
    return r.uniform(-0.1, 0.1, size=shape).astype(np.float32)


def fill_state_dict(module: torch.nn.Module, seed: int) -> None:
    """Overwrite every parameter and buffer of `module` in place."""
    sd = module.state_dict()
    bn_prefixes = {k[: -len(".running_mean")] for k in sd if k.endswith(".running_mean")}
    with torch.no_grad():
        for name, t in sd.items():
            v = fill_value(name, t.shape, seed)
            prefix, _, leaf = name.rpartition(".")
            if leaf == "weight" and prefix in bn_prefixes:
                v = (1.0 + v * 2.0).astype(np.float32)  # gamma in [0.8, 1.2]
            t.copy_(torch.from_numpy(np.asarray(v)).to(t.dtype).reshape(t.shape))


def filled_state(template: dict, seed: int) -> dict:
    """Same as fill_state_dict but for a plain {name: tensor} template; returns new CPU tensors."""
    bn_prefixes = {k[: -len(".running_mean")] for k in template if k.endswith(".running_mean")}
    out = {}
    for name, t in template.items():
        v = fill_value(name, t.shape, seed)
        prefix, _, leaf = name.rpartition(".")
        if leaf == "weight" and prefix in bn_prefixes:
            v = (1.0 + v * 2.0).astype(np.float32)
        out[name] = torch.from_numpy(np.asarray(v)).to(t.dtype).reshape(t.shape).clone()
    return out
