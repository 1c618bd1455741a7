"""Import the reference's hot-path modules on CPU (BUILD CONTAINER ONLY; TEST INFRASTRUCTURE).

Used by ``oracle/make_golden.py`` alone.  ``/root/reference`` does not exist on the GPU box and
nothing in ``tests/``, ``bench.py`` or ``smoke()`` calls this at run time.

Three shims, none touching arithmetic:
  1. ``torchfile``     — imported at algorithms.py:11, used only by the dead ``pytorch_lua_wrapper``.
  2. ``torchvision``   — imported at shape_networks.py:6-7, never used.
  3. ``Tensor.cuda``   — identity, for the hard-coded ``.cuda()`` sites
                         (algorithms.py:1162-1164,1296,1305; shape_networks.py:449-455,581,590).
Sampling is made reproducible by replaying fixture noise through ``torch.randn_like`` /
``torch.normal`` (algorithms.py:1072; shape_networks.py:507).
"""
import contextlib
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("WTPSE_REFERENCE_ROOT", "/root/reference")


def available() -> bool:
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "algorithms.py"))


def load():
    """Returns (hparams_registry, algorithms, shape_networks) of the reference."""
    import matplotlib
    matplotlib.use("Agg")
    import torch

    if "torchfile" not in sys.modules:
        sys.modules["torchfile"] = types.ModuleType("torchfile")
    if "torchvision" not in sys.modules:
        tv = types.ModuleType("torchvision")
        tvm = types.ModuleType("torchvision.models")
        tv.models = tvm
        sys.modules["torchvision"] = tv
        sys.modules["torchvision.models"] = tvm
    torch.Tensor.cuda = lambda self, *a, **k: self
    # the build's own drop-in modules carry the same module names; make sure the reference wins here
    for m in ("algorithms", "shape_networks", "hparams_registry"):
        sys.modules.pop(m, None)
    sys.path.insert(0, REFERENCE_ROOT)
    try:
        import hparams_registry
        import algorithms
        import shape_networks
    finally:
        sys.path.remove(REFERENCE_ROOT)
    assert os.path.dirname(os.path.abspath(algorithms.__file__)) == os.path.abspath(REFERENCE_ROOT)
    return hparams_registry, algorithms, shape_networks


def load_transforms():
    """The reference's custom_transforms module.  Its top-level imports pull in cv2 (not installed) and scipy/matplotlib
    names the five classes of the training pipeline never touch: cv2 is shimmed with an empty module."""
    import matplotlib
    matplotlib.use("Agg")
    if "cv2" not in sys.modules:
        sys.modules["cv2"] = types.ModuleType("cv2")
    sys.modules.pop("custom_transforms", None)
    sys.path.insert(0, REFERENCE_ROOT)
    try:
        import custom_transforms
    finally:
        sys.path.remove(REFERENCE_ROOT)
    assert os.path.dirname(os.path.abspath(custom_transforms.__file__)) == os.path.abspath(REFERENCE_ROOT)
    return custom_transforms


@contextlib.contextmanager
def replay_noise(queue):
    """Within the block, ``torch.randn_like(t)`` pops the next fixture tensor and
    ``torch.normal(mu, std)`` returns ``(mu + std * next).detach()`` (a sample has no grad_fn)."""
    import torch
    queue = list(queue)
    orig_randn_like, orig_normal = torch.randn_like, torch.normal

    def randn_like(t, **kw):
        e = queue.pop(0)
        assert e.shape == t.shape, (e.shape, t.shape)
        return e.clone()

    def normal(mu, std, **kw):
        e = queue.pop(0)
        assert e.shape == mu.shape
        return (mu + std * e).detach()

    torch.randn_like, torch.normal = randn_like, normal
    try:
        yield
    finally:
        torch.randn_like, torch.normal = orig_randn_like, orig_normal
