"""Checkpoint round trip as the reference does it: save the dict of four state_dicts (Trainer.py:282-288), load it with the
filtered load_state_dict of test_visulization.py:132-193 — into fresh networks AND into networks a TrainStep already owns (whose
packed weight copies the harness vouches for: the load must invalidate them) — and predict(): bitwise the source networks' output."""
import io

import pytest
import torch

from test_parity_gpu import build_nets, HP, DEV
from oracle.filler import fill_state_dict
from oracle.inputs import make_inputs

pytestmark = pytest.mark.gpu


def _save(nets):
    """Trainer.py:282-288."""
    buf = io.BytesIO()
    torch.save({'model': nets[0].state_dict(), 'model_shape': nets[1].state_dict(), 'model_oc': nets[2].state_dict(),
                'model_oc_shape': nets[3].state_dict()}, buf)
    buf.seek(0)
    return buf


def _load_filtered(net, pretrained_dict):
    """test_visulization.py:132-140 (and :150-157, :174-181, :188-195)."""
    model_dict = net.state_dict()
    pretrained_dict = {k: v for k, v in pretrained_dict.items() if k in model_dict}
    model_dict.update(pretrained_dict)
    net.load_state_dict(model_dict)


def _predict(nets, img):
    for n in nets:
        n.eval()
    with torch.no_grad():
        pred, att = nets[0].predict(nets[1], img)
        od = (torch.sigmoid(pred) > 0.75).float()
        roi = (img + 1) * od - 1                                  # Trainer.py:174-178
        pred_oc, _ = nets[2].predict(nets[3], torch.stack((roi, roi), 0))
    return pred, att, pred_oc


def test_checkpoint_round_trip():
    from wtpse_hip.step import TrainStep
    B, pb, H = 3, 1, 64
    img, od, oc = make_inputs(41, B, H, H)
    img, od, oc = img.to(DEV), od.to(DEV), oc.to(DEV)
    # source networks: two training steps away from the filler (BatchNorm running statistics and num_batches_tracked moved too)
    src = build_nets(pb)
    ts = TrainStep(src[0], src[1], src[2], src[3], HP)
    for n in src:
        n.seed_noise(5)
    for _ in range(2):
        ts.step(img, od, oc)
    torch.cuda.synchronize()
    want = _predict(src, img)
    ckpt = torch.load(_save(src), map_location="cpu", weights_only=True)
    assert set(ckpt) == {'model', 'model_shape', 'model_oc', 'model_oc_shape'}

    # (1) fresh networks, differently initialised
    fresh = build_nets(pb)
    for i, n in enumerate(fresh):
        fill_state_dict(n, 4321 + i)
    for n, key in zip(fresh, ('model', 'model_shape', 'model_oc', 'model_oc_shape')):
        _load_filtered(n, ckpt[key])
    got = _predict(fresh, img)
    for a, b in zip(got, want):
        assert torch.equal(a, b)

    # (2) networks already owned by a TrainStep that has stepped (`_packed_valid`: the harness vouches for the packed copies)
    live = build_nets(pb)
    for i, n in enumerate(live):
        fill_state_dict(n, 999 + i)
    ts2 = TrainStep(live[0], live[1], live[2], live[3], HP)
    ts2.step(img, od, oc)
    torch.cuda.synchronize()
    assert all(n._packed_valid for n in live)
    for n, key in zip(live, ('model', 'model_shape', 'model_oc', 'model_oc_shape')):
        _load_filtered(n, ckpt[key])
        assert n._packed_version < 0, "load_state_dict must invalidate the packed weights"
    got = _predict(live, img)
    for a, b in zip(got, want):
        assert torch.equal(a, b)
    # ... and a training forward from the loaded state sees the loaded weights too (call A's losses: everything later in the step
    # follows an Adam update, whose moments are not part of the reference's checkpoint)
    for n in src + live:
        n.train()
        n.seed_noise(9)
    la, lb = ts.step(img, od, oc), ts2.step(img, od, oc)
    torch.cuda.synchronize()
    for k in ("seg_od", "ins_od", "dom_od"):
        assert torch.equal(la[k], lb[k]), k

    # (3) a sub-module load (the reference never does it, the hazard is the same): only the DeepWT of the student
    sub = {k: v for k, v in ckpt['model_shape'].items() if k.startswith("wt_model.")}
    other = build_nets(pb)
    other[1].eval()
    other[1].ensure_ready(repack=True)
    object.__setattr__(other[1], "_packed_valid", True)
    other[1].wt_model.load_state_dict({k[len("wt_model."):]: v for k, v in sub.items()})
    assert other[1]._packed_version < 0
