"""SURVEY.md §8f row 2 (-m gpu): the validation front half (predict -> ROI -> predict OC -> bilinear resize -> threshold
-> largest component + hole fill -> Dice) against the CPU oracle on the same inputs."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import wtpse_cpu as O
from oracle.inputs import make_inputs

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("size", [(40, 56), (96, 96), (64, 64), (17, 130)])
def test_resize_bilinear(size):
    from wtpse_hip import ops
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 3, 64, 48, generator=g)
    ref = F.interpolate(x, size=size, mode="bilinear")
    got = ops.resize_bilinear(x.to(DEV), size).cpu()
    assert torch.allclose(got, ref, rtol=1e-5, atol=1e-6), float((got - ref).abs().max())


def test_validation_front_half_vs_oracle():
    from test_parity_gpu import build_nets, HP
    from wtpse_hip import validate as V
    B, pb, H = 4, 1, 64
    img, od, oc = make_inputs(31, B, H, H)
    label_size = (80, 72)
    lod = (F.interpolate(od, size=label_size) > 0.5).float()
    loc = (F.interpolate(oc, size=label_size) > 0.5).float()
    nets = build_nets(pb)
    sds = [{k: v.detach().cpu().clone() for k, v in n.state_dict().items()} for n in nets]
    for n in nets:
        n.eval()
    pred, pred_oc = V.predict_pair(*nets, img.to(DEV), label_size)
    with torch.no_grad():
        ref, ref_oc = O.validate_predict(sds[0], sds[1], sds[2], sds[3], HP, img, label_size)
    assert float((pred.cpu() - ref).abs().max()) < 1e-4
    assert float((pred_oc.cpu() - ref_oc).abs().max()) < 1e-4
    from oracle import postprocess_cpu as P
    for i in range(B):
        # product side: validate.postprocess / validate.dice on the HIP logits; checker side: the oracle's own post-processing
        # (flood-fill restatement of utils.py:267-329) and Dice (metrics.py:68-97) on the oracle's logits
        d_hip = V.dice(V.postprocess(pred[i])[0], lod[i, 0].numpy())
        d_ref = O.dice_coefficient(P.postprocessing(ref[i])[0], lod[i, 0].numpy())
        assert abs(d_hip - d_ref) <= 1e-4, (i, d_hip, d_ref)
        c_hip = V.dice(V.postprocess(pred_oc[i])[0], loc[i, 0].numpy())
        c_ref = O.dice_coefficient(P.postprocessing(ref_oc[i])[0], loc[i, 0].numpy())
        assert abs(c_hip - c_ref) <= 1e-4, (i, c_hip, c_ref)
    cup, disc = V.validate(*nets, [(img.to(DEV), lod, loc)])
    assert 0.0 <= cup <= 1.0 and 0.0 <= disc <= 1.0
    assert all(not n.training for n in nets)


def test_postprocess_keeps_largest_component_and_fills_holes():
    from wtpse_hip import validate as V
    m = np.zeros((12, 12), np.uint8)
    m[1:6, 1:6] = 1; m[3, 3] = 0            # 5x5 blob with a hole (area 24)
    m[8:10, 8:10] = 1                       # smaller blob
    m[6, 6] = 1                             # touches the big blob diagonally (8-connectivity) -> same component
    out = V.largest_fillhole(m)
    assert out[3, 3] and out[6, 6] and not out[8:10, 8:10].any()
    assert V.dice(np.zeros((4, 4)), np.zeros((4, 4))) == 1.0
