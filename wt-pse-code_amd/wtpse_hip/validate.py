"""Validation front half: counterpart of Trainer.validate (reference Trainer.py:137-256) — SURVEY.md §8f row 2.

Device part (HIP kernels): OD predict -> od_pred = sigmoid > 0.75 -> ROI -> OC predict on the stacked (roi, roi) input
-> predictions_oc * od_pred -> bilinear resize of both logit maps to the label size (Trainer.py:170-209).
Host part (as in the reference, which does it in numpy/scipy/skimage on the CPU): threshold 0.75 on the sigmoid,
largest connected component + hole filling (utils.py:267-329), Dice = (2|A&B| + 1)/(|A| + |B| + 1) (metrics.py:68-97).
ASD / HD95 come from the un-vendored `medpy` in the reference and are not reproduced (parity unpinned).
"""
import numpy as np
import torch

from . import ops


def predict_pair(model, model_shape, model_oc, model_shape_oc, data, label_size=None):
    """-> (logits_od, logits_oc*od_pred), each [B,1,h,w] resized to `label_size` when given.  Does not modify `data`."""
    with torch.no_grad():
        pred, _ = model.predict(model_shape, data)
        roi, od_pred = ops.roi(data.contiguous(), pred)
        pred_oc, _ = model_oc.predict(model_shape_oc, torch.stack((roi, roi), 0))
        # predictions_oc * od_pred: od_pred is {0,1}; reuse the ReLU-mask kernel (out = ref > 0 ? v : 0)
        pred_oc = ops.relu_mask(pred_oc, od_pred)
        if label_size is not None and tuple(label_size) != tuple(pred.shape[2:]):
            pred = ops.resize_bilinear(pred, label_size)
            pred_oc = ops.resize_bilinear(pred_oc, label_size)
    return pred, pred_oc


def largest_fillhole(binary):
    """utils.get_largest_fillhole (utils.py:267-276): keep the largest 8-connected component (skimage.measure.label's
    default connectivity in 2-D), then fill holes."""
    from scipy import ndimage
    binary = np.asarray(binary).copy()
    lab, n = ndimage.label(binary, structure=np.ones((3, 3), dtype=int))
    if n:
        areas = ndimage.sum(binary > 0, lab, index=np.arange(1, n + 1))
        binary[lab != int(np.argmax(areas)) + 1] = 0
    return ndimage.binary_fill_holes(binary.astype(int))


def postprocess(logits, threshold=0.75):
    """utils.postprocessing, label != None branch (utils.py:306-323): [1,h,w] logits -> uint8 mask [1,h,w]."""
    prob = torch.sigmoid(logits).detach().cpu().numpy()
    mask = (prob > threshold).astype(np.uint8)
    mask[0] = largest_fillhole(mask[0]).astype(np.uint8)
    return mask


def dice(seg, gt):
    seg = np.asarray(seg, dtype=np.bool_)
    gt = np.asarray(gt, dtype=np.bool_)
    inter = float(np.logical_and(seg, gt).sum())
    return (2 * inter + 1.0) / (1.0 + float(seg.sum()) + float(gt.sum()))


def validate(model, model_shape, model_oc, model_shape_oc, batches):
    """batches: iterable of (image [B,3,H,W] device, label_od [B,1,h,w], label_oc [B,1,h,w]) -> (mean cup Dice, mean disc Dice).
    Puts the four networks in eval mode for the duration (Trainer.py:138-141) and restores the previous mode."""
    nets = [model, model_shape, model_oc, model_shape_oc]
    modes = [n.training for n in nets]
    for n in nets:
        n.eval()
    cup, disc, total = 0.0, 0.0, 0
    try:
        for image, label_od, label_oc in batches:
            pred, pred_oc = predict_pair(model, model_shape, model_oc, model_shape_oc, image, label_od.shape[2:])
            lod, loc = label_od.cpu().numpy(), label_oc.cpu().numpy()
            for i in range(pred.shape[0]):
                disc += dice(postprocess(pred[i])[0], lod[i, 0])
                cup += dice(postprocess(pred_oc[i])[0], loc[i, 0])
                total += 1
    finally:
        for n, m in zip(nets, modes):
            n.train(m)
    return cup / max(total, 1), disc / max(total, 1)
