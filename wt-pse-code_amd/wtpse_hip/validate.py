"""Validation front half: counterpart of Trainer.validate (reference Trainer.py:137-256) — SURVEY.md §8f row 2.

Device part (HIP kernels): OD predict -> od_pred = sigmoid > 0.75 -> ROI -> OC predict on the stacked (roi, roi) input
-> predictions_oc * od_pred -> bilinear resize of both logit maps to the label size (Trainer.py:170-209).
Host part (as in the reference, which does it in numpy/scipy/skimage on the CPU): threshold 0.75 on the sigmoid,
largest connected component + hole filling (utils.py:267-329), Dice = (2|A&B| + 1)/(|A| + |B| + 1) (metrics.py:68-97),
ASD / HD95 (Trainer.py:226-239), the per-epoch means, the validation objective and the best-Dice checkpoint (Trainer.py:242-288).

ASD / HD95 come from the un-vendored `medpy==0.5.2` in the reference (requirements.txt:2; `medpy.metric.binary.asd / hd95`), which is
not installed here: `asd` / `hd95` below restate its PUBLISHED algorithm — surface = object XOR its erosion by the
connectivity-1 structuring element, distances = Euclidean distance transform of the complement of the other surface sampled on
this surface; asd = their mean (result -> reference, ONE direction, as medpy's `asd`; its `assd` is the symmetric one), hd95 = the
95th percentile of both directions' distances pooled.  **Parity unpinned** (no medpy run to compare with); pinned instead to an
independent brute-force surface-distance oracle on small masks (oracle/metrics_cpu.py, tests/test_validate_cpu.py).
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


def _surface_distances(result, reference):
    """medpy.metric.binary.__surface_distances (0.5.2), voxelspacing None, connectivity 1: distances from every surface pixel of
    `result` to the nearest surface pixel of `reference`.  RuntimeError on an empty mask, as medpy raises."""
    from scipy.ndimage import binary_erosion, distance_transform_edt, generate_binary_structure
    result = np.atleast_1d(np.asarray(result).astype(np.bool_))
    reference = np.atleast_1d(np.asarray(reference).astype(np.bool_))
    footprint = generate_binary_structure(result.ndim, 1)
    if not result.any():
        raise RuntimeError("The first supplied array does not contain any binary object.")
    if not reference.any():
        raise RuntimeError("The second supplied array does not contain any binary object.")
    result_border = result ^ binary_erosion(result, structure=footprint, iterations=1)
    reference_border = reference ^ binary_erosion(reference, structure=footprint, iterations=1)
    dt = distance_transform_edt(~reference_border)
    return dt[result_border]


def asd(result, reference):
    """medpy.metric.binary.asd: mean distance from the surface of `result` to the surface of `reference` (pixels)."""
    return float(_surface_distances(result, reference).mean())


def hd95(result, reference):
    """medpy.metric.binary.hd95: 95th percentile of the surface distances of both directions, pooled."""
    return float(np.percentile(np.hstack((_surface_distances(result, reference), _surface_distances(reference, result))), 95))


def surface_metrics(pred_mask, label):
    """(hd95, asd) of one image as Trainer.validate scores them (Trainer.py:218-239): 100 / 100 for an empty prediction."""
    pred_mask = np.asarray(pred_mask)
    if np.sum(pred_mask) < 1e-4:
        return 100.0, 100.0
    p, r = np.asarray(pred_mask, dtype=np.bool_), np.asarray(label, dtype=np.bool_)
    return hd95(p, r), asd(p, r)


def validate_epoch(model, model_shape, model_oc, model_shape_oc, batches):
    """One pass of Trainer.validate's loop (Trainer.py:152-249) -> per-image means
    {cup_dice, disc_dice, cup_hd, disc_hd, cup_asd, disc_asd, n}.  batches: iterable of (image [B,3,H,W] device,
    label_od [B,1,h,w], label_oc [B,1,h,w]).  Eval mode for the duration, the previous modes restored (Trainer.py:138-141,289-311)."""
    nets = [model, model_shape, model_oc, model_shape_oc]
    modes = [n.training for n in nets]
    for n in nets:
        n.eval()
    acc = dict(cup_dice=0.0, disc_dice=0.0, cup_hd=0.0, disc_hd=0.0, cup_asd=0.0, disc_asd=0.0)
    total = 0
    try:
        for image, label_od, label_oc in batches:
            pred, pred_oc = predict_pair(model, model_shape, model_oc, model_shape_oc, image, label_od.shape[2:])
            lod, loc = label_od.cpu().numpy(), label_oc.cpu().numpy()
            for i in range(pred.shape[0]):
                post, post_oc = postprocess(pred[i])[0], postprocess(pred_oc[i])[0]
                acc["disc_dice"] += dice(post, lod[i, 0])
                acc["cup_dice"] += dice(post_oc, loc[i, 0])
                hd, a = surface_metrics(post_oc, loc[i, 0])
                acc["cup_hd"] += hd
                acc["cup_asd"] += a
                hd, a = surface_metrics(post, lod[i, 0])
                acc["disc_hd"] += hd
                acc["disc_asd"] += a
                total += 1
    finally:
        for n, m in zip(nets, modes):
            n.train(m)
    out = {k: v / max(total, 1) for k, v in acc.items()}
    out["n"] = total
    return out


def best_checkpoint(model, model_shape, model_oc, model_shape_oc):
    """The dict Trainer.validate saves on a new best (Trainer.py:282-288); test_visulization.py:132-193 loads it back."""
    return {"model": model.state_dict(), "model_shape": model_shape.state_dict(), "model_oc": model_oc.state_dict(),
            "model_oc_shape": model_shape_oc.state_dict()}


class Validator:
    """The per-epoch bookkeeping of Trainer.validate (Trainer.py:258-311): the validation objective ('OD' -> disc Dice, 'OC' -> cup
    Dice, anything else their mean), best_mean_dice / best_epoch, and on a new best the four-state_dict checkpoint (returned; saved
    with torch.save when `out_dir` is given, as checkpoint_<best_epoch>.pth.tar, with the score line appended to score.txt)."""

    def __init__(self, objective="OD_OC", out_dir=None):
        self.objective, self.out_dir = objective, out_dir
        self.best_mean_dice, self.best_epoch = 0.0, -1

    def __call__(self, epoch, model, model_shape, model_oc, model_shape_oc, batches):
        """-> (is_best, cup_dice, cup_hd, cup_asd, disc_dice, disc_hd, disc_asd) on a new best, (0, 0, 0, 0, 0, 0, 0) otherwise —
        Trainer.validate's return values — plus `self.last` = the epoch's means and `self.checkpoint` = the dict just built."""
        m = validate_epoch(model, model_shape, model_oc, model_shape_oc, batches)
        self.last = m
        mean_dice = m["disc_dice"] if self.objective == "OD" else m["cup_dice"] if self.objective == "OC" else \
            (m["cup_dice"] + m["disc_dice"]) / 2
        if not mean_dice > self.best_mean_dice:
            return 0, 0, 0, 0, 0, 0, 0
        self.best_epoch, self.best_mean_dice = epoch + 1, mean_dice
        self.checkpoint = best_checkpoint(model, model_shape, model_oc, model_shape_oc)
        if self.out_dir is not None:
            import os
            with open(os.path.join(self.out_dir, "score.txt"), "a") as f:
                f.write("cd:{} dd:{} c_hd:{} d_hd:{} c_asd:{} d_asd:{}\n".format(m["cup_dice"], m["disc_dice"], m["cup_hd"], m["disc_hd"],
                                                                               m["cup_asd"], m["disc_asd"]))
            torch.save(self.checkpoint, os.path.join(self.out_dir, "checkpoint_%d.pth.tar" % self.best_epoch))
        return 1, m["cup_dice"], m["cup_hd"], m["cup_asd"], m["disc_dice"], m["disc_hd"], m["disc_asd"]


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
