"""TEST INFRASTRUCTURE (oracle): CPU restatement of the reference's validation post-processing, utils.py:267-329.

    get_largest_fillhole (utils.py:267-276):  label_image = skimage.measure.label(binary); regions = regionprops(label_image);
        idx_max = argmax(region.area); binary[label_image != idx_max + 1] = 0; scipy.ndimage.binary_fill_holes(binary)
    postprocessing, label != None branch (utils.py:306-323): sigmoid -> > 0.75 -> uint8 -> the above on channel 0

Third-party arithmetic: `skimage.measure.label` / `regionprops` come from scikit_image==0.19.2 (requirements.txt:11), which is NOT
installed in this image, and utils.py itself cannot be imported here (it imports skimage and cv2 at module level): **parity of
this function is pinned to skimage's published semantics, not to a run of the reference**.  Those semantics, restated:
  * label(): connected components of the non-zero pixels with connectivity = ndim (2-D: 8-connectivity, the default
    `connectivity=None` means full), labels 1, 2, ... assigned in raster order of each component's first pixel;
  * regionprops(): one region per label in ascending label order; `.area` = number of pixels;
  * np.argmax: the FIRST largest area wins ties.
Components are found here by an explicit flood fill in raster order (no scipy.ndimage.label: the product's validate.py uses that,
and the checker must not share its implementation); hole filling = background pixels not reachable from the border through
4-connected background (scipy.ndimage.binary_fill_holes with its default structuring element), also by flood fill.
"""
import numpy as np


def label8(binary):
    """-> (labels int32 [h,w], areas list): 8-connected components of `binary != 0`, numbered in raster order."""
    b = np.asarray(binary) != 0
    h, w = b.shape
    lab = np.zeros((h, w), np.int32)
    areas = []
    for y in range(h):
        for x in range(w):
            if not b[y, x] or lab[y, x]:
                continue
            n = len(areas) + 1
            lab[y, x] = n
            stack, area = [(y, x)], 0
            while stack:
                cy, cx = stack.pop()
                area += 1
                for dy in (-1, 0, 1):
                    for dx in (-1, 0, 1):
                        yy, xx = cy + dy, cx + dx
                        if 0 <= yy < h and 0 <= xx < w and b[yy, xx] and not lab[yy, xx]:
                            lab[yy, xx] = n
                            stack.append((yy, xx))
            areas.append(area)
    return lab, areas


def fill_holes(binary):
    """Background pixels that no 4-connected background path joins to the image border become foreground."""
    b = np.asarray(binary) != 0
    h, w = b.shape
    outside = np.zeros((h, w), bool)
    stack = [(y, x) for y in range(h) for x in (0, w - 1) if not b[y, x]] + [(y, x) for x in range(w) for y in (0, h - 1) if not b[y, x]]
    for y, x in stack:
        outside[y, x] = True
    while stack:
        cy, cx = stack.pop()
        for dy, dx in ((-1, 0), (1, 0), (0, -1), (0, 1)):
            yy, xx = cy + dy, cx + dx
            if 0 <= yy < h and 0 <= xx < w and not b[yy, xx] and not outside[yy, xx]:
                outside[yy, xx] = True
                stack.append((yy, xx))
    return ~outside


def get_largest_fillhole(binary):
    """utils.py:267-276."""
    binary = np.array(binary, copy=True)
    lab, areas = label8(binary)
    if areas:
        idx_max = int(np.argmax(areas))
        binary[lab != idx_max + 1] = 0
    return fill_holes(binary)


def postprocessing(logits, threshold=0.75):
    """utils.py:306-323 (label != None): logits [1,h,w] (numpy or torch) -> uint8 mask [1,h,w]."""
    x = np.asarray(logits.detach().cpu().numpy() if hasattr(logits, "detach") else logits, dtype=np.float32)
    prob = 1.0 / (1.0 + np.exp(-x.astype(np.float64)))
    mask = (prob.astype(np.float32) > threshold).astype(np.uint8)
    mask[0] = get_largest_fillhole(mask[0]).astype(np.uint8)
    return mask
