"""ORACLE (test infrastructure): CPU restatement of the reference's inference drivers --
sliding-window stitching (utils/inferers.py, a MONAI-0.6 fork) and the ttach-style TTA
(tta/base.py, tta/transforms.py, src/definer.py:647-658).  Index arithmetic is plain Python /
numpy; tensors are torch CPU fp32.  Pinned by tests/golden/inference.npz (window lists + stitched
output produced by the reference source under the MONAI stub: parity unpinned at the MONAI
boundary for dense_patch_slices / compute_importance_map, see oracle/refshim.py)."""
import itertools
import math

import numpy as np
import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------- sliding window
def scan_interval(image_size, roi_size, overlap):
    """_get_scan_interval, utils/inferers.py:165-186."""
    out = []
    for L, r in zip(image_size, roi_size):
        if r == L:
            out.append(int(r))
        else:
            iv = int(r * (1 - overlap))
            out.append(iv if iv > 0 else 1)
    return tuple(out)


def window_starts(image_size, roi_size, interval):
    """MONAI dense_patch_slices (called at utils/inferers.py:114): per dim the start list
    s_i = i*interval, shifted back so the last window fits; windows in row-major product order."""
    per_dim = []
    for L, p, iv in zip(image_size, roi_size, interval):
        num = int(math.ceil(float(L) / iv))
        scan = next(d for d in range(num) if d * iv + p >= L)
        per_dim.append([i * iv - max(i * iv + p - L, 0) for i in range(scan + 1)])
    return list(itertools.product(*per_dim))


def importance_map(patch, mode="constant", sigma_scale=0.125):
    """MONAI 0.6.0 compute_importance_map (utils/inferers.py:119-121); the restatement lives in oracle/refshim.py (the
    stub the reference itself is imported with when the golden vectors are generated)."""
    from .refshim import _compute_importance_map
    return _compute_importance_map(tuple(patch), mode, sigma_scale)


def sliding_window_inference(inputs, roi_size, sw_batch_size, predictor, overlap=0.25, mode="constant",
                             sigma_scale=0.125, cval=0.0, padding_mode="constant"):
    """utils/inferers.py:26-162: centre-pad up to roi (:103-109),
    windows (:111-116), weighted accumulate (:125-151), divide (:154), crop the pad (:156-162).
    The predictor may return (out, [deeps]); only the first tensor is kept (:135-136)."""
    if not 0 <= overlap < 1:
        raise AssertionError("overlap must be >= 0 and < 1.")
    img_size_ = list(inputs.shape[2:])
    nb = inputs.shape[0]
    roi = tuple(i if (r is None or r <= 0) else r for r, i in zip(roi_size, img_size_))
    image_size = tuple(max(i, r) for i, r in zip(img_size_, roi))
    pad = []
    for k in range(len(inputs.shape) - 1, 1, -1):
        diff = max(roi[k - 2] - inputs.shape[k], 0)
        half = diff // 2
        pad.extend([half, diff - half])
    x = F.pad(inputs, pad, mode=padding_mode, value=cval) if padding_mode == "constant" else F.pad(inputs, pad, mode=padding_mode)
    starts = window_starts(image_size, roi, scan_interval(image_size, roi, overlap))
    imp = importance_map(tuple(min(r, i) for r, i in zip(roi, image_size)), mode, sigma_scale)
    out = cnt = None
    total = len(starts) * nb
    for g0 in range(0, total, sw_batch_size):
        idxs = list(range(g0, min(g0 + sw_batch_size, total)))
        wins = []
        for idx in idxs:
            b, s = idx // len(starts), starts[idx % len(starts)]
            wins.append(x[b:b + 1, :, s[0]:s[0] + roi[0], s[1]:s[1] + roi[1], s[2]:s[2] + roi[2]])
        prob = predictor(torch.cat(wins))
        while isinstance(prob, (tuple, list)):
            prob = prob[0]
        if out is None:
            out = torch.zeros((nb, prob.shape[1]) + image_size, dtype=torch.float32)
            cnt = torch.zeros_like(out)
        for j, idx in enumerate(idxs):
            b, s = idx // len(starts), starts[idx % len(starts)]
            sl = (slice(b, b + 1), slice(None)) + tuple(slice(a, a + r) for a, r in zip(s, roi))
            out[sl] += imp * prob[j].float()
            cnt[sl] += imp
    out = out / cnt
    crop = [slice(None), slice(None)]
    for sp in range(3):
        p0 = pad[(2 - sp) * 2]
        crop.append(slice(p0, p0 + img_size_[sp]))
    return out[tuple(crop)]


# --------------------------------------------------------------------------- TTA
def tta_param_list():
    """get_tta_transforms, src/definer.py:647-658: product of OnAxes(['zxy','xyz']) x
    HorizontalFlip([False, True]) x Rotate90([0, 90, 180, 270]) in itertools.product order
    (tta/base.py:116) -> 16 (axe, flip, angle) tuples."""
    return list(itertools.product(["zxy", "xyz"], [False, True], [0, 90, 180, 270]))


def tta_augment(img, axe, flip, angle):
    """Image pipeline, forward order (tta/base.py:120-122): OnAxes (tta/transforms.py:30-36),
    HorizontalFlip = flip dim 3 (:56-59), Rotate90 = rot90(k, (2, 3)) (:165-167)."""
    if axe == "xyz":
        img = img.permute(0, 1, 3, 4, 2)
    elif axe == "yzx":
        img = img.permute(0, 1, 4, 2, 3)
    if flip:
        img = img.flip(3)
    k = angle // 90 if angle >= 0 else (angle + 360) // 90
    return torch.rot90(img, k, (2, 3))


def tta_deaugment(mask, axe, flip, angle):
    """Mask pipeline, reverse order (tta/base.py:113-117,123-125): Rotate90 by -angle
    (tta/transforms.py:169-170), flip dim 3, inverse permute (:38-44)."""
    a = -angle
    k = a // 90 if a >= 0 else (a + 360) // 90
    mask = torch.rot90(mask, k, (2, 3))
    if flip:
        mask = mask.flip(3)
    if axe == "xyz":
        mask = mask.permute(0, 1, 4, 2, 3)
    elif axe == "yzx":
        mask = mask.permute(0, 1, 3, 4, 2)
    return mask


def tta_predict(img, predictor, params=None):
    """Engine._apply_tta + the mean over passes of sigmoid(logits) (learning/engine.py:424-440,
    :239-249)."""
    params = params or tta_param_list()
    acc = None
    for axe, flip, angle in params:
        out = predictor(tta_augment(img, axe, flip, angle))
        while isinstance(out, (tuple, list)):
            out = out[0]
        p = torch.sigmoid(tta_deaugment(out, axe, flip, angle).float())
        acc = p if acc is None else acc + p
    return acc / len(params)
