"""ORACLE (test infrastructure): CPU restatement of the post-forward chain of the reference's
Engine.evaluate (learning/engine.py:205-285): pad to a multiple of k, ensemble mean of sigmoid
outputs, threshold, background removal, BraTS label conversion, crop back, hard Dice.

Pinned by tests/golden/post.npz, produced by the reference's own functions
(utils/transforms.py:482-550, :169-206) run in this container by tests/golden/make_golden.py.
hard_dice_metric restates MONAI 0.6.0 DiceMetric(include_background=True, reduction=none) as
wrapped by utils/metrics.py:47-67 -- MONAI is not vendored in /root/reference, so that one function
is "parity unpinned" at the MONAI boundary (known-answer tests only)."""
import numpy as np
import torch
import torch.nn.functional as F


def shape_to_divisible(data, k=16, min_shape=None):
    """utils/transforms.py:482-512: symmetric zero pad, the odd voxel goes in front."""
    assert k > 0
    shape = np.array(data.shape[-3:])
    tgt = np.ceil(shape / k).astype(int) * k
    if min_shape is not None:
        tgt[tgt < min_shape] = min_shape
    p = tgt - shape
    p_b = np.ceil(p / 2).astype(int)
    p_a = np.floor(p / 2).astype(int)
    out = F.pad(data, (int(p_b[2]), int(p_a[2]), int(p_b[1]), int(p_a[1]), int(p_b[0]), int(p_a[0])))
    return out, p_b, p_a


def shape_to_original(data, p_b, p_a):
    """utils/transforms.py:515-533."""
    up = np.array(data.shape[-3:]) - p_a
    return data[..., p_b[0]:up[0], p_b[1]:up[1], p_b[2]:up[2]].contiguous()


def remove_background_voxels(img, outputs):
    """utils/transforms.py:536-550: keep predictions only where any modality is non-zero.  (The
    reference's [B,D,H,W] mask broadcasts against [B,K,D,H,W] only at batch 1, its evaluation batch
    size; the per-sample mask here is the same thing at B=1 and the evident intent beyond.)"""
    mask = (img != 0).any(dim=1, keepdim=True).to(outputs.dtype)
    return outputs * mask


def as_discrete(prob, thresh=0.5):
    """monai AsDiscrete(threshold_values=True, logit_thresh) as configured at src/definer.py:700-703:
    img >= thresh, as float."""
    return (prob >= thresh).float()


def to_brats_labels(seg):
    """ConvertToBratsClassesBasedOnMultiChannel + ChangeLabel3To4 (utils/transforms.py:169-206) for a
    batch: channels TC/WT/ET -> labels {0,1,2,4}, assignment order et, net, ed."""
    assert seg.dim() == 5 and seg.shape[1] == 3
    tc, wt, et = seg[:, 0].bool(), seg[:, 1].bool(), seg[:, 2].bool()
    lab = torch.zeros(tc.shape, dtype=torch.uint8)
    lab[et] = 4
    lab[tc & ~et] = 1
    lab[wt & ~tc] = 2
    return lab


def ensemble_segmentation(prob_list, img, thresh=0.5):
    """learning/engine.py:239-259: mean over models x TTA passes of the sigmoid outputs, threshold,
    background removal."""
    mean = torch.stack(list(prob_list)).mean(dim=0)
    return remove_background_voxels(img, as_discrete(mean, thresh))


def hard_dice_metric(pred, target):
    """Per (batch, class) hard Dice with the reference's empty-label conventions (utils/metrics.py:47-67):
    both empty -> 1, exactly one empty -> 0, else 2|P&T| / (|P| + |T|)."""
    p = pred != 0
    t = target != 0
    axes = tuple(range(2, pred.dim()))
    inter = (p & t).sum(axes).double()
    ps, ts = p.sum(axes).double(), t.sum(axes).double()
    dice = 2 * inter / (ps + ts).clamp_min(1)
    dice = torch.where((ps == 0) & (ts == 0), torch.ones_like(dice), dice)
    dice = torch.where((ps == 0) ^ (ts == 0), torch.zeros_like(dice), dice)
    return dice.float()
