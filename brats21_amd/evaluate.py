"""Post-forward chain of the reference's ``Engine.evaluate`` (learning/engine.py:205-285) kept on the GPU.

The reference pads the volume to a multiple of 8, runs every model x TTA pass, copies each output to the CPU,
averages sigmoid outputs there, thresholds, copies back, removes background voxels, converts to BraTS labels
and crops.  Here every step is a HIP kernel on device buffers (csrc/post.hip, csrc/infer.hip); only the final
label map / metric scalars leave the GPU.

Function names and argument meaning follow utils/transforms.py (shape_to_divisible :482, shape_to_original :515,
remove_background_voxels :536) so Engine.evaluate can import them unchanged.
"""
import numpy as np
import torch

from . import _lib
from .inferers import GraphedPredictor, _first, sliding_window_inference


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _need_cuda(t, what):
    if not t.is_cuda:
        raise _lib.BratsHipError(f"brats21_amd.evaluate.{what} runs on the GPU only (no CPU fallback)")


def _pad_crop(data, out_spatial, offset, fill=0.0):
    x = data.contiguous().float()
    lead = tuple(x.shape[:-3])
    planes = int(np.prod(lead)) if lead else 1
    out = torch.empty(lead + tuple(int(v) for v in out_spatial), dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().brats_pad_crop(x.data_ptr(), out.data_ptr(), planes, *x.shape[-3:], *out.shape[-3:],
                                         *[int(o) for o in offset], float(fill), _stream()), "pad_crop")
    return out


def shape_to_divisible(data, k=16, min_shape=None):
    """utils/transforms.py:482-512 -> (padded, p_b, p_a); the odd voxel of an odd padding goes in front."""
    assert k > 0, "k need to positive"
    if data.dim() not in (4, 5):
        raise ValueError("Tensor dimension is incorrect")
    _need_cuda(data, "shape_to_divisible")
    shape = np.array(data.shape[-3:])
    tgt = np.ceil(shape / k).astype(int) * k
    if min_shape is not None:
        tgt[tgt < min_shape] = min_shape
    p = tgt - shape
    p_b, p_a = np.ceil(p / 2).astype(int), np.floor(p / 2).astype(int)
    return _pad_crop(data, tgt, p_b), p_b, p_a


def shape_to_original(data, p_b, p_a):
    """utils/transforms.py:515-533."""
    if data.dim() not in (4, 5):
        raise ValueError("Tensor dimension is incorrect")
    _need_cuda(data, "shape_to_original")
    shape = np.array(data.shape[-3:])
    return _pad_crop(data, shape - np.asarray(p_a) - np.asarray(p_b), -np.asarray(p_b))


def finalize_segmentation(prob_sum, passes=1, img=None, thresh=0.5, want_labels=False):
    """mean over passes (learning/engine.py:249) + AsDiscrete(threshold) (src/definer.py:700-703) +
    remove_background_voxels (utils/transforms.py:536-550) [+ BraTS labels, utils/transforms.py:169-206] in
    one pass.  prob_sum: [N, K, D, H, W] sum of probabilities; img: [N, C, D, H, W] or None.
    -> seg f32 0/1 [N, K, D, H, W] (and uint8 labels [N, 1, D, H, W] when want_labels)."""
    _need_cuda(prob_sum, "finalize_segmentation")
    p = prob_sum.contiguous().float()
    n, k = p.shape[:2]
    vox = p[0, 0].numel()
    im = None
    if img is not None:
        im = img.contiguous().float()
        if im.shape[0] != n or tuple(im.shape[2:]) != tuple(p.shape[2:]):
            raise ValueError(f"image {tuple(im.shape)} does not match predictions {tuple(p.shape)}")
    seg = torch.empty_like(p)
    labels = torch.empty((n, 1) + tuple(p.shape[2:]), dtype=torch.uint8, device=p.device) if want_labels else None
    _lib.check(_lib.lib().brats_post_threshold(p.data_ptr(), im.data_ptr() if im is not None else None, seg.data_ptr(),
                                               labels.data_ptr() if labels is not None else None, n, k,
                                               im.shape[1] if im is not None else 0, vox, 1.0 / passes, float(thresh),
                                               _stream()), "post_threshold")
    return (seg, labels) if want_labels else seg


def remove_background_voxels(img, outputs):
    """utils/transforms.py:536-550 (outputs are the thresholded 0/1 maps, as in Engine.evaluate)."""
    return finalize_segmentation(outputs, 1, img, thresh=0.5)


def to_brats_labels(seg):
    """ConvertToBratsClassesBasedOnMultiChannel + ChangeLabel3To4 (utils/transforms.py:169-206), batched."""
    return finalize_segmentation(seg, 1, None, thresh=0.5, want_labels=True)[1]


def overlap_counts(pred, target):
    """uint64-exact {|P&T|, |P|, |T|} per (n, k) -> int64 tensor [N, K, 3] (device)."""
    _need_cuda(pred, "overlap_counts")
    p, t = pred.contiguous().float(), target.contiguous().float()
    if p.shape != t.shape:
        raise ValueError(f"prediction {tuple(p.shape)} and target {tuple(t.shape)} differ")
    n, k = p.shape[:2]
    counts = torch.empty((n, k, 3), dtype=torch.int64, device=p.device)
    _lib.check(_lib.lib().brats_overlap_counts(p.data_ptr(), t.data_ptr(), counts.data_ptr(), n * k, p[0, 0].numel(),
                                               _stream()), "overlap_counts")
    return counts


def hard_dice_metric(pred, target):
    """Dice per (n, k) with the empty-label conventions of utils/metrics.py:47-67 (both empty -> 1, exactly
    one empty -> 0)."""
    c = overlap_counts(pred, target).double()
    inter, ps, ts = c[..., 0], c[..., 1], c[..., 2]
    dice = 2 * inter / (ps + ts).clamp_min(1)
    dice = torch.where((ps == 0) & (ts == 0), torch.ones_like(dice), dice)
    return dice.float()


class Evaluator:
    """The per-case body of Engine.evaluate (learning/engine.py:205-285) for one or several models:
    pad to k -> [TTA x] (sliding window | whole volume) -> on-GPU mean of sigmoid -> threshold ->
    background removal -> (labels) -> crop.

    ``use_graph`` (default: only with a sliding window, whose patch shape is fixed): the patch step of each model is
    captured into a hipGraph once and replayed; the graphs follow the models' weights (GraphedPredictor re-captures when
    a parameter's address / version or ops' packed-weight generation changed).  Whole-volume evaluation
    (sliding_window_size=None, the reference's default path) sees a different padded shape for almost every case, so
    it runs eagerly unless use_graph=True is passed explicitly (then at most ``max_graphs`` shapes stay captured)."""

    def __init__(self, models, tta_transforms=None, sliding_window_size=None, sw_batch_size=1, overlap=0.25,
                 k_divisible=8, thresh=0.5, amp=True, use_graph=None, max_graphs=4, amp_dtype=torch.bfloat16):
        self.models = list(models) if isinstance(models, (list, tuple)) else [models]
        self.tta, self.roi, self.swb, self.overlap = tta_transforms, sliding_window_size, sw_batch_size, overlap
        self.k, self.thresh, self.amp, self.amp_dtype = k_divisible, thresh, amp, amp_dtype
        if use_graph is None:
            use_graph = sliding_window_size is not None
        self.predictors = [GraphedPredictor(self._amp(m), modules=m, max_graphs=max_graphs) if use_graph else self._amp(m)
                           for m in self.models]

    def _amp(self, model):
        def run(x):
            with torch.autocast("cuda", dtype=self.amp_dtype, enabled=self.amp):
                return model(x)
        return run

    def _logits(self, predictor, x):
        if self.roi is not None:
            return sliding_window_inference(x, self.roi, self.swb, predictor, overlap=self.overlap)
        return _first(predictor(x)).float()

    @torch.no_grad()
    def probability_sum(self, image):
        """Sum over models x TTA passes of sigmoid(logits) on the (padded) image -> (sum, passes)."""
        acc, passes = None, 0
        for model, predictor in zip(self.models, self.predictors):
            model.eval()
            if self.tta is None:
                logits = self._logits(predictor, image)
                acc = torch.sigmoid(logits) if acc is None else acc.add_(torch.sigmoid(logits))
                passes += 1
                continue
            for t in self.tta:
                logits = self._logits(predictor, t.augment_image(image))
                if acc is None:
                    acc = torch.zeros(t.deaug_perm.out_shape(logits.shape), dtype=torch.float32, device=image.device)
                t.accumulate_probability(logits, acc)
                passes += 1
        return acc, passes

    @torch.no_grad()
    def __call__(self, image, target=None, return_original_shape=True, want_labels=False):
        """image [N, C, D, H, W] (cuda) -> dict(seg, [labels], [dice]); seg is cropped back to the input
        shape when return_original_shape (learning/engine.py:282-285)."""
        _need_cuda(image, "Evaluator")
        padded, p_b, p_a = shape_to_divisible(image, k=self.k)
        acc, passes = self.probability_sum(padded)
        res = finalize_segmentation(acc, passes, padded, self.thresh, want_labels)
        seg, labels = res if want_labels else (res, None)
        out = {}
        if target is not None:
            tp = shape_to_divisible(target, k=self.k)[0]
            out["dice"] = hard_dice_metric(seg, tp)
        if return_original_shape:
            seg = shape_to_original(seg, p_b, p_a)
            if labels is not None:
                labels = shape_to_original(labels.float(), p_b, p_a).to(torch.uint8)
        out["seg"] = seg
        if labels is not None:
            out["labels"] = labels
        return out
