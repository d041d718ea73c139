"""Loss that "stays in PyTorch-ROCm" (BASELINE.json north_star): the Dice / Jaccard criterion the
reference configures at src/definer.py:184-203 (monai.losses.DiceLoss(include_background=True,
sigmoid=True, squared_pred=True, batch=True, reduction='mean', smooth 1e-5)) and the
deep-supervision averaging of learning/engine.py:312-333."""
import torch
import torch.nn as nn


class DiceLoss(nn.Module):
    def __init__(self, jaccard=False, smooth_nr=1e-5, smooth_dr=1e-5):
        super().__init__()
        self.jaccard, self.smooth_nr, self.smooth_dr = jaccard, smooth_nr, smooth_dr

    def forward(self, logits, target):
        p = torch.sigmoid(logits.float())
        t = target.float()
        axes = (0, 2, 3, 4)  # batch=True: the batch dimension is reduced too
        inter = (t * p).sum(axes)
        denom = (t * t).sum(axes) + (p * p).sum(axes)
        if self.jaccard:
            denom = 2.0 * (denom - inter)
        return (1.0 - (2.0 * inter + self.smooth_nr) / (denom + self.smooth_dr)).mean()


def deep_supervision_loss(criterion, outputs, target):
    """mean over [main] + deep heads of criterion(head, label) (learning/engine.py:322-330)."""
    if isinstance(outputs, (tuple, list)):
        heads = [outputs[0]] + list(outputs[1])
        return torch.stack([criterion(h, target) for h in heads]).mean(), outputs[0]
    return criterion(outputs, target), outputs


class _FusedDiceFn(torch.autograd.Function):
    """loss = mean over heads and classes of (1 - (2I+eps)/(D+eps)); two HBM passes per head in HIP
    (brats_dice_stats / brats_dice_grad), the [heads, classes] algebra in torch, no host sync."""

    @staticmethod
    def forward(ctx, target, jaccard, eps, *heads):
        from . import _lib
        lib = _lib.lib()
        st = torch.cuda.current_stream().cuda_stream
        t = target.contiguous().float()
        n, k = t.shape[:2]
        vox = t[0, 0].numel()
        hs = [h.contiguous().float() for h in heads]
        sums = torch.empty((len(hs), k, 3), dtype=torch.float32, device=t.device)
        ws = torch.empty(lib.brats_dice_ws_floats(n, k), dtype=torch.float32, device=t.device)  # per-block partials
        for i, h in enumerate(hs):
            _lib.check(lib.brats_dice_stats(h.data_ptr(), t.data_ptr(), sums[i].data_ptr(), ws.data_ptr(), n, k, vox, st), "dice_stats")
        inter, p2, t2 = sums[..., 0], sums[..., 1], sums[..., 2]
        hk = float(len(hs) * k)
        if jaccard:
            den = 2.0 * (t2 + p2 - inter)
            f = 1.0 - (2.0 * inter + eps) / (den + eps)
            dden = (2.0 * inter + eps) / (den + eps) ** 2
            coef = torch.stack([(-2.0 / (den + eps) - 2.0 * dden) / hk, 2.0 * dden / hk], -1)
        else:
            den = t2 + p2
            f = 1.0 - (2.0 * inter + eps) / (den + eps)
            coef = torch.stack([-2.0 / (den + eps) / hk, (2.0 * inter + eps) / (den + eps) ** 2 / hk], -1)
        ctx.save_for_backward(t, coef.contiguous(), *hs)
        return f.mean()

    @staticmethod
    def backward(ctx, g):
        from . import _lib
        lib = _lib.lib()
        st = torch.cuda.current_stream().cuda_stream
        t, coef, *hs = ctx.saved_tensors
        n, k = t.shape[:2]
        vox = t[0, 0].numel()
        coef = (coef * g).contiguous()
        outs = []
        for i, h in enumerate(hs):
            d = torch.empty_like(h)
            _lib.check(lib.brats_dice_grad(h.data_ptr(), t.data_ptr(), coef[i].data_ptr(), d.data_ptr(), n, k, vox, st), "dice_grad")
            outs.append(d)
        return (None, None, None) + tuple(outs)


def fused_deep_supervision_dice(outputs, target, jaccard=False, eps=1e-5):
    """Same value and gradients as deep_supervision_loss(DiceLoss(jaccard), outputs, target), fused."""
    heads = [outputs[0]] + list(outputs[1]) if isinstance(outputs, (tuple, list)) else [outputs]
    return _FusedDiceFn.apply(target, jaccard, eps, *heads)
