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
