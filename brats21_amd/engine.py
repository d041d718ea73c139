"""Counterpart of the reference's ``Engine`` inner loops (learning/engine.py): the per-batch body of
``Engine.train`` (:88-123) and the per-case body of ``Engine.evaluate`` (:205-285, see evaluate.Evaluator),
with every tensor resident on the GPU.  Logging, meters, checkpoints and data loading are the reference's
own and out of scope here."""
import torch

from .evaluate import Evaluator  # noqa: F401  (re-export: the evaluate-step driver)
from .losses import DiceLoss, deep_supervision_loss, fused_deep_supervision_dice


class TrainStep:
    """One training iteration: zero_grad -> autocast forward -> deep-supervision Dice -> backward ->
    [bucketed gradient all-reduce] -> [clip] -> optimizer step.

    amp=True is the reference's default (``--no_amp`` off): bf16 MFMA kernels under autocast; bf16 needs no
    GradScaler (the reference's fp16 scaler, learning/engine.py:117-122, has nothing to scale here).
    ``buckets`` is a brats21_amd.ddp.GradientBuckets when world_size > 1."""

    def __init__(self, model, optimizer, criterion=None, amp=True, buckets=None, fused_dice=True, jaccard=False,
                 max_grad_norm=None):
        self.model, self.optimizer, self.amp, self.buckets = model, optimizer, amp, buckets
        self.fused, self.jaccard, self.max_grad_norm = fused_dice and criterion is None, jaccard, max_grad_norm
        self.criterion = criterion if criterion is not None else DiceLoss(jaccard=jaccard)

    def loss(self, outputs, target):
        """Engine._compute_loss (learning/engine.py:312-333): mean of the criterion over main + deep heads."""
        if self.fused:
            return fused_deep_supervision_dice(outputs, target, jaccard=self.jaccard)
        return deep_supervision_loss(self.criterion, outputs, target)[0]

    def __call__(self, image, target):
        self.model.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=self.amp):
            outputs = self.model(image)
            loss = self.loss(outputs, target)
        loss.backward()
        if self.buckets is not None:
            self.buckets.finish()
        if self.max_grad_norm is not None:  # Engine._unscale_and_clip (learning/engine.py:442-452)
            torch.nn.utils.clip_grad_norm_(self.model.parameters(), self.max_grad_norm)
        self.optimizer.step()
        return loss
