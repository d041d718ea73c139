"""Counterpart of the reference's ``Engine`` inner loops (learning/engine.py): the per-batch body of
``Engine.train`` (:88-123) and the per-case body of ``Engine.evaluate`` (:205-285, see evaluate.Evaluator),
with every tensor resident on the GPU.  Logging, meters, checkpoints and data loading are the reference's
own and out of scope here."""
import torch

from . import ops
from .evaluate import Evaluator  # noqa: F401  (re-export: the evaluate-step driver)
from .losses import DiceLoss, deep_supervision_loss, fused_deep_supervision_dice


class TrainStep:
    """One training iteration: zero_grad -> autocast forward -> deep-supervision Dice -> backward ->
    [bucketed gradient all-reduce] -> [clip] -> optimizer step.

    amp=True is the reference's default (``--no_amp`` off).  ``amp_dtype``:
      * torch.bfloat16 (default): bf16 MFMA kernels under autocast; bf16 needs no GradScaler;
      * torch.float16: the reference's own arithmetic (autocast fp16 + GradScaler, learning/engine.py:304,117-122,
        src/main_train.py:110): fp16 MFMA kernels (same rate, three more mantissa bits), the loss is scaled before
        backward, gradients are unscaled / checked for inf and the step is skipped on overflow by ``scaler``
        (a torch.amp.GradScaler, created here when none is passed) exactly as in the reference's loop.
    ``buckets`` is a brats21_amd.ddp.GradientBuckets when world_size > 1."""

    def __init__(self, model, optimizer, criterion=None, amp=True, buckets=None, fused_dice=True, jaccard=False,
                 max_grad_norm=None, amp_dtype=torch.bfloat16, scaler=None):
        self.model, self.optimizer, self.amp, self.buckets = model, optimizer, amp, buckets
        self.fused, self.jaccard, self.max_grad_norm = fused_dice and criterion is None, jaccard, max_grad_norm
        self.criterion = criterion if criterion is not None else DiceLoss(jaccard=jaccard)
        self.amp_dtype = amp_dtype
        self.scaler = scaler
        self._params = None
        if amp and amp_dtype == torch.float16 and scaler is None:
            self.scaler = torch.amp.GradScaler("cuda")  # src/main_train.py:110

    def loss(self, outputs, target):
        """Engine._compute_loss (learning/engine.py:312-333): mean of the criterion over main + deep heads."""
        if self.fused:
            return fused_deep_supervision_dice(outputs, target, jaccard=self.jaccard)
        return deep_supervision_loss(self.criterion, outputs, target)[0]

    def _zero_grad(self):
        # model.zero_grad(set_to_none=True) without walking the module tree every step (0.3 ms of host time for EquiUnetASSPEvo)
        params = self._params
        if params is None:
            params = self._params = list(self.model.parameters())
        for p in params:
            p.grad = None

    def __call__(self, image, target):
        self._zero_grad()
        with torch.autocast("cuda", dtype=self.amp_dtype, enabled=self.amp):
            outputs = self.model(image)
            loss = self.loss(outputs, target)
        if self.scaler is not None and self.amp:  # learning/engine.py:117-122
            self.scaler.scale(loss).backward()
            if self.buckets is not None:
                self.buckets.finish()
            if self.max_grad_norm is not None:  # Engine._unscale_and_clip (learning/engine.py:442-452)
                self.scaler.unscale_(self.optimizer)
                torch.nn.utils.clip_grad_norm_(self.model.parameters(), self.max_grad_norm)
            self.scaler.step(self.optimizer)
            self.scaler.update()
            return loss
        loss.backward()
        if self.buckets is not None:
            self.buckets.finish()
        if self.max_grad_norm is not None:  # Engine._unscale_and_clip (learning/engine.py:442-452)
            torch.nn.utils.clip_grad_norm_(self.model.parameters(), self.max_grad_norm)
        self.optimizer.step()
        return loss


class GraphedTrainStep:
    """A TrainStep captured once into a hipGraph (torch.cuda.CUDAGraph on ROCm) and replayed: forward, fused Dice,
    the backward program, and the optimizer step become ONE graph launch instead of 350-600 kernel launches, which
    takes the Python / launch path off the critical path (EquiUnetASSPEvo is launch-bound on slow hosts: 23 vs 28 ms).

    Requirements: fixed input shapes, an optimizer whose step is capturable (brats21_amd.optim.Ranger2020(capturable=
    True), or a torch optimizer constructed with capturable=True).  ``warmup`` eager steps run on the first batch before
    the capture (lazy initialisation, allocator warm-up): they are real steps.  With brats21_amd.optim.Ranger2020 the
    learning rate is a device scalar that is refreshed before every replay, so LR schedulers (the reference steps
    its scheduler once per epoch, learning/engine.py:151-155) work without a re-capture; a torch optimizer bakes lr in.

    Data parallel: with ``step.buckets`` (brats21_amd.ddp.GradientBuckets over the RCCL backend) the bucketed all-reduces
    are captured too -- the backward program pushes gradients into the buckets, each bucket's collective is forked onto
    RCCL's stream inside the graph and joined before the optimizer kernels -- so the 8-GPU step of EquiUnetASSPEvo is one
    graph launch per rank instead of a host-bound eager step.  The capture then runs in "thread_local" error mode
    (ProcessGroupNCCL's watchdog thread polls events while the stream is capturing; probed on ROCm 7 / PyTorch 2.10:
    scripts/probes/nccl_capture.py).  Needs >= 2 warm-up steps: the first one learns the bucket order.

    UNVERIFIED ON MORE THAN ONE GPU: capturing the collectives has only been exercised at world size 1 with forced
    collectives (tests/test_ddp_gpu.py) -- this pool has 1-GPU boxes.  Cross-rank ordering of the bucket launches inside the
    capture, the watchdog under real traffic and bf16 wire copies are untested, so world size > 1 must be asked for
    explicitly: ``multi_rank_capture=True`` (or BRATS_GRAPH_DDP=1); otherwise use the eager TrainStep with buckets."""

    def __init__(self, step, warmup=2, multi_rank_capture=None):
        if getattr(step, "scaler", None) is not None and step.amp:
            # torch.amp.GradScaler.step() reads found_inf on the HOST for an ordinary optimizer: a capture would bake one step's
            # decision into the graph.  With an optimizer that takes the scale and the flag as device tensors
            # (_step_supports_amp_scaling: brats21_amd.optim.Ranger2020(capturable=True), torch's fused Adam) nothing of the loop
            # touches the host -- scale(), the inf check, the skipped-or-not step and update() are all kernels -- and it captures.
            if not (getattr(step.optimizer, "_step_supports_amp_scaling", False) and getattr(step.optimizer, "capturable", False)) \
                    or step.max_grad_norm is not None:
                raise NotImplementedError("GraphedTrainStep: a TrainStep with a GradScaler (amp_dtype=torch.float16) captures only with "
                                          "an optimizer that implements the GradScaler protocol on the device "
                                          "(brats21_amd.optim.Ranger2020(capturable=True)) and without gradient clipping; otherwise use "
                                          "the eager TrainStep for fp16, or bf16 / model.precision='x3' for graph replay")
        if step.buckets is not None:
            import os
            import torch.distributed as dist
            if dist.is_initialized() and dist.get_backend(step.buckets.group) != "nccl":
                raise NotImplementedError("GraphedTrainStep with gradient buckets needs the nccl (RCCL) backend: a gloo "
                                          "all-reduce runs on the host and cannot be captured into a hipGraph")
            if multi_rank_capture is None:
                multi_rank_capture = os.environ.get("BRATS_GRAPH_DDP", "0") == "1"
            if dist.is_initialized() and dist.get_world_size(step.buckets.group) > 1 and not multi_rank_capture:
                raise NotImplementedError("GraphedTrainStep: capturing RCCL all-reduces at world size > 1 is unverified on "
                                          "multi-GPU hardware; pass multi_rank_capture=True (or BRATS_GRAPH_DDP=1) to opt in")
            warmup = max(warmup, 2)
        if not getattr(step.optimizer, "capturable", False) and not all(
                g.get("capturable", False) for g in step.optimizer.param_groups):
            raise ValueError("GraphedTrainStep needs a capturable optimizer (e.g. brats21_amd.optim.Ranger2020(capturable=True))")
        self.step, self.warmup = step, warmup
        self.graph = self.static_image = self.static_target = self.static_loss = None

    def _capture(self, image, target):
        self.static_image, self.static_target = image.clone(), target.clone()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(self.warmup):
                self.step(self.static_image, self.static_target)
        torch.cuda.current_stream().wait_stream(side)
        self.step.model.zero_grad(set_to_none=True)
        self.graph = torch.cuda.CUDAGraph()
        kw = {}
        if self.step.buckets is not None:
            self.step.buckets.measure = False  # (timing events cannot be recorded into a graph and read back per replay)
            kw["capture_error_mode"] = "thread_local"
        with torch.cuda.graph(self.graph, **kw):
            self.static_loss = self.step(self.static_image, self.static_target)

    def __call__(self, image, target):
        if self.graph is None:
            self._capture(image, target)
        else:
            if image.data_ptr() != self.static_image.data_ptr():
                self.static_image.copy_(image, non_blocking=True)
            if target.data_ptr() != self.static_target.data_ptr():
                self.static_target.copy_(target, non_blocking=True)
            sync_lr = getattr(self.step.optimizer, "sync_lr", None)
            if sync_lr is not None:
                sync_lr()  # (a scheduler may have changed group["lr"] since the last replay)
        self.graph.replay()
        ops.invalidate_packed_weights()  # the replayed optimizer kernels changed the weights behind autograd's back
        return self.static_loss
