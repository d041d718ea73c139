"""Ranger2020 with the reference's constructor and state layout (learning/optimizer.py:62-135), stepped by
two HIP launches over all parameters (csrc/ranger.hip) instead of the reference's per-tensor Python loop of
~15 small torch ops each (:145-253).

``state_dict()`` / ``load_state_dict()`` are the reference's: per parameter ``step``, ``exp_avg``,
``exp_avg_sq``, ``slow_buffer`` -- so Engine.resume (learning/engine.py:511-525) restores either way.
"""
import math
import os

import numpy as np
import torch
from torch.optim.optimizer import Optimizer

from . import _lib

_REC = np.dtype([("param", "<u8"), ("grad", "<u8"), ("exp_avg", "<u8"), ("exp_avg_sq", "<u8"), ("slow", "<u8"),
                 ("numel", "<i8"), ("rowlen", "<i4"), ("row_base", "<i4"), ("neg_step", "<f4"), ("wd", "<f4"),
                 ("flags", "<i4"), ("chunk_base", "<i4")])  # == brats_ranger_tensor (include/brats_hip.h)


def radam_step_size(step, beta1, beta2, n_sma_threshold):
    """learning/optimizer.py:198-214 -> (N_sma > threshold, step_size)."""
    beta2_t = beta2 ** step
    n_max = 2 / (1 - beta2) - 1
    n_sma = n_max - 2 * step * beta2_t / (1 - beta2_t)
    if n_sma > n_sma_threshold:
        return True, math.sqrt((1 - beta2_t) * (n_sma - 4) / (n_max - 4) * (n_sma - 2) / n_sma * n_max / (n_max - 2)) / (
            1 - beta1 ** step)
    return False, 1.0 / (1 - beta1 ** step)


class Ranger2020(Optimizer):
    def __init__(self, params, lr=1e-3, alpha=0.5, k=6, N_sma_threshhold=5, betas=(.95, 0.999), eps=1e-5, weight_decay=0,
                 use_gc=True, use_gcnorm=False, normloss=False, normloss_factor=1e-4, gc_conv_only=False, gc_loc=True,
                 capturable=False):
        if not 0.0 <= alpha <= 1.0:
            raise ValueError(f'Invalid slow update rate: {alpha}')
        if not 1 <= k:
            raise ValueError(f'Invalid lookahead steps: {k}')
        if not lr > 0:
            raise ValueError(f'Invalid Learning Rate: {lr}')
        if not eps > 0:
            raise ValueError(f'Invalid eps: {eps}')
        if normloss:
            # learning/optimizer.py:192-198: p.mul_() on a leaf that requires grad, outside torch.no_grad() -- the reference's
            # own step() raises RuntimeError there, so there is no behaviour to reproduce
            raise NotImplementedError("Ranger2020(normloss=True) is not built: the reference's step() itself fails on it "
                                      "(in-place operation on a leaf Variable, learning/optimizer.py:198)")
        if not gc_loc:
            raise NotImplementedError("brats21_amd.optim.Ranger2020 implements gc_loc=True (the reference's default)")
        defaults = dict(lr=lr, alpha=alpha, k=k, betas=betas, N_sma_threshhold=N_sma_threshhold, eps=eps,
                        weight_decay=weight_decay)
        super().__init__(params, defaults)
        self.N_sma_threshhold, self.alpha, self.k = N_sma_threshhold, alpha, k
        self.use_gc, self.gc_conv_only, self.eps = use_gc, gc_conv_only, eps
        self.use_gcnorm = use_gcnorm
        # capturable (extension, like torch.optim.Adam's): the step counter and the RAdam scalars live on the device,
        # so a step captured into a hipGraph (engine.GraphedTrainStep) replays without host-side changes.  All
        # parameters of a group then share one step count; the learning rate is a device scalar too (sync_lr() rewrites
        # it when an LR scheduler changed group["lr"]: no re-capture).
        self.capturable = capturable
        self._plans = {}
        # torch.amp.GradScaler protocol (as torch's fused Adam): GradScaler.step() then hands the loss scale and the overflow flag
        # over as DEVICE tensors (self.grad_scale / self.found_inf) instead of reading the flag on the host and unscaling in a
        # pass of its own.  capturable=True: the skip is decided on the device too (no host round trip per step); otherwise the
        # flag is read here -- the reference loop's own synchronisation (learning/engine.py:117-122), no more
        self._step_supports_amp_scaling = os.environ.get("BRATS_RANGER_AMP", "1") != "0"  # (0: GradScaler's own unscale pass + host check, for A/B runs)

    # ------------------------------------------------------------------------------------------ static plan
    def _plan(self, gi, active, dev):
        """chunk / row tables for the set of parameters that have a gradient (static per model)."""
        key = (gi, tuple(id(p) for p in active))
        plan = self._plans.get(key)
        if plan is not None:
            return plan
        chunk = _lib.lib().brats_ranger_chunk()
        chunks, rows, rowlen, rowbase, nrows, chunkbase, nch = [], [], [], [], 0, [], 0
        for t, p in enumerate(active):
            n = p.numel()
            chunkbase.append(nch)
            nch += (n + chunk - 1) // chunk
            chunks.append(np.stack([np.full((n + chunk - 1) // chunk, t, np.int32),
                                    np.arange((n + chunk - 1) // chunk, dtype=np.int32)], 1))
            gc = self.use_gc and p.dim() > (3 if self.gc_conv_only else 1)
            if gc:
                r = p.shape[0]
                rows.append(np.stack([np.full(r, t, np.int32), np.arange(r, dtype=np.int32)], 1))
                rowlen.append(n // r)
                rowbase.append(nrows)
                nrows += r
            else:
                rowlen.append(0)
                rowbase.append(0)
        plan = {
            "chunks": torch.from_numpy(np.concatenate(chunks)).to(dev),
            "rows": torch.from_numpy(np.concatenate(rows)).to(dev) if rows else None,
            "means": torch.empty(max(nrows, 1), dtype=torch.float32, device=dev),
            "nrows": nrows, "rowlen": rowlen, "rowbase": rowbase, "chunkbase": chunkbase,
            # use_gcnorm workspaces: per-chunk (sum, sum of squares) and the per-tensor standard deviation
            "chunk_stats": torch.empty((nch, 2), dtype=torch.float32, device=dev) if self.use_gcnorm else None,
            "grad_std": torch.empty(len(active), dtype=torch.float32, device=dev) if self.use_gcnorm else None,
        }
        self._plans[key] = plan
        return plan

    # ------------------------------------------------------------------------------------------ step
    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._plans = {}  # the state tensors were replaced: cached pointer tables are stale

    def sync_lr(self):
        """capturable mode: copy every group's current lr into its device scalar (call between graph replays after an LR
        scheduler stepped; GraphedTrainStep does).  A no-op while nothing changed."""
        for (gi, _), plan in self._plans.items():
            dyn = plan.get("dyn")
            if dyn is not None:
                lr = float(self.param_groups[gi]["lr"])
                if plan.get("lr_dev") != lr:
                    if torch.cuda.is_current_stream_capturing():
                        raise _lib.BratsHipError("Ranger2020: the learning rate changed inside a hipGraph capture")
                    dyn[4:6].view(torch.float64).fill_(lr)
                    plan["lr_dev"] = lr

    def sync_steps(self):
        """capturable mode: graph replays advance only the device-side step counters; copy them into state['step']."""
        for plan in self._plans.values():
            if plan.get("dyn") is not None:
                step = int(plan["dyn"][0].item())
                for st in plan["states"]:
                    st['step'] = step

    def state_dict(self):
        self.sync_steps()
        return super().state_dict()

    def _table(self, plan, active, dev):
        """Host copy of the per-tensor records.  Parameter / state pointers and shapes never change between steps
        (cached); per step only the gradient pointers and the step-dependent scalars are rewritten -- the reference's
        Python loop over parameters (learning/optimizer.py:145-253) must not come back as host time here."""
        rec = plan.get("rec")
        if rec is not None:
            # the cached records hold raw addresses: parameters or state tensors re-allocated since (model.to() / .half()
            # / .float() round trips, load_state_dict(assign=True), p.data = ..., a state replaced by hand) must not be
            # written through the old ones
            ptrs = plan["ptrs"]
            for t, p in enumerate(active):
                st = self.state[p]
                if (p.data_ptr(), st['exp_avg'].data_ptr(), st['exp_avg_sq'].data_ptr(), st['slow_buffer'].data_ptr()) != ptrs[t]:
                    rec = None
                    break
        if rec is None:
            rec = np.zeros(len(active), _REC)
            for t, p in enumerate(active):
                if p.dtype != torch.float32 or not p.is_contiguous() or p.device != dev:
                    raise _lib.BratsHipError("Ranger2020: parameters must be contiguous f32 tensors on one GPU")
                state = self.state[p]
                if len(state) == 0:
                    state['step'] = 0
                    state['exp_avg'] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    state['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    state['slow_buffer'] = p.detach().clone(memory_format=torch.contiguous_format)
                else:
                    for name in ('exp_avg', 'exp_avg_sq', 'slow_buffer'):  # after load_state_dict / .cpu() round trips
                        s = state[name]
                        if s.dtype != torch.float32 or s.device != dev or not s.is_contiguous():
                            state[name] = s.to(device=dev, dtype=torch.float32).contiguous()
                    state['step'] = int(state['step'])
                rec[t] = (p.data_ptr(), 0, state['exp_avg'].data_ptr(), state['exp_avg_sq'].data_ptr(),
                          state['slow_buffer'].data_ptr(), p.numel(), plan["rowlen"][t], plan["rowbase"][t], 0.0, 0.0, 0, plan["chunkbase"][t])
            plan["rec"] = rec
            plan["states"] = [self.state[p] for p in active]
            plan["ptrs"] = [(int(r["param"]), int(r["exp_avg"]), int(r["exp_avg_sq"]), int(r["slow"])) for r in rec]
            # pinned staging ring: the H2D copy of the table is asynchronous, a slot is reused only after its copy ran
            plan["pinned"] = [torch.empty(rec.nbytes, dtype=torch.uint8).pin_memory() for _ in range(4)]
            plan["pinned_capture"] = torch.empty(rec.nbytes, dtype=torch.uint8).pin_memory()  # see step(): capture mode
            plan["events"] = [None] * 4
            plan["dev_table"] = torch.empty(rec.nbytes, dtype=torch.uint8, device=dev)
            plan["slot"] = 0
        return rec

    @torch.no_grad()
    def step(self, closure=None):
        lib = _lib.lib()
        grad_scale, found_inf = getattr(self, "grad_scale", None), getattr(self, "found_inf", None)
        if found_inf is not None and not self.capturable:
            if float(found_inf.item()) != 0.0:  # GradScaler's skipped step: nothing moves, the step counts stay
                return None
            found_inf = None
        for t_ in (grad_scale, found_inf):
            if t_ is not None and (t_.dtype != torch.float32 or t_.numel() != 1 or not t_.is_cuda):
                raise _lib.BratsHipError("Ranger2020: grad_scale / found_inf must be one-element f32 device tensors (torch.amp.GradScaler's)")
        for gi, group in enumerate(self.param_groups):
            active = [p for p in group["params"] if p.grad is not None]
            if not active:
                continue
            dev = active[0].device
            if dev.type != "cuda":
                raise _lib.BratsHipError("brats21_amd.optim.Ranger2020 steps on the GPU only (no CPU fallback)")
            beta1, beta2 = group["betas"]
            lr, wd, k = group["lr"], group["weight_decay"], group["k"]
            plan = self._plan(gi, active, dev)
            if self.capturable and plan.get("dyn") is None and not torch.cuda.is_current_stream_capturing():
                # a plan about to get its device-side step counter: the host-side state['step'] runs AHEAD of the device after
                # overflow-skipped steps (the skip is decided on the device) -- seed it from the device's truth
                self.sync_steps()
            rec = self._table(plan, active, dev)
            keep = []
            grads = rec["grad"]
            for t, p in enumerate(active):
                g = p.grad
                if g.dtype != torch.float32 or not g.is_contiguous() or g.is_sparse:
                    if g.is_sparse:
                        raise RuntimeError('Ranger optimizer does not support sparse gradients')
                    g = g.float().contiguous()
                    keep.append(g)
                grads[t] = g.data_ptr()
            states = plan["states"]
            steps = [st['step'] + 1 for st in states]
            for st, v in zip(states, steps):
                st['step'] = v
            cache = {}
            neg, flags = rec["neg_step"], rec["flags"]
            for t, v in enumerate(steps):
                c = cache.get(v)
                if c is None:
                    adaptive, step_size = radam_step_size(v, beta1, beta2, self.N_sma_threshhold)
                    c = cache[v] = (-step_size * lr, (1 if adaptive else 0) | (2 if v % k == 0 else 0))
                neg[t], flags[t] = c
            rec["wd"] = wd
            capturing = torch.cuda.is_current_stream_capturing()
            stream = torch.cuda.current_stream().cuda_stream
            dyn = None
            if self.capturable:
                if len(cache) != 1:
                    raise _lib.BratsHipError("Ranger2020(capturable=True): all parameters of a group must share one step count")
                dyn = plan.get("dyn")
                if dyn is None:  # brats_ranger_dyn {step, flags, neg_step, reserved, double lr}; step = the count BEFORE this step
                    if capturing:
                        raise _lib.BratsHipError("Ranger2020(capturable=True): run one eager step before capturing (state allocation)")
                    dyn = plan["dyn"] = torch.tensor([steps[0] - 1, 0, 0, 0, 0, 0], dtype=torch.int32, device=dev)
                if not capturing:
                    self.sync_lr()
                elif plan.get("lr_dev") != float(lr):
                    raise _lib.BratsHipError("Ranger2020: the learning rate changed between the warm-up steps and the capture")
                _lib.check(lib.brats_ranger_advance_amp(dyn.data_ptr(), -1.0, float(beta1), float(beta2), int(k),
                                                        float(self.N_sma_threshhold),
                                                        found_inf.data_ptr() if found_inf is not None else None, stream), "ranger_advance")
            elif capturing:
                raise _lib.BratsHipError("Ranger2020.step() inside a hipGraph capture needs capturable=True")
            table = plan["dev_table"]
            if capturing:
                # the memcpy node re-reads this buffer at every replay: a buffer nobody rewrites afterwards (allocated with
                # the plan -- pinning memory is not allowed while a stream is capturing)
                pinned = plan["pinned_capture"]
                pinned.numpy()[:] = rec.view(np.uint8)
                table.copy_(pinned, non_blocking=True)
            else:
                slot = plan["slot"]
                plan["slot"] = (slot + 1) % 4
                if plan["events"][slot] is not None:
                    plan["events"][slot].synchronize()
                pinned = plan["pinned"][slot]
                pinned.numpy()[:] = rec.view(np.uint8)
                table.copy_(pinned, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record()
                plan["events"][slot] = ev
            _lib.check(lib.brats_ranger_step_amp(
                table.data_ptr(), len(active), plan["chunks"].data_ptr(), plan["chunks"].shape[0],
                plan["rows"].data_ptr() if plan["rows"] is not None else None, plan["nrows"], plan["means"].data_ptr(),
                plan["chunk_stats"].data_ptr() if self.use_gcnorm else None, plan["grad_std"].data_ptr() if self.use_gcnorm else None,
                dyn.data_ptr() if dyn is not None else None, beta1, beta2, 1 - beta1, 1 - beta2, group["eps"], self.alpha,
                grad_scale.data_ptr() if grad_scale is not None else None, found_inf.data_ptr() if found_inf is not None else None,
                stream), "ranger_step")
            for p in active:  # the kernel wrote through raw pointers: tell autograd / the packed-weight cache
                torch.autograd.graph.increment_version(p)
        return None
