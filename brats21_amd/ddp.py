"""Data-parallel training over RCCL / xGMI: one process per GPU, gradients only.

The reference is single-GPU; BASELINE.json's north_star adds: shard patients
across the 8 GPUs of a node, all-reduce the gradients.  Nothing else is exchanged: GroupNorm /
EvoNorm / SE are per-sample (no statistic sync) and DiceLoss(batch=True) reduces over the *local*
batch, so the DDP loss is the mean over ranks of per-rank batch-Dice.

Design for xGMI (7 point-to-point links x ~153 GB/s per GPU, ring collectives are per-link bound):
few, large buckets (default 32 MiB of f32 gradients -> 3 buckets for EquiUnet-48's 92.6 MB) filled in
the order the backward program produces gradients; each bucket's all-reduce is launched
asynchronously the moment its last gradient lands, so it overlaps the remaining backward kernels.
The accelerated models are a single autograd node, so instead of per-parameter autograd hooks the
backward program itself pushes every finished gradient into ``GradientBuckets.push`` (the
``_grad_sink`` attribute of the module).  Statically unused parameters (EvoNorm ``v``, SURVEY.md
Appendix B) never get a gradient and are simply not part of any bucket.
"""
import contextlib
import os

import torch
import torch.distributed as dist


def init_process_group_from_env(backend=None):
    """RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* come from torch.distributed.run (the CLI of the
    reference stays untouched: new knobs are env-only)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        # BRATS_DIST_BACKEND=gloo: debugging aid (e.g. two ranks sharing one GPU, where RCCL refuses duplicate devices)
        backend = backend or os.environ.get("BRATS_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_indices(n_items, rank, world, epoch=0, shuffle=True, seed=0):
    """Patient sharding (DistributedSampler-style over ``train_files``, src/definer.py:514-522):
    every rank gets ceil(n/world) indices, the list is padded by wrap-around so ranks stay in step."""
    if shuffle:
        g = torch.Generator().manual_seed(seed + epoch)
        order = torch.randperm(n_items, generator=g).tolist()
    else:
        order = list(range(n_items))
    per = -(-n_items // world)
    order += order[: per * world - n_items]
    return order[rank::world]


class GradientBuckets:
    """Bucketed, asynchronous gradient averaging for one model replica."""

    def __init__(self, model, bucket_bytes=32 << 20, process_group=None, comm_dtype=None):
        """comm_dtype=torch.bfloat16 (or BRATS_DDP_BF16=1): gradients travel as bf16 (half the xGMI payload: 33 MB instead
        of 66.5 MB for EquiUnetASSPEvo-48) and are summed in bf16 by the collective; p.grad stays f32.  Default: f32."""
        if comm_dtype is None and os.environ.get("BRATS_DDP_BF16", "0") != "0":
            comm_dtype = torch.bfloat16
        self.comm_dtype = comm_dtype or torch.float32
        self.force_collectives = False  # tests: launch the all-reduces even at world size 1
        self.measure = False       # bench.py: record how long finish() has to wait for the collectives (GPU time)
        self.exposed = []          # [(event before the waits, event after them)]
        self.model = model
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.params = list(model.parameters())  # indices = position in model.parameters()
        self.bucket_bytes = bucket_bytes
        self._plan = None          # list of buckets: [(param_index, offset, numel), ...]
        self._where = None         # param_index -> (bucket_id, offset)
        self._flat = None
        self._order = []           # production order seen during the first backward
        self._pending = None
        self._handles = []
        self._filled = None
        self._late = None          # buckets that must be gathered from p.grad in finish() (gradient accumulation)
        self._sync = True
        self._copies = None        # per bucket: ([destination slices], [gradients]) of pushes that still have to be copied in
        if hasattr(model, "_grad_sink"):
            model._grad_sink = self.push
            model._grad_dest = self.dest

    # -- planning ---------------------------------------------------------------------------------
    def _build_plan(self, order):
        plan, cur, size = [], [], 0
        for idx in order:
            n = self.params[idx].numel()
            if cur and (size + n) * 4 > self.bucket_bytes:
                plan.append(cur)
                cur, size = [], 0
            cur.append((idx, size, n))
            size += n
        if cur:
            plan.append(cur)
        # the last bucket's all-reduce cannot overlap any backward kernel: keep only the final ~1 MiB of gradients (the
        # shallow layers, produced last) in it and let the rest go out one bucket earlier
        if plan:
            tail, size = [], 0
            while len(plan[-1]) > 1 and (size + plan[-1][-1][2]) * 4 <= (1 << 20):
                idx, _, n = plan[-1].pop()
                tail.insert(0, (idx, n))
                size += n
            if tail and plan[-1]:
                off, bucket = 0, []
                for idx, n in tail:
                    bucket.append((idx, off, n))
                    off += n
                plan.append(bucket)
            elif tail:  # everything was small: put it back
                off = 0
                for idx, n in tail:
                    plan[-1].append((idx, off, n))
                    off += n
        self._plan = plan
        self._where = {idx: (b, off) for b, bucket in enumerate(plan) for idx, off, _ in bucket}
        dev = self.params[0].device
        self._flat = [torch.zeros(sum(n for _, _, n in bucket), dtype=torch.float32, device=dev) for bucket in plan]
        # reduced-precision transport: the collective runs on a bf16 image of the bucket
        self._wire = [torch.zeros_like(f, dtype=self.comm_dtype) for f in self._flat] if self.comm_dtype != torch.float32 else None
        # every parameter's slice of its bucket, flat and in the parameter's shape, made ONCE: slicing + view_as per parameter and
        # step was ~1 ms of host time for EquiUnetASSPEvo's 168 parameters (scripts/host_calls.py)
        self._dst = {idx: self._flat[b][off:off + n] for b, bucket in enumerate(plan) for idx, off, n in bucket}
        self._gview = {idx: d.view_as(self.params[idx]) for idx, d in self._dst.items()}

    def _start(self):
        self._filled = [0] * len(self._plan)
        self._handles = [None] * len(self._plan)
        self._late = set()
        self._copies = [([], []) for _ in self._plan]

    @contextlib.contextmanager
    def no_sync(self):
        """Gradient accumulation (the reference's --gradient_accumulation_iter, learning/engine.py:119-130): backward
        passes inside the block only accumulate into p.grad locally; the first backward outside it averages the
        accumulated total (like torch DDP's no_sync)."""
        old, self._sync = self._sync, False
        try:
            yield
        finally:
            self._sync = old

    def _launch(self, b):
        if self._copies is not None and self._copies[b][0]:
            # the small gradients of this bucket (norm weights, biases, heads): ONE multi-tensor copy instead of a copy_ launch
            # per parameter; the weight gradients (> 99 % of the bytes) were written in place by their kernels (dest())
            torch._foreach_copy_(self._copies[b][0], self._copies[b][1])
            self._copies[b] = ([], [])
        if self.world > 1 or (self.force_collectives and dist.is_initialized()):
            buf = self._flat[b]
            if self._wire is not None:
                buf = self._wire[b]
                buf.copy_(self._flat[b])  # f32 -> bf16, one pass over the bucket on the compute stream
            self._handles[b] = dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    # -- producer side ----------------------------------------------------------------------------
    def dest(self, param_index):
        """Where parameter #param_index's gradient will live: its contiguous f32 slice of the bucket, for kernels that can
        write their result anywhere (the weight-gradient slab reduce: ops.conv3d_wgrad(..., out=)) -- push() then finds the
        gradient already in place and copies nothing.  None while that is not possible (first step: no plan yet; no_sync;
        gradient accumulation into an existing p.grad)."""
        if not self._sync or self._plan is None or param_index not in self._where:
            return None
        if self.params[param_index].grad is not None:
            return None
        b, off = self._where[param_index]
        if self._handles and self._filled is not None and self._handles[b] is not None:
            return None  # (this step's collective on that bucket is already in flight)
        return self._dst[param_index]

    def push(self, param_index, grad):
        """Called by the backward program as soon as parameter #param_index's gradient exists."""
        if not self._sync:
            return
        if self._plan is None:
            self._order.append(param_index)  # first step: learn the production order, reduce in finish()
            return
        if self._filled is None:
            self._start()
        b, off = self._where[param_index]
        if self.params[param_index].grad is not None:
            # an older gradient is still there (micro-batch accumulation, or zero_grad(set_to_none=False)): autograd will
            # ADD this step's gradient to it after the backward program returns, so what has to be averaged is p.grad
            # as it stands in finish(), not `grad` -- this bucket is gathered and reduced there (no overlap, but right)
            self._late.add(b)
            return
        dst = self._dst[param_index]
        if grad.data_ptr() != dst.data_ptr():  # (a kernel that took dest() has written it in place)
            self._copies[b][0].append(dst)
            self._copies[b][1].append(grad.reshape(-1))
        self._filled[b] += 1
        if self._filled[b] == len(self._plan[b]) and b not in self._late:
            self._launch(b)

    # -- consumer side ----------------------------------------------------------------------------
    def finish(self):
        """After loss.backward(): make every p.grad the average over ranks (of the accumulated total when gradients
        were accumulated over several backward passes)."""
        if not self._sync:
            return
        if self._plan is None:
            self._order = list(dict.fromkeys(self._order))  # (accumulated micro-batches push every index repeatedly)
            seen = set(self._order)
            order = self._order + [i for i, p in enumerate(self.params) if i not in seen and p.grad is not None]
            if not order:
                order = [i for i, p in enumerate(self.params) if p.grad is not None]
            order = [i for i in order if self.params[i].grad is not None]
            self._build_plan(order)
            self._order = []
        if self._filled is None:
            self._start()
        for b, bucket in enumerate(self._plan):
            if self._filled[b] != len(bucket) or b in self._late:  # not (only) produced through push(): gather from p.grad
                if self._handles[b] is not None:  # an earlier micro-batch already sent this bucket off: let it land first
                    self._handles[b].wait()
                self._copies[b] = ([], [])  # (partial pushes of this bucket: superseded by the gather below)
                for idx, off, n in bucket:
                    g = self.params[idx].grad
                    dst = self._dst[idx]
                    if g.data_ptr() != dst.data_ptr():  # (p.grad may still be last step's view of this very bucket)
                        dst.copy_(g.reshape(-1))
                self._launch(b)
        ev = None
        if self.measure and torch.cuda.is_available():
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        for h in self._handles:
            if h is not None:
                h.wait()
        if ev is not None:
            ev[1].record()
            self.exposed.append(ev)
        if self.world > 1 or (self.force_collectives and dist.is_initialized()):
            if self._wire is not None:  # (what came back over the wire is the gradient, rounding included)
                torch._foreach_copy_(self._flat, self._wire)
            if self.world > 1:
                torch._foreach_mul_(self._flat, 1.0 / self.world)  # one multi-tensor launch instead of one mul per parameter
        params, gview = self.params, self._gview
        for idx, view in gview.items():
            # the averaged gradient is read in place from the bucket (contiguous f32 view made with the plan): no copy back
            params[idx].grad = view
        self._filled = None
        self._handles = []
        self._copies = None

    def payload_bytes(self):
        """Bytes every rank contributes to the all-reduces of one step."""
        esz = 4 if self.comm_dtype == torch.float32 else 2
        return sum(f.numel() for f in self._flat) * esz if self._flat else 0

    def exposed_ms(self):
        """Mean GPU time per measured step that finish() spent waiting for collectives (what the overlap did NOT hide)."""
        if not self.exposed:
            return None
        torch.cuda.synchronize()
        return sum(a.elapsed_time(b) for a, b in self.exposed) / len(self.exposed)

    def allreduce_ms(self, reps=5):
        """Blocking time of one step's collectives on scratch copies of the buckets (no compute beside them): the cost
        that overlap has to hide.  Call outside the timed region; every rank must call it."""
        if (self.world <= 1 and not (self.force_collectives and dist.is_initialized())) or not self._flat:
            return 0.0
        scratch = [torch.zeros_like(w) for w in (self._wire or self._flat)]
        on_gpu = scratch[0].is_cuda
        for s in scratch:
            dist.all_reduce(s, group=self.group)  # warm-up (connection set-up)
        if on_gpu:
            torch.cuda.synchronize()
        import time
        t0 = time.perf_counter()
        for _ in range(reps):
            hs = [dist.all_reduce(s, group=self.group, async_op=True) for s in scratch]
            for h in hs:
                h.wait()
        if on_gpu:
            torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3
