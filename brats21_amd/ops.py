"""Raw (non-autograd) Python entry points over the C ABI of libbrats_hip.so.

Activations are torch CUDA tensors of shape [N, D, H, W, C] (NDHWC) in torch.bfloat16 or
torch.float32; a tensor may be a channel-slice *view* of a wider buffer (pitch = stride(3)), which
is how torch.cat (networks/equiunet2020.py:478-486 of the reference) is removed.  PyTorch is only
the allocator / stream provider here; every arithmetic op is a HIP kernel of the library.
"""
import contextlib
import weakref

import numpy as np
import torch

from . import _lib
from ._lib import ACT_LEAKY, ACT_NONE, ACT_RELU, BF16, F16, F32, PACK_DGRAD, PACK_FWD, X3_BF16, X3_F16  # noqa: F401

ACTS = {"none": ACT_NONE, "relu": ACT_RELU, "leakyrelu": ACT_LEAKY, "elu": _lib.ACT_ELU, "swish": _lib.ACT_SWISH,
        "mish": _lib.ACT_MISH}


class KernelTimer:
    """Optional HIP-event timing of individual library launches on the launch stream (bench.py's
    live roofline measurement).  ops.TIMER = KernelTimer() enables it; None (default) costs nothing."""

    def __init__(self):
        self.records = {}  # key -> list of (start_event, end_event)

    def span(self, key):
        return _Span(self, key)

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for key, evs in self.records.items():
            ms = [a.elapsed_time(b) for a, b in evs]
            out[key] = (len(ms), sum(ms) / len(ms), sum(ms))
        return out


class _Span:
    def __init__(self, timer, key):
        self.timer, self.key = timer, key

    def __enter__(self):
        self.a = torch.cuda.Event(enable_timing=True)
        self.b = torch.cuda.Event(enable_timing=True)
        self.a.record()  # current stream == the stream the library launches on (_stream())

    def __exit__(self, *exc):
        self.b.record()
        self.timer.records.setdefault(self.key, []).append((self.a, self.b))


class _NoSpan:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


TIMER = None
_NOSPAN = _NoSpan()


def _span(*key):
    return TIMER.span(key) if TIMER is not None else _NOSPAN


_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_GET_DEVICE = getattr(torch._C, "_cuda_getDevice", None)


def _stream():
    """The current HIP stream of the current device as an integer handle.  torch.cuda.current_stream().cuda_stream builds a Python
    Stream object per call (~3 us, ~250 calls per training step: 0.7 ms of host time); the raw accessor is what torch's own
    generated code uses."""
    if _RAW_STREAM is not None and _GET_DEVICE is not None:
        return _RAW_STREAM(_GET_DEVICE())
    return torch.cuda.current_stream().cuda_stream


_SIDE_STREAMS = {}


def get_side_stream(device):
    """One extra HIP stream per device for work nobody waits for until the end of the backward program."""
    key = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
    st = _SIDE_STREAMS.get(key)
    if st is None:
        st = _SIDE_STREAMS[key] = torch.cuda.Stream(device=key)
    return st


@contextlib.contextmanager
def side_stream(side, *inputs):
    """Run the block on `side` after everything enqueued so far on the current stream (fork); the caller joins once, at the
    end.  `inputs` were allocated on the current stream and are read on the side stream: the caching allocator is told
    (record_stream) so that their memory is not handed out again before the side stream is done with it.  The block
    registers its results through the yielded function: they are consumed on the main stream after the join.  With
    side=None the block simply runs in line."""
    if side is None or TIMER is not None:  # (per-kernel HIP-event timing wants kernels one after the other)
        yield lambda *t: None
        return
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    with torch.cuda.stream(side):
        def produced(*tensors):
            for t in tensors:
                if t is not None:
                    t.record_stream(main)
        yield produced
    for t in inputs:
        if t is not None:
            t.record_stream(side)


# ---- split precision ("x3"): f32 tensors, 3x3x3 convolutions on three 16-bit MFMA products (csrc/conv_igemm_x3.hpp) ----
X3F = "x3_f16"    # operands split into fp16 pairs: f32-class products inside fp16's range (the forward pass)
X3B = "x3_bf16"   # ... into bf16 pairs: 2^-16 per product over f32's whole range (the gradients)
_X3 = None        # the split the 3x3x3 convolutions of f32 tensors use inside a split_precision() block; None = exact f32


@contextlib.contextmanager
def split_precision(mode):
    """Inside the block conv3d / conv3d_wgrad / pack_weights run f32 tensors' 3x3x3 convolutions on the split-precision
    kernels (mode = ops.X3F or ops.X3B; None = the exact-f32 MFMA kernels).  The network programs open it around their
    forward (X3F) and backward (X3B) when model.precision = "x3"."""
    global _X3
    if mode not in (None, X3F, X3B):
        raise _lib.BratsHipError(f"split_precision: unknown mode {mode!r}")
    old, _X3 = _X3, mode
    try:
        yield
    finally:
        _X3 = old


def x3_active():
    return _X3 is not None


def x3_mode():
    return _X3


def _conv_dtype(dtype, ksize):
    """The kernel family a convolution of `dtype` tensors runs on: the dtype itself, or the active split mode."""
    if _X3 is not None and dtype == torch.float32 and ksize == 3:
        return _X3
    return dtype


def _code(dtype):
    if dtype == X3F:
        return X3_F16
    if dtype == X3B:
        return X3_BF16
    if dtype == torch.bfloat16:
        return BF16
    if dtype == torch.float32:
        return F32
    if dtype == torch.float16:
        return F16
    raise _lib.BratsHipError(f"unsupported activation dtype {dtype}")


def is16(dtype):
    """16-bit storage (bf16 or fp16): the MFMA kernels with f32 accumulation; fp16 is the reference's autocast dtype."""
    return dtype in (torch.bfloat16, torch.float16)


def _fn16(name, dtype):
    """Entry points that move 16-bit data but have no dtype argument exist twice: brats_x (bf16) and brats_x_f16."""
    return getattr(_lib.lib(), name + ("_f16" if dtype == torch.float16 else ""))


def _desc(t):
    """(data_ptr, C, pitch) of an NDHWC tensor or channel-slice view."""
    if t.dim() != 5 or not t.is_cuda:
        raise _lib.BratsHipError("expected a CUDA tensor of shape [N, D, H, W, C]")
    n, d, h, w, c = t.shape
    p = t.stride(3)
    if t.stride(4) != 1 or t.stride(2) != w * p or t.stride(1) != h * w * p or t.stride(0) != d * h * w * p:
        raise _lib.BratsHipError(f"tensor is not NDHWC with a channel pitch: shape {tuple(t.shape)} strides {t.stride()}")
    return t.data_ptr(), c, p


def _f32(t):
    if t is None:
        return None
    if t.dtype != torch.float32 or not t.is_contiguous() or not t.is_cuda:
        raise _lib.BratsHipError("expected a contiguous CUDA float32 tensor")
    return t.data_ptr()


def new_act(n, d, h, w, c, dtype, device):
    return torch.empty((n, d, h, w, c), dtype=dtype, device=device)


# ------------------------------------------------------------------------------------------ layout
def ncdhw_to_ndhwc(x, dtype, cpad=None):
    """[N,C,D,H,W] f32 -> [N,D,H,W,cpad] dtype (extra channels zero)."""
    n, c, d, h, w = x.shape
    cpad = cpad or c
    x = x.contiguous().float()
    out = new_act(n, d, h, w, cpad, dtype, x.device)
    _lib.check(_lib.lib().brats_ncdhw_to_ndhwc(x.data_ptr(), out.data_ptr(), _code(dtype), n, c, cpad, cpad, d, h, w,
                                               _stream()), "ncdhw_to_ndhwc")
    return out


def ndhwc_to_ncdhw(x):
    ptr, c, p = _desc(x)
    n, d, h, w, _ = x.shape
    out = torch.empty((n, c, d, h, w), dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().brats_ndhwc_to_ncdhw(ptr, p, out.data_ptr(), _code(x.dtype), n, c, d, h, w, _stream()),
               "ndhwc_to_ncdhw")
    return out


# ------------------------------------------------------------------------------------------ conv
def conv_chunk(dtype, ksize, dil, c1, c2=0, cout=0):
    """K chunk of the kernel that will run a layer with c1 (+c2) input channels and `cout` GEMM rows."""
    ck = _lib.lib().brats_conv3d_chunk(_code(dtype), ksize, dil, c1, c2, cout)
    if ck <= 0:
        raise _lib.BratsHipError(f"conv3d: no channel chunk for c1={c1} c2={c2} dtype={dtype}")
    return ck


_PACK_CACHE = {}  # inference only: packed weights keyed by (tensor object, layout arguments); validated by the version
# counter, the address AND the generation below (writes through ``p.data`` -- the reference's Ranger2020 updates its
# weights with ``p.data.copy_``, learning/optimizer.py:243,253 -- do not bump a parameter's version counter)
_PACK_GEN = 0
_PACK_KEEP = None  # list collecting every packed buffer handed out while a hipGraph of the forward is being built


def invalidate_packed_weights():
    """Drop every cached packed-weight buffer.  Called by the network modules whenever their weights may have changed
    without a version bump: on every grad-enabled (training) forward and on every train() / eval() transition.  Call it
    yourself after writing weights through ``p.data`` between two no_grad forwards of an eval-mode model."""
    global _PACK_GEN
    _PACK_GEN += 1
    _PACK_CACHE.clear()


def pack_generation():
    return _PACK_GEN


@contextlib.contextmanager
def keep_packed(sink):
    """Every buffer pack_weights() / pack_weights_f8() returns inside the block is appended to ``sink``: a captured
    hipGraph bakes the buffers' addresses in, so its owner must keep them alive (inferers.GraphedPredictor)."""
    global _PACK_KEEP
    old, _PACK_KEEP = _PACK_KEEP, sink
    try:
        yield sink
    finally:
        _PACK_KEEP = old


def _cache_get(key, w):
    hit = _PACK_CACHE.get(key)
    if hit is not None and hit[0]() is w and hit[1] == w._version and hit[2] == w.data_ptr() and hit[4] == _PACK_GEN:
        if _PACK_KEEP is not None:
            _PACK_KEEP.append(hit[3])
        return hit[3]
    return None


def _cache_put(key, w, packed):
    if len(_PACK_CACHE) >= 1024:
        _PACK_CACHE.clear()
    _PACK_CACHE[key] = (weakref.ref(w), w._version, w.data_ptr(), packed, _PACK_GEN)
    if _PACK_KEEP is not None:
        _PACK_KEEP.append(packed)


_PACK_JOB = np.dtype([("w", "<u8"), ("out", "<u8"), ("dtype", "<i4"), ("mode", "<i4"), ("taps", "<i4"), ("cin_w", "<i4"),
                      ("cin_real", "<i4"), ("cin_off", "<i4"), ("rows", "<i4"), ("rows16", "<i4"), ("kdim", "<i4"), ("ck", "<i4"),
                      ("ms_n", "<i4"), ("reserved", "<i4"), ("total", "<u8")])  # == brats_pack_job (include/brats_hip.h)


class PackPlan:
    """Every packed-weight buffer a training step needs (each layer: forward and dgrad layout) as views of ONE allocation,
    filled by ONE launch per step (brats_conv3d_pack_weights_multi) instead of one ~9 us launch per layer and layout.

    The first step runs the ordinary per-layer path and records what was asked for; ``run()`` (called by the module at the
    start of a training forward) then builds the tables once and re-packs everything each step.  ``pack_weights`` hands
    out a view only while the parameter is the same object, at the same address, with the version counter ``run()`` saw;
    anything else falls back to the per-layer path (and is added to the plan)."""

    def __init__(self):
        self.recorded = {}   # key -> (weakref(w), args)
        self.entries = None  # key -> [weakref(w), data_ptr, offset, nbytes, version]
        self.dirty = True

    def record(self, key, w, args):
        if isinstance(w, torch.nn.Parameter) and key not in self.recorded:  # (temporaries would never hit again)
            self.recorded[key] = (weakref.ref(w), args)
            self.dirty = True

    def _build(self, device):
        # one job / block table per library flavour: fp16 jobs go to brats_conv3d_pack_weights_multi_f16 (where the job's
        # dtype code BF16 means "the 16-bit type"), everything else to the plain entry point
        jobs, blocks, entries, off = {False: [], True: []}, {False: [], True: []}, {}, 0
        for key, (wref, (dtype, mode, cin_pad, cin_off, cin_cnt, dil, c1)) in self.recorded.items():
            w = wref()
            if w is None or w.dtype != torch.float32 or not w.is_contiguous():
                continue
            cout_w, cin_real, k = w.shape[0], w.shape[1], w.shape[2]
            cin_w = cin_pad if cin_pad is not None else cin_real
            cnt = cin_w - cin_off if cin_cnt is None else cin_cnt
            kdim, rows = (cnt, cout_w) if mode == PACK_FWD else (cout_w, cnt)
            ck = conv_chunk(dtype, k, dil, kdim, 0, rows) if (c1 is None or mode != PACK_FWD) else conv_chunk(dtype, k, dil, c1, kdim - c1, rows)
            code = _code(dtype)
            nbytes = _lib.lib().brats_conv3d_packed_bytes(code, k, kdim, rows, ck)
            rows16 = (rows + 15) // 16
            x3 = code in (X3_BF16, X3_F16)  # (hi and lo fragments side by side)
            ms_n = nbytes // ((kdim // ck) * rows16 * (2048 if x3 else 1024))
            total = nbytes // (4 if code == F32 else 2)
            entries[key] = [wref, w.data_ptr(), off, nbytes, -1]
            fl = code in (F16, X3_F16)
            jobs[fl].append((w.data_ptr(), off, {F16: BF16, X3_F16: X3_BF16}.get(code, code), mode, k ** 3, cin_w, cin_real, cin_off,
                             rows, rows16, kdim, ck, ms_n, 0, total))
            j = len(jobs[fl]) - 1
            blocks[fl].extend((j, b) for b in range(_lib.lib().brats_conv3d_pack_blocks(kdim, ck, rows)))
            off += (nbytes + 255) // 256 * 256
        self.buf = torch.empty(max(off, 256), dtype=torch.uint8, device=device)
        self.tables = []
        for fl in (False, True):
            if not jobs[fl]:
                continue
            rec = np.array(jobs[fl], dtype=_PACK_JOB)
            rec["out"] += self.buf.data_ptr()
            self.tables.append((fl, torch.from_numpy(rec.view(np.uint8).copy()).to(device),
                                torch.tensor(blocks[fl], dtype=torch.int32, device=device).reshape(-1, 2).contiguous()))
        self.entries = entries
        self.dirty = False

    def run(self, device):
        if not self.recorded:
            return
        if self.entries is not None and not self.dirty:
            for e in self.entries.values():  # a parameter that was re-allocated (.to(), load with assign) moves: rebuild
                w = e[0]()
                if w is None or w.data_ptr() != e[1]:
                    self.dirty = True
                    break
        if self.dirty or self.entries is None:
            self._build(device)
        for fl, jobs, blocks in self.tables:
            fn = _lib.lib().brats_conv3d_pack_weights_multi_f16 if fl else _lib.lib().brats_conv3d_pack_weights_multi
            _lib.check(fn(jobs.data_ptr(), blocks.data_ptr(), blocks.shape[0], _stream()), "conv3d_pack_weights_multi")
        for e in self.entries.values():
            e[4] = e[0]()._version

    def lookup(self, key, w):
        e = self.entries.get(key) if self.entries is not None else None
        if e is not None and e[0]() is w and e[1] == w.data_ptr() and e[4] == w._version:
            return self.buf[e[2]:e[2] + e[3]]
        return None


ACTIVE_PLAN = None  # set by the network programs around a training forward / backward
_PLANS = weakref.WeakKeyDictionary()  # module -> PackPlan (kept off the module: deepcopy / state_dict stay plain)


@contextlib.contextmanager
def use_plan(plan):
    global ACTIVE_PLAN
    old, ACTIVE_PLAN = ACTIVE_PLAN, plan
    try:
        yield
    finally:
        ACTIVE_PLAN = old


def plan_for(module, device):
    """Re-pack every recorded weight of `module` (one launch) and return its plan; call at the start of a training forward."""
    plan = _PLANS.get(module)
    if plan is None:
        plan = _PLANS[module] = PackPlan()
    plan.run(device)
    return plan


def pack_weights(w, dtype, mode, cin_pad=None, cin_off=0, cin_cnt=None, dil=1, c1=None):
    """w: torch-layout [Cout, Cin, k, k, k] f32 parameter -> packed MFMA-fragment buffer.
    mode PACK_FWD: GEMM rows = Cout, K = Cin (zero-padded to cin_pad); PACK_DGRAD: rows = Cin slice,
    K = Cout, taps flipped.

    Under ``torch.no_grad()`` (sliding-window / TTA inference: 144 forwards of the same weights per volume) the
    packed buffer is cached.  A hit needs the same tensor object, address, version counter (bumped by in-place updates:
    torch optimizers, brats21_amd.optim.Ranger2020, load_state_dict, SWA averaging) and cache generation
    (``invalidate_packed_weights``: every training forward and train()/eval() switch of the modules -- that covers
    optimizers that write through ``p.data``, like the reference's Ranger2020)."""
    key = None
    dtype = _conv_dtype(dtype, w.shape[2])
    if ACTIVE_PLAN is not None:
        pkey = (id(w), str(dtype), mode, cin_pad, cin_off, cin_cnt, dil, c1)
        v = ACTIVE_PLAN.lookup(pkey, w)
        if v is not None:
            return v
        ACTIVE_PLAN.record(pkey, w, (dtype, mode, cin_pad, cin_off, cin_cnt, dil, c1))
    if not torch.is_grad_enabled():
        # keyed by the tensor OBJECT (weak reference: a new tensor that reuses a freed address must not hit) and its
        # version counter
        key = (id(w), str(dtype), mode, cin_pad, cin_off, cin_cnt, dil, c1)
        hit = _cache_get(key, w)
        if hit is not None:
            return hit
    packed = _pack_weights(w, dtype, mode, cin_pad, cin_off, cin_cnt, dil, c1)
    if key is not None:
        _cache_put(key, w, packed)
    return packed


def _pack_weights(w, dtype, mode, cin_pad, cin_off, cin_cnt, dil, c1):
    cout_w, cin_w, k = w.shape[0], w.shape[1], w.shape[2]
    w = w.detach()
    if cin_pad is not None and cin_pad != cin_w:
        wp = torch.zeros((cout_w, cin_pad, k, k, k), dtype=torch.float32, device=w.device)
        wp[:, :cin_w] = w
        w, cin_w = wp, cin_pad
    w = w.contiguous().float()
    cin_cnt = cin_w - cin_off if cin_cnt is None else cin_cnt
    code = _code(dtype)
    if mode == PACK_FWD:
        kdim, rows = cin_cnt, cout_w
    else:
        kdim, rows = cout_w, cin_cnt
    # the K chunk must be the one the kernel will pick for the (possibly two-source) input it reads
    ck = conv_chunk(dtype, k, dil, kdim, 0, rows) if (c1 is None or mode != PACK_FWD) else conv_chunk(dtype, k, dil, c1, kdim - c1, rows)
    nbytes = _lib.lib().brats_conv3d_packed_bytes(code, k, kdim, rows, ck)
    packed = torch.empty(nbytes, dtype=torch.uint8, device=w.device)
    _lib.check(_lib.lib().brats_conv3d_pack_weights(w.data_ptr(), packed.data_ptr(), code, mode, k, cout_w, cin_w,
                                                    cin_off, cin_cnt, ck, _stream()), "conv3d_pack_weights")
    return packed


def set_kp(mode):
    """brats_conv3d_set_kp: 1 / 0 = the 8-wave K-parity form of the small-grid 3x3x3 launches on / off, -1 = default.  The packed
    weights do not depend on it.  Returns the previous setting."""
    return _lib.lib().brats_conv3d_set_kp(mode)


def set_vs8(mode):
    """brats_conv3d_set_vs8 + invalidation of everything packed under the old setting (the switch changes the K chunk, i.e.
    the packed-weight layout of the layers it applies to).  Returns the previous setting."""
    old = _lib.lib().brats_conv3d_set_vs8(mode)
    invalidate_packed_weights()
    for plan in list(_PLANS.values()):
        plan.dirty = True
        plan.entries = None
    return old


def set_x3_wgrad_fused(mode):
    """brats_conv3d_set_x3_wgrad_fused: 1 = the fused split-precision weight gradient where it is built and pays, 2 = for any tile
    count (tests), 0 = the split pass + three 16-bit launches everywhere, -1 = default.  Returns the previous setting."""
    return _lib.lib().brats_conv3d_set_x3_wgrad_fused(mode)


def split_granule(cout):
    return _lib.lib().brats_conv3d_split_granule(cout)


def tiles_per_sample(d, h, w):
    return _lib.lib().brats_conv3d_tiles_per_sample(d, h, w)


class Pending:
    """z = act(y * scale + shift) that has NOT been stored: the raw convolution output `y` of a ConvBnRelu unit plus what
    its consumer needs to apply the normalisation + activation on load (conv3d(pre=...), include/brats_hip.h:
    brats_conv3d_fwd_pre), or to materialise it after all (``materialize()`` = the affine_act pass)."""

    def __init__(self, y, scale_shift, act, slope=0.01):
        self.y, self.scale_shift, self.act, self.slope = y, scale_shift, act, slope
        self.shape, self.dtype, self.device = y.shape, y.dtype, y.device

    def materialize(self):
        return affine_act(self.y, self.scale_shift, self.act, slope=self.slope)


def conv_pre_ok(dtype, ksize, dil, c1, c2, cout):
    """Is the normalise-on-load form of the convolution built for this layer?  (16-bit, 3x3x3, dilation 1, Cout % 48 == 0)"""
    return is16(dtype) and bool(_lib.lib().brats_conv3d_pre_ok(_code(dtype), ksize, dil, c1, c2, cout))


def conv3d(x, packed_w, cout, ksize=3, dil=1, bias=None, out=None, want_stats=False, x2=None, split=None, amax=None):
    if isinstance(x, Pending) or isinstance(x2, Pending):
        return _conv3d_pre(x, packed_w, cout, ksize, dil, bias, out, want_stats, x2)
    return _conv3d(x, packed_w, cout, ksize, dil, bias, out, want_stats, x2, split, amax)


def _conv3d_pre(x, packed_w, cout, ksize, dil, bias, out, want_stats, x2):
    """conv3d whose input(s) are Pending activations: normalise + act applied while the halo tile is staged."""
    p1 = x if isinstance(x, Pending) else None
    p2 = x2 if isinstance(x2, Pending) else None
    xa = p1.y if p1 is not None else x
    xb = p2.y if p2 is not None else x2
    if p1 is not None and p2 is not None and (p1.act != p2.act or p1.slope != p2.slope):
        raise _lib.BratsHipError("conv3d: the two pending inputs carry different activations")
    act, slope = (p1 or p2).act, (p1 or p2).slope
    ptr, c, p = _desc(xa)
    n, d, h, w, _ = xa.shape
    ptr2, c2, pp2 = (None, 0, 0)
    if xb is not None:
        ptr2, c2, pp2 = _desc(xb)
    if out is None:
        out = new_act(n, d, h, w, cout, xa.dtype, xa.device)
    optr, oc, op = _desc(out)
    if oc != cout or out.dtype != xa.dtype:
        raise _lib.BratsHipError("conv3d: bad output tensor")
    stats = torch.empty((n, tiles_per_sample(d, h, w), cout, 2), dtype=torch.float32, device=xa.device) if want_stats else None
    with _span("conv_igemm", c + c2, cout, ksize, dil, n, d, h, w, str(xa.dtype)):
        _lib.check(_lib.lib().brats_conv3d_fwd_pre(ptr, c, p, p1.scale_shift.data_ptr() if p1 is not None else None, ptr2, c2, pp2,
                                                   p2.scale_shift.data_ptr() if p2 is not None else None, ACTS[act], float(slope),
                                                   packed_w.data_ptr(), _f32(bias), optr, op,
                                                   stats.data_ptr() if stats is not None else None, _code(xa.dtype), dil, n, d, h, w,
                                                   cout, _stream()), "conv3d_fwd_pre")
    return out, stats


def _conv3d(x, packed_w, cout, ksize=3, dil=1, bias=None, out=None, want_stats=False, x2=None, split=None, amax=None):
    """y[N,D,H,W,cout] = conv([x | x2]) with weights from pack_weights().  Returns (y, stats|None) where
    stats = [N, tiles, cout, 2] per-tile per-channel (sum, sum of squares) of the f32 result.
    x2: optional second input (virtual concat, no torch.cat).  split: write output channels
    [0, split) and [split, cout) to two dense tensors (returns (y, y2) as y).
    amax (split precision only): 1-element f32 device tensor holding max|x| -- the input is scaled into fp16's range by the
    matching power of two and the result scaled back (brats_conv3d_x3_fwd: the input gradient, whose x = dY is tiny)."""
    ptr, c, p = _desc(x)
    n, d, h, w, _ = x.shape
    ptr2, c2, p2 = (None, 0, 0)
    if x2 is not None:
        ptr2, c2, p2 = _desc(x2)
    y2 = None
    if split is not None:
        out = new_act(n, d, h, w, split, x.dtype, x.device)
        y2 = new_act(n, d, h, w, cout - split, x.dtype, x.device)
    elif out is None:
        out = new_act(n, d, h, w, cout, x.dtype, x.device)
    optr, oc, op = _desc(out)
    if (y2 is None and oc != cout) or out.dtype != x.dtype:
        raise _lib.BratsHipError("conv3d: bad output tensor")
    stats = None
    if want_stats:
        stats = torch.empty((n, tiles_per_sample(d, h, w), cout, 2), dtype=torch.float32, device=x.device)
    kd = _conv_dtype(x.dtype, ksize)  # (split precision: f32 tensors, weights packed under the same mode)
    if amax is not None and kd in (X3F, X3B):
        with _span("conv_igemm", c + c2, cout, ksize, dil, n, d, h, w, str(kd)):
            _lib.check(_lib.lib().brats_conv3d_x3_fwd(ptr, c, p, ptr2, c2, p2, _f32(amax), packed_w.data_ptr(), _f32(bias), optr, op,
                                                      y2.data_ptr() if y2 is not None else None,
                                                      (cout - split) if y2 is not None else 0, split or 0,
                                                      stats.data_ptr() if stats is not None else None, _code(kd), dil, n, d, h, w,
                                                      cout, _stream()), "conv3d_x3_fwd")
        return ((out, y2) if y2 is not None else out), stats
    with _span("conv_igemm", c + c2, cout, ksize, dil, n, d, h, w, str(kd)):
        _lib.check(_lib.lib().brats_conv3d_fwd(ptr, c, p, ptr2, c2, p2, packed_w.data_ptr(), _f32(bias), optr, op,
                                               y2.data_ptr() if y2 is not None else None,
                                               (cout - split) if y2 is not None else 0, split or 0,
                                               stats.data_ptr() if stats is not None else None, _code(kd), ksize,
                                               dil, n, d, h, w, cout, _stream()), "conv3d_fwd")
    return ((out, y2) if y2 is not None else out), stats


# ------------------------------------------------------------------------------------------ fp8 conv
def absmax(x):
    """max|x| of an NDHWC tensor as a 1-element f32 device tensor (no host sync)."""
    ptr, c, p = _desc(x)
    out = torch.empty(1, dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().brats_absmax(ptr, p, _code(x.dtype), x.shape[0] * x.shape[1] * x.shape[2] * x.shape[3], c,
                                       out.data_ptr(), _stream()), "absmax")
    return out


F8_MIN_SIZE = 0  # e4m3 forward convolutions only at levels whose depth is >= this many voxels (0 = every level); set_f8_min_size()


def set_f8_min_size(n):
    """Restrict model.conv_fp8's FORWARD convolutions to the levels with a depth of at least n voxels (64: the 128^3 and 64^3
    levels of a 128^3 patch; 0 = all levels, the default).  An experiment knob for VERDICT r5 item 7 (does e4m3 hold the Dice bar
    when only the large, statistics-rich levels use it?); the backward kernels do not look at it.  Returns the previous value."""
    global F8_MIN_SIZE
    old, F8_MIN_SIZE = F8_MIN_SIZE, int(n)
    return old


def conv_f8_chunk(c1, c2=0):
    """Channel chunk of the fp8 kernel for an input of c1 (+c2) channels; 0 = not supported (use conv3d)."""
    return _lib.lib().brats_conv3d_f8_chunk(c1, c2)


def pack_weights_f8(w, mode, cin_pad=None, cin_off=0, cin_cnt=None, c1=None):
    """e4m3 fragments + per-row power-of-two scales for conv3d_f8 (3x3x3 only); arguments as pack_weights()."""
    key = None
    if not torch.is_grad_enabled():
        key = (id(w), "f8", mode, cin_pad, cin_off, cin_cnt, 0, c1)
        hit = _cache_get(key, w)
        if hit is not None:
            return hit
    w0 = w
    cout_w, cin_w, k = w.shape[0], w.shape[1], w.shape[2]
    if k != 3:
        raise _lib.BratsHipError("pack_weights_f8: 3x3x3 kernels only")
    w = w.detach()
    if cin_pad is not None and cin_pad != cin_w:
        wp = torch.zeros((cout_w, cin_pad, k, k, k), dtype=torch.float32, device=w.device)
        wp[:, :cin_w] = w
        w, cin_w = wp, cin_pad
    w = w.contiguous().float()
    cin_cnt = cin_w - cin_off if cin_cnt is None else cin_cnt
    kdim, rows = (cin_cnt, cout_w) if mode == PACK_FWD else (cout_w, cin_cnt)
    ck = conv_f8_chunk(kdim) if (c1 is None or mode != PACK_FWD) else conv_f8_chunk(c1, kdim - c1)
    if ck <= 0:
        raise _lib.BratsHipError(f"pack_weights_f8: K channels {kdim} (c1={c1}) are not multiples of 16")
    nbytes = _lib.lib().brats_conv3d_f8_packed_bytes(kdim, rows, ck)
    packed = torch.empty(nbytes, dtype=torch.uint8, device=w.device)
    _lib.check(_lib.lib().brats_conv3d_f8_pack_weights(w.data_ptr(), packed.data_ptr(), mode, cout_w, cin_w, cin_off,
                                                       cin_cnt, ck, _stream()), "conv3d_f8_pack_weights")
    if key is not None:
        _cache_put(key, w0, packed)
    return packed


def conv3d_f8(x, packed_w, cout, dil=1, bias=None, out=None, want_stats=False, x2=None, split=None, amax=None, amax2=None,
              xscale=None):
    """conv3d() with the e4m3 MFMA kernel (bf16 tensors in and out).  amax / amax2: 1-element f32 tensors holding
    max|x| / max|x2| (from affine_act / gn_act_bwd / absmax); when missing they are computed here (one extra pass)
    unless a static power-of-two ``xscale`` is given."""
    if not is16(x.dtype):
        raise _lib.BratsHipError("conv3d_f8: bf16 activations only")
    ptr, c, p = _desc(x)
    n, d, h, w, _ = x.shape
    ptr2, c2, p2 = (None, 0, 0)
    if x2 is not None:
        ptr2, c2, p2 = _desc(x2)
    if xscale is None:
        if amax is None:
            amax = absmax(x)
        if x2 is not None and amax2 is None:
            amax2 = absmax(x2)
    else:
        amax = amax2 = None
    y2 = None
    if split is not None:
        out = new_act(n, d, h, w, split, x.dtype, x.device)
        y2 = new_act(n, d, h, w, cout - split, x.dtype, x.device)
    elif out is None:
        out = new_act(n, d, h, w, cout, x.dtype, x.device)
    optr, oc, op = _desc(out)
    if (y2 is None and oc != cout) or out.dtype != x.dtype:
        raise _lib.BratsHipError("conv3d_f8: bad output tensor")
    stats = None
    if want_stats:
        stats = torch.empty((n, tiles_per_sample(d, h, w), cout, 2), dtype=torch.float32, device=x.device)
    with _span("conv_igemm_f8", c + c2, cout, 3, dil, n, d, h, w, "e4m3"):
        _lib.check(_fn16("brats_conv3d_f8_fwd", x.dtype)(ptr, c, p, _f32(amax), ptr2, c2, p2, _f32(amax2),
                                                  float(xscale) if xscale is not None else 0.0, packed_w.data_ptr(),
                                                  _f32(bias), optr, op, y2.data_ptr() if y2 is not None else None,
                                                  (cout - split) if y2 is not None else 0, split or 0,
                                                  stats.data_ptr() if stats is not None else None, dil, n, d, h, w, cout,
                                                  _stream()), "conv3d_f8_fwd")
    return ((out, y2) if y2 is not None else out), stats


def _grad_out(out, shape, device):
    """The f32 tensor a gradient kernel writes: `out` (e.g. the parameter's slice of a DDP bucket, ddp.GradientBuckets.dest)
    when it is given and fits, otherwise a fresh allocation."""
    if out is not None and out.numel() == int(np.prod(shape)) and out.dtype == torch.float32 and out.is_contiguous():
        return out.view(shape)
    return torch.empty(shape, dtype=torch.float32, device=device)


def conv3d_wgrad(x, dy, ksize=3, dil=1, want_dbias=False, x2=None, out=None, amax_dy=None):
    """dW [cout, cin (+cin2), k,k,k] f32 (and dbias) from the layer input [x | x2] and the output gradient dy.
    amax_dy (split precision only): 1-element f32 device tensor holding max|dy| (see conv3d's amax)."""
    ptr, c, p = _desc(x)
    ptr2, c2, p2 = (None, 0, 0)
    if x2 is not None:
        ptr2, c2, p2 = _desc(x2)
    dptr, cout, dp = _desc(dy)
    n, d, h, w, _ = x.shape
    kd = _conv_dtype(x.dtype, ksize)
    if kd in (X3F, X3B) and (c % 8 or c2 % 8 or cout % 8):
        kd = x.dtype  # (narrow test widths: the 16-bit kernels behind the split form want multiples of 8 channels)
    code = _code(kd)
    nbytes = _lib.lib().brats_conv3d_wgrad_ws_bytes(code, ksize, n, d, h, w, c, c2, cout)
    ws = torch.empty(max(nbytes, 4) // 4, dtype=torch.float32, device=x.device)
    dw = _grad_out(out, (cout, c + c2, ksize, ksize, ksize), x.device)
    db = torch.empty(cout, dtype=torch.float32, device=x.device) if want_dbias else None
    if amax_dy is not None and kd in (X3F, X3B):
        with _span("conv_wgrad", c + c2, cout, ksize, dil, n, d, h, w, str(kd)):
            _lib.check(_lib.lib().brats_conv3d_x3_wgrad(ptr, c, p, ptr2, c2, p2, dptr, dp, _f32(amax_dy), ws.data_ptr(), dw.data_ptr(),
                                                        db.data_ptr() if db is not None else None, code, dil, n, d, h, w, cout,
                                                        _stream()), "conv3d_x3_wgrad")
        return dw, db
    with _span("conv_wgrad", c + c2, cout, ksize, dil, n, d, h, w, str(kd)):
        _lib.check(_lib.lib().brats_conv3d_wgrad(ptr, c, p, ptr2, c2, p2, dptr, dp, ws.data_ptr(), dw.data_ptr(),
                                                 db.data_ptr() if db is not None else None, code, ksize, dil, n, d, h, w,
                                                 cout, _stream()), "conv3d_wgrad")
    return dw, db


def conv3d_wgrad_f8_ok(x, dy, x2=None):
    """Is the e4m3 weight-gradient kernel built for this layer (3x3x3, dilation 1, 48 x 48 / 64 x 32 channel blocks)?"""
    n, d, h, w, c = x.shape
    return (is16(x.dtype) and dy.dtype == x.dtype and
            _lib.lib().brats_conv3d_wgrad_f8_ws_bytes(n, d, h, w, c, x2.shape[-1] if x2 is not None else 0, dy.shape[-1]) > 0)


def conv3d_wgrad_f8(x, dy, amax, amax_dy, x2=None, amax2=None, out=None):
    """dW [cout, cin (+cin2), 3, 3, 3] f32 of a dilation-1 layer with X and dY rounded to e4m3 on the way into the MFMA
    (v_mfma_scale_f32_16x16x128_f8f6f4).  amax* : 1-element f32 device tensors holding max|x|, max|x2|, max|dy| (recorded
    by the kernels that produced the tensors; absmax() otherwise).  Only where conv3d_wgrad_f8_ok()."""
    ptr, c, p = _desc(x)
    ptr2, c2, p2 = (None, 0, 0)
    if x2 is not None:
        ptr2, c2, p2 = _desc(x2)
    dptr, cout, dp = _desc(dy)
    n, d, h, w, _ = x.shape
    if not is16(x.dtype) or dy.dtype != x.dtype:
        raise _lib.BratsHipError("conv3d_wgrad_f8: bf16 activations only")
    nbytes = _lib.lib().brats_conv3d_wgrad_f8_ws_bytes(n, d, h, w, c, c2, cout)
    if nbytes == 0:
        raise _lib.BratsHipError(f"conv3d_wgrad_f8: not built for {c}+{c2} -> {cout} channels at {n}x{d}x{h}x{w} (use conv3d_wgrad)")
    ws = torch.empty(nbytes // 4, dtype=torch.float32, device=x.device)
    dw = _grad_out(out, (cout, c + c2, 3, 3, 3), x.device)
    with _span("conv_wgrad_f8", c + c2, cout, 3, 1, n, d, h, w, "e4m3"):
        _lib.check(_fn16("brats_conv3d_wgrad_f8", x.dtype)(ptr, c, p, _f32(amax), ptr2, c2, p2, _f32(amax2), dptr, dp, _f32(amax_dy),
                                                    ws.data_ptr(), dw.data_ptr(), n, d, h, w, cout, _stream()), "conv3d_wgrad_f8")
    return dw


# ------------------------------------------------------------------------------------------ GroupNorm + act
def gn_finalize(stats, n, c, groups, voxels, gamma, beta, eps=1e-5):
    mean_rstd = torch.empty((n, groups, 2), dtype=torch.float32, device=stats.device)
    scale_shift = torch.empty((n, c, 2), dtype=torch.float32, device=stats.device)
    chan = torch.empty(_lib.lib().brats_gn_ws_doubles(n, c), dtype=torch.float64, device=stats.device)
    _lib.check(_lib.lib().brats_gn_finalize(stats.data_ptr(), stats.shape[1], n, c, groups, float(voxels), eps,
                                            _f32(gamma), _f32(beta), mean_rstd.data_ptr(), scale_shift.data_ptr(),
                                            chan.data_ptr(), _stream()), "gn_finalize")
    return mean_rstd, scale_shift


def affine_act(y, scale_shift, act="relu", out=None, slope=0.01, amax=None, slope_t=None):
    """amax: optional zero-initialised 1-element f32 tensor that receives max|out| (scale source of conv3d_f8).
    slope_t: 1-element f32 device tensor overriding ``slope`` (nn.PReLU's learnable weight, act = "leakyrelu")."""
    ptr, c, p = _desc(y)
    n, d, h, w, _ = y.shape
    if out is None:
        out = new_act(n, d, h, w, c, y.dtype, y.device)
    optr, oc, op = _desc(out)
    _lib.check(_lib.lib().brats_affine_act_fwd(ptr, p, scale_shift.data_ptr(), optr, op, _code(y.dtype), ACTS[act], slope,
                                               _f32(slope_t), n, d * h * w, c, _f32(amax), _stream()), "affine_act_fwd")
    return out


def dropout(x, p, state, unit, out=None):
    """nn.Dropout(p) on an NDHWC activation (or, with the same arguments, on the gradient flowing back through it): x * keep /
    (1 - p) with keep a pure function of (state[0] = seed, state[1] = step counter, unit id, element index) -- brats_dropout.
    state: int64[2] DEVICE tensor.  out=x: in place."""
    ptr, c, pitch = _desc(x)
    n, d, h, w, _ = x.shape
    if state.dtype != torch.int64 or state.numel() != 2 or not state.is_cuda:
        raise _lib.BratsHipError("dropout: state must be an int64[2] device tensor (seed, step counter)")
    if out is None:
        out = new_act(n, d, h, w, c, x.dtype, x.device)
    optr, oc, op = _desc(out)
    if oc != c or out.dtype != x.dtype:
        raise _lib.BratsHipError("dropout: bad output tensor")
    _lib.check(_lib.lib().brats_dropout(ptr, pitch, optr, op, _code(x.dtype), n * d * h * w, c, float(p), state.data_ptr(), int(unit),
                                        _stream()), "dropout")
    return out


def affine_act_pool(y, scale_shift, act="relu", slope=0.01, amax=None, slope_t=None, with_avg=False, want_argmax=False):
    """(z, pooled) = (affine_act(y), maxpool2(z)) in one pass (include/brats_hip.h: brats_affine_act_pool_fwd).
    want_argmax: z._pool_argmax = the uint8 window index of every pooled element, which maxpool2_bwd(z, ...) then reads
    instead of the window itself."""
    ptr, c, p = _desc(y)
    n, d, h, w, _ = y.shape
    z = new_act(n, d, h, w, c, y.dtype, y.device)
    pooled = new_act(n, d // 2, h // 2, w // 2, c * (2 if with_avg else 1), y.dtype, y.device)
    idx = torch.empty((n, d // 2, h // 2, w // 2, c), dtype=torch.uint8, device=y.device) if want_argmax else None
    zp, _, zpitch = _desc(z)
    pp, _, ppitch = _desc(pooled)
    _lib.check(_lib.lib().brats_affine_act_pool_fwd(ptr, p, scale_shift.data_ptr(), zp, zpitch, pp, ppitch,
                                                    idx.data_ptr() if idx is not None else None, _code(y.dtype), ACTS[act],
                                                    slope, _f32(slope_t), n, d, h, w, c, int(with_avg), _f32(amax), _stream()),
               "affine_act_pool_fwd")
    if idx is not None:
        z._pool_argmax = idx
    return z, pooled


def gn_act_bwd(dz, y, scale_shift, mean_rstd, gamma, groups=8, act="relu", slope=0.01, amax=None, slope_t=None):
    """Returns (dy, dgamma, dbeta) for z = act(GroupNorm(y)); amax (optional, zero-initialised) receives max|dy|."""
    dzp, c, dzpitch = _desc(dz)
    yp, _, ypitch = _desc(y)
    n, d, h, w, _ = y.shape
    dy = new_act(n, d, h, w, c, y.dtype, y.device)
    red = torch.empty(_lib.lib().brats_gn_bwd_ws_floats(n, c), dtype=torch.float32, device=y.device)
    dgamma = torch.empty(c, dtype=torch.float32, device=y.device)
    dbeta = torch.empty(c, dtype=torch.float32, device=y.device)
    _lib.check(_lib.lib().brats_gn_act_bwd(dzp, dzpitch, yp, ypitch, scale_shift.data_ptr(), mean_rstd.data_ptr(),
                                           _f32(gamma), dy.data_ptr(), c, red.data_ptr(), dgamma.data_ptr(),
                                           dbeta.data_ptr(), _code(y.dtype), ACTS[act], slope, _f32(slope_t), n, d * h * w, c,
                                           groups, _f32(amax), _stream()), "gn_act_bwd")
    return dy, dgamma, dbeta


def conv_bstats_ok(dtype, dil, c1, cout, act="relu", slope_t=None):
    """Is the "backward statistics" form of the 3x3x3 convolution (conv3d_bstats) built for this layer?"""
    dtype = _conv_dtype(dtype, 3)  # (f32 tensors inside a split_precision() block: the x3 kernels have the form too)
    if not (is16(dtype) or dtype in (X3F, X3B)) or slope_t is not None or act not in ("relu", "leakyrelu"):
        return False
    return bool(_lib.lib().brats_conv3d_bstats_ok(_code(dtype), 3, dil, c1, cout))


def conv3d_bstats(x, packed_w, cout, dil, fwd_y, scale_shift, act="relu", slope=0.01, amax=None):
    """The input gradient dz = conv(x = dy of a block's second unit, weights packed with PACK_DGRAD) that also leaves
    GroupNorm backward's first pass for the block's FIRST unit (include/brats_hip.h: brats_conv3d_fwd_bstats): per tile and
    channel sum u and sum u * fwd_y, u = dz * act'(fwd_y * scale + shift).  -> (dz, tile_stats) for gn_act_bwd_tiles().
    f32 tensors inside a split_precision() block run brats_conv3d_x3_fwd_bstats; amax = max|x| as for conv3d (fp16 pairs)."""
    ptr, c, p = _desc(x)
    n, d, h, w, _ = x.shape
    fp, fc, fpitch = _desc(fwd_y)
    if fc != cout or fwd_y.dtype != x.dtype or tuple(fwd_y.shape[:4]) != (n, d, h, w):
        raise _lib.BratsHipError("conv3d_bstats: the forward tensor must have the output's shape and dtype")
    out = new_act(n, d, h, w, cout, x.dtype, x.device)
    stats = torch.empty((n, tiles_per_sample(d, h, w), cout, 2), dtype=torch.float32, device=x.device)
    # (a family of its own in the bench's kernel table: these launches carry a GroupNorm-backward pass in their epilogue and
    #  are not the plain implicit GEMM the roofline line is about)
    kd = _conv_dtype(x.dtype, 3)
    if kd in (X3F, X3B):
        with _span("conv_igemm_bst", c, cout, 3, dil, n, d, h, w, str(kd)):
            _lib.check(_lib.lib().brats_conv3d_x3_fwd_bstats(ptr, c, p, _f32(amax), packed_w.data_ptr(), out.data_ptr(), cout, fp, fpitch,
                                                             scale_shift.data_ptr(), ACTS[act], float(slope), stats.data_ptr(),
                                                             _code(kd), dil, n, d, h, w, cout, _stream()), "conv3d_x3_fwd_bstats")
        return out, stats
    with _span("conv_igemm_bst", c, cout, 3, dil, n, d, h, w, str(x.dtype)):
        _lib.check(_lib.lib().brats_conv3d_fwd_bstats(ptr, c, p, packed_w.data_ptr(), out.data_ptr(), cout, fp, fpitch,
                                                      scale_shift.data_ptr(), ACTS[act], float(slope), stats.data_ptr(),
                                                      _code(x.dtype), dil, n, d, h, w, cout, _stream()), "conv3d_fwd_bstats")
    return out, stats


def gn_act_bwd_tiles(tile_stats, dz, y, scale_shift, mean_rstd, gamma, groups=8, act="relu", slope=0.01, amax=None):
    """gn_act_bwd whose first pass was taken by conv3d_bstats (tile_stats): -> (dy, dgamma, dbeta)."""
    dzp, c, dzpitch = _desc(dz)
    yp, _, ypitch = _desc(y)
    n, d, h, w, _ = y.shape
    dy = new_act(n, d, h, w, c, y.dtype, y.device)
    red = torch.empty(_lib.lib().brats_gn_bwd_ws_floats(n, c), dtype=torch.float32, device=y.device)
    dgamma = torch.empty(c, dtype=torch.float32, device=y.device)
    dbeta = torch.empty(c, dtype=torch.float32, device=y.device)
    _lib.check(_lib.lib().brats_gn_act_bwd_tiles(tile_stats.data_ptr(), tile_stats.shape[1], dzp, dzpitch, yp, ypitch,
                                                 scale_shift.data_ptr(), mean_rstd.data_ptr(), _f32(gamma), dy.data_ptr(), c,
                                                 red.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), _code(y.dtype), ACTS[act],
                                                 float(slope), n, d * h * w, c, groups, _f32(amax), _stream()), "gn_act_bwd_tiles")
    return dy, dgamma, dbeta


def gn_act_bwd_head(dlogits, head_weight, y, scale_shift, mean_rstd, gamma, groups=8, act="relu", slope=0.01, amax=None):
    """GroupNorm + activation backward of the layer under the 1x1x1 output head, the head's backward folded in
    (include/brats_hip.h: brats_gn_act_bwd_head): -> (dy, dgamma, dbeta, dhead_weight [K,C,1,1,1], dhead_bias [K])."""
    yp, c, ypitch = _desc(y)
    n, d, h, w, _ = y.shape
    k = head_weight.shape[0]
    dev = y.device
    hw = head_weight.detach().reshape(k, c).contiguous().float()
    dl = dlogits.contiguous().float()
    dy = new_act(n, d, h, w, c, y.dtype, dev)
    red = torch.empty(_lib.lib().brats_gn_bwd_ws_floats(n, c), dtype=torch.float32, device=dev)
    hws = torch.empty(_lib.lib().brats_gn_bwd_head_ws_floats(n, c, k), dtype=torch.float32, device=dev)
    dgamma = torch.empty(c, dtype=torch.float32, device=dev)
    dbeta = torch.empty(c, dtype=torch.float32, device=dev)
    dhw = torch.empty((k, c), dtype=torch.float32, device=dev)
    dhb = torch.empty(k, dtype=torch.float32, device=dev)
    _lib.check(_lib.lib().brats_gn_act_bwd_head(dl.data_ptr(), hw.data_ptr(), k, yp, ypitch, scale_shift.data_ptr(),
                                                mean_rstd.data_ptr(), _f32(gamma), dy.data_ptr(), c, red.data_ptr(), hws.data_ptr(),
                                                dgamma.data_ptr(), dbeta.data_ptr(), dhw.data_ptr(), dhb.data_ptr(), _code(y.dtype),
                                                ACTS[act], slope, n, d * h * w, c, groups, _f32(amax), _stream()), "gn_act_bwd_head")
    return dy, dgamma, dbeta, dhw.reshape(k, c, 1, 1, 1), dhb


def gn_act_bwd_pool(dskip, dpool, argmax, y, scale_shift, mean_rstd, gamma, groups=8, act="relu", slope=0.01, amax=None):
    """GroupNorm + activation backward of the layer that ends an encoder level, its output gradient dskip +
    maxpool-backward(dpool) composed inside the two passes (include/brats_hip.h: brats_gn_act_bwd_pool): -> (dy, dgamma, dbeta)."""
    yp, c, ypitch = _desc(y)
    n, d, h, w, _ = y.shape
    sp, _, spitch = _desc(dskip)
    pp, _, ppitch = _desc(dpool)
    dev = y.device
    dy = new_act(n, d, h, w, c, y.dtype, dev)
    red = torch.empty(_lib.lib().brats_gn_bwd_ws_floats(n, c), dtype=torch.float32, device=dev)
    dgamma = torch.empty(c, dtype=torch.float32, device=dev)
    dbeta = torch.empty(c, dtype=torch.float32, device=dev)
    _lib.check(_lib.lib().brats_gn_act_bwd_pool(sp, spitch, pp, ppitch, argmax.data_ptr(), yp, ypitch, scale_shift.data_ptr(),
                                                mean_rstd.data_ptr(), _f32(gamma), dy.data_ptr(), c, red.data_ptr(),
                                                dgamma.data_ptr(), dbeta.data_ptr(), _code(y.dtype), ACTS[act], slope, n, d, h, w, c,
                                                groups, _f32(amax), _stream()), "gn_act_bwd_pool")
    return dy, dgamma, dbeta


def head_fold_ok(head_weight, act, slope_t):
    """brats_gn_act_bwd_head is built for three logit planes and relu / leakyrelu without a learnable slope."""
    return head_weight.shape[0] == 3 and act in ("relu", "leakyrelu") and slope_t is None


def prelu_slope_grad(dz, y, scale_shift):
    """d loss / d slope [1] of z = PReLU(y * scale + shift) with one shared slope (nn.PReLU()): sum dz * min(pre, 0)."""
    dzp, c, dzpitch = _desc(dz)
    yp, _, ypitch = _desc(y)
    n, d, h, w, _ = y.shape
    ws = torch.empty(_lib.lib().brats_prelu_ws_floats(n), dtype=torch.float32, device=y.device)
    out = torch.empty(1, dtype=torch.float32, device=y.device)
    _lib.check(_lib.lib().brats_prelu_slope_grad(dzp, dzpitch, yp, ypitch, scale_shift.data_ptr(), ws.data_ptr(), out.data_ptr(),
                                                 _code(y.dtype), n, d * h * w, c, _stream()), "prelu_slope_grad")
    return out


# ------------------------------------------------------------------------------------------ pool / upsample
def maxpool2(x, with_avg=False, out=None, want_argmax=False):
    """want_argmax (training): x._pool_argmax = the uint8 window index of every pooled element; maxpool2_bwd(x, ...) then
    reads those bytes instead of the eight window voxels of x."""
    ptr, c, p = _desc(x)
    n, d, h, w, _ = x.shape
    co = 2 * c if with_avg else c
    if out is None:
        out = new_act(n, d // 2, h // 2, w // 2, co, x.dtype, x.device)
    optr, oc, op = _desc(out)
    idx = torch.empty((n, d // 2, h // 2, w // 2, c), dtype=torch.uint8, device=x.device) if want_argmax else None
    _lib.check(_lib.lib().brats_maxpool2_fwd(ptr, p, optr, op, idx.data_ptr() if idx is not None else None, _code(x.dtype), n, c, d,
                                             h, w, int(with_avg), _stream()), "maxpool2_fwd")
    if idx is not None:
        x._pool_argmax = idx
    return out


def maxpool2_bwd(x, dy, dx_skip=None, with_avg=False):
    ptr, c, p = _desc(x)
    dptr, _, dp = _desc(dy)
    n, d, h, w, _ = x.shape
    dx = new_act(n, d, h, w, c, x.dtype, x.device)
    sptr, sp = (None, 0)
    if dx_skip is not None:
        sptr, _, sp = _desc(dx_skip)
    idx = getattr(x, "_pool_argmax", None)
    if idx is not None:  # recorded by affine_act_pool: the window voxels of x are not read again
        _lib.check(_lib.lib().brats_maxpool2_bwd_idx(idx.data_ptr(), dptr, dp, sptr, sp, dx.data_ptr(), c, _code(x.dtype), n, c,
                                                     d, h, w, int(with_avg), _stream()), "maxpool2_bwd_idx")
        return dx
    _lib.check(_lib.lib().brats_maxpool2_bwd(ptr, p, None, 0, dptr, dp, sptr, sp, dx.data_ptr(), c, _code(x.dtype), n, c,
                                             d, h, w, int(with_avg), _stream()), "maxpool2_bwd")
    return dx


def upsample(x, scale=2, out=None):
    ptr, c, p = _desc(x)
    n, d, h, w, _ = x.shape
    if out is None:
        out = new_act(n, d * scale, h * scale, w * scale, c, x.dtype, x.device)
    optr, oc, op = _desc(out)
    _lib.check(_lib.lib().brats_upsample_fwd(ptr, p, optr, op, _code(x.dtype), n, c, d, h, w, scale, _stream()),
               "upsample_fwd")
    return out


def upsample_bwd(dy, scale=2):
    dptr, c, dp = _desc(dy)
    n, do, ho, wo, _ = dy.shape
    d, h, w = do // scale, ho // scale, wo // scale
    code = _code(dy.dtype)
    ws = torch.empty(_lib.lib().brats_upsample_bwd_ws_bytes(code, n, c, d, h, w, scale), dtype=torch.uint8, device=dy.device)
    dx = new_act(n, d, h, w, c, dy.dtype, dy.device)
    _lib.check(_lib.lib().brats_upsample_bwd(dptr, dp, dx.data_ptr(), c, ws.data_ptr(), code, n, c, d, h, w, scale,
                                             _stream()), "upsample_bwd")
    return dx


# ------------------------------------------------------------------------------------------ heads
def head(x, weight, bias, scale=1):
    """NCDHW f32 logits [N, K, D*s, H*s, W*s] = upsample_s(conv1x1(x)); weight [K, C, 1,1,1]."""
    ptr, c, p = _desc(x)
    n, d, h, w, _ = x.shape
    k = weight.shape[0]
    wf = weight.detach().reshape(k, c).contiguous().float()
    out = torch.empty((n, k, d * scale, h * scale, w * scale), dtype=torch.float32, device=x.device)
    low = torch.empty((n, k, d, h, w), dtype=torch.float32, device=x.device) if scale > 1 else None
    _lib.check(_lib.lib().brats_head_fwd(ptr, p, wf.data_ptr(), _f32(bias.detach()) if bias is not None else None,
                                         low.data_ptr() if low is not None else None, out.data_ptr(), _code(x.dtype), n, c,
                                         k, d, h, w, scale, _stream()), "head_fwd")
    return out


def gn_head(y, scale_shift, weight, bias, act="relu", slope=0.01):
    """NCDHW f32 logits of the output head on act(GroupNorm(y)) without storing that activation
    (include/brats_hip.h: brats_gn_head_fwd); bit-identical to head(affine_act(y, scale_shift, act), weight, bias)."""
    ptr, c, p = _desc(y)
    n, d, h, w, _ = y.shape
    k = weight.shape[0]
    wf = weight.detach().reshape(k, c).contiguous().float()
    out = torch.empty((n, k, d, h, w), dtype=torch.float32, device=y.device)
    _lib.check(_lib.lib().brats_gn_head_fwd(ptr, p, scale_shift.data_ptr(), ACTS[act], slope, wf.data_ptr(),
                                            _f32(bias.detach()) if bias is not None else None, out.data_ptr(), _code(y.dtype), n, c,
                                            k, d * h * w, _stream()), "gn_head_fwd")
    return out


def head_bwd(x, weight, dout, scale=1, want_dx=True):
    """Returns (dx | None, dweight [K,C,1,1,1], dbias [K])."""
    ptr, c, p = _desc(x)
    n, d, h, w, _ = x.shape
    k = weight.shape[0]
    wf = weight.detach().reshape(k, c).contiguous().float()
    dout = dout.contiguous().float()
    nbytes = _lib.lib().brats_head_bwd_ws_bytes(n, c, k, d, h, w, scale)
    ws = torch.empty(max(nbytes, 4) // 4, dtype=torch.float32, device=x.device)
    dx = new_act(n, d, h, w, c, x.dtype, x.device) if want_dx else None
    dw = torch.empty((k, c), dtype=torch.float32, device=x.device)
    db = torch.empty(k, dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().brats_head_bwd(ptr, p, wf.data_ptr(), dout.data_ptr(), ws.data_ptr(),
                                         dx.data_ptr() if dx is not None else None, c, dw.data_ptr(), db.data_ptr(),
                                         _code(x.dtype), n, c, k, d, h, w, scale, _stream()), "head_bwd")
    return dx, dw.reshape(k, c, 1, 1, 1), db


# ------------------------------------------------------------------------------------------ EvoNorm-S0 / SE helpers
def evonorm_finalize(stats, n, c, groups, voxels, eps=1e-5):
    """Returns (mean_rstd [N, groups, 2], chan [N, C, 2] f64 per-channel sums kept for the backward)."""
    mean_rstd = torch.empty((n, groups, 2), dtype=torch.float32, device=stats.device)
    chan = torch.empty(_lib.lib().brats_gn_ws_doubles(n, c), dtype=torch.float64, device=stats.device)  # totals first
    _lib.check(_lib.lib().brats_evonorm_finalize(stats.data_ptr(), stats.shape[1], n, c, groups, float(voxels), eps,
                                                 mean_rstd.data_ptr(), chan.data_ptr(), _stream()), "evonorm_finalize")
    return mean_rstd, chan


def evonorm(y, mean_rstd, gamma, beta, groups=8, out=None, want_chansum=False, amax=None):
    """z = y*sigmoid(y) * rstd_g * gamma + beta; optionally also sum_v z per (n, c) (SE's pooling)."""
    ptr, c, p = _desc(y)
    n, d, h, w, _ = y.shape
    if out is None:
        out = new_act(n, d, h, w, c, y.dtype, y.device)
    optr, _, op = _desc(out)
    cs = torch.empty(_lib.lib().brats_chan_ws_floats(n, c, 1), dtype=torch.float32, device=y.device) if want_chansum else None
    _lib.check(_lib.lib().brats_evonorm_fwd(ptr, p, mean_rstd.data_ptr(), _f32(gamma), _f32(beta), optr, op,
                                            cs.data_ptr() if cs is not None else None, _code(y.dtype), n, d * h * w, c,
                                            groups, _f32(amax), _stream()), "evonorm_fwd")
    return out, (cs[:n * c].view(n, c) if cs is not None else None)


def evonorm_bwd(dz, y, mean_rstd, gamma, groups=8, chan=None, amax=None, gscale=None, gadd=None):
    """Returns (dy, dgamma, dbeta, dconvbias|None); dconvbias = sum_v dy needs the forward's `chan` sums.
    gscale / gadd ([N, C] f32): the incoming gradient is read as dz * gscale + gadd (the SE layer's backward folded in)."""
    dzp, c, dzpitch = _desc(dz)
    yp, _, ypitch = _desc(y)
    n, d, h, w, _ = y.shape
    dy = new_act(n, d, h, w, c, y.dtype, y.device)
    red = torch.empty(_lib.lib().brats_chan_ws_floats(n, c, 3), dtype=torch.float32, device=y.device)
    dgamma = torch.empty(c, dtype=torch.float32, device=y.device)
    dbeta = torch.empty(c, dtype=torch.float32, device=y.device)
    dcb = torch.empty(c, dtype=torch.float32, device=y.device) if chan is not None else None
    _lib.check(_lib.lib().brats_evonorm_bwd(dzp, dzpitch, yp, ypitch, mean_rstd.data_ptr(), _f32(gamma), dy.data_ptr(), c,
                                            red.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(),
                                            chan.data_ptr() if chan is not None else None,
                                            dcb.data_ptr() if dcb is not None else None, _code(y.dtype), n,
                                            d * h * w, c, groups, _f32(amax), _f32(gscale.contiguous()) if gscale is not None else None,
                                            _f32(gadd.contiguous()) if gadd is not None else None, _stream()), "evonorm_bwd")
    return dy, dgamma, dbeta, dcb


def evonorm_bwd_tiles(tile_stats, dz, y, mean_rstd, gamma, beta, groups=8, chan=None, amax=None):
    """evonorm_bwd whose first pass was taken by conv3d_bstats(..., fwd_y = z, act="leakyrelu", slope=1.0) -- tile_stats = per tile
    and channel (sum dz, sum dz * z) with z the EvoNorm's stored output (include/brats_hip.h: brats_evonorm_bwd_tiles).
    -> (dy, dgamma, dbeta, dconvbias|None)."""
    dzp, c, dzpitch = _desc(dz)
    yp, _, ypitch = _desc(y)
    n, d, h, w, _ = y.shape
    dy = new_act(n, d, h, w, c, y.dtype, y.device)
    red = torch.empty(_lib.lib().brats_evonorm_bwd_tiles_ws_floats(n, c), dtype=torch.float32, device=y.device)
    dgamma = torch.empty(c, dtype=torch.float32, device=y.device)
    dbeta = torch.empty(c, dtype=torch.float32, device=y.device)
    dcb = torch.empty(c, dtype=torch.float32, device=y.device) if chan is not None else None
    _lib.check(_lib.lib().brats_evonorm_bwd_tiles(tile_stats.data_ptr(), tile_stats.shape[1], dzp, dzpitch, yp, ypitch, mean_rstd.data_ptr(),
                                                  _f32(gamma), _f32(beta), dy.data_ptr(), c, red.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(),
                                                  chan.data_ptr() if chan is not None else None, dcb.data_ptr() if dcb is not None else None,
                                                  _code(y.dtype), n, d * h * w, c, groups, _f32(amax), _stream()), "evonorm_bwd_tiles")
    return dy, dgamma, dbeta, dcb


def channel_dot(a, b=None):
    """[N, C] f32 = sum over voxels of a (* b)."""
    ap, c, apitch = _desc(a)
    bp, bpitch = (None, 0)
    if b is not None:
        bp, _, bpitch = _desc(b)
    n, d, h, w, _ = a.shape
    out = torch.empty(_lib.lib().brats_chan_ws_floats(n, c, 1), dtype=torch.float32, device=a.device)
    _lib.check(_lib.lib().brats_channel_dot(ap, apitch, bp, bpitch, out.data_ptr(), _code(a.dtype), n, d * h * w, c,
                                            _stream()), "channel_dot")
    return out[:n * c].view(n, c)


def channel_scale(a, scale, add=None, out=None, amax=None):
    """out[v][c] = a[v][c] * scale[n][c] (+ add[n][c])."""
    ap, c, apitch = _desc(a)
    n, d, h, w, _ = a.shape
    if out is None:
        out = new_act(n, d, h, w, c, a.dtype, a.device)
    optr, _, op = _desc(out)
    _lib.check(_lib.lib().brats_channel_scale(ap, apitch, _f32(scale.contiguous()), _f32(add.contiguous()) if add is not None else None,
                                              optr, op, _code(a.dtype), n, d * h * w, c, _f32(amax), _stream()), "channel_scale")
    return out


def se_gate(chansum, voxels, w1, b1, w2, b2):
    """ResidualSELayer gate from the channel sums of z: -> (1 + gate [N, C], hidden [N, C/r]) in one launch."""
    n, c = chansum.shape
    ch = w1.shape[0]
    gate1p = torch.empty((n, c), dtype=torch.float32, device=chansum.device)
    hidden = torch.empty((n, ch), dtype=torch.float32, device=chansum.device)
    _lib.check(_lib.lib().brats_se_fwd(_f32(chansum.contiguous()), 1.0 / float(voxels), _f32(w1.detach().contiguous()), _f32(b1.detach()),
                                       _f32(w2.detach().contiguous()), _f32(b2.detach()), gate1p.data_ptr(), hidden.data_ptr(),
                                       n, c, ch, _stream()), "se_fwd")
    return gate1p, hidden


def se_gate_bwd(dgate, chansum, voxels, hidden, gate1p, w1, w2):
    """-> (gadd [N, C] = d loss / d gap / V, dW1, db1, dW2, db2) in one launch."""
    n, c = chansum.shape
    ch = w1.shape[0]
    dev = chansum.device
    gadd = torch.empty((n, c), dtype=torch.float32, device=dev)
    dw1, db1 = torch.empty((ch, c), dtype=torch.float32, device=dev), torch.empty((ch,), dtype=torch.float32, device=dev)
    dw2, db2 = torch.empty((c, ch), dtype=torch.float32, device=dev), torch.empty((c,), dtype=torch.float32, device=dev)
    _lib.check(_lib.lib().brats_se_bwd(_f32(dgate.contiguous()), _f32(chansum.contiguous()), 1.0 / float(voxels), _f32(hidden), _f32(gate1p),
                                       _f32(w1.detach().contiguous()), _f32(w2.detach().contiguous()), gadd.data_ptr(), dw1.data_ptr(),
                                       db1.data_ptr(), dw2.data_ptr(), db2.data_ptr(), n, c, ch, _stream()), "se_bwd")
    return gadd, dw1, db1, dw2, db2


def evonorm_se(y, mean_rstd, gamma, beta, w1, b1, w2, b2, groups=8, out=None, amax=None, apply=True):
    """EvoNorm + ResidualSELayer in one call, the EvoNorm output never stored (csrc/se.hpp):
    -> (out = z * (1 + gate), chansum [N, C] = sum_v z, gate1p [N, C], hidden [N, C/r]).
    apply=False: statistics pass + gate only (out = None): the consumer recomputes the output on load (evonorm_head)."""
    ptr, c, p = _desc(y)
    n, d, h, w, _ = y.shape
    ch = w1.shape[0]
    dev = y.device
    if not apply:
        out, optr, op = None, None, 0
    else:
        if out is None:
            out = new_act(n, d, h, w, c, y.dtype, dev)
        optr, _, op = _desc(out)
    ws = torch.empty(_lib.lib().brats_chan_ws_floats(n, c, 1), dtype=torch.float32, device=dev)
    cs = torch.empty((n, c), dtype=torch.float32, device=dev)
    gate1p = torch.empty((n, c), dtype=torch.float32, device=dev)
    hidden = torch.empty((n, ch), dtype=torch.float32, device=dev)
    _lib.check(_lib.lib().brats_evonorm_se_fwd(ptr, p, mean_rstd.data_ptr(), _f32(gamma), _f32(beta), _f32(w1.detach().contiguous()),
                                               _f32(b1.detach()), _f32(w2.detach().contiguous()), _f32(b2.detach()), optr, op,
                                               ws.data_ptr(), cs.data_ptr(), gate1p.data_ptr(), hidden.data_ptr(), ch,
                                               _code(y.dtype), n, d * h * w, c, groups, _f32(amax), _stream()), "evonorm_se_fwd")
    return out, cs, gate1p, hidden


def evonorm_head(y, mean_rstd, gamma, beta, gate1p, weight, bias, groups=8):
    """NCDHW f32 logits of the output head on the gated EvoNorm output of y without storing it
    (include/brats_hip.h: brats_evonorm_head_fwd); bit-identical to head(evonorm_se(...)[0], weight, bias)."""
    ptr, c, p = _desc(y)
    n, d, h, w, _ = y.shape
    k = weight.shape[0]
    # the per-(n, channel) constants in the kernel's own order of operations: (rstd * gamma) * gate1p, beta * gate1p
    rstd = mean_rstd[:, :, 1].repeat_interleave(c // groups, dim=1)
    ss = torch.stack(((rstd * gamma.float()) * gate1p, beta.float() * gate1p), -1).contiguous()
    wf = weight.detach().reshape(k, c).contiguous().float()
    out = torch.empty((n, k, d, h, w), dtype=torch.float32, device=y.device)
    _lib.check(_lib.lib().brats_evonorm_head_fwd(ptr, p, ss.data_ptr(), wf.data_ptr(), _f32(bias.detach()) if bias is not None else None,
                                                 out.data_ptr(), _code(y.dtype), n, c, k, d * h * w, _stream()), "evonorm_head_fwd")
    return out


def evonorm_se_bwd(do, y, mean_rstd, gamma, beta, se_chansum, hidden, gate1p, w1, w2, groups=8, chan=None, amax=None, head=None,
                   pool=None):
    """EvoNorm backward of the layer under a ResidualSELayer with the SE backward folded in (csrc/se.hpp): `do` is the
    gradient of the SE block's OUTPUT.  -> (dy, dgamma, dbeta, dconvbias|None, dW1, db1, dW2, db2).
    head = (head_weight [K,C,1,1,1], dlogits [N,K,D,H,W]) instead of `do` (None): the block feeds only the 1x1x1 output head,
    whose backward is folded in too; two more results: dhead_weight [K,C,1,1,1], dhead_bias [K].
    pool = (d_skip, d_pooled, argmax bytes, with_avg) instead of `do` (None): the block ends an encoder level, `do` = d_skip +
    pooling-backward(d_pooled) is composed inside the passes (brats_evonorm_se_bwd_pool)."""
    yp, c, ypitch = _desc(y)
    n, d, h, w, _ = y.shape
    ch = w1.shape[0]
    dev = y.device

    def f32(*shape):
        return torch.empty(shape, dtype=torch.float32, device=dev)

    dy = new_act(n, d, h, w, c, y.dtype, dev)
    ws = f32(_lib.lib().brats_chan_ws_floats(n, c, 5) + n * c * 3)
    dgamma, dbeta, dcb = f32(c), f32(c), (f32(c) if chan is not None else None)
    gadd, dw1, db1, dw2, db2 = f32(n, c), f32(ch, c), f32(ch), f32(c, ch), f32(c)
    # the arguments the two entry points share, in their order
    norm = (yp, ypitch, mean_rstd.data_ptr(), _f32(gamma), _f32(beta), dy.data_ptr(), c, ws.data_ptr(), dgamma.data_ptr(),
            dbeta.data_ptr(), chan.data_ptr() if chan is not None else None, dcb.data_ptr() if dcb is not None else None,
            _f32(se_chansum.contiguous()), _f32(hidden), _f32(gate1p), _f32(w1.detach().contiguous()),
            _f32(w2.detach().contiguous()), gadd.data_ptr(), dw1.data_ptr(), db1.data_ptr(), dw2.data_ptr(), db2.data_ptr(), ch)
    res = (dy, dgamma, dbeta, dcb, dw1, db1, dw2, db2)
    if pool is not None:
        sp, _, spitch = _desc(pool[0])
        pp, _, ppitch = _desc(pool[1])
        _lib.check(_lib.lib().brats_evonorm_se_bwd_pool(sp, spitch, pp, ppitch, pool[2].data_ptr(), int(pool[3]), d, h, w, *norm,
                                                        _code(y.dtype), n, c, groups, _f32(amax), _stream()), "evonorm_se_bwd_pool")
        return res
    if head is None:
        dop, _, dopitch = _desc(do)
        dl = hw = hws = dhw = dhb = None
        k = 0
    else:
        dop, dopitch = None, 0
        k = head[0].shape[0]
        hw = head[0].detach().reshape(k, c).contiguous().float()
        dl = head[1].contiguous().float()
        hws, dhw, dhb = f32(_lib.lib().brats_gn_bwd_head_ws_floats(n, c, k)), f32(k, c), f32(k)
    _lib.check(_lib.lib().brats_evonorm_se_bwd(dop, dopitch, *norm, _f32(dl), _f32(hw), k, _f32(hws), _f32(dhw), _f32(dhb),
                                               _code(y.dtype), n, d * h * w, c, groups, _f32(amax), _stream()), "evonorm_se_bwd")
    return res if head is None else res + (dhw.reshape(k, c, 1, 1, 1), dhb)


_DCONV_JOB = np.dtype([("term", [("x", "<u8"), ("w", "<u8"), ("xpitch", "<i4"), ("cin", "<i4"), ("ksize", "<i4"), ("dil", "<i4")], (4,)),
                       ("nterms", "<i4"), ("rows", "<i4"), ("bias", "<u8"), ("y", "<u8"), ("ypitch", "<i4"), ("reserved", "<i4")])
# == brats_dconv_job (include/brats_hip.h)


def pack_weights_direct(w, dtype, mode, cin_off=0, cin_cnt=None):
    """w [Cout, Cin, k, k, k] f32 -> the fragment order of the direct (gather) convolution, dconv_run().  Cached under
    torch.no_grad() exactly like pack_weights()."""
    key = None
    if not torch.is_grad_enabled():
        key = (id(w), "direct", str(dtype), mode, cin_off, cin_cnt)
        hit = _cache_get(key, w)
        if hit is not None:
            return hit
    cout_w, cin_w, k = w.shape[0], w.shape[1], w.shape[2]
    cnt = cin_w - cin_off if cin_cnt is None else cin_cnt
    kdim, rows = (cnt, cout_w) if mode == PACK_FWD else (cout_w, cnt)
    code = _code(dtype)
    nbytes = _lib.lib().brats_dconv_packed_bytes(code, k, kdim, rows)
    if nbytes == 0:
        raise _lib.BratsHipError(f"pack_weights_direct: K channels {kdim} must be a multiple of {16 if code in (BF16, F16) else 8}")
    wd = w.detach()
    if wd.dtype != torch.float32 or not wd.is_contiguous():
        wd = wd.contiguous().float()
    packed = torch.empty(nbytes, dtype=torch.uint8, device=w.device)
    _lib.check(_lib.lib().brats_dconv_pack_weights(wd.data_ptr(), packed.data_ptr(), code, mode, k, cout_w, cin_w, cin_off, cnt,
                                                   _stream()), "dconv_pack_weights")
    if key is not None:
        _cache_put(key, w, packed)
    return packed


def dconv_run(jobs, n, d, h, w, dtype):
    """One launch of the direct (gather) convolution.  jobs: up to 4 of (terms, bias | None, out) with terms = up to 4 of
    (x, packed_w, ksize, dil); out[..., :] = bias + sum over the terms of conv(x, w).  x / out: NDHWC tensors or
    channel-slice views of [n, d, h, w, *]."""
    rec = np.zeros(len(jobs), _DCONV_JOB)
    keep = []
    for j, (terms, bias, out) in enumerate(jobs):
        optr, rows, opitch = _desc(out)
        if out.dtype != dtype or tuple(out.shape[:4]) != (n, d, h, w):
            raise _lib.BratsHipError("dconv_run: bad output tensor")
        rec[j]["nterms"], rec[j]["rows"], rec[j]["y"], rec[j]["ypitch"] = len(terms), rows, optr, opitch
        rec[j]["bias"] = _f32(bias) or 0
        for t, (x, wpk, ksize, dil) in enumerate(terms):
            xptr, cin, xpitch = _desc(x)
            if x.dtype != dtype or tuple(x.shape[:4]) != (n, d, h, w):
                raise _lib.BratsHipError("dconv_run: bad input tensor")
            rec[j]["term"][t] = (xptr, wpk.data_ptr(), xpitch, cin, ksize, dil)
            keep.append(wpk)
    buf = rec.tobytes()
    _lib.check(_lib.lib().brats_dconv_run(buf, len(jobs), _code(dtype), n, d, h, w, _stream()), "dconv_run")


def conv3d_wgrad_shift(x, dy, ksize=1, dil=1, want_dbias=False, out=None, amax_dy=None):
    """dW [cout, cin, k, k, k] f32 (and dbias) of a 1x1x1 convolution, or of a 3x3x3 convolution at any dilation
    (shifted-tap form of the weight-gradient kernel: the ASPP branches with dilation 4 / 6).  Inside a split_precision()
    block f32 tensors run the three-product form (amax_dy: the scale source of dy, see conv3d_wgrad)."""
    ptr, c, p = _desc(x)
    dptr, cout, dp = _desc(dy)
    n, d, h, w, _ = x.shape
    kd = x.dtype
    if _X3 is not None and x.dtype == torch.float32 and c % 8 == 0 and cout % 8 == 0:
        kd = _X3  # (1x1x1 too: the voxel GEMM is MFMA-bound in exact f32 -- 3.0 of EquiUnetASSPEvo-48's 44 ms x3 step)
    code = _code(kd)
    if amax_dy is not None and kd in (X3F, X3B):
        nbytes = _lib.lib().brats_conv3d_wgrad_shift_ws_bytes(code, ksize, n, d, h, w, c, cout)
        ws = torch.empty(max(nbytes, 4) // 4, dtype=torch.float32, device=x.device)
        dw = _grad_out(out, (cout, c, ksize, ksize, ksize), x.device)
        db = torch.empty(cout, dtype=torch.float32, device=x.device) if want_dbias else None
        with _span("conv_wgrad", c, cout, ksize, dil, n, d, h, w, str(kd)):
            _lib.check(_lib.lib().brats_conv3d_x3_wgrad_shift(ptr, c, p, dptr, dp, _f32(amax_dy), ws.data_ptr(), dw.data_ptr(),
                                                              db.data_ptr() if db is not None else None, code, ksize, dil, n, d, h, w,
                                                              cout, _stream()), "conv3d_x3_wgrad_shift")
        return dw, db
    nbytes = _lib.lib().brats_conv3d_wgrad_shift_ws_bytes(code, ksize, n, d, h, w, c, cout)
    ws = torch.empty(max(nbytes, 4) // 4, dtype=torch.float32, device=x.device)
    dw = _grad_out(out, (cout, c, ksize, ksize, ksize), x.device)
    db = torch.empty(cout, dtype=torch.float32, device=x.device) if want_dbias else None
    with _span("conv_wgrad", c, cout, ksize, dil, n, d, h, w, str(kd)):
        _lib.check(_lib.lib().brats_conv3d_wgrad_shift(ptr, c, p, dptr, dp, ws.data_ptr(), dw.data_ptr(),
                                                       db.data_ptr() if db is not None else None, code, ksize, dil, n, d, h, w,
                                                       cout, _stream()), "conv3d_wgrad_shift")
    return dw, db


# ------------------------------------------------------------------------------------------ box calibration
def probe_box(device=None, mfma_ms=50.0, stream_bytes=403 << 20, modes=(0, 1)):
    """What THIS box delivers right now on the two resources the rooflines are quoted against (csrc/probe.hip): 16-bit
    MFMA TFLOP/s with every SIMD issuing back to back (two waves per SIMD, like the convolution kernels), and TB/s of a bf16
    read + write stream over `stream_bytes` (the size of a 2 x 128^3 x 48 activation).  ~120 ms of GPU time (shorter bursts read +-3 % with the chip's power control; 50 ms per
    mode repeats to +-0.1 %: scripts/probe_noise.py), HIP-event timed
    on the current stream.
      mfma_TFLOPs       : dense pseudo-random operands -- the matrix rate the chip SUSTAINS (it holds 1.8-2.0 GHz of its
                          nominal 2.4 GHz under this load: sclk_MHz = 16 cycles per MFMA and SIMD); roofline.frac_of_box
                          is quoted against this number;
      mfma_zeros_TFLOPs : every second operand value zero (post-ReLU activations): less switching power, higher clock --
                          the ceiling for kernels fed with such data.
    (modes 2 / 3 of the probe add vector instructions between the matrix groups: scripts/box_calib.py.)"""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    l = _lib.lib()
    ncu = torch.cuda.get_device_properties(dev).multi_processor_count
    blocks = 2 * ncu  # 8 waves per CU = two per SIMD, like the convolution kernels
    out = torch.empty(blocks, dtype=torch.float32, device=dev)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]

    def mfma(iters, mode):
        ev[0].record()
        _lib.check(l.brats_probe_mfma(out.data_ptr(), blocks, iters, mode, _stream()), "probe_mfma")
        ev[1].record()
        ev[1].synchronize()
        return ev[0].elapsed_time(ev[1])

    res = {}
    for mode in modes:
        mfma(2000, mode)  # warm-up (code object load, clocks)
        t0 = mfma(20000, mode)
        iters = max(20000, int(20000 * mfma_ms / max(t0, 1e-3)))
        ms = mfma(iters, mode)
        tf = blocks * 4.0 * iters * 8 * 16384 / (ms * 1e-3) / 1e12
        res[mode] = (tf, ms, iters)
    src = torch.empty(stream_bytes, dtype=torch.uint8, device=dev)
    src.view(torch.int16).fill_(0x3c00)
    dst = torch.empty_like(src)
    for _ in range(2):
        _lib.check(l.brats_probe_stream(src.data_ptr(), dst.data_ptr(), stream_bytes, _stream()), "probe_stream")
    reps = 8
    ev[2].record()
    for _ in range(reps):
        _lib.check(l.brats_probe_stream(src.data_ptr(), dst.data_ptr(), stream_bytes, _stream()), "probe_stream")
    ev[3].record()
    ev[3].synchronize()
    sms = ev[2].elapsed_time(ev[3]) / reps
    rec = {"stream_TBps": round(2.0 * stream_bytes / (sms * 1e-3) / 1e12, 3)}
    names = {0: "mfma", 1: "mfma_zeros", 2: "mfma_duty_dense", 3: "mfma_duty"}
    for mode, (tf, ms, iters) in res.items():
        rec[names[mode] + "_TFLOPs"] = round(tf, 1)
    if 0 in res:  # two waves per SIMD x iters x 8 MFMAs x 16 cycles
        rec["sclk_MHz"] = round(2.0 * res[0][2] * 8 * 16 / (res[0][1] * 1e-3) / 1e6, 0)
    rec["probe"] = (f"csrc/probe.hip: {blocks} x 4 waves issuing v_mfma_f32_16x16x32_bf16 back to back, ~{mfma_ms:.0f} ms per mode (mfma = dense "
                    f"random operands: the sustained matrix rate, sclk = the clock behind it at 16 cycles per MFMA and SIMD; mfma_zeros = "
                    f"every second value zero); bf16 read + write stream over {stream_bytes >> 20} MiB, {sms * 1e3:.0f} us per pass")
    return rec
