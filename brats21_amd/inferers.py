"""Sliding-window inference with the reference's signature (utils/inferers.py:26-40), executed
entirely on the GPU: windows are gathered (padding fused) by a HIP kernel, the predictor's fixed-shape
patch step is captured once into a hipGraph and replayed per window batch, and the weighted stitching /
count map / final division run as HIP kernels on device buffers (the reference moves every window
result to the CPU and stitches there, learning/engine.py:305-307).

Host-side index logic (scan interval, window origins, importance map) restates MONAI 0.6.0's
dense_patch_slices / compute_importance_map as used by the reference.
"""
import collections
import itertools
import math

import torch

from . import _lib


# --------------------------------------------------------------------------------------- host logic
def fall_back_tuple(roi, image_size):
    if isinstance(roi, int):
        roi = (roi,) * len(image_size)
    return tuple(i if (r is None or r <= 0) else r for r, i in zip(roi, image_size))


def get_scan_interval(image_size, roi_size, overlap):
    """_get_scan_interval, utils/inferers.py:165-186."""
    out = []
    for L, r in zip(image_size, roi_size):
        if r == L:
            out.append(int(r))
        else:
            iv = int(r * (1 - overlap))
            out.append(iv if iv > 0 else 1)
    return tuple(out)


def dense_window_starts(image_size, roi_size, interval):
    """MONAI dense_patch_slices: last window shifted back to fit; row-major product order."""
    per_dim = []
    for L, p, iv in zip(image_size, roi_size, interval):
        num = int(math.ceil(float(L) / iv))
        scan = next(d for d in range(num) if d * iv + p >= L)
        per_dim.append([i * iv - max(i * iv + p - L, 0) for i in range(scan + 1)])
    return list(itertools.product(*per_dim))


def _gaussian_taps(sigma, truncated=4.0):
    """MONAI 0.6.0 gaussian_1d(sigma, truncated=4.0, approx="erf"): the Gaussian integrated over each unit cell,
    round(truncated * sigma) taps either side, float32."""
    sigma = torch.as_tensor(float(sigma), dtype=torch.float32)
    tail = int(max(float(sigma) * truncated, 0.5) + 0.5)
    x = torch.arange(-tail, tail + 1, dtype=torch.float32)
    t = 0.70710678 / torch.abs(sigma)
    return (0.5 * (torch.erf(t * (x + 0.5)) - torch.erf(t * (x - 0.5)))).clamp(min=0)


def importance_map(patch, mode="constant", sigma_scale=0.125, device=None):
    """MONAI 0.6.0 compute_importance_map as the reference calls it (utils/inferers.py:119-121).  ``gaussian`` is
    GaussianFilter(sigma_scale * patch) applied to a unit delta at patch // 2 (erf-integrated taps, truncated at 4
    sigma, zero padding), divided by its maximum, zeros replaced by the smallest non-zero weight.  The filter of a delta
    is the outer product of the (shifted, cut) tap vectors, multiplied axis 0 first in float32 like the separable
    convolutions do."""
    mode = getattr(mode, "value", mode)
    if mode == "constant":
        return torch.ones(tuple(patch), dtype=torch.float32, device=device)
    if mode != "gaussian":
        raise ValueError(f"unsupported blend mode {mode}")
    if isinstance(sigma_scale, (int, float)):
        sigma_scale = (sigma_scale,) * len(patch)
    m = torch.ones((), dtype=torch.float32)
    for ax, (p, s) in enumerate(zip(patch, sigma_scale)):
        taps = _gaussian_taps(p * s)
        tail = (taps.numel() - 1) // 2
        d = (torch.arange(p) - p // 2).abs()
        g = torch.where(d <= tail, taps[(tail + d).clamp(max=2 * tail)], torch.zeros((), dtype=torch.float32))
        shape = [1] * len(patch)
        shape[ax] = p
        m = m * g.view(shape)
    m = (m / m.max()).float()
    m[m == 0] = m[m != 0].min()
    return m.to(device)


def _first(out):
    while isinstance(out, (tuple, list)):  # deep supervision: keep the main head (inferers.py:135-136)
        out = out[0]
    return out


# --------------------------------------------------------------------------------------- hipGraph patch step
class GraphedPredictor:
    """Captures ``predictor(window_batch)`` for one fixed input shape into a hipGraph (torch.cuda.CUDAGraph
    is hipGraph on ROCm) and replays it: one graph launch instead of ~100 kernel launches per patch.

    A captured graph bakes in the addresses of the packed-weight buffers the warm-up produced, so it is only valid
    for the weights it was captured with.  Pass the model(s) as ``modules``: every call compares their parameters'
    (object, address, version counter) and ops' packed-weight generation with what the graph saw and re-captures on a
    mismatch (optimizer steps, load_state_dict, SWA, train()/eval() switches); ``reset()`` drops all graphs by hand.
    Each graph keeps its packed buffers alive.  At most ``max_graphs`` shapes stay captured (least recently used
    evicted): every graph owns a private memory pool the size of a forward."""

    def __init__(self, predictor, modules=None, max_graphs=4):
        self.predictor = predictor
        self.modules = [] if modules is None else (list(modules) if isinstance(modules, (list, tuple)) else [modules])
        self.max_graphs = max_graphs
        self.graphs = collections.OrderedDict()
        self.captures = 0

    def reset(self):
        self.graphs.clear()

    def _signature(self):
        from . import ops
        sig = [ops.pack_generation()]
        for m in self.modules:
            sig.append(m.training)
            for p in m.parameters():
                sig.append((id(p), p.data_ptr(), p._version))
        return tuple(sig)

    def _capture(self, x):
        from . import ops
        keep = []
        static_in = torch.empty_like(x)
        static_in.copy_(x)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with ops.keep_packed(keep):
            with torch.cuda.stream(side), torch.no_grad():
                for _ in range(2):  # warm-up outside capture (lazy allocations, attribute setup, packed weights)
                    _first(self.predictor(static_in))
            torch.cuda.current_stream().wait_stream(side)
            g = torch.cuda.CUDAGraph()
            with torch.no_grad(), torch.cuda.graph(g):
                static_out = _first(self.predictor(static_in))
        self.captures += 1
        return g, static_in, static_out, keep

    def __call__(self, x):
        key = (tuple(x.shape), x.dtype)
        sig = self._signature()
        hit = self.graphs.get(key)
        if hit is not None and hit[4] != sig:
            del self.graphs[key]  # weights changed since the capture
            hit = None
        if hit is None:
            while len(self.graphs) >= self.max_graphs:
                self.graphs.popitem(last=False)
            g, static_in, static_out, keep = self._capture(x)
            hit = self.graphs[key] = (g, static_in, static_out, keep, self._signature())
        else:
            self.graphs.move_to_end(key)
        g, static_in, static_out = hit[:3]
        static_in.copy_(x)
        g.replay()
        return static_out


# --------------------------------------------------------------------------------------- the inferer
def sliding_window_inference(inputs, roi_size, sw_batch_size, predictor, overlap=0.25, mode="constant",
                             sigma_scale=0.125, padding_mode="constant", cval=0.0, sw_device=None, device=None,
                             *args, **kwargs):
    """Same contract as utils/inferers.py:26-162.  ``device`` / ``sw_device`` are accepted for API
    compatibility; everything stays on ``inputs.device`` (the point of this implementation), and the
    result is moved to ``device`` only at the very end if one was requested."""
    if inputs.dim() != 5:
        raise ValueError("expected NCDHW input")
    if overlap < 0 or overlap >= 1:
        raise AssertionError("overlap must be >= 0 and < 1.")
    pad_mode = {"constant": 0, "reflect": 1, "replicate": 2, "circular": 3}.get(getattr(padding_mode, "value", padding_mode))
    if pad_mode is None:
        raise ValueError(f"padding_mode {padding_mode!r} is not a PytorchPadMode (constant, reflect, replicate, circular)")
    if not inputs.is_cuda:
        raise _lib.BratsHipError("brats21_amd.sliding_window_inference runs on the GPU only (no CPU fallback)")
    lib = _lib.lib()
    stream = torch.cuda.current_stream().cuda_stream
    x = inputs.contiguous().float()
    nb, c = x.shape[:2]
    image_size_ = tuple(x.shape[2:])
    roi = fall_back_tuple(roi_size, image_size_)
    image_size = tuple(max(i, r) for i, r in zip(image_size_, roi))
    pad = [(max(r - i, 0)) // 2 for r, i in zip(roi, image_size_)]  # leading pad per dim (inferers.py:103-108)
    interval = get_scan_interval(image_size, roi, overlap)
    starts = dense_window_starts(image_size, roi, interval)
    num_win = len(starts)
    total = num_win * nb
    imp = importance_map(tuple(min(r, i) for r, i in zip(roi, image_size)), mode, sigma_scale, x.device)
    win_tbl = torch.tensor([[idx // num_win, *starts[idx % num_win]] for idx in range(total)], dtype=torch.int32,
                           device=x.device)
    out = cnt = None
    for g0 in range(0, total, sw_batch_size):
        b = min(sw_batch_size, total - g0)
        window = torch.empty((b, c) + roi, dtype=torch.float32, device=x.device)
        _lib.check(lib.brats_sw_gather(x.data_ptr(), window.data_ptr(), win_tbl[g0:g0 + b].data_ptr(), b, c, *image_size_,
                                       *roi, *pad, float(cval), pad_mode, stream), "sw_gather")
        prob = _first(predictor(window, *args, **kwargs)).contiguous().float()
        if out is None:
            k = prob.shape[1]
            out = torch.zeros((nb, k) + image_size, dtype=torch.float32, device=x.device)
            cnt = torch.zeros_like(out)
        # all windows of the batch in one launch (output-centric, windows added in order: bit-identical to one launch per window)
        _lib.check(lib.brats_sw_accumulate_multi(prob.data_ptr(), imp.data_ptr(), out.data_ptr(), cnt.data_ptr(),
                                                 win_tbl[g0:g0 + b].data_ptr(), b, nb, k, *image_size, *roi, stream), "sw_accumulate_multi")
    res = torch.empty((nb, k) + image_size_, dtype=torch.float32, device=x.device)
    _lib.check(lib.brats_sw_finalize(out.data_ptr(), cnt.data_ptr(), res.data_ptr(), nb * k, *image_size, *image_size_, *pad,
                                     stream), "sw_finalize")
    if device is not None and torch.device(device) != res.device:
        res = res.to(device)
    return res


def tta_predict(img, predictor, transforms, out=None):
    """Engine._apply_tta + the mean over passes of sigmoid(logits) (learning/engine.py:424-440, :239-249)
    without leaving the GPU: returns the TTA-averaged probabilities [N, K, D, H, W]."""
    acc = None
    n = 0
    for t in transforms:
        logits = _first(predictor(t.augment_image(img))).contiguous().float()
        if acc is None:
            shape = t.deaug_perm.out_shape(logits.shape)
            acc = torch.zeros(shape, dtype=torch.float32, device=img.device) if out is None else out.zero_()
        t.accumulate_probability(logits, acc)
        n += 1
    return acc.div_(n)
