"""EquiUnetASSPEvo (EvoNorm-S0 + MaxAvgPool + ResidualSE + dilated ASPP head) on the HIP kernels.

Drop-in for ``networks.equiunet2021.EquiUnetASSPEvo`` of the reference (networks/equiunet2021.py:225-333):
same constructor signature (``norm_layer`` / ``act`` are accepted and ignored exactly like the reference
does, :233), same ``state_dict`` keys / shapes -- including the statically unused EvoNorm ``v`` parameter
and ``running_var`` buffer (:76-83) -- same ``forward(x) -> (logits, [deep3, deep2])`` contract.
One autograd node; explicit forward / backward programs over NDHWC activations (see equiunet.py).
"""
import warnings

import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from .._lib import PACK_DGRAD, PACK_FWD, BratsHipError
from .equiunet import douts_device, _AmaxSlots, _ConvParams, _PackedWeightsModule, _inherit_amax


# ------------------------------------------------------------------------------------------ parameter holders
class EvoNorm3D(nn.Module):
    """Parameters of networks/equiunet2021.py:55-118 (S0, affine, non_linear): gamma, beta, v + running_var."""

    def __init__(self, c):
        super().__init__()
        self.gamma = nn.Parameter(torch.ones(1, c, 1, 1, 1))
        self.beta = nn.Parameter(torch.zeros(1, c, 1, 1, 1))
        self.v = nn.Parameter(torch.ones(1, c, 1, 1, 1))  # unused on the efficient S0 path (:101-103)
        self.register_buffer("running_var", torch.ones(1, c, 1, 1, 1))


class _SEParams(nn.Module):
    """MONAI ResidualSELayer(3, C, r=2) parameters: fc.0 / fc.2 Linear layers."""

    def __init__(self, c):
        super().__init__()
        self.fc = nn.ModuleList([nn.Linear(c, c // 2), nn.Identity(), nn.Linear(c // 2, c), nn.Identity()])


class ConvEvoBlockCorrected(nn.Module):
    """conv3+bias -> EvoNorm -> conv3+bias -> EvoNorm -> ResidualSE (networks/equiunet2021.py:192-209).
    ``conv_conv_se`` is indexed like the reference's nn.Sequential (0,1,3,4,6; 2 and 5 are Dropout)."""

    def __init__(self, cin, cout):
        super().__init__()
        self.conv_conv_se = nn.ModuleList([
            _ConvParams(cin, cout, 3, True), EvoNorm3D(cout), nn.Identity(),
            _ConvParams(cout, cout, 3, True), EvoNorm3D(cout), nn.Identity(), _SEParams(cout)])


class ConvEvo(nn.Module):
    """1x1x1 conv + bias -> EvoNorm (networks/equiunet2021.py:212-222)."""

    def __init__(self, cin, cout):
        super().__init__()
        self.conv = _ConvParams(cin, cout, 1, True)
        self.evo = EvoNorm3D(cout)


class SimpleASPPEVO(nn.Module):
    """networks/equiunet2021.py:121-189: four parallel convs (k1, k3 d2/d4/d6) -> cat -> ConvEvo 1x1."""

    def __init__(self, cin, cout_each, kernel_sizes=(1, 3, 3, 3), dilations=(1, 2, 4, 6)):
        super().__init__()
        self.kernel_sizes, self.dilations = tuple(kernel_sizes), tuple(dilations)
        self.convs = nn.ModuleList([_ConvParams(cin, cout_each, k, True) for k in kernel_sizes])
        self.conv_k1 = ConvEvo(cout_each * len(kernel_sizes), cout_each * len(kernel_sizes))


# ------------------------------------------------------------------------------------------ unit programs
def _flat(p):
    return p.detach().reshape(-1).contiguous()


class _Ctx:
    """Per-forward state shared by the unit programs."""

    def __init__(self, model, dtype):
        self.m, self.dtype = model, dtype
        # e4m3 convolutions (EquiUnet.conv_fp8 semantics): the EvoNorm / SE kernels record the |max| of what they write
        # into slots taken here; a tensor without a recorded |max| falls back to ops.absmax inside ops.conv3d_f8
        self.fp8 = getattr(model, "conv_fp8", None) if ops.is16(dtype) else None
        # split precision on fp16 pairs (backward): dy is scaled by a power of two from its recorded |max| (EquiUnet._x3_modes)
        self.x3s = ops.x3_mode() == ops.X3F and dtype == torch.float32
        self.slots = None
        self.names = {p: i for i, p in enumerate(model.parameters())}
        self.grads = {}
        # nn.Dropout(p) behind every EvoNorm but the ASPP's (networks/equiunet2021.py:200,203,219; :178 pins the ASPP's to 0),
        # training mode only: (p, state) with state = (seed, step counter) on the device; None = off
        self.drop = None

    def dropped(self, t, uid, out=None):
        """t * keep / (1 - p) with unit uid's mask of this step (forward: in place on the activation; backward: on the gradient)."""
        if self.drop is None or uid is None:
            return t
        return ops.dropout(t, self.drop[0], self.drop[1], uid, out=out)

    def slot(self, device, force=False):
        if not self.fp8 and not force:
            return None
        if self.slots is None or self.slots.i >= self.slots.buf.numel():
            self.slots = _AmaxSlots(64, device)
        return self.slots.take()

    def identity_ss(self, n, c, device):
        """{scale, shift} = {1, 0} per (sample, channel): ops.conv3d_bstats' activation argument when there is no activation."""
        key = (n, c)
        if getattr(self, "_ss", None) is None:
            self._ss = {}
        if key not in self._ss:
            ss = torch.zeros((n, c, 2), dtype=torch.float32, device=device)
            ss[..., 0] = 1.0
            self._ss[key] = ss
        return self._ss[key]

    def dest(self, param):
        """The parameter's slice of a DDP all-reduce bucket (ddp.GradientBuckets.dest) for kernels that can write there."""
        d = getattr(self.m, "_grad_dest", None)
        return d(self.names[param]) if d is not None else None

    def put(self, param, grad):
        i = self.names[param]
        self.grads[i] = grad.reshape(param.shape)
        if self.m._grad_sink is not None:
            self.m._grad_sink(i, self.grads[i])


def _conv_any_fwd(cx, conv, x, dil, want_stats, out=None):
    """conv (3x3x3 with dilation 1 | 2, or 1x1x1) + bias.  Returns (y, stats, saved) with what backward needs."""
    w = conv.weight
    cout, cin, k = w.shape[0], w.shape[1], w.shape[2]
    if cx.fp8 and k == 3 and x.shape[1] >= ops.F8_MIN_SIZE and ops.conv_f8_chunk(x.shape[-1]) > 0:
        wpk = ops.pack_weights_f8(w, PACK_FWD, cin_pad=x.shape[-1])
        y, stats = ops.conv3d_f8(x, wpk, cout, dil, bias=_flat(conv.bias), out=out, want_stats=want_stats,
                                 amax=getattr(x, "_amax", None))
        return y, stats, (x, dil)
    wpk = ops.pack_weights(w, cx.dtype, PACK_FWD, cin_pad=x.shape[-1], dil=dil)
    y, stats = ops.conv3d(x, wpk, cout, k, dil, bias=_flat(conv.bias), out=out, want_stats=want_stats,
                          amax=getattr(x, "_x3amax", None) if k == 3 else None)  # (the network input in split-precision mode: EquiUnet)
    return y, stats, (x, dil)


def _conv_any_bwd(cx, conv, saved, dy, need_dx=True, db=None, bstats=False):
    """bstats: this is the SECOND convolution of a block, whose input xin is the first EvoNorm's output z1 -- where the kernel form
    is built, the input-gradient launch also leaves (sum dz1, sum dz1 * z1) per tile and channel (ops.conv3d_bstats with the
    identity "activation": leakyrelu, slope 1) and (dz1, tile sums) is returned for ops.evonorm_bwd_tiles."""
    xin, dil = saved
    w = conv.weight
    cout, cin, k = w.shape[0], w.shape[1], w.shape[2]
    if db is None:
        db = ops.channel_dot(dy).sum(0)
    if cx.x3s and getattr(dy, "_amax", None) is None:
        dy._amax = ops.absmax(dy)  # (a producer that records no |max|: one extra pass)
    if k == 1:
        # a GEMM over the voxels: shifted-tap kernel, 1 tap
        dw, _ = ops.conv3d_wgrad_shift(xin, dy, 1, out=cx.dest(conv.weight), amax_dy=dy._amax if cx.x3s else None)
        cx.put(conv.weight, dw)
    else:
        ax, ady = getattr(xin, "_amax", None), getattr(dy, "_amax", None)
        if cx.fp8 == "all" and dil == 1 and ax is not None and ady is not None and ops.conv3d_wgrad_f8_ok(xin, dy):
            dw = ops.conv3d_wgrad_f8(xin, dy, ax, ady, out=cx.dest(conv.weight))  # e4m3 operands, scales from the recorded |max|
        else:
            dw, _ = ops.conv3d_wgrad(xin, dy, 3, dil, out=cx.dest(conv.weight), amax_dy=ady if cx.x3s else None)
        cx.put(conv.weight, dw[:, :cin].contiguous() if dw.shape[1] != cin else dw)
    cx.put(conv.bias, db)
    if not need_dx:
        return None
    if cx.fp8 == "all" and k == 3 and ops.conv_f8_chunk(cout) > 0:
        dx, _ = ops.conv3d_f8(dy, ops.pack_weights_f8(w, PACK_DGRAD), cin, dil, amax=getattr(dy, "_amax", None))
        return dx
    wpk = ops.pack_weights(w, cx.dtype, PACK_DGRAD, dil=dil)
    if (bstats and k == 3 and xin.shape[-1] == cin and xin.dtype == dy.dtype
            and ops.conv_bstats_ok(cx.dtype, dil, cout, cin, "leakyrelu")):
        n = dy.shape[0]
        ss = cx.identity_ss(n, cin, dy.device)
        return ops.conv3d_bstats(dy, wpk, cin, dil, xin, ss, "leakyrelu", slope=1.0,  # (dz1, tile sums)
                                 amax=getattr(dy, "_amax", None) if cx.x3s else None)
    dx, _ = ops.conv3d(dy, wpk, cin, k, dil, amax=getattr(dy, "_amax", None) if (cx.x3s and k == 3) else None)
    return dx


def _aspp_fwd(cx, aspp, x, acat):
    """SimpleASPPEVO's parallel branches (networks/equiunet2021.py:179-187): ONE launch of the direct (gather) convolution,
    each branch writing its channel slice of the concat buffer `acat` (no torch.cat, no im2col buffer)."""
    n, d, h, w, _ = x.shape
    q = aspp.convs[0].weight.shape[0]
    jobs = []
    for i, (k, dl) in enumerate(zip(aspp.kernel_sizes, aspp.dilations)):
        conv = aspp.convs[i]
        wpk = ops.pack_weights_direct(conv.weight, cx.dtype, PACK_FWD)
        jobs.append(([(x, wpk, k, dl)], _flat(conv.bias), acat[..., i * q:(i + 1) * q]))
    for j0 in range(0, len(jobs), 4):
        ops.dconv_run(jobs[j0:j0 + 4], n, d, h, w, cx.dtype)
    return x


def _aspp_bwd(cx, aspp, x, d_acat):
    """Weight / bias gradients of the branches (shifted-tap weight-gradient kernel: 1 tap for k = 1, 27 shifted taps for
    the dilated ones) and the input gradient as ONE launch whose accumulators sum the four branches."""
    n, d, h, w, _ = x.shape
    q = aspp.convs[0].weight.shape[0]
    db_all = ops.channel_dot(d_acat).sum(0)
    terms = []
    for i, (k, dl) in enumerate(zip(aspp.kernel_sizes, aspp.dilations)):
        conv = aspp.convs[i]
        dyi = d_acat[..., i * q:(i + 1) * q]
        dw, _ = ops.conv3d_wgrad_shift(x, dyi, k, dl, amax_dy=ops.absmax(dyi) if cx.x3s else None)  # (16^3: the extra pass is 3 MB)
        cx.put(conv.weight, dw)
        cx.put(conv.bias, db_all[i * q:(i + 1) * q])
        terms.append((dyi, ops.pack_weights_direct(conv.weight, cx.dtype, PACK_DGRAD), k, dl))
    dx = ops.new_act(n, d, h, w, x.shape[-1], cx.dtype, x.device)
    if len(terms) <= 4:
        ops.dconv_run([(terms, None, dx)], n, d, h, w, cx.dtype)
    else:  # (more than four branches: the reference's constructor allows any number)
        parts = []
        for j0 in range(0, len(terms), 4):
            part = dx if j0 == 0 else torch.empty_like(dx)
            ops.dconv_run([(terms[j0:j0 + 4], None, part)], n, d, h, w, cx.dtype)
            parts.append(part)
        for part in parts[1:]:
            dx += part
    return dx


def _conv_evo_fwd(cx, conv, evo, x, out=None, want_chansum=False, uid=None):
    """uid: the unit's dropout stream (None: no dropout behind this EvoNorm)."""
    y, stats, saved = _conv_any_fwd(cx, conv, x, 1, True)
    n, d, h, w, c = y.shape
    mr, chan = ops.evonorm_finalize(stats, n, c, 8, d * h * w)
    amax = cx.slot(y.device)
    z, cs = ops.evonorm(y, mr, _flat(evo.gamma), _flat(evo.beta), 8, out=out, want_chansum=want_chansum, amax=amax)
    if amax is not None:
        z._amax = amax
    cx.dropped(z, uid, out=z)
    return z, cs, (conv, evo, saved, y, mr, chan, uid)


def _conv_evo_bwd(cx, rec, dz, need_dx=True, gscale=None, gadd=None):
    conv, evo, saved, y, mr, chan, uid = rec
    if isinstance(dz, tuple):  # (dz, tile sums of the pass-1 quantities): _conv_any_bwd(bstats=True)
        dz, tiles = dz
        amax = cx.slot(y.device, cx.x3s) if (cx.fp8 == "all" or cx.x3s) else None
        dy, dgamma, dbeta, dcb = ops.evonorm_bwd_tiles(tiles, dz, y, mr, _flat(evo.gamma), _flat(evo.beta), 8, chan=chan, amax=amax)
        if amax is not None:
            dy._amax = amax
        cx.put(evo.gamma, dgamma)
        cx.put(evo.beta, dbeta)
        return _conv_any_bwd(cx, conv, saved, dy, need_dx, db=dcb)
    dz = cx.dropped(dz, uid)
    amax = cx.slot(y.device, cx.x3s) if (cx.fp8 == "all" or cx.x3s) else None  # scale source of the e4m3 / fp16-pair gradients
    dy, dgamma, dbeta, dcb = ops.evonorm_bwd(dz, y, mr, _flat(evo.gamma), 8, chan=chan, amax=amax, gscale=gscale, gadd=gadd)
    if amax is not None:
        dy._amax = amax
    cx.put(evo.gamma, dgamma)
    cx.put(evo.beta, dbeta)
    return _conv_any_bwd(cx, conv, saved, dy, need_dx, db=dcb)


def _block_fwd(cx, blk, x, out=None, head=None):
    """head: the 1x1x1 output head module when the block's output feeds nothing else -- the block then returns the head's
    logits instead of its output tensor, which is recomputed on load inside the head kernel and never stored."""
    s = blk.conv_conv_se
    uid = cx.m._unit_ids[blk] if cx.drop is not None else None
    z1, _, r1 = _conv_evo_fwd(cx, s[0], s[1], x, uid=uid)
    if cx.drop is not None:
        # dropout between the second EvoNorm and the SE layer (networks/equiunet2021.py:201-205): the fused EvoNorm + SE forms derive
        # the layer's global average from sums of the UN-dropped values, so the three steps run one by one on a stored z2
        conv, evo = s[3], s[4]
        y, stats, saved = _conv_any_fwd(cx, conv, z1, 1, True)
        n, d, h, w, c = y.shape
        mr, chan = ops.evonorm_finalize(stats, n, c, 8, d * h * w)
        z2, _ = ops.evonorm(y, mr, _flat(evo.gamma), _flat(evo.beta), 8)
        cx.dropped(z2, uid + 1, out=z2)
        fc1, fc2 = s[6].fc[0], s[6].fc[2]
        cs = ops.channel_dot(z2)
        gate1p, hidden = ops.se_gate(cs, d * h * w, fc1.weight, fc1.bias, fc2.weight, fc2.bias)
        o = ops.channel_scale(z2, gate1p, out=out)
        return o, (blk, r1, (conv, evo, saved, y, mr, chan), cs, hidden, gate1p, z2)
    # second conv + EvoNorm + ResidualSELayer (out = z2 + z2 * sigmoid(W2 relu(W1 gap + b1) + b2)) as one call: a statistics
    # pass over y, the gate (csrc/se.hip), then the EvoNorm pass writes z2 * (1 + gate) -- z2 itself is never stored
    conv, evo = s[3], s[4]
    y, stats, saved = _conv_any_fwd(cx, conv, z1, 1, True)
    n, d, h, w, c = y.shape
    mr, chan = ops.evonorm_finalize(stats, n, c, 8, d * h * w)
    fc1, fc2 = s[6].fc[0], s[6].fc[2]
    rec2 = (conv, evo, saved, y, mr, chan)
    if head is not None:
        _, cs, gate1p, hidden = ops.evonorm_se(y, mr, _flat(evo.gamma), _flat(evo.beta), fc1.weight, fc1.bias, fc2.weight, fc2.bias, 8,
                                               apply=False)
        logits = ops.evonorm_head(y, mr, _flat(evo.gamma), _flat(evo.beta), gate1p, head.weight, head.bias, 8)
        return logits, (blk, r1, rec2, cs, hidden, gate1p)
    amax = cx.slot(y.device)
    o, cs, gate1p, hidden = ops.evonorm_se(y, mr, _flat(evo.gamma), _flat(evo.beta), fc1.weight, fc1.bias, fc2.weight, fc2.bias, 8,
                                           out=out, amax=amax)
    if amax is not None:
        o._amax = amax
    return o, (blk, r1, rec2, cs, hidden, gate1p)


def _block_bwd(cx, rec, do, need_dx=True, head=None, pool=None):
    """head = (head module, dlogits) instead of `do`: the block's output feeds only the 1x1x1 output head, whose backward is
    folded into the same call (d(up1) is never written)."""
    s = rec[0].conv_conv_se
    fc1, fc2 = s[6].fc[0], s[6].fc[2]
    if len(rec) == 7:  # the dropout form of _block_fwd: SE backward, dropout, EvoNorm backward one by one
        blk, r1, r2, cs, hidden, gate1p, z2 = rec
        conv, evo, saved, y, mr, chan = r2
        uid = cx.m._unit_ids[blk]
        n, d, h, w, c = y.shape
        dgate = ops.channel_dot(do, z2)  # out = z2 * (1 + gate): d loss / d gate = sum_v do * z2
        gadd, dw1, db1, dw2, db2 = ops.se_gate_bwd(dgate, cs, d * h * w, hidden, gate1p, fc1.weight, fc2.weight)
        dz2 = cx.dropped(ops.channel_scale(do, gate1p, add=gadd), uid + 1)
        amax = cx.slot(y.device, cx.x3s) if cx.x3s else None
        dy, dgamma, dbeta, dcb = ops.evonorm_bwd(dz2, y, mr, _flat(evo.gamma), 8, chan=chan, amax=amax)
        if amax is not None:
            dy._amax = amax
        for prm, g in ((fc1.weight, dw1), (fc1.bias, db1), (fc2.weight, dw2), (fc2.bias, db2), (evo.gamma, dgamma), (evo.beta, dbeta)):
            cx.put(prm, g)
        dz1 = _conv_any_bwd(cx, conv, saved, dy, True, db=dcb)
        return _conv_evo_bwd(cx, r1, dz1, need_dx)
    blk, r1, r2, cs, hidden, gate1p = rec
    # one call: pass 1 of the EvoNorm backward over (do, y) also yields d loss / d gate = sum_v do * z2 (linear in its sums),
    # the SE backward runs on those, pass 2 reads the gradient as do * (1 + gate) + dgap / V  (csrc/se.hpp)
    conv, evo, saved, y, mr, chan = r2
    amax = cx.slot(y.device, cx.x3s) if (cx.fp8 == "all" or cx.x3s) else None  # scale source of the e4m3 / fp16-pair gradients
    res = ops.evonorm_se_bwd(do, y, mr, _flat(evo.gamma), _flat(evo.beta), cs, hidden, gate1p, fc1.weight, fc2.weight, 8, chan=chan,
                             amax=amax, head=(head[0].weight, head[1]) if head is not None else None, pool=pool)
    dy, dgamma, dbeta, dcb, dw1, db1, dw2, db2 = res[:8]
    if head is not None:
        cx.put(head[0].weight, res[8])
        cx.put(head[0].bias, res[9])
    if amax is not None:
        dy._amax = amax
    for prm, g in ((fc1.weight, dw1), (fc1.bias, db1), (fc2.weight, dw2), (fc2.bias, db2), (evo.gamma, dgamma), (evo.beta, dbeta)):
        cx.put(prm, g)
    # the first EvoNorm's backward statistics ride in this input-gradient launch where that kernel form is built (model.fold_bwd_stats)
    dz1 = _conv_any_bwd(cx, conv, saved, dy, True, db=dcb, bstats=cx.m.fold_bwd_stats)
    return _conv_evo_bwd(cx, r1, dz1, need_dx)


class _AsspFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, x, dtype, *params):
        # training: the module packed all layers' weights up front (ops.plan_for); pack_weights() then returns views
        ctx.plan = ops._PLANS.get(model) if (model.training and model.pack_plan) else None
        ctx.x3 = model._x3_modes() if dtype == torch.float32 else (None, None)
        with ops.use_plan(ctx.plan), ops.split_precision(ctx.x3[0]):
            return _AsspFn._forward(ctx, model, x, dtype, *params)

    @staticmethod
    def backward(ctx, *douts):
        with ops.use_plan(ctx.plan), ops.split_precision(ctx.x3[1]):
            return _AsspFn._backward(ctx, *douts)

    @staticmethod
    def _forward(ctx, model, x, dtype, *params):
        m = model
        f = m.features
        cx = _Ctx(m, dtype)
        if m.training and m.dropout_p > 0.0:
            if cx.fp8:
                raise NotImplementedError("--dropout > 0 with the e4m3 convolution path is not implemented")
            cx.drop = (m.dropout_p, m._advance_dropout(x.device))

        def uid(mod):  # dropout stream of a ConvEvo unit (the ASPP's has p = 0: networks/equiunet2021.py:178)
            return m._unit_ids[mod] if cx.drop is not None else None
        n, _, d, h, w = x.shape
        dev = x.device
        h0, h1, h2 = f[0] // 2, f[1] // 2, f[2] // 2
        x0 = ops.ncdhw_to_ndhwc(x, dtype, cpad=8 if (ops.is16(dtype) or ops.x3_active()) else 4)
        if ops.x3_mode() == ops.X3F:
            x0._x3amax = ops.absmax(x0)  # fp16 pairs: the first layer's forward scales the un-normalised input (networks/equiunet.py)
        will_bwd = any(ctx.needs_input_grad) and model._fwd_grad  # (needs_input_grad ignores no_grad)

        def pool(t):  # max and average of a 2x2x2 cell never exceed the |max| of the input
            # (training: the arg-max bytes go along, the pooling backward reads them instead of the window)
            return _inherit_amax(ops.maxpool2(t, with_avg=True, want_argmax=will_bwd), t)

        def cat_amax(cat, a, b):  # a concat buffer written by two producers: |max| = the larger of theirs
            if cx.fp8 and hasattr(a, "_amax") and hasattr(b, "_amax"):
                cat._amax = torch.maximum(a._amax, b._amax)

        # encoder (networks/equiunet2021.py:291-298)
        down1, rb1 = _block_fwd(cx, m.encoder1, x0)
        down2, rb2 = _block_fwd(cx, m.encoder2, pool(down1))
        down3, rb3 = _block_fwd(cx, m.encoder3, pool(down2))
        down4, rb4 = _block_fwd(cx, m.encoder4, pool(down3))
        # ASPP (:299, :187-189): the four branches write into channel slices of one buffer
        acat = ops.new_act(n, d // 8, h // 8, w // 8, f[3], dtype, dev)
        ra = _aspp_fwd(cx, m.aspp, down4, acat)
        assp, _, rk1 = _conv_evo_fwd(cx, m.aspp.conv_k1.conv, m.aspp.conv_k1.evo, acat)
        # bridges (:302-304) and decoder (:306-320): bridge / up-sample outputs land in concat buffers
        cat1 = ops.new_act(n, d, h, w, 2 * h0, dtype, dev)
        cat2 = ops.new_act(n, d // 2, h // 2, w // 2, 2 * h1, dtype, dev)
        cat3 = ops.new_act(n, d // 4, h // 4, w // 4, 2 * h2, dtype, dev)
        br1, _, rbr1 = _conv_evo_fwd(cx, m.bridge1.conv, m.bridge1.evo, down1, out=cat1[..., :h0], uid=uid(m.bridge1))
        br2, _, rbr2 = _conv_evo_fwd(cx, m.bridge2.conv, m.bridge2.evo, down2, out=cat2[..., :h1], uid=uid(m.bridge2))
        br3, _, rbr3 = _conv_evo_fwd(cx, m.bridge3.conv, m.bridge3.evo, down3, out=cat3[..., :h2], uid=uid(m.bridge3))
        uc3, _, ru3 = _conv_evo_fwd(cx, m.upconv3.conv, m.upconv3.evo, assp, uid=uid(m.upconv3))
        ops.upsample(uc3, 2, out=cat3[..., h2:])
        cat_amax(cat3, br3, uc3)
        up3, rd3 = _block_fwd(cx, m.decoder3, cat3)
        uc2, _, ru2 = _conv_evo_fwd(cx, m.upconv2.conv, m.upconv2.evo, up3, uid=uid(m.upconv2))
        ops.upsample(uc2, 2, out=cat2[..., h1:])
        cat_amax(cat2, br2, uc2)
        up2, rd2 = _block_fwd(cx, m.decoder2, cat2)
        uc1, _, ru1 = _conv_evo_fwd(cx, m.upconv1.conv, m.upconv1.evo, up2, uid=uid(m.upconv1))
        ops.upsample(uc1, 2, out=cat1[..., h0:])
        cat_amax(cat1, br1, uc1)
        # decoder1's output feeds only the output head: where the kernels for it are built the head recomputes it on load
        # (ops.evonorm_head) and the backward folds the head in (ops.evonorm_se_bwd(head=...)) -- up1 is never stored
        nk = m.out_conv.weight.shape[0]
        fuse_top = m.fold_head_fwd and cx.drop is None and nk <= 4 and (not will_bwd or (m.fold_head_bwd and nk == 3))
        if fuse_top:
            logits, rd1 = _block_fwd(cx, m.decoder1, cat1, head=m.out_conv)
            up1 = None
            outs = [logits]
        else:
            up1, rd1 = _block_fwd(cx, m.decoder1, cat1)
            outs = [ops.head(up1, m.out_conv.weight, m.out_conv.bias, 1)]
        ctx.top_fused = fuse_top
        ctx.out_shape = tuple(outs[0].shape)
        heads = [(m.out_conv, up1, 1)]
        if m.deep_supervision and not (m.skip_deep_heads_in_eval and not m.training):
            for hd, src, sc in ((m.deep3[0], up3, 4), (m.deep2[0], up2, 2)):
                outs.append(ops.head(src, hd.weight, hd.bias, sc))
                heads.append((hd, src, sc))
        ctx.cx, ctx.heads, ctx.nparams = cx, heads, len(params)
        ctx.recs = dict(rb1=rb1, rb2=rb2, rb3=rb3, rb4=rb4, ra=ra, rk1=rk1, rbr1=rbr1, rbr2=rbr2, rbr3=rbr3, ru3=ru3,
                        ru2=ru2, ru1=ru1, rd3=rd3, rd2=rd2, rd1=rd1)
        ctx.bufs = (down1, down2, down3, up3, up2, up1)
        return tuple(outs)

    @staticmethod
    def _backward(ctx, *douts):
        cx, R = ctx.cx, ctx.recs
        m = cx.m
        cx.x3s = ops.x3_mode() == ops.X3F and cx.dtype == torch.float32  # (the BACKWARD split decides about the dy scales)
        f = m.features
        h0, h1, h2 = f[0] // 2, f[1] // 2, f[2] // 2
        down1, down2, down3, up3, up2, up1 = ctx.bufs
        dsrc = {}
        top = None  # the output head on up1: folded into the backward of the decoder1 block (three logit planes)
        for (hd, src, sc), dout in zip(ctx.heads, douts):
            if dout is None and hd is m.out_conv and ctx.top_fused:
                # a loss built from the deep heads only: the fused top has no stored up1 to fall back on -- zero logit gradients
                dout = torch.zeros((ctx.out_shape), dtype=torch.float32, device=douts_device(douts))
            if dout is None:
                continue
            if hd is m.out_conv and (ctx.top_fused or (m.fold_head_bwd and cx.drop is None and hd.weight.shape[0] == 3)):
                top = (hd, dout)
                continue
            dx, dw, db = ops.head_bwd(src, hd.weight, dout, sc)
            cx.put(hd.weight, dw)
            cx.put(hd.bias, db)
            dsrc[src.data_ptr()] = dx

        def plus(a, t):
            b = dsrc.get(t.data_ptr())
            return a if b is None else a + b

        dcat1 = _block_bwd(cx, R["rd1"], None, head=top) if top is not None else _block_bwd(cx, R["rd1"], dsrc[up1.data_ptr()] if up1.data_ptr() in dsrc else torch.zeros_like(up1))
        d_up2 = plus(_conv_evo_bwd(cx, R["ru1"], ops.upsample_bwd(dcat1[..., h0:], 2)), up2)
        dcat2 = _block_bwd(cx, R["rd2"], d_up2)
        d_up3 = plus(_conv_evo_bwd(cx, R["ru2"], ops.upsample_bwd(dcat2[..., h1:], 2)), up3)
        dcat3 = _block_bwd(cx, R["rd3"], d_up3)
        d_assp = _conv_evo_bwd(cx, R["ru3"], ops.upsample_bwd(dcat3[..., h2:], 2))
        d_acat = _conv_evo_bwd(cx, R["rk1"], d_assp)
        d_down4 = _aspp_bwd(cx, m.aspp, R["ra"], d_acat)
        def level_bwd(rec, down, d_pooled, d_skip, need_dx=True):
            """Backward of an encoder block: its output gradient = d_skip (bridge) + MaxAvgPool backward(d_pooled), composed
            inside the block's EvoNorm / SE backward where the pooling forward recorded its arg-max bytes."""
            idx = getattr(down, "_pool_argmax", None)
            if idx is not None and m.fold_pool_bwd and cx.drop is None:
                return _block_bwd(cx, rec, None, need_dx, pool=(d_skip, d_pooled, idx, True))
            return _block_bwd(cx, rec, ops.maxpool2_bwd(down, d_pooled, dx_skip=d_skip, with_avg=True), need_dx)

        d_p3 = _block_bwd(cx, R["rb4"], d_down4)
        d_p2 = level_bwd(R["rb3"], down3, d_p3, _conv_evo_bwd(cx, R["rbr3"], dcat3[..., :h2]))
        d_p1 = level_bwd(R["rb2"], down2, d_p2, _conv_evo_bwd(cx, R["rbr2"], dcat2[..., :h1]))
        level_bwd(R["rb1"], down1, d_p1, _conv_evo_bwd(cx, R["rbr1"], dcat1[..., :h0]), need_dx=False)
        grads = cx.grads
        ctx.recs = ctx.bufs = ctx.cx = None
        return (None, None, None) + tuple(grads.get(i) for i in range(ctx.nparams))


# ------------------------------------------------------------------------------------------ module
class EquiUnetASSPEvo(_PackedWeightsModule):
    """Constructor signature of networks/equiunet2021.py:230-231."""
    name = "EquiUnetASSPEvo"

    def __init__(self, inplanes, num_classes, features, norm_layer=None, act="relu", deep_supervision=False, dropout=0,
                 refinement=False):
        super().__init__()
        warnings.warn("norm layer and activation specified will not be used ! only EVO !!")
        if refinement:
            raise NotImplementedError("equiunet_assp_evo_ref raises AttributeError in the reference too (SURVEY App. B)")
        if inplanes != 4 or num_classes > 4 or any(c % 16 for c in features):
            raise NotImplementedError("EquiUnetASSPEvo needs inplanes=4, num_classes<=4, widths multiple of 16 (F10)")
        print(f"EquiUnetASSP features: {features}")
        self.deep_supervision = deep_supervision
        self.act = act.upper()
        self.features = list(features)
        # "auto" = follow torch.autocast; BRATS_PRECISION=x3 makes the split-precision parity mode the default of an unmodified
        # training script run with --no_amp
        self.precision = os.environ.get("BRATS_PRECISION", "auto")
        self.conv_fp8 = None  # None | "fwd" | "all": e4m3 kernel for the 3x3x3 convolutions (see EquiUnet.conv_fp8)
        self.pack_plan = os.environ.get("BRATS_PACK_PLAN", "1") != "0"  # training: one multi-tensor weight-packing launch per step (ops.PackPlan)
        # the output head's backward inside the backward of the decoder1 block (brats_evonorm_se_bwd with dlogits); 0: the
        # separate brats_head_bwd pass, for same-box A/B runs
        self.fold_head_bwd = os.environ.get("BRATS_FOLD_HEAD", "1") != "0"
        # the pooling backward + bridge-gradient add inside the block's EvoNorm / SE backward (brats_evonorm_se_bwd_pool)
        self.fold_pool_bwd = os.environ.get("BRATS_FOLD_POOL", "1") != "0"
        # EvoNorm backward's first pass of a block's first unit inside the input-gradient launch of its second convolution
        # (ops.conv3d_bstats on the stored EvoNorm output + ops.evonorm_bwd_tiles): dz1 and y1 are read once instead of twice
        self.fold_bwd_stats = os.environ.get("BRATS_FOLD_BWD_STATS", "1") != "0"
        # ... and its forward on the last block's raw convolution output (brats_evonorm_head_fwd): up1 is never stored
        self.fold_head_fwd = os.environ.get("BRATS_FOLD_HEAD_FWD", os.environ.get("BRATS_FOLD_HEAD", "1")) != "0"
        self.skip_deep_heads_in_eval = False
        self._grad_sink = None
        self._grad_dest = None
        f = self.features
        self.encoder1 = ConvEvoBlockCorrected(inplanes, f[0])
        self.encoder2 = ConvEvoBlockCorrected(2 * f[0], f[1])
        self.encoder3 = ConvEvoBlockCorrected(2 * f[1], f[2])
        self.encoder4 = ConvEvoBlockCorrected(2 * f[2], f[3])
        self.bridge1 = ConvEvo(f[0], f[0] // 2)
        self.bridge2 = ConvEvo(f[1], f[1] // 2)
        self.bridge3 = ConvEvo(f[2], f[2] // 2)
        self.aspp = SimpleASPPEVO(f[3], f[3] // 4)
        self.upconv3 = ConvEvo(f[3], f[3] // 4)
        self.decoder3 = ConvEvoBlockCorrected(f[2], f[2])
        self.upconv2 = ConvEvo(f[2], f[2] // 4)
        self.decoder2 = ConvEvoBlockCorrected(f[1], f[1])
        self.upconv1 = ConvEvo(f[1], f[1] // 4)
        self.decoder1 = ConvEvoBlockCorrected(f[0], f[0])
        self.out_conv = _ConvParams(f[0], num_classes, 1, bias=True)
        if deep_supervision:
            self.deep3 = nn.ModuleList([_ConvParams(f[2], num_classes, 1, bias=True)])
            self.deep2 = nn.ModuleList([_ConvParams(f[1], num_classes, 1, bias=True)])
        # (the reference leaves torch's default init here: init_weights is commented out, :287)
        # --dropout p: nn.Dropout behind both EvoNorms of a block and behind every ConvEvo's but the ASPP's (networks/equiunet2021.py:
        # 200,203,219; :178).  Dropout streams: a block owns ids (u, u + 1), a ConvEvo one id.
        self._init_dropout(dropout)
        ids, nxt = {}, 0
        for mod in self.modules():
            if isinstance(mod, ConvEvoBlockCorrected):
                ids[mod], nxt = nxt, nxt + 2
            elif isinstance(mod, ConvEvo):
                ids[mod], nxt = nxt, nxt + 1
        self._unit_ids = ids

    def _dtype(self):
        if self.precision == "bf16":
            return torch.bfloat16
        if self.precision == "fp16":
            return torch.float16
        if self.precision in ("fp32", "x3", "fp16x3", "bf16x3", "x3fwd", "x3bwd"):
            return torch.float32
        if torch.is_autocast_enabled():  # the reference's switch (learning/engine.py:304): its autocast dtype is fp16
            return torch.float16 if torch.get_autocast_dtype("cuda") == torch.float16 else torch.bfloat16
        return torch.float32

    def _x3_modes(self):
        """(forward, backward) split of the 3x3x3 convolutions when the activations are f32 (ops.split_precision):
        see EquiUnet._x3_modes."""
        if self.precision in ("x3", "fp16x3"):
            return ops.X3F, ops.X3F
        if self.precision == "bf16x3":
            return ops.X3B, ops.X3B
        if self.precision in ("x3fwd", "x3bwd"):  # (diagnostic: one pass split, the other exact f32)
            return (ops.X3F, None) if self.precision == "x3fwd" else (None, ops.X3F)
        return None, None

    def forward(self, x):
        if not x.is_cuda:
            raise BratsHipError("brats21_amd.EquiUnetASSPEvo runs on the GPU only (no CPU fallback)")
        if x.dim() != 5 or x.shape[1] != 4 or any(s % 8 for s in x.shape[2:]):
            raise ValueError("expected input [N, 4, D, H, W] with D, H, W divisible by 8")
        self._fwd_grad = torch.is_grad_enabled()  # (inside autograd.Function.forward grad mode is always off)
        self._weights_may_have_changed()
        if self.training and self.pack_plan and torch.is_grad_enabled():
            ops.plan_for(self, x.device)  # all layers' weights (forward + dgrad layouts) packed by one launch
        outs = _AsspFn.apply(self, x.float(), self._dtype(), *tuple(self.parameters()))
        if self.deep_supervision:
            return outs[0], list(outs[1:])
        return outs[0]
