"""EquiUnet (GroupNorm + ReLU 3D U-Net) on the HIP kernels of libbrats_hip.so.

Drop-in for ``networks.equiunet2020.EquiUnet`` of the reference (networks/equiunet2020.py:407-500):
same constructor signature, same ``state_dict`` keys / shapes, same
``forward(x) -> (logits, [deep heads])`` contract on NCDHW float32 input, trainable through
``loss.backward()``.  Internally the whole network is ONE autograd node whose forward / backward are
explicit programs over NDHWC activations: torch.cat is replaced by producers writing into channel
slices of shared buffers, GroupNorm statistics come out of the convolution epilogue, and the skip /
pool gradient sum is fused into the pooling backward.

Precision follows the reference's switch (learning/engine.py:304 ``autocast(enabled=not no_amp)``):
under autocast the bf16-storage / f32-accumulate kernels run, otherwise the exact-f32 MFMA kernels
(the 1e-3 logit parity mode).  ``model.precision = "bf16" | "fp32"`` overrides it.
"""
import contextlib
import math

import os

import torch
import torch.nn as nn

from .. import ops
from .._lib import PACK_DGRAD, PACK_FWD, BratsHipError


# ------------------------------------------------------------------------------------------ parameter holders
class _ConvParams(nn.Module):
    """Parameter holder with nn.Conv3d's names, shapes and default init (weight [Co,Ci,k,k,k])."""

    def __init__(self, cin, cout, k, bias):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin, k, k, k))
        self.bias = nn.Parameter(torch.empty(cout)) if bias else None
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if bias:
            bound = 1.0 / math.sqrt(cin * k * k * k)
            nn.init.uniform_(self.bias, -bound, bound)


class _NormParams(nn.Module):
    """weight / bias of the unit's norm layer; with ``batch`` also nn.BatchNorm3d's buffers (same state-dict names)."""

    def __init__(self, c, batch=False):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(c))
        self.bias = nn.Parameter(torch.zeros(c))
        if batch:
            self.register_buffer("running_mean", torch.zeros(c))
            self.register_buffer("running_var", torch.ones(c))
            self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))


class _EstBNParams(nn.Module):
    """EstBN's parameters and buffers under its state-dict names (networks/factory.py:150-160)."""

    def __init__(self, c):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(c))
        self.bias = nn.Parameter(torch.zeros(c))
        self.register_buffer("running_mean", torch.zeros(c))
        self.register_buffer("running_var", torch.ones(c))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))
        self.register_buffer("estbn_moving_speed", torch.zeros(1))


class _BCNormParams(nn.Module):
    """``--norm bcn`` = BCNorm(C, 8, estimate=True) (networks/factory.py:125-147,189-190): per-GROUP weight / bias [1, 8, 1]
    around a per-(sample, group) normalisation of EstBN's output.  How it runs here (no new kernel): EstBN is a per-channel
    affine map with constants -- u = a_c * y + b_c, a_c = weight_c / sqrt(running_var_c + 1e-5), b_c = bias_c - running_mean_c *
    a_c -- because nothing in the reference ever sets ``estbn_moving_speed`` away from 0, so its running buffers never move.  A
    per-output-channel affine map behind a convolution IS a convolution: the unit packs a_c * W_c and passes b_c as the
    bias, the implicit-GEMM kernel writes u directly, and what is left of BCNorm is GroupNorm(8) whose per-channel gamma /
    beta are the group's weight / bias -- the existing statistics, apply and backward kernels.  The chain rule back to W, EstBN's
    weight / bias and the group parameters is [C]- and weight-sized torch algebra (_bcn_param_grads)."""

    def __init__(self, c, groups=8):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(1, groups, 1))
        self.bias = nn.Parameter(torch.zeros(1, groups, 1))
        self.bn = _EstBNParams(c)
        self.num_groups = groups
        self._speed_checked = False

    def _load_from_state_dict(self, *args, **kwargs):
        self._speed_checked = False
        return super()._load_from_state_dict(*args, **kwargs)

    def tables(self):
        """(a_c, b_c, gamma_c, beta_c): EstBN as a per-channel affine map; the group weight / bias expanded per channel."""
        if not self._speed_checked:  # (the reference reads it with .item() in EVERY forward; once per load is enough here)
            if float(self.bn.estbn_moving_speed.reshape(-1)[0]) != 0.0:
                raise NotImplementedError("--norm bcn: EstBN with estbn_moving_speed != 0 is not implemented (the reference "
                                          "never sets it: networks/factory.py:160 is its only mention)")
            self._speed_checked = True
        c = self.bn.weight.numel()
        a = self.bn.weight.detach() * torch.rsqrt(self.bn.running_var + 1e-5)
        b = self.bn.bias.detach() - self.bn.running_mean * a
        cpg = c // self.num_groups
        gamma = self.weight.detach().reshape(-1).repeat_interleave(cpg)
        beta = self.bias.detach().reshape(-1).repeat_interleave(cpg)
        return a.float().contiguous(), b.float().contiguous(), gamma.float().contiguous(), beta.float().contiguous()


def _bcn_param_grads(unit, dw_eff, db_eff, dgamma_c, dbeta_c):
    """Chain rule of the BCNorm unit (see _BCNormParams): W' = a_c W, bias' = b_c, gamma_c = group weight, beta_c = group bias.
    Returns {parameter: gradient} for conv.weight, bn.weight, bn.bias, bn.bn.weight, bn.bn.bias."""
    bcn = unit.bn
    a, _, _, _ = bcn.tables()
    w = unit.conv.weight.detach()
    k = torch.rsqrt(bcn.bn.running_var + 1e-5)
    da = (dw_eff * w).sum((1, 2, 3, 4))
    g = bcn.num_groups
    return {unit.conv.weight: dw_eff * a.view(-1, 1, 1, 1, 1),
            bcn.weight: dgamma_c.view(g, -1).sum(1).view(1, g, 1),
            bcn.bias: dbeta_c.view(g, -1).sum(1).view(1, g, 1),
            bcn.bn.weight: k * (da - bcn.bn.running_mean * db_eff),
            bcn.bn.bias: db_eff}


class ConvBnRelu(nn.Module):
    """conv3x3x3 (no bias) -> norm -> act -> Dropout(p)   (networks/equiunet2020.py:51-75).  The norm is GroupNorm(8)
    (``--norm group``) or InstanceNorm3d(affine=True) (``--norm instance``, the CLI default; networks/factory.py:
    179-188): the same kernels with 8 or ``planes`` statistics groups.  ``--norm batch`` (nn.BatchNorm3d(affine=True),
    round 4) is the same kernels again: NDHWC is contiguous, so the batch [N, D, H, W, C] viewed as ONE sample
    [1, N*D, H, W, C] with ``planes`` groups gives exactly the batch statistics (biased variance) in training mode, forward
    and backward; the running buffers are updated from the finalised mean / rstd, and eval mode builds the per-channel
    scale / shift from them (_bn_forward_tables).  ``--act prelu``: MONAI's Act["prelu"] = nn.PReLU()
    -- one learnable slope per unit, state-dict key ``<unit>.prelu.weight`` like the reference's nn.Sequential entry."""

    def __init__(self, inplanes, planes, dilation=1, norm="group", act="relu"):
        super().__init__()
        self.conv = _ConvParams(inplanes, planes, 3, bias=False)
        self.batch_norm = norm == "batch"
        self.bcn = norm == "bcn"
        self.bn = _BCNormParams(planes, 8) if self.bcn else _NormParams(planes, batch=self.batch_norm)
        if act == "prelu":
            self.prelu = nn.PReLU()
        self.dilation = dilation
        self.groups = 8 if norm in ("group", "bcn") else planes


class UBlock(nn.Module):
    """networks/equiunet2020.py:105-123"""

    def __init__(self, inplanes, midplanes, outplanes, dilation=(1, 1), norm="group", act="relu"):
        super().__init__()
        self.ConvBnRelu1 = ConvBnRelu(inplanes, midplanes, dilation[0], norm, act)
        self.ConvBnRelu2 = ConvBnRelu(midplanes, outplanes, dilation[1], norm, act)


def _head(cin, k):
    return nn.ModuleList([_ConvParams(cin, k, 1, bias=True)])  # key "<name>.0.weight" like nn.Sequential


class _PackedWeightsModule(nn.Module):
    """Keeps ops' no_grad packed-weight cache honest: the cache is dropped on every train() / eval() transition and on
    every grad-enabled forward, so weights changed without a version bump (``p.data.copy_`` of the reference's
    Ranger2020, learning/optimizer.py:243,253) are never served stale to a later evaluation."""

    PRECISIONS = ("auto", "bf16", "fp16", "fp32", "x3", "fp16x3", "bf16x3", "x3fwd", "x3bwd")

    @property
    def precision(self):
        return self._precision

    @precision.setter
    def precision(self, value):
        # a typo ("X3", "x3 ") must not fall through to autocast / exact f32 under a parity-mode label
        if value not in self.PRECISIONS:
            raise ValueError(f"model.precision / BRATS_PRECISION must be one of {self.PRECISIONS}, got {value!r}")
        self._precision = value

    def train(self, mode=True):
        if mode != self.training:
            ops.invalidate_packed_weights()
        return super().train(mode)

    def _init_dropout(self, p):
        """--dropout p (src/arguments_train.py:52).  The masks come from this library's Philox stream (csrc/dropout.hip), seeded
        from torch's CPU generator at construction (torch.manual_seed makes a run repeatable); the (seed, step counter) pair is a
        non-persistent buffer: not part of the state dict, like torch's own RNG state."""
        if not 0.0 <= float(p) < 1.0:
            raise ValueError(f"dropout probability has to be in [0, 1), got {p}")
        self.dropout_p = float(p)
        # The seed comes from a PRIVATE generator seeded with torch.initial_seed(): constructing a model never advances the global
        # CPU generator (the initial weights under a given torch.manual_seed do not depend on --dropout), and p = 0 draws
        # nothing.  Data-parallel ranks build identical replicas under one torch seed: the rank is mixed in at the first forward
        # (RANK of the launcher, else torch.distributed's rank) so that they draw different masks.
        seed = 0
        if self.dropout_p > 0.0:
            seed = int(torch.randint(0, 2 ** 62, (1,), generator=torch.Generator().manual_seed(torch.initial_seed())))
        self._dropout_seed_base, self._dropout_rank_mixed = seed, self.dropout_p == 0.0
        self.register_buffer("_dropout_state", torch.tensor([seed, 0], dtype=torch.int64), persistent=False)

    def _advance_dropout(self, device):
        """Next step's dropout state: the counter moves ON THE DEVICE (a captured step draws fresh masks at every replay); the
        returned copy belongs to this forward / backward pair."""
        if self._dropout_state.device != device:
            raise BratsHipError("brats21_amd: module and input are on different devices")
        if not self._dropout_rank_mixed:
            rank = os.environ.get("RANK")
            if rank is None:
                import torch.distributed as dist
                rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
            self._dropout_state[0] = (self._dropout_seed_base + 0x9E3779B97F4A7C15 * int(rank)) % (2 ** 63)
            self._dropout_rank_mixed = True
        self._dropout_state[1] += 1
        return self._dropout_state.clone()

    def _weights_may_have_changed(self):
        if torch.is_grad_enabled():
            ops.invalidate_packed_weights()


# ------------------------------------------------------------------------------------------ programs
class _AmaxSlots:
    """Zero-initialised device scalars for the |max| side outputs of the producer kernels (one fill per pass)."""

    def __init__(self, n, device):
        self.buf = torch.zeros(n, dtype=torch.float32, device=device)
        self.i = 0

    def take(self):
        s = self.buf[self.i:self.i + 1]
        self.i += 1
        return s


def _inherit_amax(dst, src):
    """max-pooling / trilinear interpolation never exceed the |max| of their input."""
    a = getattr(src, "_amax", None)
    if a is not None:
        dst._amax = a
    return dst


def _f8_ok(fp8, dtype, x, x2=None):
    return (bool(fp8) and ops.is16(dtype) and x.shape[1] >= ops.F8_MIN_SIZE and
            ops.conv_f8_chunk(x.shape[-1], x2.shape[-1] if x2 is not None else 0) > 0)


def _unit_act(unit, act):
    """(kernel activation, device slope | None): PReLU = leakyrelu whose slope is the unit's learnable scalar."""
    if act == "prelu":
        return "leakyrelu", unit.prelu.weight.detach()
    return act, None


def _one_sample(t):
    """[N, D, H, W, C] (dense) as one sample [1, N*D, H, W, C]: what turns per-sample statistics into batch statistics."""
    n, d, h, w, c = t.shape
    if not t.is_contiguous():
        raise BratsHipError("--norm batch needs dense activations")
    return t.view(1, n * d, h, w, c)


def _bn_fwd_tail(unit, x, x2, y, stats, act, pool, training, drop=None):
    """BatchNorm3d + activation behind the convolution (see ConvBnRelu)."""
    n, d, h, wd, c = y.shape
    bn = unit.bn
    if training:
        mean_rstd, scale_shift = ops.gn_finalize(stats.view(1, -1, c, 2), 1, c, c, n * d * h * wd, bn.weight.detach(), bn.bias.detach())
        with torch.no_grad():  # running buffers: momentum 0.1, UNBIASED variance (torch.nn.functional.batch_norm)
            cnt = float(n * d * h * wd)
            mean = mean_rstd[0, :, 0]
            var = (1.0 / mean_rstd[0, :, 1].double() ** 2 - 1e-5).clamp_min(0).float()
            bn.running_mean.mul_(0.9).add_(mean, alpha=0.1)
            bn.running_var.mul_(0.9).add_(var, alpha=0.1 * cnt / max(cnt - 1.0, 1.0))
            bn.num_batches_tracked += 1
    else:
        mean_rstd = None
        scale = bn.weight.detach() * torch.rsqrt(bn.running_var + 1e-5)
        scale_shift = torch.stack([scale, bn.bias.detach() - bn.running_mean * scale], -1).reshape(1, c, 2).contiguous().float()
    kact, slope_t = _unit_act(unit, act)
    z = ops.affine_act(_one_sample(y), scale_shift, kact, slope_t=slope_t).view(n, d, h, wd, c)
    if drop is not None:
        ops.dropout(z, drop[0], drop[1], drop[2], out=z)
    rec = (unit, x, x2, y, mean_rstd, scale_shift)
    if pool:
        return (z, ops.maxpool2(z, want_argmax=pool == "argmax")), rec
    return z, rec


def _cgr_fwd(unit, x, dtype, act, out=None, x2=None, fp8=None, slots=None, no_act=False, pool=False, lazy=False, training=True,
             drop=None):
    """One ConvBnRelu: pack -> implicit-GEMM conv over the virtual concat [x | x2] (+ tile statistics)
    -> finalize -> normalise+act.  fp8: the convolution runs on the e4m3 kernel (scales from the |max| the producer of
    x recorded); the normalise+act pass records the |max| of its own output for the next layer."""
    # drop = (p, state, unit id): nn.Dropout(p) behind the activation (networks/equiunet2020.py:62; training mode only).  The
    # activation is then always materialised (no fused pooling / head / on-load forms) and dropped in place (ops.dropout).
    if drop is not None:
        lazy = False
    w = unit.conv.weight
    cbias, gamma_c, beta_c = None, unit.bn.weight.detach(), unit.bn.bias.detach()
    if unit.bcn:
        # BCNorm (see _BCNormParams): EstBN's per-channel affine map rides in the convolution (weights a_c * W, bias b_c), the
        # rest is GroupNorm(8) with the group's weight / bias as per-channel gamma / beta
        if torch.is_grad_enabled():
            a_c, cbias, gamma_c, beta_c = unit.bn.tables()
            w = w.detach() * a_c.view(-1, 1, 1, 1, 1)
        else:
            # inference (~144 forwards of the same weights per volume): fold ONCE per unit -- the folded tensor stays the same
            # object, so ops.pack_weights' cache hits and no dead entry piles up in it.  Valid while none of the
            # tensors it was made from changed: version counters, addresses and the pack generation (writes through p.data)
            srcs = (unit.conv.weight, unit.bn.weight, unit.bn.bias, unit.bn.bn.weight, unit.bn.bn.bias, unit.bn.bn.running_var,
                    unit.bn.bn.running_mean)
            key = (ops.pack_generation(),) + tuple((t._version, t.data_ptr()) for t in srcs)
            hit = getattr(unit, "_bcn_fold", None)
            if hit is None or hit[0] != key:
                a_c, cbias, gamma_c, beta_c = unit.bn.tables()
                hit = unit._bcn_fold = (key, (w.detach() * a_c.view(-1, 1, 1, 1, 1), cbias, gamma_c, beta_c))
            w, cbias, gamma_c, beta_c = hit[1]
        fp8, lazy = None, False
    cout = w.shape[0]
    cin_pad = x.shape[-1] + (x2.shape[-1] if x2 is not None else 0)
    c1 = x.shape[-1] if x2 is not None else None
    if isinstance(x, ops.Pending) or isinstance(x2, ops.Pending):
        # inference: an input whose normalised activation was never stored -- applied on load where that form of the
        # convolution is built for this layer, materialised (the affine_act pass) otherwise
        if fp8 or not ops.conv_pre_ok(dtype, 3, unit.dilation, x.shape[-1], x2.shape[-1] if x2 is not None else 0, cout):
            x = x.materialize() if isinstance(x, ops.Pending) else x
            x2 = x2.materialize() if isinstance(x2, ops.Pending) else x2
    if _f8_ok(fp8, dtype, x, x2):
        wpk = ops.pack_weights_f8(w, PACK_FWD, cin_pad=cin_pad, c1=c1)
        y, stats = ops.conv3d_f8(x, wpk, cout, unit.dilation, want_stats=True, x2=x2, amax=getattr(x, "_amax", None),
                                 amax2=getattr(x2, "_amax", None) if x2 is not None else None)
    else:
        with ops.use_plan(None) if unit.bcn else contextlib.nullcontext():  # (a_c * W is a new tensor every step: not a plan entry)
            wpk = ops.pack_weights(w, dtype, PACK_FWD, cin_pad=cin_pad, dil=unit.dilation, c1=c1)
        # (_x3amax: the network input in split-precision mode -- the only conv input no normalisation has bounded; see _EquiUnetFn)
        y, stats = ops.conv3d(x, wpk, cout, 3, unit.dilation, want_stats=True, x2=x2, amax=getattr(x, "_x3amax", None), bias=cbias)
    n, d, h, wd, _ = y.shape
    if unit.batch_norm:
        return _bn_fwd_tail(unit, x, x2, y, stats, act, pool, training, drop)
    mean_rstd, scale_shift = ops.gn_finalize(stats, n, cout, unit.groups, d * h * wd, gamma_c, beta_c)
    if no_act:  # the last layer under the fused output head (ops.gn_head): the activation is applied on load there
        return y, (unit, x, x2, y, mean_rstd, scale_shift)
    amax = slots.take() if slots is not None else None
    kact, slope_t = _unit_act(unit, act)
    if lazy and slope_t is None and kact in ("relu", "leakyrelu") and amax is None and not pool:
        return ops.Pending(y, scale_shift, kact), None  # (no_grad only: nothing is taped)
    if pool and drop is None and kact in ("relu", "leakyrelu") and y.numel() * y.element_size() >= (256 << 20):
        # the layer ends an encoder level: normalise + act and the 2x2x2 max pool of the result in one pass -- for tensors
        # beyond the Infinity Cache (the 128^3 level: 188 us against 148 + 85); smaller ones are re-read from the cache by
        # the pooling kernel at no HBM cost and the two plain kernels are as fast (measured: 70 against 63 us)
        z, pooled = ops.affine_act_pool(y, scale_shift, kact, amax=amax, slope_t=slope_t, want_argmax=pool == "argmax")
        if amax is not None:
            z._amax = pooled._amax = amax  # (max|pool(z)| <= max|z|: the pooled tensor inherits the scale source)
        return (z, pooled), (unit, x, x2, y, mean_rstd, scale_shift)
    z = ops.affine_act(y, scale_shift, kact, out=out, amax=amax, slope_t=slope_t)
    if amax is not None:
        z._amax = amax
    if drop is not None:
        ops.dropout(z, drop[0], drop[1], drop[2], out=z)
    if pool:
        return (z, _inherit_amax(ops.maxpool2(z, want_argmax=pool == "argmax"), z)), (unit, x, x2, y, mean_rstd, scale_shift)
    return z, (unit, x, x2, y, mean_rstd, scale_shift)


def _cgr_bwd(rec, dz, dtype, act, grads, names, need_dx=True, sink=None, fp8=None, slots=None, side=None, dest=None, head=None,
             pool=None, bst=None, drop=None):
    """Returns dx, or (dx1, dx2) -- two dense tensors from one dgrad launch -- for a two-source unit.
    bst: the record of the unit that PRODUCED this unit's input (the first unit of the block).  Where the kernel form is built,
    the input-gradient launch also takes the first pass of that unit's GroupNorm backward (ops.conv3d_bstats) and dx is
    returned as (dx, tile_stats); handed on as `dz`, such a pair makes this function finish from the tile sums
    (ops.gn_act_bwd_tiles) instead of reading dz and y twice.
    fp8 == "all": the input gradient (dgrad) and -- for the dilation-1 layers the all-taps kernel covers -- the weight
    gradient run on the e4m3 kernels too, scaled by the |max| of dy that the GroupNorm backward records (and the |max| of
    the layer input recorded in the forward pass)."""
    unit, x, x2, y, mean_rstd, scale_shift = rec
    cin = unit.conv.weight.shape[1]
    gamma_c = unit.bn.weight.detach()
    w_bwd = unit.conv.weight
    if unit.bcn:
        a_c, _, gamma_c, _ = unit.bn.tables()
        w_bwd = unit.conv.weight.detach() * a_c.view(-1, 1, 1, 1, 1)  # the weights the forward convolved with
        fp8 = None
    tiles = None
    if isinstance(dz, tuple):
        dz, tiles = dz
    if drop is not None:  # the forward's mask, regenerated: d(dropout(z)) / dz = the same multiplier (ops.dropout)
        dz = ops.dropout(dz, drop[0], drop[1], drop[2])
    all8 = fp8 == "all" and ops.is16(dtype)
    f8 = all8 and need_dx and ops.conv_f8_chunk(y.shape[-1]) > 0
    # e4m3 weight gradient: where the all-taps kernel is built for the layer and the producers of x (x2) recorded |max|
    ax, ax2 = getattr(x, "_amax", None), (getattr(x2, "_amax", None) if x2 is not None else None)
    w8 = (all8 and unit.dilation == 1 and ax is not None and (x2 is None or ax2 is not None) and ops.is16(y.dtype)
          and ops.conv3d_wgrad_f8_ok(x, y, x2))
    # split precision on fp16 pairs: dy (values of 1e-6 and below) is scaled by a power of two taken from its recorded |max|
    x3s = ops.x3_mode() == ops.X3F and dtype == torch.float32
    amax = slots.take() if ((f8 or w8 or x3s) and slots is not None) else None
    kact, slope_t = _unit_act(unit, act)
    if slope_t is not None:
        grads[names[unit.prelu.weight]] = ops.prelu_slope_grad(dz, y, scale_shift)
        if sink is not None:
            sink(names[unit.prelu.weight], grads[names[unit.prelu.weight]])
    if unit.batch_norm:
        if mean_rstd is None:
            raise NotImplementedError("--norm batch: backward through an eval-mode forward (running statistics) is not implemented")
        n_, d_, h_, w_, c_ = y.shape
        dy, dgamma, dbeta = ops.gn_act_bwd(_one_sample(dz), _one_sample(y), scale_shift, mean_rstd, gamma_c, c_, kact,
                                           amax=amax, slope_t=slope_t)
        dy = dy.view(n_, d_, h_, w_, c_)
        if amax is not None:
            dy._amax = amax
    elif head is not None:
        # the last layer: its output feeds only the 1x1x1 head, whose backward is folded into the GroupNorm backward --
        # d(up1) is never written, the head's weight / bias gradients come out of the same passes (ops.gn_act_bwd_head)
        hd, dout = head
        dy, dgamma, dbeta, dhw, dhb = ops.gn_act_bwd_head(dout, hd.weight, y, scale_shift, mean_rstd, gamma_c,
                                                          unit.groups, kact, amax=amax)
        for prm, g in ((hd.weight, dhw), (hd.bias, dhb)):
            grads[names[prm]] = g
            if sink is not None:
                sink(names[prm], g)
    elif pool is not None:
        # the layer ends an encoder level: dz = skip gradient + max-pool backward, composed inside the GroupNorm backward from
        # (d_skip, d_pooled, arg-max bytes) -- the pooling backward's output tensor is never written (ops.gn_act_bwd_pool)
        dy, dgamma, dbeta = ops.gn_act_bwd_pool(pool[0], pool[1], pool[2], y, scale_shift, mean_rstd, gamma_c,
                                                unit.groups, kact, amax=amax)
    elif tiles is not None:
        dy, dgamma, dbeta = ops.gn_act_bwd_tiles(tiles, dz, y, scale_shift, mean_rstd, gamma_c, unit.groups, kact,
                                                 amax=amax)
    else:
        dy, dgamma, dbeta = ops.gn_act_bwd(dz, y, scale_shift, mean_rstd, gamma_c, unit.groups, kact, amax=amax,
                                           slope_t=slope_t)
    # data-parallel: the weight gradient is written straight into its slice of the all-reduce bucket
    wdst = dest(names[unit.conv.weight]) if dest is not None else None
    # (model.wgrad_stream = "small": only the 16^3 / 32^3 levels, whose weight-gradient grids are <= 1 workgroup per CU and leave
    #  most of the chip idle beside the next layer's equally small input-gradient launch)
    if side is not None and getattr(unit, "_side_small_only", False) and x.shape[1] > 32:
        side = None
    with ops.side_stream(side, dy, x, x2) as on_side:
        # (the weight gradient depends only on dy and the saved input and nobody but the optimizer waits for it: on the
        # side stream it fills the CUs that the tail of the input-gradient kernel and the small GroupNorm launches leave idle)
        if w8 and amax is not None:
            dw = ops.conv3d_wgrad_f8(x, dy, ax, amax, x2=x2, amax2=ax2, out=wdst)
        elif x2 is not None and x.shape[-1] % 16:  # narrow test widths only: the wgrad ci tile (16) would straddle x | x2
            dw, db = ops.conv3d_wgrad(torch.cat([x, x2], -1), dy, 3, unit.dilation, amax_dy=amax if x3s else None, want_dbias=unit.bcn)
        else:
            dw, db = ops.conv3d_wgrad(x, dy, 3, unit.dilation, x2=x2, out=None if unit.bcn else wdst, amax_dy=amax if x3s else None,
                                      want_dbias=unit.bcn)
        dw = dw[:, :cin].contiguous() if dw.shape[1] != cin else dw
        on_side(dw)
    if unit.bcn:
        pg = _bcn_param_grads(unit, dw, db, dgamma, dbeta)  # dw / db are the gradients of a_c * W and b_c
    else:
        pg = {unit.conv.weight: dw, unit.bn.weight: dgamma, unit.bn.bias: dbeta}
    for prm, g in pg.items():
        grads[names[prm]] = g
        if sink is not None:  # data-parallel: hand finished gradients to the bucketed all-reduce right away
            sink(names[prm], g)
    if not need_dx:
        return None
    if f8:
        wpk = ops.pack_weights_f8(unit.conv.weight, PACK_DGRAD)

        def dgrad(**kw):
            return ops.conv3d_f8(dy, wpk, cin, unit.dilation, amax=amax, **kw)
    else:
        with ops.use_plan(None) if unit.bcn else contextlib.nullcontext():
            wpk = ops.pack_weights(w_bwd, dtype, PACK_DGRAD, dil=unit.dilation)

        def dgrad(**kw):
            return ops.conv3d(dy, wpk, cin, 3, unit.dilation, amax=amax if x3s else None, **kw)
    if x2 is None:
        if bst is not None and not f8:
            u1, _, _, y1, mr1, ss1 = bst
            kact1, slope1 = _unit_act(u1, act)
            if (not u1.batch_norm and not u1.bcn and not unit.bcn and mr1 is not None and y1.shape[-1] == cin and y1.dtype == dy.dtype
                    and ops.conv_bstats_ok(dtype, unit.dilation, dy.shape[-1], cin, kact1, slope1)):
                # (dx, tile sums of u1's GroupNorm backward)
                return ops.conv3d_bstats(dy, wpk, cin, unit.dilation, y1, ss1, kact1, amax=amax if x3s else None)
        dx, _ = dgrad()
        return dx
    c1 = x.shape[-1]
    if c1 % ops.split_granule(cin) == 0:
        dx, _ = dgrad(split=c1)  # (dx1, dx2): two dense tensors, one launch
        return dx
    dx, _ = dgrad()  # narrow test widths: slice views of one tensor
    return dx[..., :c1], dx[..., c1:]


def douts_device(douts):
    return next(d.device for d in douts if d is not None)


class _EquiUnetFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, x, dtype, *params):
        # training: the module packed all layers' weights up front (ops.plan_for); pack_weights() then returns views
        ctx.plan = ops._PLANS.get(model) if (model.training and model.pack_plan) else None
        ctx.x3 = model._x3_modes() if dtype == torch.float32 else (None, None)
        with ops.use_plan(ctx.plan), ops.split_precision(ctx.x3[0]):
            return _EquiUnetFn._forward(ctx, model, x, dtype, *params)

    @staticmethod
    def backward(ctx, *douts):
        with ops.use_plan(ctx.plan), ops.split_precision(ctx.x3[1]):
            return _EquiUnetFn._backward(ctx, *douts)

    @staticmethod
    def _forward(ctx, model, x, dtype, *params):
        m = model
        act = m.act
        f = m.features
        n, _, d, h, w = x.shape
        dev = x.device
        tape = []

        fp8 = m.conv_fp8 if ops.is16(dtype) else None
        slots = _AmaxSlots(32, dev) if fp8 else None

        will_bwd = any(ctx.needs_input_grad) and model._fwd_grad  # (needs_input_grad ignores no_grad)

        # inference (no_grad): the activation between the two units of a block has ONE reader, the block's second
        # convolution -- it is never stored (ops.Pending: normalise + act applied on load, model.norm_on_load)
        lazy_ok = (not will_bwd) and (not torch.is_grad_enabled()) and m.norm_on_load and ops.is16(dtype)

        # nn.Dropout(p) behind every unit's activation, training mode only (networks/equiunet2020.py:62): the state (seed, step
        # counter) is advanced on the device and a copy travels to the backward, which regenerates the masks from it
        drop_state = m._advance_dropout(dev) if (m.training and m.dropout_p > 0.0) else None
        ctx.drop_state = drop_state
        if drop_state is not None and fp8:
            raise NotImplementedError("--dropout > 0 with the e4m3 convolution path is not implemented")

        def drop_of(unit):
            return (m.dropout_p, drop_state, m._unit_ids[unit]) if drop_state is not None else None

        def cgr(unit, xin, x2=None, pool=False, lazy=False):
            # (training: the fused pooling pass also records the arg-max bytes its backward reads)
            z, rec = _cgr_fwd(unit, xin, dtype, act, None, x2, fp8, slots, pool=("argmax" if will_bwd else True) if pool else False,
                              lazy=lazy and lazy_ok, training=m.training, drop=drop_of(unit))
            tape.append(rec)
            return z

        def up(t):
            return _inherit_amax(ops.upsample(t, 2), t)

        x0 = ops.ncdhw_to_ndhwc(x, dtype, cpad=8 if (ops.is16(dtype) or ops.x3_active()) else 4)
        if ops.x3_mode() == ops.X3F:
            # fp16 pairs overflow at |x| >= 65504 (hi = inf, lo = NaN: silently NaN logits).  Every other convolution
            # input is a normalised activation; the network INPUT is whatever the caller passes (un-normalised volumes reach
            # 3e4 and more), so its |max| is recorded (one 134 MB pass at 2 x 128^3) and the first layer's forward scales by the
            # matching power of two (brats_conv3d_x3_fwd's xamax: exact).  The first layer's WEIGHT gradient still splits the
            # unscaled input: split-precision training expects z-scored inputs like the reference's pipeline produces
            # (utils/transforms.py:364-385), inference does not.
            x0._x3amax = ops.absmax(x0)
        # encoder (networks/equiunet2020.py:469-475); every tensor is dense NDHWC, the decoder convolutions
        # read the virtual concat [skip | up-sampled] from two pointers (no torch.cat, no strided slices)
        # (the last layer of a level writes its activation -- the skip connection -- and the max-pooled tensor in one pass)
        down1, p1 = cgr(m.encoder1.ConvBnRelu2, cgr(m.encoder1.ConvBnRelu1, x0, lazy=True), pool=True)
        down2, p2 = cgr(m.encoder2.ConvBnRelu2, cgr(m.encoder2.ConvBnRelu1, p1, lazy=True), pool=True)
        down3, p3 = cgr(m.encoder3.ConvBnRelu2, cgr(m.encoder3.ConvBnRelu1, p2, lazy=True), pool=True)
        down4 = cgr(m.encoder4.ConvBnRelu2, cgr(m.encoder4.ConvBnRelu1, p3, lazy=True))
        # bottom (:477-478): dilated block, then conv over cat[down4, bottom]
        deep_heads = m.deep_supervision and not (m.skip_deep_heads_in_eval and not m.training)
        bottom = cgr(m.bottom.ConvBnRelu2, cgr(m.bottom.ConvBnRelu1, down4, lazy=True), lazy=not deep_heads)
        bottom_2 = cgr(m.bottom_2, down4, x2=bottom)
        # decoder (:481-486)
        up3 = cgr(m.decoder3.ConvBnRelu2, cgr(m.decoder3.ConvBnRelu1, down3, x2=up(bottom_2), lazy=True))
        up2 = cgr(m.decoder2.ConvBnRelu2, cgr(m.decoder2.ConvBnRelu1, down2, x2=up(up3), lazy=True))
        u1 = cgr(m.decoder1.ConvBnRelu1, down1, x2=up(up2), lazy=True)
        # the last layer's activation up1 feeds only the output head: where the kernels for it are built, the head reads
        # the raw convolution output and applies GroupNorm + act on load (ops.gn_head), the backward recomputes what it
        # needs (ops.gn_act_bwd_head) -- up1 (2 x 403 MB written + read at 2 x 48 x 128^3) is never stored
        kact, slope_t = _unit_act(m.decoder1.ConvBnRelu2, act)
        nk = m.outconv.weight.shape[0]
        fuse_top = (m.fold_head_fwd and drop_state is None and not m.decoder1.ConvBnRelu2.batch_norm and slope_t is None and kact in ("relu", "leakyrelu") and nk <= 4
                    and (not will_bwd or (m.fold_head_bwd and ops.head_fold_ok(m.outconv.weight, kact, slope_t))))
        if fuse_top:
            y1, rec1 = _cgr_fwd(m.decoder1.ConvBnRelu2, u1, dtype, act, None, None, fp8, slots, no_act=True)
            tape.append(rec1)
            up1 = None
            outs = [ops.gn_head(y1, rec1[5], m.outconv.weight, m.outconv.bias, kact)]
        else:
            up1 = cgr(m.decoder1.ConvBnRelu2, u1)
            outs = [ops.head(up1, m.outconv.weight, m.outconv.bias, 1)]
        ctx.top_fused = fuse_top
        ctx.out_shape = tuple(outs[0].shape)
        heads = [(m.outconv, up1, 1)]
        if deep_heads:
            for hd, src, sc in ((m.deep_bottom[0], bottom, 8), (m.deep_bottom2[0], bottom_2, 8), (m.deep3[0], up3, 4),
                                (m.deep2[0], up2, 2)):
                outs.append(ops.head(src, hd.weight, hd.bias, sc))
                heads.append((hd, src, sc))
        ctx.model, ctx.dtype, ctx.tape, ctx.heads = m, dtype, tape, heads
        ctx.bufs = (down1, down2, down3, down4, bottom, bottom_2, up3, up2, up1)
        ctx.nparams = len(params)
        return tuple(outs)

    @staticmethod
    def _backward(ctx, *douts):
        m, dtype, tape = ctx.model, ctx.dtype, ctx.tape
        act, f = m.act, m.features
        names = {p: i for i, p in enumerate(m.parameters())}
        grads = {}
        down1, down2, down3, down4, bottom, bottom_2, up3, up2, up1 = ctx.bufs
        rec = {r[0]: r for r in tape}

        fp8 = m.conv_fp8 if ops.is16(dtype) else None
        slots = _AmaxSlots(32, douts[0].device) if (fp8 == "all" or ops.x3_mode() == ops.X3F) else None

        # weight gradients on a side stream (model.wgrad_stream); with gradient buckets they stay on the main stream: the
        # buckets' copies and collectives are ordered against it
        side = ops.get_side_stream(douts[0].device) if (m.wgrad_stream and m._grad_sink is None) else None
        for u in rec:
            u._side_small_only = m.wgrad_stream == "small"

        drop_state = ctx.drop_state  # dropout: the folds that never materialise a unit's output gradient are off

        def cbw(unit, dz, need_dx=True, head=None, pool=None, first=None):
            # first: the block's first unit, whose output is this unit's only input -- its GroupNorm backward's first pass rides
            # in this unit's input-gradient launch (model.fold_bwd_stats)
            bst = rec[first] if (first is not None and m.fold_bwd_stats and drop_state is None) else None
            drop = (m.dropout_p, drop_state, m._unit_ids[unit]) if drop_state is not None else None
            return _cgr_bwd(rec[unit], dz, dtype, act, grads, names, need_dx, m._grad_sink, fp8, slots, side, m._grad_dest, head, pool, bst,
                            drop)

        def level_bwd(unit, down, d_pooled, d_skip, need_dx=True, first=None):
            """Backward of the last layer of an encoder level: its output gradient = d_skip + max-pool backward(d_pooled)."""
            idx = getattr(down, "_pool_argmax", None)
            kact, slope_t = _unit_act(unit, act)
            if (idx is not None and m.fold_pool_bwd and drop_state is None and not unit.batch_norm and slope_t is None
                    and kact in ("relu", "leakyrelu")):
                return cbw(unit, None, need_dx, pool=(d_skip, d_pooled, idx), first=first)
            return cbw(unit, ops.maxpool2_bwd(down, d_pooled, dx_skip=d_skip), need_dx, first=first)

        # heads: d(logits) -> gradient w.r.t. their NDHWC source tensors
        dsrc = {}
        top = None  # the output head on up1: folded into the GroupNorm backward of the last layer where that is built
        for (hd, src, sc), dout in zip(ctx.heads, douts):
            if dout is None and hd is m.outconv and ctx.top_fused:
                # a loss built from the deep heads only: the fused top has no stored up1 to fall back on -- zero logit gradients
                dout = torch.zeros((ctx.out_shape), dtype=torch.float32, device=douts_device(douts))
            if dout is None:
                continue
            if hd is m.outconv and (ctx.top_fused or (m.fold_head_bwd and drop_state is None and not m.decoder1.ConvBnRelu2.batch_norm
                                                       and ops.head_fold_ok(hd.weight, *_unit_act(m.decoder1.ConvBnRelu2, act)))):
                top = (hd, dout)
                continue
            dx, dw, db = ops.head_bwd(src, hd.weight, dout, sc)
            grads[names[hd.weight]] = dw
            grads[names[hd.bias]] = db
            if m._grad_sink is not None:
                m._grad_sink(names[hd.weight], dw)
                m._grad_sink(names[hd.bias], db)
            key = src.data_ptr()
            dsrc[key] = dx if key not in dsrc else dsrc[key] + dx

        def extra(t):
            return dsrc.get(t.data_ptr())

        def plus(a, b):
            return a if b is None else a + b

        def blk(b):  # (first, second) unit of a block
            return b.ConvBnRelu1, b.ConvBnRelu2

        c1, c2 = blk(m.decoder1)
        d_c1 = cbw(c2, None, head=top, first=c1) if top is not None else cbw(c2, extra(up1) if extra(up1) is not None else torch.zeros_like(up1), first=c1)
        d_skip1, d_u1 = cbw(c1, d_c1)
        d_up2 = plus(ops.upsample_bwd(d_u1, 2), extra(up2))
        c1, c2 = blk(m.decoder2)
        d_skip2, d_u2 = cbw(c1, cbw(c2, d_up2, first=c1))
        d_up3 = plus(ops.upsample_bwd(d_u2, 2), extra(up3))
        c1, c2 = blk(m.decoder3)
        d_skip3, d_u3 = cbw(c1, cbw(c2, d_up3, first=c1))
        d_b2 = plus(ops.upsample_bwd(d_u3, 2), extra(bottom_2))
        d_skip4, d_bot = cbw(m.bottom_2, d_b2)
        d_bottom = plus(d_bot, extra(bottom))
        c1, c2 = blk(m.bottom)
        d_down4 = d_skip4 + cbw(c1, cbw(c2, d_bottom, first=c1))
        c1, c2 = blk(m.encoder4)
        d_p3 = cbw(c1, cbw(c2, d_down4, first=c1))
        c1, c2 = blk(m.encoder3)
        d_p2 = cbw(c1, level_bwd(c2, down3, d_p3, d_skip3, first=c1))
        c1, c2 = blk(m.encoder2)
        d_p1 = cbw(c1, level_bwd(c2, down2, d_p2, d_skip2, first=c1))
        c1, c2 = blk(m.encoder1)
        cbw(c1, level_bwd(c2, down1, d_p1, d_skip1, first=c1), need_dx=False)
        if side is not None:
            torch.cuda.current_stream().wait_stream(side)  # every weight gradient is complete before autograd hands them on
        ctx.tape = ctx.bufs = None
        return (None, None, None) + tuple(grads.get(i) for i in range(ctx.nparams))


# ------------------------------------------------------------------------------------------ module
class EquiUnet(_PackedWeightsModule):
    """Constructor signature of networks/equiunet2020.py:413-414."""
    name = "EquiUnet"

    def __init__(self, inplanes, num_classes, features, norm_layer=None, act="relu", deep_supervision=False, dropout=0,
                 refinement=False):
        super().__init__()
        if norm_layer not in ("group", "instance", "batch", "bcn"):
            raise NotImplementedError(f"brats21_amd.EquiUnet implements --norm group|instance|batch|bcn (got {norm_layer!r})")
        if norm_layer == "batch" and act == "prelu":
            raise NotImplementedError("--norm batch with --act prelu is not implemented")
        if act not in ("relu", "leakyrelu", "elu", "prelu", "swish", "mish"):
            raise NotImplementedError(f"brats21_amd.EquiUnet implements --act relu|leakyrelu|elu|prelu|swish|mish (got {act!r})")
        if refinement:
            raise NotImplementedError("equiunet_ref (RefUnet) is outside the accelerated hot path")
        if inplanes != 4 or num_classes > 4 or any(c % 8 for c in features):
            raise NotImplementedError("EquiUnet kernels need inplanes=4, num_classes<=4, widths multiple of 8")
        print(f"EquiUnet features: {features}")
        self.deep_supervision = deep_supervision
        self.act = act
        self.features = list(features)
        # nn.Dropout(p) behind every ConvBnRelu's activation (networks/equiunet2020.py:62; --dropout, src/arguments_train.py:52)
        self._init_dropout(dropout)
        # "auto" = follow torch.autocast; BRATS_PRECISION=x3 makes the split-precision parity mode the default of an unmodified
        # training script run with --no_amp
        self.precision = os.environ.get("BRATS_PRECISION", "auto")
        # None | "fwd" | "all": run the 3x3x3 convolutions (forward / forward + input gradients) on the e4m3 MFMA kernel
        # when the activations are bf16 (BASELINE.json configs[4]); the weight gradients stay bf16
        self.conv_fp8 = None
        self.skip_deep_heads_in_eval = False
        # inference (no_grad, 16-bit): the activation between the two convolutions of a block is applied on load by the second
        # one and never stored (ops.Pending / brats_conv3d_fwd_pre); BRATS_NORM_ON_LOAD=0: the two-pass path, for A/B runs
        self.norm_on_load = os.environ.get("BRATS_NORM_ON_LOAD", "1") != "0"
        # weight gradients on a second HIP stream (they depend only on dy and the saved input).  Off: measured 16.40 ->
        # 16.63 ms / step same-box -- the all-taps kernel owns a CU's whole LDS, so the two streams only take CUs from
        # each other, and the tails they could fill are shorter than the interference they add
        ws = os.environ.get("BRATS_WGRAD_STREAM", "0")
        self.wgrad_stream = "small" if ws == "small" else ws != "0"
        self._grad_sink = None  # set by brats21_amd.ddp.GradientBuckets
        self._grad_dest = None  # (ditto: parameter index -> its slice of an all-reduce bucket, or None)
        # training: one multi-tensor weight-packing launch per step (ops.PackPlan).  Off by default here: this network's
        # step is GPU-bound; the single launch (0.11 ms) saves 0.1 ms over the 33 per-layer ones (6.6 us each), and the
        # convolutions lose 0.10-0.18 ms per step on weights that were packed long before their layer runs and have left
        # L2 (same-box A/B); EquiUnetASSPEvo (host-bound eager) gains 10 %.
        self.pack_plan = os.environ.get("BRATS_PACK_PLAN", "0") != "0"
        # the output head's backward inside the GroupNorm backward of the last layer (brats_gn_act_bwd_head); 0: the two-call
        # path (brats_head_bwd + brats_gn_act_bwd) for same-box A/B runs
        self.fold_head_bwd = os.environ.get("BRATS_FOLD_HEAD", "1") != "0"
        # the pooling backward + skip add inside the GroupNorm backward of the level's last layer (brats_gn_act_bwd_pool)
        self.fold_pool_bwd = os.environ.get("BRATS_FOLD_POOL", "1") != "0"
        # GroupNorm backward's first pass (sum u, sum u * xhat) of a block's first unit inside the input-gradient launch of its
        # second unit (brats_conv3d_fwd_bstats + brats_gn_act_bwd_tiles): dz and y are read once instead of twice
        self.fold_bwd_stats = os.environ.get("BRATS_FOLD_BWD_STATS", "1") != "0"
        # ... and its forward on the last layer's raw convolution output (brats_gn_head_fwd): up1 is never stored
        self.fold_head_fwd = os.environ.get("BRATS_FOLD_HEAD_FWD", os.environ.get("BRATS_FOLD_HEAD", "1")) != "0"
        f = self.features
        nl = norm_layer
        self.encoder1 = UBlock(inplanes, f[0], f[0], norm=nl, act=act)
        self.encoder2 = UBlock(f[0], f[1], f[1], norm=nl, act=act)
        self.encoder3 = UBlock(f[1], f[2], f[2], norm=nl, act=act)
        self.encoder4 = UBlock(f[2], f[3], f[3], norm=nl, act=act)
        self.bottom = UBlock(f[3], f[3], f[3], (2, 2), norm=nl, act=act)
        self.bottom_2 = ConvBnRelu(f[3] * 2, f[2], norm=nl, act=act)
        self.decoder3 = UBlock(f[2] * 2, f[2], f[1], norm=nl, act=act)
        self.decoder2 = UBlock(f[1] * 2, f[1], f[0], norm=nl, act=act)
        self.decoder1 = UBlock(f[0] * 2, f[0], f[0], norm=nl, act=act)
        self.outconv = _ConvParams(f[0], num_classes, 1, bias=True)
        if deep_supervision:
            self.deep_bottom = _head(f[3], num_classes)
            self.deep_bottom2 = _head(f[2], num_classes)
            self.deep3 = _head(f[1], num_classes)
            self.deep2 = _head(f[0], num_classes)
        self._unit_ids = {u: i for i, u in enumerate(mod for mod in self.modules() if isinstance(mod, ConvBnRelu))}
        # init_weights(self, "kaiming"), networks/factory.py:203-224: kaiming-normal fan_out on conv weights
        print("initialize network with kaiming")
        for mod in self.modules():
            if isinstance(mod, _ConvParams):
                nn.init.kaiming_normal_(mod.weight.data, a=0.0, mode="fan_out")
            elif isinstance(mod, _NormParams) and norm_layer == "batch":  # factory.py:219-221: BatchNorm3d weights ~ N(1, 0.02)
                nn.init.normal_(mod.weight.data, 1.0, 0.02)
                nn.init.constant_(mod.bias.data, 0.0)

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
        precision "x3" (= "fp16x3") = fp16 pairs (f32-class: 2^-22 per product) forward AND backward -- dY, whose values lie far
        below fp16's range, is scaled by a power of two from the |max| its producer kernel records (brats_conv3d_x3_fwd /
        _x3_wgrad; the reference needs a GradScaler for the same reason); "bf16x3" = bf16 pairs everywhere (2^-16 per product:
        logits ~5e-4, gradients ~5e-3 from f64 -- measured, tests/test_x3_gpu.py); anything else = the exact-f32 MFMA kernels."""
        if self.precision in ("x3", "fp16x3"):
            return ops.X3F, ops.X3F
        if self.precision == "bf16x3":
            return ops.X3B, ops.X3B
        if self.precision in ("x3fwd", "x3bwd"):  # (diagnostic: one pass split, the other exact f32)
            return (ops.X3F, None) if self.precision == "x3fwd" else (None, ops.X3F)
        return None, None

    def forward(self, x):
        if not x.is_cuda:
            raise BratsHipError("brats21_amd.EquiUnet runs on the GPU only (no CPU fallback); move input/model to cuda")
        if x.dim() != 5 or x.shape[1] != 4 or any(s % 8 for s in x.shape[2:]):
            raise ValueError("expected input [N, 4, D, H, W] with D, H, W divisible by 8")
        params = tuple(self.parameters())
        self._fwd_grad = torch.is_grad_enabled()  # (inside autograd.Function.forward grad mode is always off)
        self._weights_may_have_changed()
        if self.training and self.pack_plan and torch.is_grad_enabled():
            ops.plan_for(self, x.device)  # all layers' weights (forward + dgrad layouts) packed by one launch
        outs = _EquiUnetFn.apply(self, x.float(), self._dtype(), *params)
        if self.deep_supervision:
            return outs[0], list(outs[1:])
        return outs[0]
