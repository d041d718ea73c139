"""ORACLE (test infrastructure, NOT product code): CPU fp32 restatement of the reference's two
in-scope networks as *pure functions over a state dict*.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
The product path (brats21_amd/) never does and has no CPU fallback.

Every function cites the reference lines it restates (paths relative to /root/reference).  The
state-dict key names are the reference's own, so the same ``sd`` can be loaded
into the reference modules (tests/golden/make_golden.py does exactly that to pin this oracle) and
into the product modules.  Layout here is the reference's NCDHW; arithmetic is stock torch CPU ops
(F.conv3d, F.group_norm, F.max_pool3d, F.interpolate), i.e. the same ATen kernels the reference's
CPU path executes.

Pinning status: ``equiunet_forward`` is pinned by tests/golden/equiunet_*.npz (generated from the
reference source itself).  ``assp_evo_forward`` is pinned against the reference source *under the
MONAI stub* (oracle/refshim.py) -- parity unpinned at the MONAI boundary (MaxAvgPool,
ResidualSELayer), see DESIGN.md.
"""
import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------- EquiUnet (GN + act)
def _act(x, act, slope=None):
    # networks/factory.py:195-200 -> MONAI Act lookup (src/arguments_train.py:49-50: elu, relu, leakyrelu, prelu, swish, mish)
    if act == "relu":
        return F.relu(x)
    if act == "leakyrelu":
        return F.leaky_relu(x, 0.01)
    if act == "elu":
        return F.elu(x)            # MONAI Act["elu"] = nn.ELU (alpha 1)
    if act == "prelu":
        return F.prelu(x, slope)   # MONAI Act["prelu"] = nn.PReLU(): one learnable slope, "<unit>.prelu.weight"
    if act == "swish":
        return x * torch.sigmoid(x)  # MONAI Swish(alpha=1.0): input * sigmoid(alpha * input)
    if act == "mish":
        return F.mish(x)           # MONAI Mish: x * tanh(softplus(x))
    raise ValueError(act)


def conv_gn_act(sd, pre, x, dilation=1, act="relu", norm="group", training=True, new_stats=None, drop=None):
    """ConvBnRelu, networks/equiunet2020.py:51-75: conv3x3x3 (no bias, pad=dil) -> norm -> act -> Dropout(p=0)
    (identity).  norm "group" = GroupNorm(8, C, affine), "instance" = InstanceNorm3d(C, affine=True) (the CLI
    default, src/arguments_train.py:48) = per-(sample, channel) statistics, biased variance, eps 1e-5, "batch" =
    BatchNorm3d(C, affine=True) (networks/factory.py:179-188): batch statistics in training mode (momentum 0.1 update of the
    running buffers, unbiased variance there), the running buffers in eval mode.  The oracle is pure: the updated buffers
    of a training-mode call are returned through ``new_stats`` ({key: tensor}) instead of being written into ``sd``."""
    y = F.conv3d(x, sd[pre + ".conv.weight"], None, 1, dilation, dilation)
    if norm == "group":
        y = F.group_norm(y, 8, sd[pre + ".bn.weight"], sd[pre + ".bn.bias"], 1e-5)
    elif norm == "instance":
        y = F.instance_norm(y, None, None, sd[pre + ".bn.weight"], sd[pre + ".bn.bias"], True, 0.1, 1e-5)
    elif norm == "batch":
        rm, rv = sd[pre + ".bn.running_mean"].detach().clone().to(y.dtype), sd[pre + ".bn.running_var"].detach().clone().to(y.dtype)
        y = F.batch_norm(y, rm, rv, sd[pre + ".bn.weight"], sd[pre + ".bn.bias"], training, 0.1, 1e-5)
        if training and new_stats is not None:
            new_stats[pre + ".bn.running_mean"], new_stats[pre + ".bn.running_var"] = rm, rv
    elif norm == "bcn":
        y = bcnorm(sd, pre + ".bn", y)
    else:
        raise ValueError(norm)
    y = _act(y, act, sd.get(pre + ".prelu.weight"))
    # nn.Dropout(p) behind the activation (networks/equiunet2020.py:62).  The oracle is a pure function: the caller passes the
    # multiplier keep / (1 - p) of every unit ({unit prefix: tensor}), e.g. the masks the product's generator drew
    return y if drop is None else y * drop[pre]


def bcnorm(sd, pre, x, groups=8, eps=1e-5):
    """--norm bcn = BCNorm(C, 8, estimate=True), networks/factory.py:125-176,189-190.
    (i) EstBN (:150-176): (x - running_mean_c) / sqrt(running_var_c + 1e-5) * weight_c + bias_c with the running BUFFERS in
    training and eval mode alike; its training-mode update of the buffers moves them by ``estbn_moving_speed``, a buffer
    initialised to 0 that nothing in the reference ever sets (grep: only factory.py mentions it) -- so the buffers stay what
    they were initialised / loaded as, and the oracle (a pure function) refuses a non-zero speed instead of mutating sd.
    (ii) torch.batch_norm over the view [1, N * groups, -1] with training=True, no affine (:143-144): per (sample, group)
    mean and BIASED variance over C/groups x D x H x W, eps 1e-5.  (iii) per-GROUP weight / bias of shape [1, groups, 1] (:146)."""
    ms = float(sd[pre + ".bn.estbn_moving_speed"].reshape(-1)[0])
    if ms != 0.0:
        raise NotImplementedError("EstBN with estbn_moving_speed != 0 (the reference never sets it)")
    shp = (1, -1, 1, 1, 1)
    k = torch.rsqrt(sd[pre + ".bn.running_var"].detach().to(x.dtype) + 1e-5).view(shp)
    u = (x - sd[pre + ".bn.running_mean"].detach().to(x.dtype).view(shp)) * k * sd[pre + ".bn.weight"].view(shp) + sd[pre + ".bn.bias"].view(shp)
    n = x.shape[0]
    ug = u.reshape(n, groups, -1)
    mean = ug.mean(-1, keepdim=True)
    var = ug.var(-1, unbiased=False, keepdim=True)
    out = (ug - mean) * torch.rsqrt(var + eps) * sd[pre + ".weight"] + sd[pre + ".bias"]
    return out.reshape(x.shape)


def ublock(sd, pre, x, dilation=(1, 1), act="relu", norm="group", training=True, new_stats=None, drop=None):
    """UBlock, networks/equiunet2020.py:105-123."""
    x = conv_gn_act(sd, pre + ".ConvBnRelu1", x, dilation[0], act, norm, training, new_stats, drop)
    return conv_gn_act(sd, pre + ".ConvBnRelu2", x, dilation[1], act, norm, training, new_stats, drop)


def _up(x, s):
    # nn.Upsample(scale_factor=s, mode="trilinear", align_corners=True), equiunet2020.py:439
    return F.interpolate(x, scale_factor=s, mode="trilinear", align_corners=True)


def _c1(sd, pre, x):
    # conv1x1 with bias, networks/equiunet2020.py:37-41
    return F.conv3d(x, sd[pre + ".weight"], sd[pre + ".bias"])


def equiunet_forward(sd, x, act="relu", deep_supervision=True, norm="group", training=True, new_stats=None, drop=None):
    """EquiUnet.forward, networks/equiunet2020.py:467-500. Returns (logits, [4 deep heads]).  training / new_stats: only
    --norm batch distinguishes the two modes (conv_gn_act)."""
    kw = dict(act=act, norm=norm, training=training, new_stats=new_stats, drop=drop)
    down1 = ublock(sd, "encoder1", x, **kw)
    down2 = ublock(sd, "encoder2", F.max_pool3d(down1, 2, 2), **kw)
    down3 = ublock(sd, "encoder3", F.max_pool3d(down2, 2, 2), **kw)
    down4 = ublock(sd, "encoder4", F.max_pool3d(down3, 2, 2), **kw)
    bottom = ublock(sd, "bottom", down4, (2, 2), **kw)
    bottom_2 = conv_gn_act(sd, "bottom_2", torch.cat([down4, bottom], 1), 1, act, norm, training, new_stats, drop)
    up3 = ublock(sd, "decoder3", torch.cat([down3, _up(bottom_2, 2)], 1), **kw)
    up2 = ublock(sd, "decoder2", torch.cat([down2, _up(up3, 2)], 1), **kw)
    up1 = ublock(sd, "decoder1", torch.cat([down1, _up(up2, 2)], 1), **kw)
    out = _c1(sd, "outconv", up1)
    if not deep_supervision:
        return out
    deeps = [
        _up(_c1(sd, "deep_bottom.0", bottom), 8),     # equiunet2020.py:444-446
        _up(_c1(sd, "deep_bottom2.0", bottom_2), 8),  # :448-450
        _up(_c1(sd, "deep3.0", up3), 4),              # :452-454
        _up(_c1(sd, "deep2.0", up2), 2),              # :456-458
    ]
    return out, deeps


def equiunet_state_shapes(width, inplanes=4, num_classes=3, act="relu", norm="group"):
    """Ordered {key: shape} of EquiUnet(features=[width*2**i]) with GroupNorm, deep supervision
    (networks/equiunet2020.py:424-458). Checked against the reference in tests/golden.  act "prelu" adds the unit's
    nn.PReLU weight (the activation is a named entry of the reference's nn.Sequential, :51-65)."""
    f = [width * 2 ** i for i in range(4)]
    shapes = {}

    def cbr(pre, cin, cout):
        shapes[pre + ".conv.weight"] = (cout, cin, 3, 3, 3)
        if norm == "bcn":  # BCNorm: per-group weight / bias, then its EstBN (networks/factory.py:127-139,152-160), in state-dict order
            shapes[pre + ".bn.weight"] = (1, 8, 1)
            shapes[pre + ".bn.bias"] = (1, 8, 1)
            shapes[pre + ".bn.bn.weight"] = (cout,)
            shapes[pre + ".bn.bn.bias"] = (cout,)
            shapes[pre + ".bn.bn.running_mean"] = (cout,)
            shapes[pre + ".bn.bn.running_var"] = (cout,)
            shapes[pre + ".bn.bn.num_batches_tracked"] = ()
            shapes[pre + ".bn.bn.estbn_moving_speed"] = (1,)
        else:
            shapes[pre + ".bn.weight"] = (cout,)
            shapes[pre + ".bn.bias"] = (cout,)
        if norm == "batch":  # nn.BatchNorm3d's buffers, in its state-dict order
            shapes[pre + ".bn.running_mean"] = (cout,)
            shapes[pre + ".bn.running_var"] = (cout,)
            shapes[pre + ".bn.num_batches_tracked"] = ()
        if act == "prelu":
            shapes[pre + ".prelu.weight"] = (1,)

    def ub(pre, cin, mid, cout):
        cbr(pre + ".ConvBnRelu1", cin, mid)
        cbr(pre + ".ConvBnRelu2", mid, cout)

    def c1(pre, cin, cout):
        shapes[pre + ".weight"] = (cout, cin, 1, 1, 1)
        shapes[pre + ".bias"] = (cout,)

    ub("encoder1", inplanes, f[0], f[0])
    ub("encoder2", f[0], f[1], f[1])
    ub("encoder3", f[1], f[2], f[2])
    ub("encoder4", f[2], f[3], f[3])
    ub("bottom", f[3], f[3], f[3])
    cbr("bottom_2", f[3] * 2, f[2])
    ub("decoder3", f[2] * 2, f[2], f[1])
    ub("decoder2", f[1] * 2, f[1], f[0])
    ub("decoder1", f[0] * 2, f[0], f[0])
    c1("outconv", f[0], num_classes)
    c1("deep_bottom.0", f[3], num_classes)
    c1("deep_bottom2.0", f[2], num_classes)
    c1("deep3.0", f[1], num_classes)
    c1("deep2.0", f[0], num_classes)
    return shapes


# --------------------------------------------------------------------------- EquiUnetASSPEvo
def evonorm_s0(x, gamma, beta, groups=8, eps=1e-5):
    """EvoNorm3D S0, efficient=True path: networks/equiunet2021.py:95-103 with group_std :48-52.
    x*sigmoid(x) / sqrt(var_unbiased over (C/groups, D, H, W) + eps) * gamma + beta.
    Parameter ``v`` and buffer ``running_var`` are unused on this path."""
    n, c = x.shape[:2]
    xg = x.reshape(n, groups, c // groups, *x.shape[2:])
    var = torch.var(xg, dim=(2, 3, 4, 5), keepdim=True)  # unbiased (torch.var default)
    std = torch.sqrt(var + eps).expand_as(xg).reshape(x.shape)
    return x * torch.sigmoid(x) / std * gamma + beta


def residual_se(sd, pre, x):
    """MONAI ResidualSELayer(3, C, r=2, relu, sigmoid) as used at equiunet2021.py:204-205
    (MONAI 0.6.0 semantics restated)."""
    y = x.mean(dim=(2, 3, 4))
    y = F.relu(F.linear(y, sd[pre + ".fc.0.weight"], sd[pre + ".fc.0.bias"]))
    y = torch.sigmoid(F.linear(y, sd[pre + ".fc.2.weight"], sd[pre + ".fc.2.bias"]))
    return x + x * y[:, :, None, None, None]


def conv_evo_block(sd, pre, x, drop=None):
    """ConvEvoBlockCorrected, networks/equiunet2021.py:192-209 (Sequential indices 0,1,3,4,6; 2 and 5 = nn.Dropout(p)).
    drop: {key: multiplier keep / (1 - p)} with keys pre + ".2" / pre + ".5" (the oracle is pure: the caller supplies the masks)."""
    p = pre + ".conv_conv_se"
    x = F.conv3d(x, sd[p + ".0.weight"], sd[p + ".0.bias"], 1, 1)
    x = evonorm_s0(x, sd[p + ".1.gamma"], sd[p + ".1.beta"])
    if drop is not None:
        x = x * drop[pre + ".2"]
    x = F.conv3d(x, sd[p + ".3.weight"], sd[p + ".3.bias"], 1, 1)
    x = evonorm_s0(x, sd[p + ".4.gamma"], sd[p + ".4.beta"])
    if drop is not None:
        x = x * drop[pre + ".5"]
    return residual_se(sd, p + ".6", x)


def conv_evo(sd, pre, x, drop=None):
    """ConvEvo (1x1x1 conv + bias -> EvoNorm -> nn.Dropout(p)), networks/equiunet2021.py:212-222; drop: {pre: multiplier}."""
    x = F.conv3d(x, sd[pre + ".conv.weight"], sd[pre + ".conv.bias"])
    x = evonorm_s0(x, sd[pre + ".evo.gamma"], sd[pre + ".evo.beta"])
    return x if drop is None else x * drop[pre]


def max_avg_pool(x):
    """MONAI MaxAvgPool(spatial_dims=3, kernel_size=2), equiunet2021.py:261: cat([max, avg], 1)."""
    return torch.cat([F.max_pool3d(x, 2), F.avg_pool3d(x, 2)], 1)


def aspp(sd, pre, x, dilations=(1, 2, 4, 6), kernels=(1, 3, 3, 3)):
    """SimpleASPPEVO, networks/equiunet2021.py:121-189; pad = same_padding(k, d) = (k-1)/2*d."""
    outs = []
    for i, (k, d) in enumerate(zip(kernels, dilations)):
        pad = (k - 1) // 2 * d
        outs.append(F.conv3d(x, sd[f"{pre}.convs.{i}.weight"], sd[f"{pre}.convs.{i}.bias"], 1, pad, d))
    return conv_evo(sd, pre + ".conv_k1", torch.cat(outs, 1))


def assp_evo_forward(sd, x, deep_supervision=True, drop=None):
    """EquiUnetASSPEvo.forward, networks/equiunet2021.py:289-333. Returns (logits, [2 deeps]).
    drop (training with --dropout p): the multipliers keep / (1 - p) of every nn.Dropout but the ASPP's, whose p is pinned to 0
    (:178): {"<block>.2", "<block>.5", "<bridge | upconv>": tensor}."""
    down1 = conv_evo_block(sd, "encoder1", x, drop)
    down2 = conv_evo_block(sd, "encoder2", max_avg_pool(down1), drop)
    down3 = conv_evo_block(sd, "encoder3", max_avg_pool(down2), drop)
    down4 = conv_evo_block(sd, "encoder4", max_avg_pool(down3), drop)
    a = aspp(sd, "aspp", down4)
    down1b = conv_evo(sd, "bridge1", down1, drop)
    down2b = conv_evo(sd, "bridge2", down2, drop)
    down3b = conv_evo(sd, "bridge3", down3, drop)
    up3 = conv_evo_block(sd, "decoder3", torch.cat([down3b, _up(conv_evo(sd, "upconv3", a, drop), 2)], 1), drop)
    up2 = conv_evo_block(sd, "decoder2", torch.cat([down2b, _up(conv_evo(sd, "upconv2", up3, drop), 2)], 1), drop)
    up1 = conv_evo_block(sd, "decoder1", torch.cat([down1b, _up(conv_evo(sd, "upconv1", up2, drop), 2)], 1), drop)
    out = _c1(sd, "out_conv", up1)
    if not deep_supervision:
        return out
    deeps = [_up(_c1(sd, "deep3.0", up3), 4), _up(_c1(sd, "deep2.0", up2), 2)]  # :326-332
    return out, deeps


def assp_evo_state_shapes(width, inplanes=4, num_classes=3):
    """Ordered {key: shape} of EquiUnetASSPEvo (networks/equiunet2021.py:246-281), incl. the unused
    EvoNorm ``v`` parameter and ``running_var`` buffer (:76-83)."""
    f = [width * 2 ** i for i in range(4)]
    shapes = {}

    def evo(pre, c):
        for k in ("gamma", "beta", "v", "running_var"):
            shapes[f"{pre}.{k}"] = (1, c, 1, 1, 1)

    def block(pre, cin, cout):
        p = pre + ".conv_conv_se"
        shapes[p + ".0.weight"] = (cout, cin, 3, 3, 3)
        shapes[p + ".0.bias"] = (cout,)
        evo(p + ".1", cout)
        shapes[p + ".3.weight"] = (cout, cout, 3, 3, 3)
        shapes[p + ".3.bias"] = (cout,)
        evo(p + ".4", cout)
        shapes[p + ".6.fc.0.weight"] = (cout // 2, cout)
        shapes[p + ".6.fc.0.bias"] = (cout // 2,)
        shapes[p + ".6.fc.2.weight"] = (cout, cout // 2)
        shapes[p + ".6.fc.2.bias"] = (cout,)

    def cevo(pre, cin, cout):
        shapes[pre + ".conv.weight"] = (cout, cin, 1, 1, 1)
        shapes[pre + ".conv.bias"] = (cout,)
        evo(pre + ".evo", cout)

    def c1(pre, cin, cout):
        shapes[pre + ".weight"] = (cout, cin, 1, 1, 1)
        shapes[pre + ".bias"] = (cout,)

    block("encoder1", inplanes, f[0])
    block("encoder2", 2 * f[0], f[1])
    block("encoder3", 2 * f[1], f[2])
    block("encoder4", 2 * f[2], f[3])
    cevo("bridge1", f[0], f[0] // 2)
    cevo("bridge2", f[1], f[1] // 2)
    cevo("bridge3", f[2], f[2] // 2)
    for i, k in enumerate((1, 3, 3, 3)):
        shapes[f"aspp.convs.{i}.weight"] = (f[3] // 4, f[3], k, k, k)
        shapes[f"aspp.convs.{i}.bias"] = (f[3] // 4,)
    cevo("aspp.conv_k1", f[3], f[3])
    cevo("upconv3", f[3], f[3] // 4)
    block("decoder3", f[2], f[2])
    cevo("upconv2", f[2], f[2] // 4)
    block("decoder2", f[1], f[1])
    cevo("upconv1", f[1], f[1] // 4)
    block("decoder1", f[0], f[0])
    c1("out_conv", f[0], num_classes)
    c1("deep3.0", f[2], num_classes)
    c1("deep2.0", f[1], num_classes)
    return shapes


# --------------------------------------------------------------------------- loss / train step
def dice_loss(logits, target, jaccard=False, smooth=1e-5):
    """monai.losses.DiceLoss(include_background, sigmoid, squared_pred, batch=True, mean) as
    configured at src/definer.py:184-203 (formula restated in SURVEY.md a16)."""
    p = torch.sigmoid(logits.float())
    t = target.float()
    axes = (0, 2, 3, 4)
    inter = (t * p).sum(axes)
    denom = (t * t).sum(axes) + (p * p).sum(axes)
    if jaccard:
        denom = 2.0 * (denom - inter)
    return (1.0 - (2.0 * inter + smooth) / (denom + smooth)).mean()


def deep_supervision_loss(outputs, target, jaccard=False):
    """Engine._compute_loss, learning/engine.py:312-333: mean over [main] + deep heads of the
    criterion against the same full-resolution label."""
    if isinstance(outputs, (tuple, list)):
        heads = [outputs[0]] + list(outputs[1])
    else:
        heads = [outputs]
    return torch.stack([dice_loss(h, target, jaccard) for h in heads]).mean()


def hard_dice(logits, target):
    """Hard Dice of thresholded sigmoid(logits) per class (post_trans threshold 0.5,
    src/definer.py:696-697) -- the 'Dice within 1e-3' check of BASELINE.md."""
    p = (torch.sigmoid(logits.float()) > 0.5).float()
    t = target.float()
    axes = (0, 2, 3, 4)
    inter = (p * t).sum(axes)
    return (2 * inter + 1e-5) / (p.sum(axes) + t.sum(axes) + 1e-5)
