"""TEST INFRASTRUCTURE ONLY -- import shim that lets the *reference* hot-path modules be imported
in this container so that golden vectors can be generated from them (tests/golden/make_golden.py).

Why it exists: every hot-path module of the reference imports
``monai==0.6.0`` (requirements.txt:17), which is not installed here and cannot be installed
(no network); ``utils/misc.py:6`` also needs ``collections.Sequence`` (Python <= 3.9).  This file
restates, from MONAI 0.6.0's documented behaviour, only the handful of symbols the reference path
touches.  **Parity is therefore unpinned at the MONAI boundary** (MaxAvgPool, ResidualSELayer,
dense_patch_slices, compute_importance_map, DiceLoss): nothing under /root/reference tests them.
``EquiUnet`` (networks/equiunet2020.py) touches MONAI only for the activation *lookup*
(networks/factory.py:195-200), so its golden vectors are pinned by reference source + torch alone.

Nothing here is product code and nothing here travels as reference source: the GPU box never has
/root/reference, so this module is only ever imported by the golden generator and by the optional
``-m "not gpu"`` cross-check that runs when /root/reference is present.
"""
import collections
import collections.abc
import enum
import math
import sys
import types

import torch
import torch.nn as nn
import torch.nn.functional as F

REFERENCE_ROOT = "/root/reference"


# ----------------------------------------------------------------------------- monai.networks
class _ActFactory:
    """``Act[name](**kw)``: case-insensitive lookup (Appendix A of SURVEY.md)."""

    _table = {
        "RELU": nn.ReLU,
        "LEAKYRELU": nn.LeakyReLU,
        "ELU": nn.ELU,
        "PRELU": nn.PReLU,
        "SIGMOID": nn.Sigmoid,
    }

    def __getitem__(self, name):
        return self._table[str(name).upper()]


class _ConvFactory:
    CONV = "conv"
    CONVTRANS = "convtrans"

    def __getitem__(self, key):
        kind, dims = key
        if kind != self.CONV or dims != 3:
            raise KeyError(key)
        return nn.Conv3d


def _same_padding(kernel_size, dilation=1):
    # (k - 1) / 2 * d as int
    return int((kernel_size - 1) / 2 * dilation)


class _MaxAvgPool(nn.Module):
    """cat([max_pool(x), avg_pool(x)], dim=1); stride = kernel, no padding."""

    def __init__(self, spatial_dims, kernel_size, stride=None, padding=0, ceil_mode=False):
        super().__init__()
        assert spatial_dims == 3
        self.max_pool = nn.MaxPool3d(kernel_size, stride, padding, ceil_mode=ceil_mode)
        self.avg_pool = nn.AvgPool3d(kernel_size, stride, padding, ceil_mode=ceil_mode)

    def forward(self, x):
        return torch.cat([self.max_pool(x), self.avg_pool(x)], dim=1)


class _ResidualSELayer(nn.Module):
    """x + x * sigmoid(W2 relu(W1 gap(x) + b1) + b2); keys fc.0.*, fc.2.*"""

    def __init__(self, spatial_dims, in_channels, r=2, acti_type_1="relu", acti_type_2="sigmoid"):
        super().__init__()
        assert spatial_dims == 3
        self.avg_pool = nn.AdaptiveAvgPool3d(1)
        hidden = int(in_channels // r)
        self.fc = nn.Sequential(
            nn.Linear(in_channels, hidden, bias=True),
            nn.ReLU(inplace=True),
            nn.Linear(hidden, in_channels, bias=True),
            nn.Sigmoid(),
        )

    def forward(self, x):
        b, c = x.shape[:2]
        y = self.avg_pool(x).view(b, c)
        y = self.fc(y).view(b, c, 1, 1, 1)
        return x + x * y


# ----------------------------------------------------------------------------- monai.utils / data
class _BlendMode(enum.Enum):
    CONSTANT = "constant"
    GAUSSIAN = "gaussian"


class _PytorchPadMode(enum.Enum):
    CONSTANT = "constant"
    REFLECT = "reflect"
    REPLICATE = "replicate"
    CIRCULAR = "circular"


def _fall_back_tuple(user, default):
    if isinstance(user, int):
        user = (user,) * len(default)
    return tuple(d if (u is None or u <= 0) else u for u, d in zip(user, default))


def _get_valid_patch_size(image_size, patch_size):
    if isinstance(patch_size, int):
        patch_size = (patch_size,) * len(image_size)
    return tuple(min(p if p else i, i) for p, i in zip(patch_size, image_size))


def _dense_patch_slices(image_size, patch_size, scan_interval):
    import itertools

    starts = []
    for L, p, iv in zip(image_size, patch_size, scan_interval):
        if iv == 0:
            starts.append([0])
            continue
        num = int(math.ceil(float(L) / iv))
        scan = next(d for d in range(num) if d * iv + p >= L)
        dim_starts = []
        for i in range(scan + 1):
            s = i * iv
            s -= max(s + p - L, 0)
            dim_starts.append(s)
        starts.append(dim_starts)
    return [tuple(slice(s, s + p) for s, p in zip(st, patch_size)) for st in itertools.product(*starts)]


def _gaussian_1d(sigma, truncated=4.0):
    """monai.networks.layers.convutils.gaussian_1d(sigma, truncated=4.0, approx="erf", normalize=False) of MONAI 0.6.0:
    the Gaussian integrated over each unit cell, cut at round(truncated * sigma) taps either side (float32)."""
    sigma = torch.as_tensor(sigma, dtype=torch.float)
    tail = int(max(float(sigma) * truncated, 0.5) + 0.5)
    x = torch.arange(-tail, tail + 1, dtype=torch.float)
    t = 0.70710678 / torch.abs(sigma)
    out = 0.5 * ((t * (x + 0.5)).erf() - (t * (x - 0.5)).erf())
    return out.clamp(min=0)


def _gaussian_filter(x, sigmas, truncated=4.0):
    """monai.networks.layers.GaussianFilter(spatial_dims, sigma, truncated=4.0, approx="erf").forward of MONAI 0.6.0 on
    x [B, C, *spatial]: separable_filtering with zero padding -- one grouped convolution per spatial axis (every channel
    filtered by the same kernel), axis 0 first.  Restated (MONAI is absent here): parity at this boundary stays unpinned."""
    nsp = x.dim() - 2
    conv = [F.conv1d, F.conv2d, F.conv3d][nsp - 1]
    c = x.shape[1]
    for ax, s in enumerate(sigmas):
        k = _gaussian_1d(s, truncated)
        if k.numel() == 1 and float(k[0]) == 1:  # (a unit kernel is skipped)
            continue
        shape = [1, 1] + [1] * nsp
        shape[ax + 2] = -1
        pad = [0] * nsp
        pad[ax] = (k.numel() - 1) // 2
        x = conv(x, k.reshape(shape).repeat([c, 1] + [1] * nsp), padding=pad, groups=c)
    return x


def _compute_importance_map(patch_size, mode="constant", sigma_scale=0.125, device=None):
    """monai.data.utils.compute_importance_map of MONAI 0.6.0 (called at utils/inferers.py:119-121).  gaussian: a unit
    delta at patch // 2 filtered by GaussianFilter(sigmas = sigma_scale * patch) -- separable zero-padded convolutions
    with the erf-integrated, 4-sigma-truncated kernel above, axis 0 first -- divided by its maximum, zeros replaced by
    the smallest non-zero weight.  Restated (MONAI is absent here): parity at this boundary stays unpinned."""
    mode = _BlendMode(mode)
    if mode == _BlendMode.CONSTANT:
        return torch.ones(patch_size, device=device, dtype=torch.float)
    patch_size = tuple(int(p) for p in patch_size)
    if isinstance(sigma_scale, (int, float)):
        sigma_scale = (sigma_scale,) * len(patch_size)
    m = torch.zeros(patch_size, dtype=torch.float)
    m[tuple(p // 2 for p in patch_size)] = 1
    m = m[None, None]
    conv = [F.conv1d, F.conv2d, F.conv3d][len(patch_size) - 1]
    for ax, (p, s) in enumerate(zip(patch_size, sigma_scale)):
        k = _gaussian_1d(p * s)
        shape = [1, 1] + [1] * len(patch_size)
        shape[ax + 2] = -1
        pad = [0] * len(patch_size)
        pad[ax] = (k.numel() - 1) // 2
        m = conv(m, k.reshape(shape), padding=pad)
    m = m[0, 0]
    m = (m / m.max()).float()
    m[m == 0] = m[m != 0].min().item()
    return m.to(device)


class _DiceLoss(nn.Module):
    """monai.losses.DiceLoss restricted to the options src/definer.py:184-203 uses."""

    def __init__(self, include_background=True, sigmoid=False, softmax=False, squared_pred=False,
                 jaccard=False, batch=False, smooth_nr=1e-5, smooth_dr=1e-5, reduction="mean"):
        super().__init__()
        assert include_background and not softmax and reduction == "mean"
        self.sigmoid, self.squared_pred, self.jaccard, self.batch = sigmoid, squared_pred, jaccard, batch
        self.smooth_nr, self.smooth_dr = float(smooth_nr), float(smooth_dr)

    def forward(self, input, target):
        if self.sigmoid:
            input = torch.sigmoid(input)
        reduce_axis = list(range(2, input.dim()))
        if self.batch:
            reduce_axis = [0] + reduce_axis
        inter = torch.sum(target * input, dim=reduce_axis)
        if self.squared_pred:
            target = torch.pow(target, 2)
            input = torch.pow(input, 2)
        denom = torch.sum(target, dim=reduce_axis) + torch.sum(input, dim=reduce_axis)
        if self.jaccard:
            denom = 2.0 * (denom - inter)
        f = 1.0 - (2.0 * inter + self.smooth_nr) / (denom + self.smooth_dr)
        return torch.mean(f)


class _Randomizable:
    pass


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    m.__path__ = []  # behave as a package
    sys.modules[name] = m
    return m


def install():
    """Put the stub ``monai`` in sys.modules and /root/reference on sys.path (idempotent)."""
    if not hasattr(collections, "Sequence"):
        collections.Sequence = collections.abc.Sequence
    if not hasattr(collections, "Iterable"):
        collections.Iterable = collections.abc.Iterable
    if "monai" not in sys.modules:
        act, conv = _ActFactory(), _ConvFactory()
        _mod("monai")
        _mod("monai.networks")
        _mod("monai.networks.layers", same_padding=_same_padding)
        _mod("monai.networks.layers.factories", Act=act, Conv=conv)
        _mod("monai.networks.blocks", MaxAvgPool=_MaxAvgPool, ResidualSELayer=_ResidualSELayer)
        _mod("monai.transforms")
        _mod("monai.transforms.compose", Randomizable=_Randomizable)
        _mod("monai.data")
        _mod("monai.data.utils", compute_importance_map=_compute_importance_map,
             dense_patch_slices=_dense_patch_slices, get_valid_patch_size=_get_valid_patch_size)
        _mod("monai.utils", BlendMode=_BlendMode, PytorchPadMode=_PytorchPadMode,
             fall_back_tuple=_fall_back_tuple)
        _mod("monai.losses", DiceLoss=_DiceLoss)
        # utils/transforms.py imports these at module level; the functions the golden generator calls
        # (shape_to_divisible / shape_to_original / remove_background_voxels / the two label
        # converters) use none of them beyond `Transform` as an empty base class.
        sys.modules["monai.transforms"].Transform = type("Transform", (), {})
        sys.modules["monai.transforms"].MapTransform = type("MapTransform", (), {})
        sys.modules["monai.transforms"].BorderPad = type("BorderPad", (), {})
        _mod("monai.config", DtypeLike=object, KeysCollection=object)
        for absent in ("SimpleITK", "skimage", "skimage.morphology"):
            if absent not in sys.modules:
                _mod(absent)
        sys.modules["skimage"].morphology = sys.modules["skimage.morphology"]
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)


def reference_available():
    import os

    return os.path.isdir(REFERENCE_ROOT + "/networks")
