"""Input pipeline on the GPU: what the reference's CPU transform chain does to every
training patch (src/definer.py:449-467), as a handful of fused HIP kernels on batched NCDHW f32 tensors
(csrc/prep.hip).  Random draws stay on the host (numpy RandomState, like MONAI's Randomizable); every function is
deterministic given its arguments, which is how the parity tests drive it.

Names follow the reference / MONAI transforms they replace: NormalizeIntensity (utils/transforms.py:328),
ConvertToMultiChannelBasedOnBratsClasses (:145 / MONAI), CropForeground (bounding box by integer atomics) + SpatialPad
/ DivisiblePad, RandSpatialCrop + RandRotate90 + RandFlip + RandShiftIntensity (one gather), RandAdjustContrast +
RandGaussianNoise (one pass), RandGaussianSmooth (three separable passes).
"""
import numpy as np
import torch

from . import _lib
from .tta.base import SignedPerm


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _need(t, what):
    if not t.is_cuda:
        raise _lib.BratsHipError(f"brats21_amd.transforms.{what} runs on the GPU only (no CPU fallback)")
    return t.contiguous().float()


def normalize_intensity(img, nonzero=True, channel_wise=True, remove_outliers=False, outliers_value=3.0):
    """NormalizeIntensity(nonzero, channel_wise[, remove_outliers]) of utils/transforms.py:328-406 on [N, C, D, H, W]
    (or [C, D, H, W]): per (sample, channel) z-score over the non-zero voxels, zeros stay zero."""
    x = _need(img, "normalize_intensity")
    if not channel_wise:
        raise NotImplementedError("the reference pipeline normalises channel-wise (src/definer.py:466)")
    planes = int(np.prod(x.shape[:-3]))
    vox = x[(0,) * (x.dim() - 3)].numel()
    out = torch.empty_like(x)
    stats = torch.empty((planes, 3), dtype=torch.float64, device=x.device)
    _lib.check(_lib.lib().brats_zscore_normalize(x.data_ptr(), out.data_ptr(), stats.data_ptr(), planes, vox, int(bool(nonzero)),
                                                 float(outliers_value) if remove_outliers else 0.0, _stream()), "zscore_normalize")
    return out


def convert_to_multichannel(label, order="monai"):
    """BraTS label map [N, D, H, W] (or [N, 1, D, H, W]) with values {0, 1, 2, 4} -> [N, 3, D, H, W] f32.
    order "monai" = (TC, WT, ET) as the training pipeline (src/definer.py:451), "utils" = (WT, TC, ET)
    (utils/transforms.py:155-166)."""
    l = _need(label, "convert_to_multichannel")
    if l.dim() == 5:
        if l.shape[1] != 1:
            raise ValueError("label must have one channel")
        l = l[:, 0].contiguous()
    if l.dim() != 4:
        raise ValueError("expected [N, D, H, W] labels")
    n = l.shape[0]
    vox = l[0].numel()
    out = torch.empty((n, 3) + tuple(l.shape[1:]), dtype=torch.float32, device=l.device)
    _lib.check(_lib.lib().brats_label_to_channels(l.data_ptr(), out.data_ptr(), n, vox, {"monai": 0, "utils": 1}[order], _stream()),
               "label_to_channels")
    return out


def rot90_perm(k, spatial_axes=(0, 2)):
    """np.rot90(img, k, axes) over two spatial axes as a signed permutation (RandRotate90d, src/definer.py:459)."""
    a, b = spatial_axes
    p = SignedPerm()
    for _ in range(k % 4):
        perm, flip = [0, 1, 2], [False, False, False]
        perm[a], perm[b] = b, a      # rot90 = flip(swapaxes): out[.., i_a, .., i_b, ..] = in[.., i_b, .., n_b-1-i_a, ..]
        flip[a] = True
        p = p.then(SignedPerm(perm, flip))
    return p


def crop_perm(x, crop_start, crop_size, perm=None, scale=None, shift=None):
    """out = scale * perm(x[..., crop box]) + shift in one gather.  x: [N, C, D, H, W]; perm: SignedPerm; scale /
    shift: None, a float, or a tensor broadcastable to [N, C]."""
    x = _need(x, "crop_perm")
    n, c = x.shape[:2]
    perm = perm or SignedPerm()
    e = [int(v) for v in crop_size]
    out = torch.empty((n, c) + tuple(e[perm.perm[a]] for a in range(3)), dtype=torch.float32, device=x.device)

    def plane_param(v):
        if v is None:
            return None
        return torch.as_tensor(v, dtype=torch.float32, device=x.device).expand(n, c).contiguous()
    sc, sh = plane_param(scale), plane_param(shift)
    _lib.check(_lib.lib().brats_crop_perm(x.data_ptr(), out.data_ptr(), n * c, *x.shape[2:], *[int(v) for v in crop_start], *e,
                                          *perm.perm, *[int(f) for f in perm.flip], sc.data_ptr() if sc is not None else None,
                                          sh.data_ptr() if sh is not None else None, _stream()), "crop_perm")
    return out


def gamma_noise(x, gamma=None, noise=None):
    """MONAI AdjustContrast(gamma) over the whole tensor followed by + noise (RandAdjustContrastd +
    RandGaussianNoised, src/definer.py:462-463); either part may be None."""
    x = _need(x, "gamma_noise")
    out = torch.empty_like(x)
    mn = rg = 0.0
    if gamma is not None:
        lo, hi = torch.aminmax(x)
        mn, rg = float(lo), float(hi - lo)
    nz = _need(noise, "gamma_noise") if noise is not None else None
    _lib.check(_lib.lib().brats_gamma_noise(x.data_ptr(), out.data_ptr(), x.numel(), mn, rg, float(gamma) if gamma is not None else 0.0,
                                            nz.data_ptr() if nz is not None else None, _stream()), "gamma_noise")
    return out


def gaussian_smooth(x, sigma):
    """MONAI 0.6 GaussianSmooth(sigma=(s0, s1, s2), approx="erf") on [N, C, D, H, W] f32 (RandGaussianSmoothd,
    src/definer.py:464): GaussianFilter's zero-padded separable correlations, axis 0 first, the same kernel for every
    channel; taps = the Gaussian integrated over unit cells, cut at round(4 sigma) either side."""
    from .inferers import _gaussian_taps
    x = _need(x, "gaussian_smooth")
    if x.dim() != 5 or len(sigma) != 3:
        raise ValueError("gaussian_smooth: expected [N, C, D, H, W] and three sigmas")
    n, c, d, h, w = x.shape
    cur = x
    for ax, sg in enumerate(sigma):
        taps = _gaussian_taps(sg)
        if taps.numel() == 1 and float(taps[0]) == 1:
            continue
        taps = taps.to(x.device)
        out = torch.empty_like(cur)
        outer, length, inner = (n * c, d, h * w) if ax == 0 else ((n * c * d, h, w) if ax == 1 else (n * c * d * h, w, 1))
        _lib.check(_lib.lib().brats_blur_axis(cur.data_ptr(), out.data_ptr(), outer, length, inner, taps.data_ptr(), taps.numel(),
                                              _stream()), "blur_axis")
        cur = out
    return cur if cur is not x else x.clone()


def foreground_bbox(img):
    """Per-sample box of MONAI CropForegroundd(source_key="img") (src/definer.py:452: voxels with any channel > 0, margin
    0) of [N, C, D, H, W]: int32 device tensor [N, 6] = (z0, y0, x0, z1, y1, x1), ends exclusive."""
    x = _need(img, "foreground_bbox")
    if x.dim() != 5:
        raise ValueError("foreground_bbox: expected [N, C, D, H, W]")
    box = torch.empty((x.shape[0], 6), dtype=torch.int32, device=x.device)
    _lib.check(_lib.lib().brats_foreground_bbox(x.data_ptr(), *x.shape, box.data_ptr(), _stream()), "foreground_bbox")
    return box


def crop_foreground(img, seg=None):
    """CropForegroundd(keys=["img", "seg"], source_key="img") for ONE volume ([1, C, D, H, W]; boxes differ between
    volumes).  The box comes back to the host (it decides tensor shapes); an image without foreground raises ValueError
    like MONAI 0.6.0 (np.min of an empty index list)."""
    if img.shape[0] != 1:
        raise ValueError("crop_foreground: one volume at a time (the boxes of different volumes differ)")
    z0, y0, x0, z1, y1, x1 = (int(v) for v in foreground_bbox(img)[0].cpu())
    if z0 > z1:
        raise ValueError("crop_foreground: the image has no voxel > 0")
    crop = img[:, :, z0:z1, y0:y1, x0:x1].contiguous()
    return crop if seg is None else (crop, seg[:, :, z0:z1, y0:y1, x0:x1].contiguous())


def spatial_pad(x, size):
    """SpatialPadd(spatial_size, method="symmetric") (src/definer.py:453): zero-pad [N, C, D, H, W] up to `size` per axis
    (floor half in front); never crops."""
    pads = []
    for have, want in zip(reversed(x.shape[2:]), reversed(tuple(size))):
        w = max(int(want) - have, 0)
        pads += [w // 2, w - w // 2]
    return torch.nn.functional.pad(x, pads) if any(pads) else x


def divisible_pad(x, k=8):
    """DivisiblePadd(k=8) (src/definer.py:465)."""
    return spatial_pad(x, [-(-s // k) * k for s in x.shape[2:]])


class TrainAugment:
    """The random part of the reference's training chain (src/definer.py:458-466) on GPU-resident, already padded
    volumes: RandSpatialCrop(roi) -> RandRotate90(p=0.7, axes (0, 2)) -> RandFlip(p=0.7, all axes) ->
    RandShiftIntensity(p=0.7, 0.1) -> RandAdjustContrast(p=0.2, gamma 0.5..4.5) -> RandGaussianNoise(p=0.5, std 0.1)
    -> RandGaussianSmooth(p=0.2, sigma 0.25..1.5 per axis) -> NormalizeIntensity(nonzero, channel-wise).  (The patch size
    is a multiple of 8, so DivisiblePadd(8) is the identity here; CropForeground / SpatialPad run once per volume:
    crop_foreground, spatial_pad.)"""

    def __init__(self, roi_size, seed=None, remove_outliers=False):
        self.roi, self.R, self.remove_outliers = tuple(roi_size), np.random.RandomState(seed), remove_outliers

    def draw(self, spatial):
        start = tuple(int(self.R.randint(0, s - r + 1)) for s, r in zip(spatial, self.roi))
        k = int(self.R.randint(3)) + 1 if self.R.rand() < 0.7 else 0
        if k and self.roi[0] != self.roi[2]:
            k = 2 * (k // 2)  # odd quarter turns would change the patch shape
        return {
            "start": start, "k_rot": k, "flip": bool(self.R.rand() < 0.7),
            "offset": float(self.R.uniform(-0.1, 0.1)) if self.R.rand() < 0.7 else 0.0,
            "gamma": float(self.R.uniform(0.5, 4.5)) if self.R.rand() < 0.2 else None,
            "noise_std": float(self.R.uniform(0, 0.1)) if self.R.rand() < 0.5 else None,
            "smooth": tuple(float(self.R.uniform(0.25, 1.5)) for _ in range(3)) if self.R.rand() < 0.2 else None,
        }

    def __call__(self, img, seg, params=None):
        """img [N, 4, D, H, W], seg [N, 3, D, H, W] (cuda) -> (patch, label) of roi_size."""
        p = params or self.draw(img.shape[2:])
        perm = rot90_perm(p["k_rot"])
        if p["flip"]:
            perm = perm.then(SignedPerm((0, 1, 2), (True, True, True)))
        x = crop_perm(img, p["start"], self.roi, perm, shift=p["offset"] if p["offset"] else None)
        y = crop_perm(seg, p["start"], self.roi, perm)
        if p["gamma"] is not None or p["noise_std"] is not None:
            noise = torch.randn_like(x) * p["noise_std"] if p["noise_std"] is not None else None
            x = gamma_noise(x, p["gamma"], noise)
        if p.get("smooth") is not None:
            x = gaussian_smooth(x, p["smooth"])
        return normalize_intensity(x, nonzero=True, channel_wise=True, remove_outliers=self.remove_outliers), y
