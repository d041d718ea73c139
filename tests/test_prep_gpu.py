"""-m gpu: the GPU input pipeline (brats21_amd/transforms.py, csrc/prep.hip) against the reference's golden vectors
(tests/golden/prep.npz) and the CPU oracle (oracle/prep.py)."""
import os

import numpy as np
import pytest
import torch

from oracle import prep as oprep

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_normalize_and_labels_match_reference_golden(golden_dir):
    from brats21_amd import transforms as T
    g = np.load(os.path.join(golden_dir, "prep.npz"))
    img = torch.from_numpy(g["img"]).to(DEV)
    # f32 tolerance: the reference's mean / std are numpy float32 reductions, ours f64 -> 1e-5 abs on z-scores of O(1)
    np.testing.assert_allclose(T.normalize_intensity(img[None])[0].cpu().numpy(), g["norm_nz"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(T.normalize_intensity(img[None], remove_outliers=True, outliers_value=1.5)[0].cpu().numpy(),
                               g["norm_nz_clip"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(T.normalize_intensity(img[None], nonzero=False)[0].cpu().numpy(), g["norm_all"], rtol=0, atol=2e-5)
    lab = torch.from_numpy(g["label"]).to(DEV)[None]
    np.testing.assert_array_equal(T.convert_to_multichannel(lab, "utils")[0].cpu().numpy(), g["label_utils"])
    np.testing.assert_array_equal(T.convert_to_multichannel(lab, "monai")[0].cpu().numpy(), oprep.convert_to_multichannel(g["label"], "monai"))
    with pytest.raises(Exception, match="GPU only"):
        T.normalize_intensity(torch.zeros(1, 4, 4, 4, 4))


def test_normalize_full_volume_vs_oracle():
    """4 x 240 x 240 x 155 with a zero background: per-channel non-zero z-score at the BraTS volume size."""
    from brats21_amd import transforms as T
    rng = np.random.RandomState(3)
    img = (rng.rand(4, 155, 240, 240).astype(np.float32) * 800 + 5) * (rng.rand(1, 155, 240, 240) > 0.4)
    ref = oprep.normalize_intensity(img, remove_outliers=True)
    out = T.normalize_intensity(torch.from_numpy(img).to(DEV)[None], remove_outliers=True)[0].cpu().numpy()
    np.testing.assert_allclose(out, ref, rtol=0, atol=5e-4)   # f32 mean of 5e6 values vs f64
    nz = img != 0
    assert abs(float(out[nz].mean())) < 1e-3 and float(np.abs(out[~nz]).max()) == 0.0


@pytest.mark.parametrize("k_rot,do_flip", [(0, False), (1, False), (2, True), (3, True), (0, True)])
def test_augment_chain_vs_oracle(k_rot, do_flip):
    from brats21_amd import transforms as T
    rng = np.random.RandomState(5 + k_rot)
    img = rng.randn(2, 4, 20, 18, 22).astype(np.float32)
    seg = (rng.rand(2, 3, 20, 18, 22) > 0.5).astype(np.float32)
    noise = (rng.randn(2, 4, 12, 12, 12) * 0.05).astype(np.float32)
    params = {"start": (3, 2, 5), "k_rot": k_rot, "flip": do_flip, "offset": 0.07, "gamma": None, "noise_std": None}
    aug = T.TrainAugment((12, 12, 12), seed=0)
    x, y = aug(torch.from_numpy(img).to(DEV), torch.from_numpy(seg).to(DEV), params)
    for n in range(2):
        xi, yi = oprep.augment(img[n], seg[n], (3, 2, 5), (12, 12, 12), k_rot, do_flip, 0.07)
        np.testing.assert_allclose(x[n].cpu().numpy(), oprep.normalize_intensity(xi), rtol=0, atol=2e-5)
        np.testing.assert_array_equal(y[n].cpu().numpy(), yi)
    # contrast + noise pass on its own (whole-tensor min / range like MONAI's AdjustContrast on one image)
    xs = torch.from_numpy(img[:1, :, :12, :12, :12].copy()).to(DEV)
    gn = T.gamma_noise(xs, 1.7, torch.from_numpy(noise[:1]).to(DEV))[0].cpu().numpy()
    ref = oprep.adjust_contrast(img[0, :, :12, :12, :12], 1.7) + noise[0]
    np.testing.assert_allclose(gn, ref, rtol=1e-5, atol=2e-5)
    # the random driver draws valid parameters and keeps the roi shape
    xr, yr = aug(torch.from_numpy(img).to(DEV), torch.from_numpy(seg).to(DEV))
    assert tuple(xr.shape) == (2, 4, 12, 12, 12) and tuple(yr.shape) == (2, 3, 12, 12, 12) and bool(torch.isfinite(xr).all())


@pytest.mark.parametrize("sigma", [(0.25, 0.9, 1.5), (1.5, 1.5, 0.25), (0.6, 0.3, 1.1)])
def test_gaussian_smooth_vs_oracle(sigma):
    """RandGaussianSmoothd's filter (MONAI 0.6 GaussianFilter restated in oracle/refshim.py) on a ragged volume: zero
    padding at the faces, a different kernel length per axis (sigma 0.25 -> 3 taps, 1.5 -> 13 taps)."""
    from brats21_amd import transforms as T
    rng = np.random.RandomState(11)
    img = rng.randn(2, 4, 19, 14, 23).astype(np.float32)
    out = T.gaussian_smooth(torch.from_numpy(img).to(DEV), sigma).cpu().numpy()
    for n in range(2):
        np.testing.assert_allclose(out[n], oprep.gaussian_smooth(img[n], sigma), rtol=0, atol=2e-6)
    # a unit delta comes back as the outer product of the three tap vectors
    from brats21_amd.inferers import _gaussian_taps
    d = torch.zeros(1, 1, 17, 17, 17, device=DEV)
    d[0, 0, 8, 8, 8] = 1
    k = [_gaussian_taps(s) for s in sigma]
    got = T.gaussian_smooth(d, sigma)[0, 0].cpu()
    r = [(t.numel() - 1) // 2 for t in k]
    want = torch.einsum("i,j,k->ijk", *k)
    np.testing.assert_allclose(got[8 - r[0]:9 + r[0], 8 - r[1]:9 + r[1], 8 - r[2]:9 + r[2]].numpy(), want.numpy(), rtol=0, atol=1e-7)
    assert abs(float(got.sum()) - float(want.sum())) < 1e-6


def test_crop_foreground_and_pads_vs_oracle():
    from brats21_amd import transforms as T
    rng = np.random.RandomState(12)
    img = np.zeros((3, 4, 21, 26, 18), np.float32)
    seg = (rng.rand(3, 3, 21, 26, 18) > 0.5).astype(np.float32)
    img[0, 1, 3:17, 5:20, 2:11] = rng.rand(14, 15, 9) + 0.1      # one channel carries the foreground
    img[0, 2, 16:19, 4:6, 12] = 0.5                               # another channel widens the box
    img[0, 0] = -np.abs(rng.randn(21, 26, 18))                    # negative values are background (select_fn x > 0)
    img[1, :, 0, 0, 0] = 1.0                                      # a single corner voxel
    img[1, 3, 20, 25, 17] = 2.0                                   # ... and the opposite corner
    box = T.foreground_bbox(torch.from_numpy(img).to(DEV)).cpu().numpy()
    for n in range(2):
        start, end = oprep.foreground_bbox(img[n])
        assert list(box[n, :3]) == start and list(box[n, 3:]) == end, (n, box[n], start, end)
    assert box[2, 0] > box[2, 3]  # no foreground in sample 2
    x, y = T.crop_foreground(torch.from_numpy(img[:1]).to(DEV), torch.from_numpy(seg[:1]).to(DEV))
    xr, yr = oprep.crop_foreground(img[0], seg[0])
    np.testing.assert_array_equal(x[0].cpu().numpy(), xr)
    np.testing.assert_array_equal(y[0].cpu().numpy(), yr)
    with pytest.raises(ValueError):
        T.crop_foreground(torch.from_numpy(img[2:3]).to(DEV))
    with pytest.raises(ValueError):
        oprep.foreground_bbox(img[2])
    # SpatialPadd(patch) + DivisiblePadd(8): symmetric zero padding, never cropping
    xp = T.spatial_pad(x, (32, 16, 24))
    np.testing.assert_array_equal(xp[0].cpu().numpy(), oprep.spatial_pad(xr, (32, 16, 24)))
    np.testing.assert_array_equal(T.divisible_pad(x)[0].cpu().numpy(), oprep.divisible_pad(xr, 8))


def test_augment_with_smoothing_vs_oracle():
    from brats21_amd import transforms as T
    rng = np.random.RandomState(13)
    img = rng.randn(1, 4, 20, 18, 22).astype(np.float32)
    seg = (rng.rand(1, 3, 20, 18, 22) > 0.5).astype(np.float32)
    params = {"start": (1, 2, 3), "k_rot": 2, "flip": True, "offset": -0.03, "gamma": None, "noise_std": None, "smooth": (0.4, 1.2, 0.8)}
    x, y = T.TrainAugment((16, 16, 16), seed=0)(torch.from_numpy(img).to(DEV), torch.from_numpy(seg).to(DEV), params)
    xi, yi = oprep.augment(img[0], seg[0], (1, 2, 3), (16, 16, 16), 2, True, -0.03, smooth=(0.4, 1.2, 0.8))
    np.testing.assert_allclose(x[0].cpu().numpy(), oprep.normalize_intensity(xi), rtol=0, atol=3e-5)
    np.testing.assert_array_equal(y[0].cpu().numpy(), yi)
    draws = [T.TrainAugment((16, 16, 16), seed=s).draw((20, 18, 22)) for s in range(40)]
    sm = [d["smooth"] for d in draws if d["smooth"] is not None]
    assert 0 < len(sm) < 20 and all(0.25 <= v <= 1.5 for t in sm for v in t)
