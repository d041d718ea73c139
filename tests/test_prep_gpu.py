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
