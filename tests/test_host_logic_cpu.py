"""-m "not gpu": host-side logic of the inference drivers against the golden vectors / torch semantics."""
import json
import os

import numpy as np
import pytest
import torch

from brats21_amd import inferers, tta
from brats21_amd.tta.base import SignedPerm


def _perm_torch(sp, x):
    """torch reference of a SignedPerm (what the HIP gather kernel computes)."""
    y = x.permute(0, 1, *(2 + p for p in sp.perm))
    dims = [2 + a for a in range(3) if sp.flip[a]]
    return y.flip(dims) if dims else y


def _compose():
    # get_tta_transforms, src/definer.py:653-657
    return tta.Compose([tta.OnAxes(axes=["zxy", "xyz"]), tta.HorizontalFlip(), tta.Rotate90(angles=[0, 90, 180, 270])])


def test_window_starts_match_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "inference.npz"))
    for size in ((240, 240, 155), (240, 240, 160)):
        for ov in (0.25, 0.5):
            iv = inferers.get_scan_interval(size, (128,) * 3, ov)
            st = inferers.dense_window_starts(size, (128,) * 3, iv)
            np.testing.assert_array_equal(np.array(st), g[f"starts_{size[2]}_{int(ov * 100)}"])
    assert inferers.get_scan_interval((128, 100, 64), (128, 64, 64), 0.5) == (128, 32, 64)
    assert inferers.fall_back_tuple((64, -1, None), (10, 20, 30)) == (64, 20, 30)
    m = inferers.importance_map((4, 4, 4), "gaussian")
    assert float(m.max()) == 1.0 and float(m.min()) > 0


def test_tta_compose_order_and_signed_perms_match_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "inference.npz"))
    comp = _compose()
    assert len(comp) == 16
    assert [[a, f, r] for a, f, r in comp.aug_transform_parameters] == json.loads(str(g["tta_params"]))
    from oracle import synth
    v = synth.closed_form("ttav", (1, 2, 4, 6, 6))
    for i, tr in enumerate(comp):
        a = _perm_torch(tr.aug_perm, v)
        np.testing.assert_array_equal(a.contiguous().numpy().ravel(), g["tta_aug"][i])   # same augmented tensor
        back = _perm_torch(tr.deaug_perm, a)
        np.testing.assert_array_equal(back.contiguous().numpy(), v.numpy())               # exactly invertible (F7)
        assert tr.aug_perm.then(tr.deaug_perm).is_identity


def test_signed_perm_algebra_against_torch_ops():
    x = torch.arange(2 * 3 * 4 * 5 * 6, dtype=torch.float32).view(2, 3, 4, 5, 6)
    rot = tta.Rotate90([0, 90, 180, 270])
    for ang, k in ((0, 0), (90, 1), (180, 2), (270, 3)):
        torch.testing.assert_close(_perm_torch(rot.aug(angle=ang), x), torch.rot90(x, k, (2, 3)))
        torch.testing.assert_close(_perm_torch(rot.deaug(angle=ang), torch.rot90(x, k, (2, 3))), x)
    torch.testing.assert_close(_perm_torch(tta.HorizontalFlip().aug(apply=True), x), x.flip(3))
    torch.testing.assert_close(_perm_torch(tta.VerticalFlip().aug(apply=True), x), x.flip(2))
    ax = tta.OnAxes(["zxy", "xyz", "yzx"])
    torch.testing.assert_close(_perm_torch(ax.aug(axe="xyz"), x), x.permute(0, 1, 3, 4, 2))
    torch.testing.assert_close(_perm_torch(ax.aug(axe="yzx"), x), x.permute(0, 1, 4, 2, 3))
    torch.testing.assert_close(_perm_torch(ax.deaug(axe="xyz"), x.permute(0, 1, 3, 4, 2)), x)
    a, b = ax.aug(axe="xyz"), rot.aug(angle=90)
    torch.testing.assert_close(_perm_torch(a.then(b), x), _perm_torch(b, _perm_torch(a, x).contiguous()))
    assert SignedPerm().is_identity and a.out_shape(x.shape) == (2, 3, 5, 6, 4)


def test_gaussian_importance_map_is_monai_gaussian_filter_of_a_delta():
    """MONAI 0.6.0 compute_importance_map(gaussian) (utils/inferers.py:119-121): GaussianFilter (erf-integrated taps,
    4 sigma) over a unit delta, / max, zeros -> smallest non-zero.  Known-answer check of the closed form, and the
    product's outer-product construction against the oracle's convolution construction (bit for bit)."""
    import math
    from brats21_amd.inferers import importance_map
    from oracle.refshim import _compute_importance_map
    m = importance_map((16, 16, 16), "gaussian", 0.125)
    sig = 2.0

    def tap(d):
        t = 1.0 / (math.sqrt(2.0) * sig)
        return 0.5 * (math.erf(t * (d + 0.5)) - math.erf(t * (d - 0.5)))

    assert float(m[8, 8, 8]) == 1.0
    for i in range(16):
        assert abs(float(m[i, 8, 8]) - tap(abs(i - 8)) / tap(0)) < 1e-6
    assert abs(float(m[0, 0, 0]) - (tap(8) / tap(0)) ** 3) < 1e-12
    for patch in ((16, 16, 16), (12, 20, 9), (5, 3, 7), (32, 32, 32)):
        a, b = importance_map(patch, "gaussian", 0.125), _compute_importance_map(patch, "gaussian", 0.125)
        assert torch.equal(a, b) and float(a.min()) > 0
    # sigma so small that the 4-sigma cut leaves cells untouched: they get the smallest non-zero weight, never 0
    a, b = importance_map((16, 16, 16), "gaussian", 0.03), _compute_importance_map((16, 16, 16), "gaussian", 0.03)
    assert torch.equal(a, b) and float(a.min()) > 0 and float(a[0, 0, 0]) == float(a.min())


def test_oracle_input_pipeline_known_answers():
    """CropForeground / SpatialPad / DivisiblePad / GaussianSmooth restatements (oracle/prep.py; MONAI 0.6.0 is absent, so
    these are known-answer checks, not pinned parity)."""
    import numpy as np
    from oracle import prep
    img = np.zeros((2, 6, 7, 8), np.float32)
    img[0, 1:4, 2:5, 3] = 1.0
    img[1, 5, 6, 7] = 0.5
    img[1, 0, 0, 0] = -3.0                         # not foreground (x > 0)
    assert prep.foreground_bbox(img) == ([1, 2, 3], [6, 7, 8])
    x, y = prep.crop_foreground(img, img[:1])
    assert x.shape == (2, 5, 5, 5) and y.shape == (1, 5, 5, 5)
    p = prep.spatial_pad(x, (8, 5, 6))             # 3 missing -> 1 in front, 2 behind; 1 missing -> 0 in front, 1 behind
    assert p.shape == (2, 8, 5, 6) and np.array_equal(p[:, 1:6, :, 0:5], x) and p[:, 0].sum() == 0 and p[:, 6:].sum() == 0
    assert prep.divisible_pad(x, 8).shape == (2, 8, 8, 8) and prep.divisible_pad(p[:, :8, :, :], 4).shape == (2, 8, 8, 8)
    # GaussianFilter: the response to a unit delta is the outer product of the erf-integrated tap vectors; sigma 0.25 ->
    # 3 taps, sigma 1.5 -> 13 taps (tail = int(max(4 sigma, 0.5) + 0.5)); mass is preserved away from the faces
    d = np.zeros((1, 15, 15, 15), np.float32)
    d[0, 7, 7, 7] = 1
    g = prep.gaussian_smooth(d, (0.25, 1.5, 1.0))[0]
    nz = np.nonzero(g > 0)
    assert [int(i.max() - i.min()) + 1 for i in nz] == [3, 13, 9]
    assert abs(float(g.sum()) - 1.0) < 2e-4 and np.isclose(g[7, 7, 7], g.max())
    assert np.allclose(g, g[::-1, ::-1, ::-1])


def test_precision_is_validated_at_construction_and_on_assignment(monkeypatch):
    """a mistyped model.precision / BRATS_PRECISION must raise instead of silently running another mode."""
    import argparse
    import contextlib
    import io
    import warnings
    from brats21_amd import get_model

    def make(model, width):
        with contextlib.redirect_stdout(io.StringIO()), warnings.catch_warnings():
            warnings.simplefilter("ignore")
            return get_model(argparse.Namespace(model=model, width=width, norm="group", act="relu", num_classes=3, dropout=0))

    for model, width in (("equiunet", 8), ("equiunet_assp_evo", 16)):
        m = make(model, width)
        assert m.precision == "auto"
        for ok in ("bf16", "fp16", "fp32", "x3", "bf16x3", "auto"):
            m.precision = ok
            assert m.precision == ok
        for bad in ("X3", "x3 ", "float32", None):
            with pytest.raises(ValueError):
                m.precision = bad
        assert m.precision == "auto"
        monkeypatch.setenv("BRATS_PRECISION", "x3")
        assert make(model, width).precision == "x3"
        monkeypatch.setenv("BRATS_PRECISION", "x3 ")
        with pytest.raises(ValueError):
            make(model, width)
        monkeypatch.delenv("BRATS_PRECISION")
