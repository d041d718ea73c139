"""-m gpu: BASELINE.json configs[3] at size -- sliding-window inference (utils/inferers.py:26-162) of ONE synthetic
4x240x240x155 volume, padded to a multiple of 8 like Engine.evaluate (learning/engine.py:217 -> 160), EquiUnet width 48,
128^3 window, overlap 0.5 (18 windows), with the bench's own volume and the model's own kaiming initialisation:

  (i)   f32 mode AND the split-precision mode ("x3"), identity TTA: the stitched LOGITS of the HIP path (window gather ->
        exact-f32 MFMA / three-fp16-product network -> weighted accumulate -> divide -> crop) against oracle/inference.py +
        oracle/unet.py on the GPU box's host cores, over the whole volume, within the north-star bar of 1e-3 abs;
  (ii)  the benchmarked configuration (bf16, 8-flip TTA, the Evaluator's full chain down to the thresholded, background-
        removed segmentation): hard Dice against the synthetic target within 1e-3 of the same chain in f32 mode (whose
        network arithmetic (i) has just pinned to the oracle and whose flips (iii) pins to torch.flip);
  (iii) the 8 flip transformers bench.py uses == torch.flip on the same dims (image and mask pipelines), bit for bit,
        on a non-cubic tensor, and the 16 reference transformers (src/definer.py:647-658) on the full padded volume
        shape == oracle/inference.py's tta_augment / tta_deaugment.
"""
import argparse
import contextlib
import io
import itertools
import os
import warnings

import pytest
import torch

from oracle import inference as oinf
from oracle import synth, unet

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")
VOL = (240, 240, 155)
ROI = (128, 128, 128)


def _model():
    from brats21_amd import get_model
    torch.manual_seed(0)
    ns = argparse.Namespace(model="equiunet", width=48, norm="group", act="relu", num_classes=3, dropout=0)
    with contextlib.redirect_stdout(io.StringIO()), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return get_model(ns)


def _volume():
    """bench.py's inference volume: random image, zero outside the 'brain' (the largest of the nested spheres)."""
    x = synth.random_image(1, 4, VOL, seed=99)
    return x * (synth.nested_spheres(1, VOL)[:, 0:1] > 0)


def test_flip8_and_reference_tta_are_exact_index_maps():
    from brats21_amd import tta
    g = torch.Generator().manual_seed(5)
    x = torch.randn((2, 3, 6, 10, 7), generator=g)
    xd = x.to(DEV)
    flips = tta.flip8()
    assert len(flips) == 8
    for t, f in zip(flips, itertools.product([False, True], repeat=3)):
        dims = [2 + a for a in range(3) if f[a]]
        ref = torch.flip(x, dims) if dims else x
        assert torch.equal(t.augment_image(xd).cpu(), ref), f
        assert torch.equal(t.deaugment_mask(xd).cpu(), ref), f
        assert torch.equal(t.deaugment_mask(t.augment_image(xd)).cpu(), x), f
    # the reference's 16 transformers at the padded configs[3] volume shape (non-cubic: 'xyz' changes the shape)
    v = torch.randn((1, 1, 24, 24, 16), generator=g)  # same aspect as 240 x 240 x 160
    vd = v.to(DEV)
    params = oinf.tta_param_list()
    trs = list(tta.get_tta_transforms())
    assert len(trs) == len(params) == 16
    for t, (axe, flip, angle) in zip(trs, params):
        a_ref = oinf.tta_augment(v, axe, flip, angle)
        a = t.augment_image(vd)
        assert torch.equal(a.cpu(), a_ref), (axe, flip, angle)
        assert torch.equal(t.deaugment_mask(a).cpu(), oinf.tta_deaugment(a_ref, axe, flip, angle)), (axe, flip, angle)
        assert torch.equal(t.deaugment_mask(a).cpu(), v)


def test_config3_stitched_logits_vs_oracle_and_bf16_flip8_dice():
    from brats21_amd import tta
    from brats21_amd.evaluate import Evaluator, hard_dice_metric, shape_to_divisible
    from brats21_amd.inferers import sliding_window_inference
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    m = _model()
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    m = m.to(DEV).eval()
    m.skip_deep_heads_in_eval = True
    x = _volume()
    target = synth.nested_spheres(1, VOL)

    # ---- (i) f32, identity TTA: stitched logits against the oracle, whole volume ----
    xd = x.to(DEV)
    padded, p_b, p_a = shape_to_divisible(xd, k=8)
    assert tuple(padded.shape[2:]) == (240, 240, 160)
    m.precision = "fp32"
    with torch.no_grad():
        logits = sliding_window_inference(padded, ROI, 3, lambda w: m(w), overlap=0.5)
        m.precision = "x3"  # the split-precision parity mode (csrc/conv_igemm_x3.hpp) through the same chain
        logits_x3 = sliding_window_inference(padded, ROI, 3, lambda w: m(w), overlap=0.5)
        m.precision = "fp32"
        torch.cuda.synchronize()
        calls = []

        def oracle_predictor(w):
            calls.append(w.shape[0])
            return unet.equiunet_forward(sd, w)[0]

        ref = oinf.sliding_window_inference(padded.cpu(), ROI, 1, oracle_predictor, overlap=0.5)
    assert sum(calls) == 18  # 3 x 3 x 2 windows
    assert tuple(logits.shape) == tuple(ref.shape) == (1, 3, 240, 240, 160)
    err = float((logits.cpu() - ref).abs().max())
    err_x3 = float((logits_x3.cpu() - ref).abs().max())
    print(f"\nconfigs[3] stitched logits vs oracle over {ref.numel()} values: max abs err f32 {err:.3e}, x3 {err_x3:.3e} "
          f"(|logits| max {float(ref.abs().max()):.2f})")
    assert err < 1e-3 and err_x3 < 1e-3, (err, err_x3)

    # ---- (ii) the benchmarked chain: bf16 + 8-flip TTA against the same chain in f32 ----
    td = target.to(DEV)
    res = {}
    for prec, amp in (("fp32", False), ("bf16", True)):
        m.precision = "auto" if amp else "fp32"
        ev = Evaluator(m, tta_transforms=tta.flip8(), sliding_window_size=ROI, sw_batch_size=3, overlap=0.5, k_divisible=8, amp=amp)
        out = ev(xd, target=td)
        assert tuple(out["seg"].shape) == (1, 3) + VOL
        res[prec] = (out["seg"].cpu(), hard_dice_metric(out["seg"], td).cpu())
        del ev
    diff = float((res["fp32"][0] != res["bf16"][0]).float().mean())
    d32, d16 = res["fp32"][1], res["bf16"][1]
    print(f"configs[3] 8-flip TTA: hard Dice vs target f32 {d32.flatten().tolist()} bf16 {d16.flatten().tolist()}; "
          f"segmentation voxels that differ {diff:.3e}")
    assert float((d32 - d16).abs().max()) <= 1e-3, (d32, d16)
    # and the f32 chain's thresholded identity-TTA prediction is the oracle's: sigmoid(ref) > 0.5 <=> ref > 0, away from 0
    seg_ref = (ref[..., p_b[2]:160 - p_a[2]] > 0) & (x.abs().sum(1, keepdim=True) > 0)
    m.precision = "fp32"
    ev1 = Evaluator(m, tta_transforms=None, sliding_window_size=ROI, sw_batch_size=3, overlap=0.5, k_divisible=8, amp=False)
    seg1 = ev1(xd)["seg"].cpu() > 0.5
    sure = (ref[..., p_b[2]:160 - p_a[2]].abs() > 2e-3).expand_as(seg_ref) | ~seg_ref
    assert torch.equal(seg1[sure], seg_ref[sure])
    m.precision = "auto"


def test_whole_volume_path_vs_oracle_f32_and_x3():
    """The reference's PUBLISHED evaluation path (learning/engine.py:305-309, README.md:134-170): no sliding window, the whole
    padded 240 x 240 x 160 volume through the network.  bench.py times it (inference_whole_volume); here the network's logits
    at that size -- non-cubic, 4.4x the voxels of a training patch, every level's tile count different from the 128^3 case --
    are checked against the CPU oracle on the bench's own volume and weights: exact-f32 mode AND the split-precision mode
    within the north-star bar (identity TTA; the 16 transformers are pinned bit for bit above), then the Evaluator's
    whole-volume chain (16 TTA transforms, bf16) against the same chain in f32: hard Dice within 1e-3."""
    from brats21_amd import tta
    from brats21_amd.evaluate import Evaluator, hard_dice_metric, shape_to_divisible
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    m = _model()
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    m = m.to(DEV).eval()
    m.skip_deep_heads_in_eval = True
    x = _volume()
    xd = x.to(DEV)
    padded, p_b, p_a = shape_to_divisible(xd, k=8)
    assert tuple(padded.shape[2:]) == (240, 240, 160)
    with torch.no_grad():
        ref = unet.equiunet_forward(sd, padded.cpu(), deep_supervision=False)
        ref = ref[0] if isinstance(ref, (tuple, list)) else ref
        errs = {}
        for prec in ("fp32", "x3"):
            m.precision = prec
            out = m(padded)
            out = out[0] if isinstance(out, (tuple, list)) else out
            errs[prec] = float((out.cpu() - ref).abs().max())
    print(f"\nwhole padded volume 240x240x160 through the network, max abs logit err vs the CPU oracle over {ref.numel()} values: {errs} "
          f"(|logits| max {float(ref.abs().max()):.2f})")
    assert errs["fp32"] < 1e-3 and errs["x3"] < 1e-3, errs
    # the Evaluator's whole-volume chain with the reference's 16 transforms: bf16 against f32
    td = synth.nested_spheres(1, VOL).to(DEV)
    dice = {}
    for prec, amp in (("fp32", False), ("bf16", True)):
        m.precision = "auto" if amp else "fp32"
        ev = Evaluator(m, tta_transforms=list(tta.get_tta_transforms()), sliding_window_size=None, k_divisible=8, amp=amp)
        dice[prec] = hard_dice_metric(ev(xd, target=td)["seg"], td).cpu()
        del ev
    print(f"whole-volume chain, 16 TTA transforms: hard Dice vs target f32 {dice['fp32'].flatten().tolist()} bf16 {dice['bf16'].flatten().tolist()}")
    assert float((dice["fp32"] - dice["bf16"]).abs().max()) <= 1e-3, dice
    m.precision = "auto"
