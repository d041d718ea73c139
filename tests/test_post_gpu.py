"""-m gpu: the post-forward chain of Engine.evaluate on the GPU (brats21_amd/evaluate.py, csrc/post.hip)
against the reference's golden vectors (tests/golden/post.npz) and the CPU oracle (oracle/evaluate.py)."""
import argparse
import os

import numpy as np
import pytest
import torch

from oracle import evaluate as oev
from oracle import inference as oinf
from oracle import synth, unet

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_pad_crop_background_labels_match_reference_golden(golden_dir):
    from brats21_amd import evaluate as ev
    g = np.load(os.path.join(golden_dir, "post.npz"))
    for tag in "abc":
        d, h, w, k, ms = (int(v) for v in g[f"div_{tag}_meta"])
        x = synth.closed_form("pad" + tag, (2, 4, d, h, w)).to(DEV)
        y, p_b, p_a = ev.shape_to_divisible(x, k=k, min_shape=None if ms < 0 else ms)
        np.testing.assert_array_equal(p_b, g[f"div_{tag}_pb"])
        np.testing.assert_array_equal(p_a, g[f"div_{tag}_pa"])
        np.testing.assert_array_equal(y.cpu().numpy(), g[f"div_{tag}_out"])
        np.testing.assert_array_equal(ev.shape_to_original(y * 2.0 + 1.0, p_b, p_a).cpu().numpy(), g[f"orig_{tag}_out"])
        y4, _, _ = ev.shape_to_divisible(x[0], k=k, min_shape=None if ms < 0 else ms)  # 4-D input (CDHW)
        np.testing.assert_array_equal(y4.cpu().numpy(), g[f"div_{tag}_out"][0])
    out = ev.remove_background_voxels(torch.from_numpy(g["bg_img"]).to(DEV), torch.from_numpy(g["bg_pred"]).to(DEV))
    np.testing.assert_array_equal(out.cpu().numpy(), g["bg_out"])
    lab = ev.to_brats_labels(torch.from_numpy(g["lab_seg"]).to(DEV))
    assert lab.dtype == torch.uint8
    np.testing.assert_array_equal(lab.cpu().numpy(), g["lab_out"].astype(np.uint8))
    with pytest.raises(ValueError):
        ev.shape_to_divisible(torch.zeros(3, 4, 5, device=DEV), k=8)
    with pytest.raises(Exception, match="GPU only"):
        ev.shape_to_divisible(torch.zeros(1, 4, 5, 6, 7), k=8)


def test_fused_finalize_vs_oracle_full_volume():
    """mean over passes + threshold + background mask + labels in one kernel, at the BraTS volume size."""
    from brats21_amd import evaluate as ev
    gen = torch.Generator().manual_seed(7)
    shape = (1, 3, 160, 240, 240)
    probs = [torch.rand(shape, generator=gen) for _ in range(3)]
    img = torch.randn((1, 4) + shape[2:], generator=gen) * (torch.rand((1, 1) + shape[2:], generator=gen) > 0.3)
    ref = oev.ensemble_segmentation(probs, img)
    acc = sum(p.to(DEV) for p in probs)
    seg, lab = ev.finalize_segmentation(acc, 3, img.to(DEV), 0.5, want_labels=True)
    mean = torch.stack(probs).mean(0)
    unsure = (mean - 0.5).abs() < 1e-6  # summation order may flip exact ties only
    assert bool(((seg.cpu() == ref) | unsure).all())
    np.testing.assert_array_equal(lab.cpu().numpy()[:, 0], oev.to_brats_labels(seg.cpu()).numpy())
    # hard Dice: exact integer counts
    tgt = (torch.rand(shape, generator=gen) > 0.6).float()
    tgt[:, 2] = 0
    seg2 = seg.clone()
    seg2[:, 2] = 0                                  # both empty -> 1
    seg2[:, 1] = 0                                  # prediction empty, target not -> 0
    d = ev.hard_dice_metric(seg2, tgt.to(DEV)).cpu()
    d_ref = oev.hard_dice_metric(seg2.cpu(), tgt)
    torch.testing.assert_close(d, d_ref, atol=1e-7, rtol=0)
    assert float(d[0, 2]) == 1.0 and float(d[0, 1]) == 0.0 and 0.2 < float(d[0, 0]) < 0.8
    c = ev.overlap_counts(seg, tgt.to(DEV)).cpu()
    assert int(c[0, 0, 1]) == int(seg[0, 0].sum().item()) and int(c[0, 0, 2]) == int(tgt[0, 0].sum().item())


def test_evaluator_case_vs_oracle_chain():
    """Engine.evaluate's per-case body: pad to 8 -> 16 TTA passes x sliding window -> mean -> threshold ->
    background removal -> labels -> crop, EquiUnet w8 in exact-f32 mode vs the CPU oracle doing the same."""
    from brats21_amd import get_model, tta
    from brats21_amd.evaluate import Evaluator
    sd = synth.fill_state_dict(unet.equiunet_state_shapes(8))
    m = get_model(argparse.Namespace(model="equiunet", width=8, norm="group", act="relu", num_classes=3, dropout=0))
    m.load_state_dict(sd)
    m = m.to(DEV).eval()
    m.skip_deep_heads_in_eval = True
    x = synth.closed_form_image(1, 4, (21, 18, 22), "evalcase")
    x = x * (synth.closed_form("evalmask", (1, 1, 21, 18, 22)) > -0.3)  # background voxels
    tgt = synth.nested_spheres(1, (21, 18, 22))
    comp = tta.Compose([tta.OnAxes(axes=["zxy", "xyz"]), tta.HorizontalFlip(), tta.Rotate90(angles=[0, 90, 180, 270])])
    ev = Evaluator(m, tta_transforms=comp, sliding_window_size=(16, 16, 16), overlap=0.5, amp=False)
    res = ev(x.to(DEV), tgt.to(DEV), want_labels=True)
    # oracle chain
    ref_pred = lambda p: oinf.sliding_window_inference(p, (16, 16, 16), 1, lambda q: unet.equiunet_forward(sd, q),  # noqa: E731
                                                       overlap=0.5)
    with torch.no_grad():
        xp, p_b, p_a = oev.shape_to_divisible(x, k=8)
        mean = oinf.tta_predict(xp, ref_pred)
        seg_ref = oev.remove_background_voxels(xp, oev.as_discrete(mean))
        dice_ref = oev.hard_dice_metric(seg_ref, oev.shape_to_divisible(tgt, k=8)[0])
        lab_ref = oev.shape_to_original(oev.to_brats_labels(seg_ref)[:, None].float(), p_b, p_a)
        seg_ref_c = oev.shape_to_original(seg_ref, p_b, p_a)
        unsure = oev.shape_to_original(((mean - 0.5).abs() < 1e-3).float(), p_b, p_a).bool()
    assert tuple(res["seg"].shape) == (1, 3, 21, 18, 22) and tuple(res["labels"].shape) == (1, 1, 21, 18, 22)
    mism = (res["seg"].cpu() != seg_ref_c)
    assert not bool((mism & ~unsure).any()), "segmentation differs away from the 0.5 threshold"
    assert float(mism.float().mean()) < 2e-3
    lab_m = (res["labels"].cpu().float() != lab_ref) & ~unsure.any(1, keepdim=True)
    assert not bool(lab_m.any())
    assert float((res["dice"].cpu() - dice_ref).abs().max()) < 1e-3   # Dice within 1e-3
    assert 0.0 < float(seg_ref.mean()) < 1.0  # the case is not degenerate


def test_evaluator_two_model_ensemble_vs_oracle():
    """Engine.evaluate with a list of models (learning/engine.py:196-199, :239-249): mean over models of the sigmoid
    outputs on the padded volume (no TTA, no sliding window), threshold, background removal."""
    from brats21_amd import get_model
    from brats21_amd.evaluate import Evaluator
    ns = argparse.Namespace(model="equiunet", width=8, norm="instance", act="relu", num_classes=3, dropout=0)
    sds, models = [], []
    for tag in ("a", "b"):
        shapes = unet.equiunet_state_shapes(8)
        sd = {k: synth.closed_form(f"ens{tag}.{k}", s, 0.6 if k.endswith("conv.weight") or ".0.weight" in k or k == "outconv.weight" else 1.0)
              for k, s in shapes.items()}
        for k in sd:
            if k.endswith("bn.weight"):
                sd[k] = sd[k].abs() + 0.5
        m = get_model(ns)
        m.load_state_dict(sd)
        m = m.to(DEV).eval()
        m.skip_deep_heads_in_eval = True
        sds.append(sd)
        models.append(m)
    x = synth.closed_form_image(1, 4, (19, 22, 17), "ensx")
    x = x * (synth.closed_form("ensmask", (1, 1, 19, 22, 17)) > -0.5)
    res = Evaluator(models, amp=False, use_graph=False)(x.to(DEV))
    with torch.no_grad():
        xp, p_b, p_a = oev.shape_to_divisible(x, k=8)
        probs = [torch.sigmoid(unet.equiunet_forward(sd, xp, norm="instance")[0]) for sd in sds]
        mean = torch.stack(probs).mean(0)
        seg_ref = oev.shape_to_original(oev.remove_background_voxels(xp, oev.as_discrete(mean)), p_b, p_a)
        unsure = oev.shape_to_original(((mean - 0.5).abs() < 1e-3).float(), p_b, p_a).bool()
    mism = res["seg"].cpu() != seg_ref
    assert tuple(res["seg"].shape) == (1, 3, 19, 22, 17)
    assert not bool((mism & ~unsure).any())
    assert 0.0 < float(seg_ref.mean()) < 1.0
