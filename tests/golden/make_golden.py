"""Generates the committed golden vectors in tests/golden/*.npz by running the REFERENCE source
(/root/reference, imported under oracle/refshim.py) on closed-form inputs/weights (oracle/synth.py).

Run here only (the GPU box has no /root/reference):  python tests/golden/make_golden.py
The fixtures hold inputs' *recipes* (names/sizes) and the reference's outputs -- data only, no
reference source in any encoding.
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import refshim, synth, unet  # noqa: E402

refshim.install()
from networks.equiunet2020 import EquiUnet, ConvBnRelu  # noqa: E402
from networks.equiunet2021 import (EquiUnetASSPEvo, EvoNorm3D, SimpleASPPEVO,  # noqa: E402
                                   ConvEvoBlockCorrected, ConvEvo)
from networks.factory import get_norm_layer  # noqa: E402
from utils.inferers import sliding_window_inference, _get_scan_interval  # noqa: E402
from monai.data.utils import dense_patch_slices  # noqa: E402  (the stub)
from monai.losses import DiceLoss  # noqa: E402  (the stub)
import tta  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
torch.set_num_threads(8)


def _criterion():
    # src/definer.py:184-193
    return DiceLoss(include_background=True, sigmoid=True, softmax=False, squared_pred=True,
                    jaccard=False, batch=True)


def _ds_loss(outputs, target, crit):
    # learning/engine.py:322-330 (flatten -> mean over heads)
    heads = [outputs[0]] + list(outputs[1])
    return torch.mean(torch.stack([crit(h, target) for h in heads]))


def _model_fixture(model, shapes_fn, width, size, fname, sub):
    sd = synth.fill_state_dict(shapes_fn(width))
    ref_sd = model.state_dict()
    assert list(ref_sd.keys()) == list(sd.keys()), "state-dict key order/name mismatch vs reference"
    for k in sd:
        assert tuple(ref_sd[k].shape) == tuple(sd[k].shape), k
    model.load_state_dict(sd, strict=True)
    model.train()
    x = synth.closed_form_image(1, 4, size)
    t = synth.nested_spheres(1, size)
    out = model(x)
    loss = _ds_loss(out, t, _criterion())
    loss.backward()
    res = {
        "meta": json.dumps({"width": width, "size": list(size), "sub": sub,
                            "keys": list(sd.keys()), "shapes": [list(v.shape) for v in sd.values()]}),
        "logits": out[0].detach().numpy()[:, :, ::sub, ::sub, ::sub],
        "loss": np.float64(loss.item()),
    }
    for i, d in enumerate(out[1]):
        res[f"deep{i}"] = d.detach().numpy()[:, :, ::2 * sub, ::2 * sub, ::2 * sub]
    names, gn = [], []
    for k, p in model.named_parameters():
        if p.grad is None:
            continue
        names.append(k)
        gn.append(float(p.grad.double().norm()))
        if p.grad.numel() <= 4096:
            res["grad:" + k] = p.grad.numpy().copy()
    res["grad_names"] = json.dumps(names)
    res["grad_norms"] = np.array(gn)
    np.savez_compressed(os.path.join(OUT, fname), **res)
    print(fname, "loss", loss.item(), "logits absmax", float(out[0].abs().max()))


def equiunet_fixtures():
    for size, fname, sub in (((32, 32, 32), "equiunet_w8_32.npz", 1), ((64, 64, 64), "equiunet_w8_64.npz", 4)):
        m = EquiUnet(4, 3, [8, 16, 32, 64], norm_layer="group", act="relu", deep_supervision=True, dropout=0)
        _model_fixture(m, unet.equiunet_state_shapes, 8, size, fname, sub)


def equiunet_prelu_fixture():
    """--act prelu (MONAI Act["prelu"] = torch.nn.PReLU(): one learnable slope per ConvBnRelu) with --norm group."""
    import functools
    m = EquiUnet(4, 3, [8, 16, 32, 64], norm_layer="group", act="prelu", deep_supervision=True, dropout=0)
    _model_fixture(m, functools.partial(unet.equiunet_state_shapes, act="prelu"), 8, (16, 16, 16), "equiunet_w8_16_prelu.npz", 1)


def equiunet_elu_fixture():
    """--act elu (MONAI Act["elu"] = torch.nn.ELU) with the default --norm instance."""
    m = EquiUnet(4, 3, [8, 16, 32, 64], norm_layer="instance", act="elu", deep_supervision=True, dropout=0)
    _model_fixture(m, unet.equiunet_state_shapes, 8, (16, 16, 16), "equiunet_w8_16_elu.npz", 1)


def equiunet_bcn_fixture():
    """--norm bcn (BCNorm(C, 8, estimate=True) = EstBN + per-(sample, group) normalisation + per-group affine,
    networks/factory.py:125-176,189-190) with non-trivial running buffers (as after loading a checkpoint)."""
    import functools
    m = EquiUnet(4, 3, [8, 16, 32, 64], norm_layer="bcn", act="relu", deep_supervision=True, dropout=0)
    _model_fixture(m, functools.partial(unet.equiunet_state_shapes, norm="bcn"), 8, (16, 16, 16), "equiunet_w8_16_bcn.npz", 1)


def equiunet_instance_fixture():
    """--norm instance is the CLI default (src/arguments_train.py:48): InstanceNorm3d(affine=True)."""
    m = EquiUnet(4, 3, [8, 16, 32, 64], norm_layer="instance", act="relu", deep_supervision=True, dropout=0)
    _model_fixture(m, unet.equiunet_state_shapes, 8, (32, 32, 32), "equiunet_w8_32_instance.npz", 2)


def equiunet_batch_fixture():
    """--norm batch (nn.BatchNorm3d(affine=True), networks/factory.py:185-186): a training-mode step on a batch of TWO
    patches (batch statistics, running buffers updated), then the eval-mode forward on the updated buffers."""
    import functools
    width, size, fname = 8, (16, 16, 16), "equiunet_w8_16_batchnorm.npz"
    m = EquiUnet(4, 3, [8, 16, 32, 64], norm_layer="batch", act="relu", deep_supervision=True, dropout=0)
    shapes = unet.equiunet_state_shapes(width, norm="batch")
    sd = synth.fill_state_dict(shapes)
    ref_sd = m.state_dict()
    assert list(ref_sd.keys()) == list(sd.keys()), "state-dict key order/name mismatch vs reference"
    for k in sd:
        assert tuple(ref_sd[k].shape) == tuple(sd[k].shape), k
    m.load_state_dict(sd, strict=True)
    m.train()
    x = synth.closed_form_image(2, 4, size)
    t = synth.nested_spheres(2, size)
    out = m(x)
    loss = _ds_loss(out, t, _criterion())
    loss.backward()
    res = {"meta": json.dumps({"width": width, "size": list(size), "batch": 2, "keys": list(sd.keys())}),
           "logits": out[0].detach().numpy(), "loss": np.float64(loss.item())}
    for i, d in enumerate(out[1]):
        res[f"deep{i}"] = d.detach().numpy()[:, :, ::2, ::2, ::2]
    names, gn = [], []
    for k, p in m.named_parameters():
        names.append(k)
        gn.append(float(p.grad.double().norm()))
        if p.grad.numel() <= 4096:
            res["grad:" + k] = p.grad.numpy().copy()
    res["grad_names"] = json.dumps(names)
    res["grad_norms"] = np.array(gn)
    for k, v in m.state_dict().items():  # the running buffers after one training forward
        if "running_" in k or "num_batches" in k:
            res["buf:" + k] = v.numpy().copy()
    m.eval()
    with torch.no_grad():
        res["eval_logits"] = m(x)[0].numpy()
    np.savez_compressed(os.path.join(OUT, fname), **res)
    print(fname, "loss", loss.item(), "logits absmax", float(out[0].abs().max()), "eval absmax", float(np.abs(res["eval_logits"]).max()))


def assp_fixture():
    m = EquiUnetASSPEvo(4, 3, [16, 32, 64, 128], norm_layer="group", act="relu", deep_supervision=True, dropout=0)
    _model_fixture(m, unet.assp_evo_state_shapes, 16, (32, 32, 32), "assp_w16_32.npz", 1)


def op_fixtures():
    res = {}
    x = synth.closed_form_image(1, 16, (12, 12, 12), "opx")
    # ConvBnRelu with dilation 2 (the `bottom` block, equiunet2020.py:429)
    m = ConvBnRelu(16, 16, "relu", get_norm_layer("group"), dilation=2)
    sd = synth.fill_state_dict({"conv.weight": (16, 16, 3, 3, 3), "bn.weight": (16,), "bn.bias": (16,)})
    m.load_state_dict(sd)
    res["cbr_d2"] = m(x).detach().numpy()
    # EvoNorm3D S0 (equiunet2021.py:55-118), forward + input gradient
    e = EvoNorm3D(16)
    esd = synth.fill_state_dict({k: (1, 16, 1, 1, 1) for k in ("gamma", "beta", "v", "running_var")})
    e.load_state_dict(esd)
    xe = x.clone().requires_grad_(True)
    ye = e(xe)
    (ye * synth.closed_form("evo_go", ye.shape)).sum().backward()
    res["evo_y"] = ye.detach().numpy()
    res["evo_dx"] = xe.grad.numpy()
    res["evo_dgamma"] = e.gamma.grad.numpy()
    res["evo_dbeta"] = e.beta.grad.numpy()
    # SimpleASPPEVO (equiunet2021.py:121-189)
    xa = synth.closed_form_image(1, 32, (8, 8, 8), "asppx")
    a = SimpleASPPEVO(32, 8)
    shapes = {}
    for i, k in enumerate((1, 3, 3, 3)):
        shapes[f"convs.{i}.weight"] = (8, 32, k, k, k)
        shapes[f"convs.{i}.bias"] = (8,)
    shapes["conv_k1.conv.weight"] = (32, 32, 1, 1, 1)
    shapes["conv_k1.conv.bias"] = (32,)
    for k in ("gamma", "beta", "v", "running_var"):
        shapes[f"conv_k1.evo.{k}"] = (1, 32, 1, 1, 1)
    a.load_state_dict(synth.fill_state_dict(shapes))
    res["aspp_y"] = a(xa).detach().numpy()
    # ConvEvoBlockCorrected (equiunet2021.py:192-209) incl. the MONAI ResidualSELayer stub
    b = ConvEvoBlockCorrected(16, 16, 0)
    bsh = {}
    for idx in ("0", "3"):
        bsh[f"conv_conv_se.{idx}.weight"] = (16, 16, 3, 3, 3)
        bsh[f"conv_conv_se.{idx}.bias"] = (16,)
    for idx in ("1", "4"):
        for k in ("gamma", "beta", "v", "running_var"):
            bsh[f"conv_conv_se.{idx}.{k}"] = (1, 16, 1, 1, 1)
    bsh["conv_conv_se.6.fc.0.weight"] = (8, 16)
    bsh["conv_conv_se.6.fc.0.bias"] = (8,)
    bsh["conv_conv_se.6.fc.2.weight"] = (16, 8)
    bsh["conv_conv_se.6.fc.2.bias"] = (16,)
    order = list(b.state_dict().keys())
    bsd = synth.fill_state_dict({k: bsh[k] for k in order})
    b.load_state_dict(bsd)
    res["block_y"] = b(x).detach().numpy()
    np.savez_compressed(os.path.join(OUT, "ops.npz"), **res)
    print("ops.npz", {k: v.shape for k, v in res.items()})


def inference_fixtures():
    res = {}
    # window lists for config 4 and the reference default
    for size in ((240, 240, 155), (240, 240, 160)):
        for ov in (0.25, 0.5):
            iv = _get_scan_interval(size, (128, 128, 128), 3, ov)
            sl = dense_patch_slices(size, (128, 128, 128), iv)
            res[f"starts_{size[2]}_{int(ov * 100)}"] = np.array([[s.start for s in w] for w in sl])
    # stitched output of a cheap analytic predictor that returns (out, [deep]) like the nets do
    x = synth.closed_form_image(1, 4, (20, 27, 17), "swx")
    w = synth.closed_form("swpred", (3, 4), 0.5)

    def predictor(p):
        zz = torch.arange(p.shape[2], dtype=torch.float32).view(1, 1, -1, 1, 1) * 0.01
        out = torch.einsum("oc,ncdhw->nodhw", w, p) + zz  # position-in-window dependent
        return out, [out * 2]

    for mode in ("constant", "gaussian"):
        for ov in (0.25, 0.5):
            y = sliding_window_inference(x, (16, 16, 16), 1, predictor, overlap=ov, mode=mode)
            res[f"sw_{mode}_{int(ov * 100)}"] = y.numpy()
    y = sliding_window_inference(x[..., :12], (16, 16, 16), 2, predictor, overlap=0.5)  # roi > image: padded
    res["sw_pad"] = y.numpy()
    for pm in ("reflect", "replicate", "circular"):  # PytorchPadMode of utils/inferers.py:34,109 (roi > image in two dims)
        y = sliding_window_inference(x[:, :, :, :11, :12], (16, 16, 16), 2, predictor, overlap=0.5, padding_mode=pm)
        res[f"sw_pad_{pm}"] = y.numpy()
    # TTA: parameter order + augmented/deaugmented tensors of a tiny non-cubic volume
    from src_definer_tta import get_tta  # noqa
    comp = get_tta(tta)
    res["tta_params"] = json.dumps([[str(a), bool(f), int(r)] for a, f, r in comp.aug_transform_parameters])
    v = synth.closed_form("ttav", (1, 2, 4, 6, 6))
    augs, deaugs = [], []
    for tr in comp:
        a = tr.augment_image(v)
        augs.append(a.contiguous().numpy().ravel())
        deaugs.append(tr.deaugment_mask(a).contiguous().numpy())
    res["tta_aug"] = np.stack(augs)
    res["tta_roundtrip"] = np.stack(deaugs)
    np.savez_compressed(os.path.join(OUT, "inference.npz"), **res)
    print("inference.npz", {k: getattr(v, "shape", None) for k, v in res.items()})


def post_fixtures():
    """Post-forward chain of Engine.evaluate (learning/engine.py:205-285) from the reference's own
    utils/transforms.py.  numpy >= 1.24 dropped the `np.int` alias the reference still uses, and
    remove_background_voxels calls `.to(img.get_device())`, which only resolves for CUDA tensors:
    both are environment shims applied around the unchanged reference functions."""
    np.int = int
    from utils.transforms import (shape_to_divisible, shape_to_original, remove_background_voxels,  # noqa: E402
                                  ConvertToBratsClassesBasedOnMultiChannel, ChangeLabel3To4)
    res = {}
    rng = np.random.RandomState(2021)
    for tag, size, k, min_shape in (("a", (13, 10, 7), 8, None), ("b", (16, 9, 24), 8, None),
                                    ("c", (5, 6, 7), 4, 12)):
        x = synth.closed_form("pad" + tag, (2, 4) + size)
        y, p_b, p_a = shape_to_divisible(x, k=k, min_shape=min_shape)
        res[f"div_{tag}_meta"] = np.array(list(size) + [k, -1 if min_shape is None else min_shape])
        res[f"div_{tag}_out"] = y.numpy()
        res[f"div_{tag}_pb"], res[f"div_{tag}_pa"] = np.asarray(p_b), np.asarray(p_a)
        res[f"orig_{tag}_out"] = shape_to_original(y * 2.0 + 1.0, p_b, p_a).numpy()
    img = synth.closed_form("bgimg", (1, 4, 9, 10, 11))  # the reference broadcast only works at batch 1
    keep = torch.from_numpy(rng.rand(1, 4, 9, 10, 11) > 0.7)
    img = img * keep  # ~24 % of the voxels are zero in every modality
    out = torch.from_numpy((rng.rand(1, 3, 9, 10, 11) > 0.4).astype(np.float32))
    get_device = torch.Tensor.get_device
    torch.Tensor.get_device = lambda self: self.device
    try:
        res["bg_out"] = remove_background_voxels(img, out).numpy()
    finally:
        torch.Tensor.get_device = get_device
    res["bg_img"], res["bg_pred"] = img.numpy(), out.numpy()
    seg = torch.from_numpy((rng.rand(1, 3, 6, 7, 8) > 0.5).astype(np.float32))  # every TC/WT/ET combination
    lab = ChangeLabel3To4()(ConvertToBratsClassesBasedOnMultiChannel()(seg))
    res["lab_seg"], res["lab_out"] = seg.numpy(), lab.numpy()
    np.savez_compressed(os.path.join(OUT, "post.npz"), **res)
    print("post.npz", {k: getattr(v, "shape", None) for k, v in res.items()})


RANGER_SHAPES = {"conv.weight": (6, 4, 3, 3, 3), "conv.bias": (6,), "fc.weight": (5, 7), "gamma": (1, 6, 1, 1, 1),
                 "unused": (3,)}
RANGER_CASES = {"gc_wd": dict(use_gc=True, gc_conv_only=False, weight_decay=1e-2),
                "convonly": dict(use_gc=True, gc_conv_only=True, weight_decay=0.0),
                "plain": dict(use_gc=False, gc_conv_only=False, weight_decay=1e-5),
                # use_gcnorm: off by default in the reference (learning/optimizer.py:189-190).  (normloss, :192-198, cannot
                # be pinned: the reference's step() raises "a leaf Variable that requires grad is being used in an in-place
                # operation" at :198.)
                "gcnorm": dict(use_gc=True, gc_conv_only=False, weight_decay=0.0, use_gcnorm=True),
                "gcnorm_nogc": dict(use_gc=False, gc_conv_only=False, weight_decay=1e-3, use_gcnorm=True)}


def ranger_fixture():
    """13 steps of the reference's Ranger2020 (learning/optimizer.py) on closed-form parameters / gradients:
    steps 1-5 take the non-adaptive branch (N_sma <= 5), 6.. the RAdam branch, lookahead fires at 6 and 12;
    the parameter 'unused' never receives a gradient (like EvoNorm's v)."""
    import contextlib
    import io
    from learning.optimizer import Ranger2020
    res = {"meta": json.dumps({"shapes": {k: list(v) for k, v in RANGER_SHAPES.items()}, "steps": 13, "lr": 1e-2,
                               "cases": RANGER_CASES})}
    for case, kw in RANGER_CASES.items():
        params = {n: torch.nn.Parameter(synth.closed_form("rp." + n, s)) for n, s in RANGER_SHAPES.items()}
        with contextlib.redirect_stdout(io.StringIO()):
            opt = Ranger2020(list(params.values()), lr=1e-2, alpha=0.5, k=6, N_sma_threshhold=5, betas=(.95, 0.999),
                             eps=1e-5, gc_loc=True, **{"use_gcnorm": False, "normloss": False, **kw})
        for step in range(1, 14):
            for n, p in params.items():
                p.grad = None if n == "unused" else synth.closed_form(f"rg.{n}.{step}", RANGER_SHAPES[n], 0.1 * step)
            opt.step()
            if step in (5, 6, 13):
                for n, p in params.items():
                    res[f"{case}.{step}.{n}"] = p.detach().numpy().copy()
        for n, p in params.items():
            if n != "unused":
                st = opt.state[p]
                res[f"{case}.exp_avg.{n}"] = st["exp_avg"].numpy().copy()
                res[f"{case}.exp_avg_sq.{n}"] = st["exp_avg_sq"].numpy().copy()
                res[f"{case}.slow.{n}"] = st["slow_buffer"].numpy().copy()
    np.savez_compressed(os.path.join(OUT, "ranger.npz"), **res)
    print("ranger.npz", len(res), "arrays")


def prep_fixture():
    """Input pipeline: the reference's own NormalizeIntensity (utils/transforms.py:328-406) and its
    ConvertToMultiChannelBasedOnBratsClassesd (:145-166) on seeded inputs."""
    from utils.transforms import NormalizeIntensity, ConvertToMultiChannelBasedOnBratsClassesd
    rng = np.random.RandomState(7)
    img = (rng.rand(4, 9, 10, 11).astype(np.float32) * 300.0 + 20.0) * (rng.rand(4, 9, 10, 11) > 0.35)
    img[3] = 0.0                      # a channel without any non-zero voxel is returned unchanged
    img[2][img[2] != 0] = 57.0        # zero variance -> divisor 1
    res = {"img": img.astype(np.float32)}
    for tag, kw in (("nz", dict(nonzero=True, channel_wise=True)),
                    ("nz_clip", dict(nonzero=True, channel_wise=True, remove_outliers=True, outliers_value=1.5)),
                    ("all", dict(nonzero=False, channel_wise=True))):
        res["norm_" + tag] = NormalizeIntensity(**kw)(img.astype(np.float32).copy())
    label = rng.choice(np.array([0, 1, 2, 4], dtype=np.float32), size=(6, 7, 8))
    conv = ConvertToMultiChannelBasedOnBratsClassesd.__new__(ConvertToMultiChannelBasedOnBratsClassesd)
    conv.keys = ["seg"]  # the stub MapTransform base has no __init__
    res["label"] = label
    res["label_utils"] = conv({"seg": label})["seg"]
    np.savez_compressed(os.path.join(OUT, "prep.npz"), **res)
    print("prep.npz", {k: getattr(v, "shape", None) for k, v in res.items()})


if __name__ == "__main__":
    # src/definer.py imports half of MONAI at module import; restate only its 6-line TTA list
    # constructor call (src/definer.py:653-657) against the reference's own tta package.
    import types

    m = types.ModuleType("src_definer_tta")
    m.get_tta = lambda t: t.Compose([t.OnAxes(axes=["zxy", "xyz"]), t.HorizontalFlip(),
                                     t.Rotate90(angles=[0, 90, 180, 270])])
    sys.modules["src_definer_tta"] = m
    which = sys.argv[1:] or ["equiunet", "assp", "ops", "inference", "post", "ranger", "prep", "equiunet_instance", "equiunet_elu", "equiunet_prelu", "equiunet_batch", "equiunet_bcn"]
    if "equiunet" in which:
        equiunet_fixtures()
    if "equiunet_instance" in which:
        equiunet_instance_fixture()
    if "equiunet_elu" in which:
        equiunet_elu_fixture()
    if "equiunet_prelu" in which:
        equiunet_prelu_fixture()
    if "equiunet_batch" in which:
        equiunet_batch_fixture()
    if "equiunet_bcn" in which:
        equiunet_bcn_fixture()
    if "assp" in which:
        assp_fixture()
    if "ops" in which:
        op_fixtures()
    if "inference" in which:
        inference_fixtures()
    if "post" in which:
        post_fixtures()
    if "ranger" in which:
        ranger_fixture()
    if "prep" in which:
        prep_fixture()
