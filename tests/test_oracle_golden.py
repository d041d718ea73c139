"""-m "not gpu": the CPU oracle (oracle/) against the golden vectors produced by the reference
source (tests/golden/make_golden.py).  This is what pins the oracle."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import inference as oinf
from oracle import synth, unet

TOL = 2e-5  # fp32 CPU vs fp32 CPU, same ATen kernels, different op grouping


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def _run(forward, shapes_fn, g):
    meta = json.loads(str(g["meta"]))
    shapes = shapes_fn(meta["width"])
    assert list(shapes.keys()) == meta["keys"]
    assert [list(s) for s in shapes.values()] == meta["shapes"]
    sd = {k: v.requires_grad_(True) for k, v in synth.fill_state_dict(shapes).items()}
    size = tuple(meta["size"])
    x = synth.closed_form_image(1, 4, size)
    t = synth.nested_spheres(1, size)
    out = forward(sd, x)
    loss = unet.deep_supervision_loss(out, t)
    loss.backward()
    return meta, sd, out, loss


@pytest.mark.parametrize("fname", ["equiunet_w8_32.npz", "equiunet_w8_64.npz"])
def test_equiunet_oracle_matches_reference(golden_dir, fname):
    g = _load(golden_dir, fname)
    meta, sd, out, loss = _run(unet.equiunet_forward, unet.equiunet_state_shapes, g)
    s = meta["sub"]
    np.testing.assert_allclose(out[0].detach().numpy()[:, :, ::s, ::s, ::s], g["logits"], atol=TOL, rtol=0)
    for i, d in enumerate(out[1]):
        np.testing.assert_allclose(d.detach().numpy()[:, :, ::2 * s, ::2 * s, ::2 * s], g[f"deep{i}"], atol=TOL, rtol=0)
    assert abs(loss.item() - float(g["loss"])) < 1e-6
    names = json.loads(str(g["grad_names"]))
    norms = np.array([float(sd[k].grad.double().norm()) for k in names])
    np.testing.assert_allclose(norms, g["grad_norms"], rtol=2e-4, atol=1e-9)
    for k in g.files:
        if k.startswith("grad:"):
            np.testing.assert_allclose(sd[k[5:]].grad.numpy(), g[k], atol=1e-6, rtol=1e-4)


def test_assp_oracle_matches_reference(golden_dir):
    g = _load(golden_dir, "assp_w16_32.npz")
    meta, sd, out, loss = _run(unet.assp_evo_forward, unet.assp_evo_state_shapes, g)
    np.testing.assert_allclose(out[0].detach().numpy(), g["logits"], atol=TOL, rtol=0)
    for i, d in enumerate(out[1]):
        np.testing.assert_allclose(d.detach().numpy()[:, :, ::2, ::2, ::2], g[f"deep{i}"], atol=TOL, rtol=0)
    assert abs(loss.item() - float(g["loss"])) < 1e-6
    names = json.loads(str(g["grad_names"]))
    assert not any(n.endswith(".v") for n in names)  # EvoNorm `v` is statically unused
    norms = np.array([float(sd[k].grad.double().norm()) for k in names])
    np.testing.assert_allclose(norms, g["grad_norms"], rtol=2e-4, atol=1e-9)


def test_op_vectors(golden_dir):
    g = _load(golden_dir, "ops.npz")
    x = synth.closed_form_image(1, 16, (12, 12, 12), "opx")
    sd = synth.fill_state_dict({"conv.weight": (16, 16, 3, 3, 3), "bn.weight": (16,), "bn.bias": (16,)})
    y = unet.conv_gn_act({"p." + k: v for k, v in sd.items()}, "p", x, 2)
    np.testing.assert_allclose(y.numpy(), g["cbr_d2"], atol=TOL)
    esd = synth.fill_state_dict({k: (1, 16, 1, 1, 1) for k in ("gamma", "beta", "v", "running_var")})
    xe = x.clone().requires_grad_(True)
    gam, bet = esd["gamma"].requires_grad_(True), esd["beta"].requires_grad_(True)
    ye = unet.evonorm_s0(xe, gam, bet)
    (ye * synth.closed_form("evo_go", ye.shape)).sum().backward()
    np.testing.assert_allclose(ye.detach().numpy(), g["evo_y"], atol=TOL)
    np.testing.assert_allclose(xe.grad.numpy(), g["evo_dx"], atol=TOL)
    np.testing.assert_allclose(gam.grad.numpy(), g["evo_dgamma"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(bet.grad.numpy(), g["evo_dbeta"], rtol=1e-4, atol=1e-4)
    shapes = {}
    for i, k in enumerate((1, 3, 3, 3)):
        shapes[f"convs.{i}.weight"] = (8, 32, k, k, k)
        shapes[f"convs.{i}.bias"] = (8,)
    shapes["conv_k1.conv.weight"] = (32, 32, 1, 1, 1)
    shapes["conv_k1.conv.bias"] = (32,)
    for k in ("gamma", "beta", "v", "running_var"):
        shapes[f"conv_k1.evo.{k}"] = (1, 32, 1, 1, 1)
    asd = {"a." + k: v for k, v in synth.fill_state_dict(shapes).items()}
    xa = synth.closed_form_image(1, 32, (8, 8, 8), "asppx")
    np.testing.assert_allclose(unet.aspp(asd, "a", xa).numpy(), g["aspp_y"], atol=TOL)
    order = []
    for idx in ("0", "1", "3", "4"):
        if idx in ("0", "3"):
            order += [(f"conv_conv_se.{idx}.weight", (16, 16, 3, 3, 3)), (f"conv_conv_se.{idx}.bias", (16,))]
        else:
            order += [(f"conv_conv_se.{idx}.{k}", (1, 16, 1, 1, 1)) for k in ("gamma", "beta", "v", "running_var")]
    order += [("conv_conv_se.6.fc.0.weight", (8, 16)), ("conv_conv_se.6.fc.0.bias", (8,)),
              ("conv_conv_se.6.fc.2.weight", (16, 8)), ("conv_conv_se.6.fc.2.bias", (16,))]
    bsd = {"b." + k: v for k, v in synth.fill_state_dict(dict(order)).items()}
    np.testing.assert_allclose(unet.conv_evo_block(bsd, "b", x).numpy(), g["block_y"], atol=TOL)


def test_sliding_window_and_tta_vectors(golden_dir):
    g = _load(golden_dir, "inference.npz")
    for size in ((240, 240, 155), (240, 240, 160)):
        for ov in (0.25, 0.5):
            st = oinf.window_starts(size, (128,) * 3, oinf.scan_interval(size, (128,) * 3, ov))
            assert len(st) == 18
            np.testing.assert_array_equal(np.array(st), g[f"starts_{size[2]}_{int(ov * 100)}"])
    x = synth.closed_form_image(1, 4, (20, 27, 17), "swx")
    w = synth.closed_form("swpred", (3, 4), 0.5)

    def predictor(p):
        zz = torch.arange(p.shape[2], dtype=torch.float32).view(1, 1, -1, 1, 1) * 0.01
        out = torch.einsum("oc,ncdhw->nodhw", w, p) + zz
        return out, [out * 2]

    for mode in ("constant", "gaussian"):
        for ov in (0.25, 0.5):
            y = oinf.sliding_window_inference(x, (16, 16, 16), 1, predictor, overlap=ov, mode=mode)
            np.testing.assert_allclose(y.numpy(), g[f"sw_{mode}_{int(ov * 100)}"], atol=1e-6)
    y = oinf.sliding_window_inference(x[..., :12], (16, 16, 16), 2, predictor, overlap=0.5)
    np.testing.assert_allclose(y.numpy(), g["sw_pad"], atol=1e-6)
    for pm in ("reflect", "replicate", "circular"):
        y = oinf.sliding_window_inference(x[:, :, :, :11, :12], (16, 16, 16), 2, predictor, overlap=0.5, padding_mode=pm)
        np.testing.assert_allclose(y.numpy(), g[f"sw_pad_{pm}"], atol=1e-6)
    params = oinf.tta_param_list()
    assert [[a, f, r] for a, f, r in params] == json.loads(str(g["tta_params"]))
    v = synth.closed_form("ttav", (1, 2, 4, 6, 6))
    for i, p in enumerate(params):
        a = oinf.tta_augment(v, *p)
        np.testing.assert_array_equal(a.contiguous().numpy().ravel(), g["tta_aug"][i])
        np.testing.assert_array_equal(oinf.tta_deaugment(a, *p).contiguous().numpy(), g["tta_roundtrip"][i])
        np.testing.assert_array_equal(g["tta_roundtrip"][i], v.numpy())  # all 16 exactly invertible (F7)


def test_post_forward_chain_vectors(golden_dir):
    """oracle/evaluate.py against the reference's utils/transforms.py outputs (tests/golden/post.npz)."""
    from oracle import evaluate as oev
    g = np.load(os.path.join(golden_dir, "post.npz"))
    for tag in "abc":
        d, h, w, k, ms = (int(v) for v in g[f"div_{tag}_meta"])
        x = synth.closed_form("pad" + tag, (2, 4, d, h, w))
        y, p_b, p_a = oev.shape_to_divisible(x, k=k, min_shape=None if ms < 0 else ms)
        np.testing.assert_array_equal(p_b, g[f"div_{tag}_pb"])
        np.testing.assert_array_equal(p_a, g[f"div_{tag}_pa"])
        np.testing.assert_array_equal(y.numpy(), g[f"div_{tag}_out"])
        np.testing.assert_array_equal(oev.shape_to_original(y * 2.0 + 1.0, p_b, p_a).numpy(), g[f"orig_{tag}_out"])
    out = oev.remove_background_voxels(torch.from_numpy(g["bg_img"]), torch.from_numpy(g["bg_pred"]))
    np.testing.assert_array_equal(out.numpy(), g["bg_out"])
    assert 0.1 < 1 - float(g["bg_out"].sum() / g["bg_pred"].sum()) < 0.5  # the mask really removes something
    lab = oev.to_brats_labels(torch.from_numpy(g["lab_seg"]))
    np.testing.assert_array_equal(lab.numpy()[:, None], g["lab_out"])
    assert set(np.unique(g["lab_out"]).tolist()) == {0, 1, 2, 4}


def test_hard_dice_known_answers():
    from oracle import evaluate as oev
    p = torch.zeros(1, 3, 2, 2, 2)
    t = torch.zeros(1, 3, 2, 2, 2)
    p[0, 0, 0] = 1            # 4 voxels predicted
    t[0, 0, :, 0] = 1         # 4 voxels true, 2 shared -> 2*2/8
    p[0, 1, 0, 0, 0] = 1      # predicted, nothing true -> 0
    np.testing.assert_allclose(oev.hard_dice_metric(p, t).numpy(), [[0.5, 0.0, 1.0]])


def test_ranger_oracle_matches_reference(golden_dir):
    from _replay import ranger_replay
    from oracle import ranger as orang

    def step_fn(params, lr, kw):
        states = {n: None for n in params}

        def step(grads):
            for n, p in params.items():
                if grads[n] is None:
                    continue
                if states[n] is None:
                    states[n] = orang.new_state(p)
                orang.ranger_step(p, grads[n], states[n], lr=lr, **kw)
        return {"step": step, "state": lambda n: states[n]}

    ranger_replay(golden_dir, lambda t: t.clone(), step_fn, lambda t: t.numpy())


def test_input_pipeline_vectors(golden_dir):
    """oracle/prep.py against the reference's NormalizeIntensity / label conversion (tests/golden/prep.npz) and
    known answers for the MONAI-side augmentations."""
    from oracle import prep
    g = np.load(os.path.join(golden_dir, "prep.npz"))
    img = g["img"]
    np.testing.assert_allclose(prep.normalize_intensity(img), g["norm_nz"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(prep.normalize_intensity(img, remove_outliers=True, outliers_value=1.5), g["norm_nz_clip"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(prep.normalize_intensity(img, nonzero=False), g["norm_all"], rtol=0, atol=1e-6)
    assert np.array_equal(g["norm_nz"][3], img[3]) and float(np.abs(g["norm_nz"][2]).max()) == 0.0
    np.testing.assert_array_equal(prep.convert_to_multichannel(g["label"], "utils"), g["label_utils"])
    m = prep.convert_to_multichannel(g["label"], "monai")
    np.testing.assert_array_equal(m[0], g["label_utils"][1])
    np.testing.assert_array_equal(m[1], g["label_utils"][0])
    # known answers: rot90 over spatial axes (0, 2), flip of all axes, AdjustContrast end points
    a = np.arange(2 * 2 * 1 * 3, dtype=np.float32).reshape(2, 2, 1, 3)
    r = prep.rotate90(a, 1)
    assert r.shape == (2, 3, 1, 2) and r[0, 0, 0, 0] == a[0, 0, 0, 2] and r[0, 2, 0, 1] == a[0, 1, 0, 0]
    f = prep.flip(a)
    assert f[1, 0, 0, 0] == a[1, 1, 0, 2]
    c = prep.adjust_contrast(np.array([[[[1.0, 3.0, 5.0]]]], dtype=np.float32), 2.0)
    np.testing.assert_allclose(c.ravel(), [1.0, 2.0, 5.0], atol=1e-5)


def test_equiunet_instance_norm_oracle_matches_reference(golden_dir):
    """--norm instance (the reference CLI's default) through the same oracle with per-channel statistics."""
    g = _load(golden_dir, "equiunet_w8_32_instance.npz")
    meta, sd, out, loss = _run(lambda sd, x: unet.equiunet_forward(sd, x, norm="instance"), unet.equiunet_state_shapes, g)
    s = meta["sub"]
    np.testing.assert_allclose(out[0].detach().numpy()[:, :, ::s, ::s, ::s], g["logits"], atol=TOL, rtol=0)
    for i, d in enumerate(out[1]):
        np.testing.assert_allclose(d.detach().numpy()[:, :, ::2 * s, ::2 * s, ::2 * s], g[f"deep{i}"], atol=TOL, rtol=0)
    assert abs(loss.item() - float(g["loss"])) < 1e-6
    names = json.loads(str(g["grad_names"]))
    norms = np.array([float(sd[k].grad.double().norm()) for k in names])
    np.testing.assert_allclose(norms, g["grad_norms"], rtol=2e-4, atol=1e-9)


def test_equiunet_elu_oracle_matches_reference(golden_dir):
    g = _load(golden_dir, "equiunet_w8_16_elu.npz")
    meta, sd, out, loss = _run(lambda sd, x: unet.equiunet_forward(sd, x, act="elu", norm="instance"), unet.equiunet_state_shapes, g)
    np.testing.assert_allclose(out[0].detach().numpy(), g["logits"], atol=TOL, rtol=0)
    assert abs(loss.item() - float(g["loss"])) < 1e-6
    names = json.loads(str(g["grad_names"]))
    norms = np.array([float(sd[k].grad.double().norm()) for k in names])
    np.testing.assert_allclose(norms, g["grad_norms"], rtol=2e-4, atol=1e-9)


def test_equiunet_prelu_oracle_matches_reference(golden_dir):
    """--act prelu: the reference's EquiUnet with nn.PReLU units (state-dict keys "<unit>.prelu.weight"), logits, loss,
    gradient norms and the 17 slope gradients themselves."""
    import functools
    g = _load(golden_dir, "equiunet_w8_16_prelu.npz")
    meta, sd, out, loss = _run(lambda sd, x: unet.equiunet_forward(sd, x, act="prelu", norm="group"),
                               functools.partial(unet.equiunet_state_shapes, act="prelu"), g)
    np.testing.assert_allclose(out[0].detach().numpy(), g["logits"], atol=TOL, rtol=0)
    assert abs(loss.item() - float(g["loss"])) < 1e-6
    names = json.loads(str(g["grad_names"]))
    assert sum(n.endswith(".prelu.weight") for n in names) == 17
    norms = np.array([float(sd[k].grad.double().norm()) for k in names])
    np.testing.assert_allclose(norms, g["grad_norms"], rtol=2e-4, atol=1e-9)
    for k in g.files:
        if k.startswith("grad:") and k.endswith(".prelu.weight"):
            np.testing.assert_allclose(sd[k[5:]].grad.numpy(), g[k], atol=1e-6, rtol=1e-4)


def test_equiunet_bcn_oracle_matches_reference(golden_dir):
    """--norm bcn (BCNorm(C, 8, estimate=True): EstBN on its running buffers + per-(sample, group) normalisation + per-group
    affine, networks/factory.py:125-176,189-190) against the reference's own outputs: state-dict keys and shapes, logits, deep
    heads, loss, gradient norms (the parameters whose gradient is analytically zero -- EstBN's weight / bias where a group
    holds ONE channel -- are noise of 1e-8 in the reference itself: absolute tolerance)."""
    import functools
    g = _load(golden_dir, "equiunet_w8_16_bcn.npz")
    meta = json.loads(str(g["meta"]))
    shapes = functools.partial(unet.equiunet_state_shapes, norm="bcn")(meta["width"])
    assert list(shapes.keys()) == meta["keys"] and [list(v) for v in shapes.values()] == meta["shapes"]
    buffers = ("running_mean", "running_var", "num_batches_tracked", "estbn_moving_speed")
    sd = {k: (v if k.rsplit(".", 1)[-1] in buffers else v.requires_grad_(True)) for k, v in synth.fill_state_dict(shapes).items()}
    size = tuple(meta["size"])
    x, t = synth.closed_form_image(1, 4, size), synth.nested_spheres(1, size)
    out = unet.equiunet_forward(sd, x, norm="bcn")
    loss = unet.deep_supervision_loss(out, t)
    loss.backward()
    np.testing.assert_allclose(out[0].detach().numpy(), g["logits"], atol=TOL, rtol=0)
    for i, d in enumerate(out[1]):
        np.testing.assert_allclose(d.detach().numpy()[:, :, ::2, ::2, ::2], g[f"deep{i}"], atol=TOL, rtol=0)
    assert abs(loss.item() - float(g["loss"])) < 1e-6
    names = json.loads(str(g["grad_names"]))
    assert sum(n.endswith(".bn.bn.weight") for n in names) == 17 and sum(n.endswith(".bn.weight") for n in names) == 34
    norms = np.array([float(sd[k].grad.double().norm()) for k in names])
    np.testing.assert_allclose(norms, g["grad_norms"], rtol=2e-4, atol=2e-8)
    for k in g.files:
        if k.startswith("grad:") and ".bn." in k:
            np.testing.assert_allclose(sd[k[5:]].grad.numpy(), g[k], atol=1e-6 * max(1.0, float(np.abs(g[k]).max()) * 1e3), rtol=1e-3)
    # a moving EstBN is refused, not silently frozen
    sd2 = dict(sd)
    sd2["encoder1.ConvBnRelu1.bn.bn.estbn_moving_speed"] = torch.full((1,), 0.1)
    with pytest.raises(NotImplementedError):
        unet.equiunet_forward(sd2, x, norm="bcn")


def test_equiunet_batch_norm_oracle_matches_reference(golden_dir):
    """--norm batch (nn.BatchNorm3d, networks/factory.py:185-186): training-mode step on two patches (logits, loss, gradients,
    updated running buffers) and the eval-mode forward on those buffers, against the reference's own outputs."""
    g = _load(golden_dir, "equiunet_w8_16_batchnorm.npz")
    meta = json.loads(str(g["meta"]))
    shapes = unet.equiunet_state_shapes(meta["width"], norm="batch")
    assert list(shapes.keys()) == meta["keys"]
    sd = {k: (v.requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else v)
          for k, v in synth.fill_state_dict(shapes).items()}
    size = tuple(meta["size"])
    x, t = synth.closed_form_image(meta["batch"], 4, size), synth.nested_spheres(meta["batch"], size)
    new_stats = {}
    out = unet.equiunet_forward(sd, x, norm="batch", training=True, new_stats=new_stats)
    loss = unet.deep_supervision_loss(out, t)
    loss.backward()
    np.testing.assert_allclose(out[0].detach().numpy(), g["logits"], atol=TOL, rtol=0)
    for i, d in enumerate(out[1]):
        np.testing.assert_allclose(d.detach().numpy()[:, :, ::2, ::2, ::2], g[f"deep{i}"], atol=TOL, rtol=0)
    assert abs(loss.item() - float(g["loss"])) < 1e-6
    names = json.loads(str(g["grad_names"]))
    norms = np.array([float(sd[k].grad.double().norm()) for k in names])
    np.testing.assert_allclose(norms, g["grad_norms"], rtol=2e-4, atol=1e-9)
    for k, v in new_stats.items():
        np.testing.assert_allclose(v.numpy(), g["buf:" + k], atol=1e-6, rtol=1e-5)
    sd_eval = {k: v.detach() for k, v in sd.items()}
    sd_eval.update(new_stats)
    with torch.no_grad():
        ev = unet.equiunet_forward(sd_eval, x, norm="batch", training=False)[0]
    np.testing.assert_allclose(ev.numpy(), g["eval_logits"], atol=TOL, rtol=0)
