"""-m gpu: end-to-end parity of brats21_amd.EquiUnet (HIP kernels through the C ABI) with
(a) the committed golden vectors produced by the reference source and (b) the CPU oracle on the
same closed-form inputs.  Bar (BASELINE.json north_star): logits within 1e-3 abs in the f32 mode;
the bf16 mode's deviation is asserted at a stated looser bound."""
import argparse
import contextlib
import io
import json
import os

import numpy as np
import pytest
import torch

from oracle import synth, unet

pytestmark = pytest.mark.gpu
LOGIT_ATOL = 1e-3  # north_star: "within 1e-3 abs on identical inputs"


def _model(width, sd=None, precision="fp32", norm="group"):
    from brats21_amd import get_model
    m = get_model(argparse.Namespace(model="equiunet", width=width, norm=norm, act="relu", num_classes=3, dropout=0))
    if sd is not None:
        m.load_state_dict(sd, strict=True)
    m.precision = precision
    return m.cuda()


def _golden(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


@pytest.mark.parametrize("fname", ["equiunet_w8_32.npz", "equiunet_w8_64.npz", "equiunet_w8_32_instance.npz"])
def test_equiunet_f32_matches_reference_golden(golden_dir, fname):
    """--norm group and --norm instance (the CLI default: InstanceNorm3d(affine=True))."""
    g = _golden(golden_dir, fname)
    meta = json.loads(str(g["meta"]))
    size, s = tuple(meta["size"]), meta["sub"]
    sd = synth.fill_state_dict(unet.equiunet_state_shapes(meta["width"]))
    m = _model(meta["width"], sd, "fp32", norm="instance" if "instance" in fname else "group").train()
    x = synth.closed_form_image(1, 4, size).cuda()
    t = synth.nested_spheres(1, size).cuda()
    out, deeps = m(x)
    assert out.shape == (1, 3, *size) and len(deeps) == 4 and all(d.shape == out.shape for d in deeps)
    err = np.abs(out.detach().cpu().numpy()[:, :, ::s, ::s, ::s] - g["logits"]).max()
    assert err < LOGIT_ATOL, f"logit max abs err {err}"
    for i, d in enumerate(deeps):
        e = np.abs(d.detach().cpu().numpy()[:, :, ::2 * s, ::2 * s, ::2 * s] - g[f"deep{i}"]).max()
        assert e < LOGIT_ATOL, f"deep head {i} max abs err {e}"
    # loss + gradients through the PyTorch-side deep-supervision Dice loss (learning/engine.py:312-333)
    loss = unet.deep_supervision_loss((out, deeps), t)
    assert abs(loss.item() - float(g["loss"])) < 1e-4
    loss.backward()
    names = json.loads(str(g["grad_names"]))
    params = dict(m.named_parameters())
    norms = np.array([float(params[k].grad.double().norm()) for k in names])
    np.testing.assert_allclose(norms, g["grad_norms"], rtol=2e-3, atol=1e-7)
    for k in g.files:
        if k.startswith("grad:"):
            ref = g[k]
            got = params[k[5:]].grad.cpu().numpy()
            np.testing.assert_allclose(got, ref, atol=2e-3 * max(np.abs(ref).max(), 1e-6), rtol=2e-3)


def test_equiunet_bf16_deviation_bounded(golden_dir):
    """bf16 storage cannot meet 1e-3 through 20 layers; assert the deviation
    stays at bf16 level: logits mean abs dev < 0.05 (max < 1.0 on |logits| <= 5.2) / hard-Dice of the thresholded masks within 1e-2."""
    g = _golden(golden_dir, "equiunet_w8_32.npz")
    sd = synth.fill_state_dict(unet.equiunet_state_shapes(8))
    m = _model(8, sd, "bf16").eval()
    x = synth.closed_form_image(1, 4, (32, 32, 32)).cuda()
    with torch.no_grad():
        out, deeps = m(x)
    ref = torch.from_numpy(g["logits"])
    err = (out.cpu() - ref).abs()
    assert float(err.max()) < 1.0 and float(err.mean()) < 0.05, (float(err.max()), float(err.mean()))
    t = synth.nested_spheres(1, (32, 32, 32))
    d_ref, d_got = unet.hard_dice(ref, t), unet.hard_dice(out.cpu(), t)
    assert float((d_ref - d_got).abs().max()) < 1e-2


def test_equiunet_autocast_selects_bf16_and_width48_runs():
    """Width-48 channel counts (CK=48 chunks, 96-wide concat buffers) at a small volume vs the oracle."""
    # random (seeded) image and perturbed weights: the closed-form volume is constant outside its
    # ellipsoid, i.e. full of exact max-pool / ReLU ties, which makes single gradients ill-conditioned
    g = torch.Generator().manual_seed(7)
    sd = {k: v + 0.02 * torch.randn(v.shape, generator=g) for k, v in
          synth.fill_state_dict(unet.equiunet_state_shapes(48)).items()}
    m = _model(48, sd, "auto").train()
    size = (16, 16, 16)
    x = synth.random_image(2, 4, size)
    t = synth.nested_spheres(2, size)
    # ground truth = the oracle evaluated in float64: ReLU / max-pool make some gradients ill-conditioned
    # (the f32 CPU oracle itself is 2e-3..4e-3 off the f64 one on these inputs), so both f32
    # implementations are judged against f64 rather than against each other.
    sd_ref = {k: v.clone().double().requires_grad_(True) for k, v in sd.items()}
    out_ref = unet.equiunet_forward(sd_ref, x.double())
    loss_ref = unet.deep_supervision_loss(out_ref, t.double())
    loss_ref.backward()
    out, deeps = m(x.cuda())  # no autocast -> exact f32 kernels
    err = float((out.detach().cpu().double() - out_ref[0].detach()).abs().max())
    assert err < LOGIT_ATOL, err
    loss = unet.deep_supervision_loss((out, deeps), t.cuda())
    loss.backward()
    assert abs(loss.item() - loss_ref.item()) < 1e-4
    for k, p in m.named_parameters():
        ref = sd_ref[k].grad
        rel = float((p.grad.cpu().double() - ref).norm() / (ref.norm() + 1e-30))
        assert rel < 5e-3, (k, rel)
    m.zero_grad()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out_b, deeps_b = m(x.cuda())
        loss_b = unet.deep_supervision_loss((out_b, deeps_b), t.cuda())
    loss_b.backward()
    assert out_b.dtype == torch.float32
    assert float((out_b.detach().cpu().double() - out_ref[0].detach()).abs().max()) < 0.25
    assert abs(loss_b.item() - loss_ref.item()) < 5e-3
    # yardstick for "what bf16 storage costs": torch's own CPU bf16 autocast of the oracle on the same
    # inputs (measured here: median 4-9 %, worst 24-27 % relative gradient error vs f64 -- far-from-loss
    # encoder gradients carry the most rounding noise).  The HIP bf16 path must be in that league.
    sd_b = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    with torch.autocast("cpu", dtype=torch.bfloat16):
        out_c = unet.equiunet_forward(sd_b, x)
    unet.deep_supervision_loss(out_c, t).backward()
    e_hip, e_ref = [], []
    for k, p in m.named_parameters():
        ref = sd_ref[k].grad
        e_hip.append(float((p.grad.cpu().double() - ref).norm() / (ref.norm() + 1e-30)))
        e_ref.append(float((sd_b[k].grad.double() - ref).norm() / (ref.norm() + 1e-30)))
    med = lambda v: sorted(v)[len(v) // 2]  # noqa: E731
    assert med(e_hip) <= 1.5 * med(e_ref) + 0.02, (med(e_hip), med(e_ref))
    assert max(e_hip) <= 1.5 * max(e_ref) + 0.05, (max(e_hip), max(e_ref))


def test_cpu_input_fails_loudly():
    from brats21_amd import BratsHipError
    m = _model(8)
    with pytest.raises(BratsHipError):
        m(torch.zeros(1, 4, 16, 16, 16))


@pytest.mark.parametrize("size,batch", [((24, 40, 16), 1), ((16, 16, 48), 3)])
def test_equiunet_ragged_volumes_and_batches(size, batch):
    """Non-cubic volumes (partial tiles in every dimension of the 4x4x16 tiling) and odd batch sizes, f32 mode,
    forward + gradients vs the CPU oracle."""
    sd = synth.fill_state_dict(unet.equiunet_state_shapes(8))
    m = _model(8, sd, "fp32").train()
    x = synth.random_image(batch, 4, size, seed=11)
    t = synth.nested_spheres(batch, size)
    sd_ref = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    out_ref = unet.equiunet_forward(sd_ref, x)
    loss_ref = unet.deep_supervision_loss(out_ref, t)
    loss_ref.backward()
    out, deeps = m(x.cuda())
    assert float((out.detach().cpu() - out_ref[0].detach()).abs().max()) < LOGIT_ATOL
    for d, dr in zip(deeps, out_ref[1]):
        assert float((d.detach().cpu() - dr.detach()).abs().max()) < LOGIT_ATOL
    loss = unet.deep_supervision_loss((out, deeps), t.cuda())
    loss.backward()
    assert abs(loss.item() - loss_ref.item()) < 1e-4
    worst = max(float((p.grad.cpu() - sd_ref[k].grad).norm() / (sd_ref[k].grad.norm() + 1e-12)) for k, p in m.named_parameters())
    assert worst < 1e-2, worst


def test_state_dict_roundtrip_and_eval_fast_path():
    """load_state_dict(strict) from reference-shaped tensors, eval() without deep heads, determinism."""
    sd = synth.fill_state_dict(unet.equiunet_state_shapes(8))
    m = _model(8, sd, "fp32").eval()
    for k, v in m.state_dict().items():
        torch.testing.assert_close(v.cpu(), sd[k])
    x = synth.random_image(1, 4, (16, 16, 16)).cuda()
    with torch.no_grad():
        a, deeps = m(x)
        m.skip_deep_heads_in_eval = True
        b, none = m(x)
    assert len(deeps) == 4 and len(none) == 0
    assert torch.equal(a, b)  # bitwise reproducible (no float atomics on the logits path)


def test_full_size_step_is_deterministic_and_learns():
    """BASELINE configs[1] at full size (EquiUnet-48, 2 x 4 x 128^3, bf16): two runs of the same 4 steps are BITWISE
    identical (every reduction on the path -- conv tile statistics, norm backward, Dice sums, head gradients, split-K
    weight gradients -- adds per-block partial sums in a fixed order; there is no float atomic on the EquiUnet path),
    every value is finite, and the fused Dice loss goes down on a fixed batch."""
    import contextlib
    import io
    from brats21_amd import get_model, synth as gsynth
    from brats21_amd.engine import TrainStep
    from brats21_amd.optim import Ranger2020
    dev = torch.device("cuda:0")
    ns = argparse.Namespace(model="equiunet", width=48, norm="group", act="relu", num_classes=3, dropout=0)
    x = gsynth.random_image(2, 4, (128, 128, 128), seed=5, device=dev)
    t = gsynth.nested_spheres(2, (128, 128, 128), device=dev)
    runs = []
    for _ in range(2):
        torch.manual_seed(0)
        with contextlib.redirect_stdout(io.StringIO()):
            m = get_model(ns).to(dev).train()
            opt = Ranger2020(m.parameters(), lr=3e-3, weight_decay=1e-5, use_gc=False)
        step = TrainStep(m, opt, amp=True)
        losses = [float(step(x, t).detach()) for _ in range(4)]
        runs.append((losses, torch.cat([p.detach().flatten()[:1000] for p in m.parameters()]).clone()))
        del m, opt, step
        torch.cuda.empty_cache()
    (l0, p0), (l1, p1) = runs
    assert all(np.isfinite(l0)) and bool(torch.isfinite(p0).all())
    assert l0 == l1, (l0, l1)
    assert torch.equal(p0, p1)
    assert l0[-1] < l0[0] - 1e-3, l0


def test_instance_norm_bf16_width48_vs_oracle():
    """--norm instance at the flagship width in bf16: logits against the f32 oracle on the same weights, bounded like
    the GroupNorm path (bf16 storage), and gradients finite."""
    torch.manual_seed(1)
    from brats21_amd import get_model
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        m = get_model(argparse.Namespace(model="equiunet", width=48, norm="instance", act="relu", num_classes=3, dropout=0)).cuda().train()
    sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    x = synth.random_image(1, 4, (32, 32, 32), seed=3)
    with torch.no_grad():
        ref = unet.equiunet_forward(sd, x, norm="instance")[0]
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out, deeps = m(x.cuda())
    got = out.float().cpu()
    dev = (got - ref).abs()
    scale = float(ref.abs().max())
    # bf16 storage of every activation: the GroupNorm path shows the same ~0.4 worst-case logit deviation (smoke())
    assert float(dev.max()) < 0.15 * scale + 0.3 and float(dev.mean()) < 0.03 * scale + 0.03, (float(dev.max()), float(dev.mean()), scale)
    corr = float(torch.corrcoef(torch.stack([got.flatten(), ref.flatten()]))[0, 1])
    assert corr > 0.995, corr
    (out.float().mean() + sum(d.float().mean() for d in deeps)).backward()
    assert all(bool(torch.isfinite(p.grad).all()) for p in m.parameters())


@pytest.mark.parametrize("act", ["elu", "swish", "mish"])
def test_other_activations_f32_network_vs_oracle(golden_dir, act):
    """--act elu | swish | mish end to end in the exact-f32 mode: logits, loss and gradients against the oracle (and,
    for elu, against the reference's own golden vectors; MONAI's Swish / Mish are restated in the oracle)."""
    from brats21_amd import get_model
    sd = synth.fill_state_dict(unet.equiunet_state_shapes(8))
    m = get_model(argparse.Namespace(model="equiunet", width=8, norm="instance", act=act, num_classes=3, dropout=0))
    m.load_state_dict(sd)
    m.precision = "fp32"
    m = m.cuda().train()
    size = (16, 16, 16)
    x, t = synth.closed_form_image(1, 4, size), synth.nested_spheres(1, size)
    out, deeps = m(x.cuda())
    loss = unet.deep_supervision_loss((out, deeps), t.cuda())
    loss.backward()
    sd_ref = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    out_ref = unet.equiunet_forward(sd_ref, x, act=act, norm="instance")
    loss_ref = unet.deep_supervision_loss(out_ref, t)
    loss_ref.backward()
    assert float((out.detach().cpu() - out_ref[0].detach()).abs().max()) < LOGIT_ATOL
    assert abs(loss.item() - loss_ref.item()) < 1e-4
    worst = max(float((p.grad.cpu() - sd_ref[k].grad).norm() / (sd_ref[k].grad.norm() + 1e-12)) for k, p in m.named_parameters())
    assert worst < 5e-3, worst
    if act == "elu":
        g = _golden(golden_dir, "equiunet_w8_16_elu.npz")
        assert np.abs(out.detach().cpu().numpy() - g["logits"]).max() < LOGIT_ATOL
        assert abs(loss.item() - float(g["loss"])) < 1e-4


def test_dropout_kernel_statistics_and_mask_reuse():
    """brats_dropout: keep rate 1 - p, survivors scaled by 1 / (1 - p), the mask a function of (seed, step, unit, element) only --
    the same for f32 and 16-bit storage, for a channel-slice view of a wider buffer, and for the gradient in the backward pass."""
    from brats21_amd import ops
    dev = torch.device("cuda:0")
    p = 0.3
    state = torch.tensor([123456789, 7], dtype=torch.int64, device=dev)
    x = torch.ones(2, 8, 16, 16, 24, device=dev)
    y = ops.dropout(x, p, state, 5)
    keep = (y != 0)
    rate = float(keep.float().mean())
    assert abs(rate - (1 - p)) < 5e-3, rate
    assert torch.equal(y[keep], torch.full_like(y[keep], 1.0 / (1 - p)))
    assert float(keep.float().mean((0, 1, 2, 3)).min()) > 0.65 and float(keep.float().mean(-1).min()) >= 0.25  # no dead channel / voxel pattern
    assert torch.equal(ops.dropout(x, p, state, 5), y)                          # deterministic
    assert not torch.equal(ops.dropout(x, p, state, 6) != 0, keep)              # another unit, another mask
    state2 = state.clone(); state2[1] += 1
    assert not torch.equal(ops.dropout(x, p, state2, 5) != 0, keep)             # another step, another mask
    for dt in (torch.bfloat16, torch.float16):
        assert torch.equal(ops.dropout(x.to(dt), p, state, 5) != 0, keep)       # the same mask in 16-bit storage
    wide = torch.ones(2, 8, 16, 16, 48, device=dev)
    assert torch.equal(ops.dropout(wide[..., 24:], p, state, 5) != 0, keep)     # ... and through a channel-slice view (pitch 48)
    g = torch.randn_like(x)
    assert torch.equal(ops.dropout(g, p, state, 5), g * y)                      # backward = the same multiplier on the gradient
    z = x.clone()
    assert ops.dropout(z, p, state, 5, out=z) is z and torch.equal(z, y)        # in place
    assert torch.equal(ops.dropout(g, 0.0, state, 5), g)                        # p = 0: identity


@pytest.mark.parametrize("norm", ["group", "batch"])
def test_dropout_network_matches_oracle_given_the_masks(norm):
    """--dropout p > 0 (nn.Dropout behind every ConvBnRelu's activation, networks/equiunet2020.py:62): the masks of a training
    forward are re-drawn from the model's (seed, step) state with ops.dropout on ones and handed to the CPU oracle -- logits,
    loss and every parameter gradient then agree at the f32 bars (the masks are shared by forward and backward, and every fold
    that skips a materialised activation is correctly switched off).  p = 0 is bit-identical to a model built without dropout;
    eval mode ignores p; two training forwards draw different masks."""
    from brats21_amd import get_model, ops
    dev = torch.device("cuda:0")
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    p = 0.25
    size = (16, 16, 16)

    def make(dropout):
        torch.manual_seed(0)
        with contextlib.redirect_stdout(io.StringIO()):
            m = get_model(argparse.Namespace(model="equiunet", width=16, norm=norm, act="relu", num_classes=3, dropout=dropout))
        m.precision = "fp32"
        return m.to(dev)

    m = make(p).train()
    sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    x, t = synth.random_image(2, 4, size, seed=3), synth.nested_spheres(2, size)
    out, deeps = m(x.to(dev))
    loss = unet.deep_supervision_loss((out, deeps), t.to(dev))
    loss.backward()
    state = m._dropout_state.clone()  # what that forward used
    assert int(state[1]) == 1
    drop = {}
    widths = {k[:-len(".conv.weight")]: v.shape[0] for k, v in sd.items() if k.endswith(".conv.weight")}
    level = {"encoder1": 1, "encoder2": 2, "encoder3": 4, "encoder4": 8, "bottom": 8, "bottom_2": 8, "decoder3": 4, "decoder2": 2, "decoder1": 1}
    names = {u: k for k, u in m.named_modules() if u in m._unit_ids}
    for unit, uid in m._unit_ids.items():
        pre = names[unit]
        s3 = tuple(s // level[pre.split(".")[0]] for s in size)
        ones = torch.ones(2, *s3, widths[pre], device=dev)
        drop[pre] = ops.dropout(ones, p, state, uid).permute(0, 4, 1, 2, 3).contiguous().cpu()
        assert abs(float((drop[pre] != 0).float().mean()) - (1 - p)) < 0.05
    sd_ref = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else v.clone()) for k, v in sd.items()}
    out_ref = unet.equiunet_forward(sd_ref, x, norm=norm, training=True, new_stats={}, drop=drop)
    loss_ref = unet.deep_supervision_loss(out_ref, t)
    loss_ref.backward()
    err = float((out.detach().cpu() - out_ref[0].detach()).abs().max())
    worst = max(float((q.grad.cpu() - sd_ref[k].grad).norm() / (sd_ref[k].grad.norm() + 1e-12)) for k, q in m.named_parameters())
    print(f"\n--dropout {p} --norm {norm}: logits max abs err vs the oracle given the masks {err:.2e}, loss {loss.item():.6f} vs "
          f"{loss_ref.item():.6f}, worst gradient rel err {worst:.2e}")
    assert err < LOGIT_ATOL and abs(loss.item() - loss_ref.item()) < 1e-4 and worst < 5e-3
    # a second training forward draws new masks; eval ignores p
    with torch.no_grad():
        out2 = m(x.to(dev))[0]
        assert int(m._dropout_state[1]) == 2 and not torch.equal(out2, out.detach())
        m.eval()
        e1, e2 = m(x.to(dev))[0], m(x.to(dev))[0]
        assert torch.equal(e1, e2) and int(m._dropout_state[1]) == 2
    if norm == "group":
        m0, mz = make(0.0).train(), make(0).train()
        mz.load_state_dict(m0.state_dict())
        a, b = m0(x.to(dev)), mz(x.to(dev))
        assert torch.equal(a[0], b[0]) and all(torch.equal(u, v) for u, v in zip(a[1], b[1]))


@pytest.mark.parametrize("precision", ["fp32", "x3"])
def test_bcn_network_vs_reference_golden(golden_dir, precision):
    """--norm bcn (the reference's BCNorm with EstBN, networks/factory.py:125-176,189-190; a choice of the unchanged CLI,
    src/arguments_train.py:48): state-dict keys in the reference's order, logits / deep heads / loss / every parameter gradient
    of a training step against the reference's own outputs (exact-f32 and split-precision modes); eval mode gives the same logits
    (EstBN reads its running buffers in both modes); then a bf16 step with the fused optimizer moves all five parameter kinds."""
    import functools
    from brats21_amd import get_model
    from brats21_amd.engine import TrainStep
    from brats21_amd.optim import Ranger2020
    g = _golden(golden_dir, "equiunet_w8_16_bcn.npz")
    sd = synth.fill_state_dict(functools.partial(unet.equiunet_state_shapes, norm="bcn")(8))
    with contextlib.redirect_stdout(io.StringIO()):
        m = get_model(argparse.Namespace(model="equiunet", width=8, norm="bcn", act="relu", num_classes=3, dropout=0))
    assert list(m.state_dict().keys()) == list(sd.keys()) == json.loads(str(g["meta"]))["keys"]
    m.load_state_dict(sd, strict=True)
    m.precision = precision
    m = m.cuda().train()
    size = (16, 16, 16)
    x, t = synth.closed_form_image(1, 4, size), synth.nested_spheres(1, size)
    out, deeps = m(x.cuda())
    loss = unet.deep_supervision_loss((out, deeps), t.cuda())
    loss.backward()
    err = np.abs(out.detach().cpu().numpy() - g["logits"]).max()
    derr = max(np.abs(d.detach().cpu().numpy()[:, :, ::2, ::2, ::2] - g[f"deep{i}"]).max() for i, d in enumerate(deeps))
    print(f"\n--norm bcn, {precision}: logits max abs err {err:.2e}, deep heads {derr:.2e}, loss {loss.item():.6f} vs {float(g['loss']):.6f}")
    assert err < LOGIT_ATOL and derr < LOGIT_ATOL and abs(loss.item() - float(g["loss"])) < 1e-4
    params = dict(m.named_parameters())
    names = json.loads(str(g["grad_names"]))
    norms = np.array([float(params[k].grad.double().norm()) for k in names])
    np.testing.assert_allclose(norms, g["grad_norms"], rtol=2e-3, atol=1e-7)
    for k in g.files:
        if k.startswith("grad:"):
            ref = g[k]
            got = params[k[5:]].grad.cpu().numpy()
            assert got.shape == ref.shape, k
            assert np.abs(got - ref).max() <= 5e-3 * np.abs(ref).max() + 1e-7, (k, np.abs(got - ref).max(), np.abs(ref).max())
    with torch.no_grad():
        out_eval = m.eval()(x.cuda())[0]
    assert float((out_eval - out.detach()).abs().max()) < 1e-5
    # inference folds EstBN into the convolution weights ONCE per unit: further windows neither re-fold nor add
    # packed-weight cache entries, and an in-place weight update is seen
    from brats21_amd import ops as _ops
    with torch.no_grad():
        entries = len(_ops._PACK_CACHE)
        folds = [id(u._bcn_fold[1][0]) for u in m.modules() if getattr(u, "_bcn_fold", None) is not None]
        assert len(folds) == 17
        again = m(x.cuda())[0]
        assert torch.equal(again, out_eval) and len(_ops._PACK_CACHE) == entries
        assert folds == [id(u._bcn_fold[1][0]) for u in m.modules() if getattr(u, "_bcn_fold", None) is not None]
        m.encoder1.ConvBnRelu1.bn.bn.weight.mul_(1.5)
        assert not torch.equal(m(x.cuda())[0], out_eval)
        m.encoder1.ConvBnRelu1.bn.bn.weight.div_(1.5)
    if precision == "fp32":
        m.train()
        m.precision = "auto"
        before = {k: p.detach().clone() for k, p in m.named_parameters()}
        with contextlib.redirect_stdout(io.StringIO()):
            opt = Ranger2020(m.parameters(), lr=1e-2)
        step = TrainStep(m, opt, criterion=None, amp=True)
        l0 = float(step(x.cuda(), t.cuda()))
        for _ in range(3):
            l1 = float(step(x.cuda(), t.cuda()))
        assert l1 < l0
        for kind in (".conv.weight", "decoder3.ConvBnRelu1.bn.weight", "decoder3.ConvBnRelu1.bn.bias", "decoder3.ConvBnRelu1.bn.bn.weight",
                     "decoder3.ConvBnRelu1.bn.bn.bias"):
            moved = [k for k, p in m.named_parameters() if k.endswith(kind) and not torch.equal(p.detach(), before[k])]
            assert moved, kind


def test_prelu_network_vs_reference_golden_and_oracle(golden_dir):
    """--act prelu (nn.PReLU per ConvBnRelu; reference networks/factory.py:195-200): state-dict keys, logits, loss and every
    gradient -- the 17 learnable slopes included -- against the reference's golden vectors (f32 mode) and the oracle; then
    a bf16 step with the fused optimizer moves the slopes."""
    import functools
    from brats21_amd import get_model
    from brats21_amd.engine import TrainStep
    from brats21_amd.optim import Ranger2020
    g = _golden(golden_dir, "equiunet_w8_16_prelu.npz")
    sd = synth.fill_state_dict(functools.partial(unet.equiunet_state_shapes, act="prelu")(8))
    with contextlib.redirect_stdout(io.StringIO()):
        m = get_model(argparse.Namespace(model="equiunet", width=8, norm="group", act="prelu", num_classes=3, dropout=0))
    assert list(m.state_dict().keys()) == list(sd.keys()) == json.loads(str(g["meta"]))["keys"]
    m.load_state_dict(sd)
    m.precision = "fp32"
    m = m.cuda().train()
    size = (16, 16, 16)
    x, t = synth.closed_form_image(1, 4, size), synth.nested_spheres(1, size)
    out, deeps = m(x.cuda())
    loss = unet.deep_supervision_loss((out, deeps), t.cuda())
    loss.backward()
    assert np.abs(out.detach().cpu().numpy() - g["logits"]).max() < LOGIT_ATOL
    assert abs(loss.item() - float(g["loss"])) < 1e-4
    params = dict(m.named_parameters())
    nslopes = 0
    for k in g.files:
        if k.startswith("grad:"):
            ref = g[k]
            got = params[k[5:]].grad.cpu().numpy()
            assert np.abs(got - ref).max() <= 5e-3 * np.abs(ref).max() + 1e-6, k
            nslopes += k.endswith(".prelu.weight")
    assert nslopes == 17
    names = json.loads(str(g["grad_names"]))
    norms = np.array([float(params[k].grad.double().norm()) for k in names])
    np.testing.assert_allclose(norms, g["grad_norms"], rtol=5e-3, atol=1e-9)
    # bf16 + fused optimizer: the slopes are ordinary parameters of the step
    m.precision = "auto"
    before = torch.stack([p.detach().clone().flatten()[0] for k, p in m.named_parameters() if k.endswith("prelu.weight")])
    with contextlib.redirect_stdout(io.StringIO()):
        opt = Ranger2020(m.parameters(), lr=1e-2)
    step = TrainStep(m, opt, amp=True)
    l0 = float(step(x.cuda(), t.cuda()).detach())
    for _ in range(3):
        l1 = float(step(x.cuda(), t.cuda()).detach())
    after = torch.stack([p.detach().flatten()[0] for k, p in m.named_parameters() if k.endswith("prelu.weight")])
    assert l1 < l0 and bool((after != before).all()) and bool(torch.isfinite(after).all())


@pytest.mark.parametrize("precision", ["bf16", "fp16"])
@pytest.mark.parametrize("act,norm", [("relu", "group"), ("leakyrelu", "instance")])
def test_backward_statistics_fold_leaves_the_step_unchanged(precision, act, norm):
    """model.fold_bwd_stats (round 4): the first pass of a block's first GroupNorm backward inside the input-gradient launch of
    the block's second convolution (brats_conv3d_fwd_bstats + brats_gn_act_bwd_tiles), EquiUnet-48: all eight blocks take the
    fused form, the forward is untouched, and the gradients equal the two-pass form's up to the rounding of dz (the fused
    sums see the f32 accumulators, the two-pass sums the stored 16-bit values)."""
    import argparse, contextlib, copy, io
    from brats21_amd import get_model, ops, synth
    from brats21_amd.losses import DiceLoss
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    ns = argparse.Namespace(model="equiunet", width=48, norm=norm, act=act, num_classes=3, dropout=0)
    with contextlib.redirect_stdout(io.StringIO()):
        base = get_model(ns).to(dev).train()
    base.precision = precision
    x = synth.random_image(2, 4, (32, 32, 32), seed=7, device=dev)
    t = synth.nested_spheres(2, (32, 32, 32), device=dev)
    crit = DiceLoss().to(dev)
    calls = []
    real = ops.conv3d_bstats

    def counted(*a, **k):
        calls.append(1)
        return real(*a, **k)

    def run(fold):
        model = copy.deepcopy(base)
        model.fold_bwd_stats = fold
        out, deep = model(x)
        loss = crit(out.float(), t) + sum(crit(d.float(), t) for d in deep)
        loss.backward()
        return [p.grad.detach().clone() if p.grad is not None else None for p in model.parameters()]

    ops.conv3d_bstats = counted
    try:
        g1 = run(True)
        n_fused = len(calls)
        g0 = run(False)
    finally:
        ops.conv3d_bstats = real
    assert n_fused == 8 and len(calls) == 8
    tol = 4e-2 if precision == "bf16" else 1e-2
    for (n, _), a, b in zip(base.named_parameters(), g1, g0):
        assert (a is None) == (b is None), n
        if a is None:
            continue
        assert bool(torch.isfinite(a).all()), n
        scale = float(b.abs().max()) + 1e-12
        assert float((a - b).abs().max()) <= tol * scale, (n, float((a - b).abs().max()) / scale)


@pytest.mark.parametrize("offset", [0.0, 0.03])
def test_backward_statistics_fold_is_no_worse_than_the_two_pass_form_against_the_f64_oracle(offset):
    """the fused sums (f32 accumulators) and the two-pass sums (stored 16-bit dz) are both judged against the
    oracle's float64 gradients -- not only against each other -- on EquiUnet-48 in fp16 storage (the forward, hence every ReLU
    mask, is bit-identical in both forms).  offset > 0 adds a constant to every 3x3x3 weight and feeds an image with a non-zero
    mean, so that each channel's raw convolution output has a mean several times its standard deviation: the regime where
    sum(u*y) - mean*sum(u)  cancels."""
    import copy
    from brats21_amd import get_model
    dev = torch.device("cuda:0")
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    torch.manual_seed(0)
    ns = argparse.Namespace(model="equiunet", width=48, norm="group", act="relu", num_classes=3, dropout=0)
    with contextlib.redirect_stdout(io.StringIO()):
        base = get_model(ns)
    sd = {k: (v + offset if (v.dim() == 5 and v.shape[2] == 3) else v).detach().clone() for k, v in base.state_dict().items()}
    base.load_state_dict(sd)
    base = base.to(dev).train()
    base.precision = "fp16"
    size = (32, 32, 32)
    x, t = synth.random_image(1, 4, size, seed=11), synth.nested_spheres(1, size)
    if offset:  # an image with a mean of several standard deviations (un-normalised intensities), not zeroed outside a mask
        x = 0.3 * torch.randn(1, 4, *size, generator=torch.Generator().manual_seed(11)) + 2.0
    sd64 = {k: (v.double().requires_grad_(True) if v.dtype.is_floating_point else v) for k, v in sd.items()}
    unet.deep_supervision_loss(unet.equiunet_forward(sd64, x.double()), t.double()).backward()
    if offset:  # the premise: |channel mean| >> channel std of the raw convolution outputs
        with torch.no_grad():
            y = torch.nn.functional.conv3d(x, sd["encoder1.ConvBnRelu1.conv.weight"], None, 1, 1)
        ratio = float((y.mean((0, 2, 3, 4)).abs() / y.std((0, 2, 3, 4))).median())
        print(f"\n|mean| / std of encoder1.ConvBnRelu1's raw output, median over channels: {ratio:.2f}")
        assert ratio > 3.0, ratio

    def run(fold):
        model = copy.deepcopy(base)
        model.fold_bwd_stats = fold
        out, deep = model(x.to(dev))
        unet.deep_supervision_loss((out.float(), [d.float() for d in deep]), t.to(dev)).backward()
        return {k: p.grad.detach().double().cpu() for k, p in model.named_parameters()}

    g_f, g_2 = run(True), run(False)
    rel = lambda g, k: float((g[k] - sd64[k].grad).norm() / (sd64[k].grad.norm() + 1e-30))  # noqa: E731
    e_f = {k: rel(g_f, k) for k in g_f}
    e_2 = {k: rel(g_2, k) for k in g_2}
    med = lambda d: sorted(d.values())[len(d) // 2]  # noqa: E731
    worst = max(e_f, key=lambda k: e_f[k] / (e_2[k] + 1e-6))
    print(f"\nfold_bwd_stats vs f64 oracle (offset {offset}): fused median {med(e_f):.3e} max {max(e_f.values()):.3e}; two-pass median "
          f"{med(e_2):.3e} max {max(e_2.values()):.3e}; worst ratio {e_f[worst] / (e_2[worst] + 1e-6):.2f} ({worst})")
    assert med(e_f) <= 1.2 * med(e_2) + 1e-4 and max(e_f.values()) <= 1.2 * max(e_2.values()) + 1e-3
    for k in e_f:
        assert e_f[k] <= 1.2 * e_2[k] + 0.25 * med(e_2) + 1e-4, (k, e_f[k], e_2[k])


@pytest.mark.parametrize("name", ["equiunet", "equiunet_assp_evo"])
@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_fold_forms_leave_the_step_unchanged(name, precision):
    """The round-3 fold forms (output head inside the last layer's normalisation passes, pooling backward inside the next
    normalisation backward: model.fold_head_fwd / fold_head_bwd / fold_pool_bwd) against the separate kernels they replace:
    the forward is bit-identical (logits of all heads); the gradients are equal up to the rounding of the tensors that are no
    longer stored (none in f32 but the summation order of two reductions; one bf16 rounding of two gradient tensors)."""
    import argparse, contextlib, copy, io
    from brats21_amd import get_model, synth
    from brats21_amd.losses import DiceLoss
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    ns = argparse.Namespace(model=name, width=16, norm="group", act="relu", num_classes=3, dropout=0)
    with contextlib.redirect_stdout(io.StringIO()):
        base = get_model(ns).to(dev).train()
    base.precision = precision
    x = synth.random_image(2, 4, (32, 32, 32), seed=7, device=dev)
    t = synth.nested_spheres(2, (32, 32, 32), device=dev)
    crit = DiceLoss().to(dev)

    def run(fold):
        model = copy.deepcopy(base)
        model.fold_head_fwd = model.fold_head_bwd = model.fold_pool_bwd = fold
        out, deep = model(x)
        loss = crit(out.float(), t) + sum(crit(d.float(), t) for d in deep)
        loss.backward()
        return ([out.detach()] + [d.detach() for d in deep],
                [p.grad.detach().clone() if p.grad is not None else None for p in model.parameters()])

    o1, g1 = run(True)
    o0, g0 = run(False)
    assert all(torch.equal(a, b) for a, b in zip(o1, o0))
    tol = 2e-4 if precision == "fp32" else 4e-2
    names = [n for n, _ in base.named_parameters()]
    for n, a, b in zip(names, g1, g0):
        assert (a is None) == (b is None), n
        if a is None:  # (a parameter the forward does not use)
            continue
        scale = float(b.abs().max()) + 1e-12
        assert float((a - b).abs().max()) <= tol * scale, (n, float((a - b).abs().max()) / scale)


@pytest.mark.parametrize("name", ["equiunet", "equiunet_assp_evo"])
def test_multi_tensor_weight_packing_is_transparent(name):
    """ops.PackPlan (one packing launch per training step) must not change a single bit: three optimizer steps with and
    without it, from the same initial weights; the plan must follow in-place updates (version counters) and survive
    deepcopy / load_state_dict of the module."""
    import argparse, contextlib, copy, io
    from brats21_amd import get_model, synth, ops
    from brats21_amd.losses import DiceLoss
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    ns = argparse.Namespace(model=name, width=16, norm="group", act="relu", num_classes=3, dropout=0)
    with contextlib.redirect_stdout(io.StringIO()):
        base = get_model(ns).to(dev).train()
    x = synth.random_image(2, 4, (32, 32, 32), seed=5, device=dev)
    t = synth.nested_spheres(2, (32, 32, 32), device=dev)
    crit = DiceLoss().to(dev)

    def run(plan):
        model = copy.deepcopy(base)
        model.pack_plan = plan
        opt = torch.optim.SGD(model.parameters(), lr=0.05)
        losses = []
        for _ in range(3):
            opt.zero_grad()
            with torch.autocast("cuda", dtype=torch.bfloat16):
                out, deep = model(x)
            loss = crit(out.float(), t) + sum(crit(d.float(), t) for d in deep)
            loss.backward()
            opt.step()
            losses.append(float(loss.detach()))
        return losses, [p.detach().clone() for p in model.parameters()], model

    l0, p0, _ = run(False)
    l1, p1, m1 = run(True)
    assert l0 == l1
    assert all(torch.equal(a, b) for a, b in zip(p0, p1))
    plan = ops._PLANS.get(m1)
    assert plan is not None and plan.entries and all(e[4] >= 0 for e in plan.entries.values())  # the plan was really used
    # load_state_dict copies in place (same storage, new version): the next step must see the new weights
    m1.load_state_dict(base.state_dict())
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out1, _ = m1(x)
        base.pack_plan = False
        out0, _ = base(x)
    assert torch.equal(out0, out1)


@pytest.mark.parametrize("width", [32, 64])
def test_equiunet_other_widths_f32_vs_oracle(width):
    """Widths other than the flagship 48 take other channel chunks / tile roles (32-channel chunks, 2-fragment cout tiles,
    K-split in f32; BASELINE.json configs[4] names width 64): logits, loss and gradients against the oracle in the
    exact-f32 mode, and the bf16 mode (y-split / cout-split roles) bounded against it."""
    from brats21_amd import get_model
    import contextlib
    import io
    torch.manual_seed(2)
    with contextlib.redirect_stdout(io.StringIO()):
        m = get_model(argparse.Namespace(model="equiunet", width=width, norm="group", act="relu", num_classes=3, dropout=0)).cuda().train()
    sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    size = (16, 16, 32)
    x, t = synth.random_image(1, 4, size, seed=4), synth.nested_spheres(1, size)
    m.precision = "fp32"
    out, deeps = m(x.cuda())
    loss = unet.deep_supervision_loss((out, deeps), t.cuda())
    loss.backward()
    sd_ref = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    out_ref = unet.equiunet_forward(sd_ref, x)
    loss_ref = unet.deep_supervision_loss(out_ref, t)
    loss_ref.backward()
    assert float((out.detach().cpu() - out_ref[0].detach()).abs().max()) < LOGIT_ATOL
    assert abs(loss.item() - loss_ref.item()) < 1e-4
    worst = max(float((p.grad.cpu() - sd_ref[k].grad).norm() / (sd_ref[k].grad.norm() + 1e-12)) for k, p in m.named_parameters())
    assert worst < 5e-3, worst
    m.precision = "bf16"
    m.zero_grad()
    out16, deeps16 = m(x.cuda())
    dev = (out16.float().cpu() - out_ref[0].detach()).abs()
    scale = float(out_ref[0].detach().abs().max())
    assert float(dev.max()) < 0.15 * scale + 0.3, (float(dev.max()), scale)
    unet.deep_supervision_loss((out16.float(), [d.float() for d in deeps16]), t.cuda()).backward()
    g16 = torch.cat([p.grad.flatten().cpu() for p in m.parameters()])
    gref = torch.cat([sd_ref[k].grad.flatten() for k, _ in m.named_parameters()])
    assert float(torch.nn.functional.cosine_similarity(g16, gref, dim=0)) > 0.98


@pytest.mark.parametrize("name", ["equiunet", "equiunet_assp_evo"])
def test_loss_from_deep_heads_only_with_fused_top(name):
    """A loss that does not touch the main logits: autograd hands the network program None for them.  With the output head
    folded into the last layer's passes (top_fused: up1 is never stored) the backward used to fall through to code that
    needs up1; it now runs the fold with zero logit gradients -- same result as the unfused program."""
    import warnings
    from brats21_amd import get_model
    grads = {}
    for fold in ("1", "0"):
        os.environ["BRATS_FOLD_HEAD"] = fold
        os.environ["BRATS_FOLD_HEAD_FWD"] = fold
        try:
            torch.manual_seed(0)
            with contextlib.redirect_stdout(io.StringIO()), warnings.catch_warnings():
                warnings.simplefilter("ignore")
                m = get_model(argparse.Namespace(model=name, width=8 if name == "equiunet" else 16, norm="group", act="relu", num_classes=3,
                                                 dropout=0)).cuda().train()
        finally:
            os.environ.pop("BRATS_FOLD_HEAD", None)
            os.environ.pop("BRATS_FOLD_HEAD_FWD", None)
        m.precision = "fp32"
        x = synth.random_image(1, 4, (16, 16, 16), seed=3).cuda()
        out, deeps = m(x)
        (deeps[0].square().mean() + deeps[-1].square().mean()).backward()
        grads[fold] = {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}
    main = "outconv.weight" if name == "equiunet" else "out_conv.weight"
    assert main in grads["1"] and float(grads["1"][main].abs().max()) == 0.0
    for k, g in grads["0"].items():
        if k in grads["1"]:
            torch.testing.assert_close(grads["1"][k], g, atol=1e-6, rtol=1e-4)


@pytest.mark.parametrize("act", ["relu", "leakyrelu"])
@pytest.mark.parametrize("prec", ["bf16", "fp16"])
@pytest.mark.parametrize("size,n", [((32, 32, 32), 2), ((40, 24, 16), 1)])
def test_normalise_on_load_inference_is_bit_identical(act, prec, size, n):
    """no_grad inference with the activation between the two convolutions of every block applied ON LOAD by the second one
    (ops.Pending / brats_conv3d_fwd_pre: z is never stored) against the two-pass path (affine_act, then the plain kernel):
    the staged values are computed by the same f32 operations and rounded once, so the logits must be IDENTICAL -- on a
    ragged volume too (out-of-volume voxels must enter as zeros, not as act(shift)), with and without the deep heads."""
    import warnings
    from brats21_amd import get_model, ops
    torch.manual_seed(0)
    with contextlib.redirect_stdout(io.StringIO()), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = get_model(argparse.Namespace(model="equiunet", width=48, norm="group", act=act, num_classes=3, dropout=0)).cuda().eval()
    g = torch.Generator().manual_seed(4)
    with torch.no_grad():
        for p in m.parameters():  # (negative GroupNorm weights too: the affine map is not monotone)
            if p.dim() == 1:
                p.add_(0.5 * torch.randn(p.shape, generator=g).cuda())
    m.precision = prec
    x = synth.random_image(n, 4, size, seed=9).cuda()
    outs = {}
    with torch.no_grad():
        for skip in (False, True):
            m.skip_deep_heads_in_eval = skip
            for on in (True, False):
                m.norm_on_load = on
                o = m(x)
                outs[(skip, on)] = [t.clone() for t in ([o[0]] + list(o[1]) if isinstance(o, tuple) else [o])]
    for skip in (False, True):
        for a, b in zip(outs[(skip, True)], outs[(skip, False)]):
            assert torch.equal(a, b)
    assert torch.equal(outs[(True, True)][0], outs[(False, True)][0])
    # and the fused kernels really ran: one affine_act launch less per block
    calls = {"n": 0}
    orig = ops.affine_act

    def counting(*a, **k):
        calls["n"] += 1
        return orig(*a, **k)
    ops.affine_act = counting
    try:
        with torch.no_grad():
            m.norm_on_load = True
            m(x)
            n_on = calls["n"]
            m.norm_on_load = False
            m(x)
            n_off = calls["n"] - n_on
    finally:
        ops.affine_act = orig
        m.norm_on_load = True
    assert n_off - n_on >= 8, (n_on, n_off)


@pytest.mark.parametrize("precision", ["fp32", "x3"])
def test_equiunet_batch_norm_matches_reference_golden(golden_dir, precision):
    """--norm batch (nn.BatchNorm3d(affine=True), networks/factory.py:185-186) against the reference's own outputs
    (tests/golden/equiunet_w8_16_batchnorm.npz): a training-mode step on TWO patches -- logits, deep heads, loss, per-parameter
    gradients, the running buffers after the step -- then the eval-mode forward on the updated buffers.  The HIP path is the
    GroupNorm kernels on the batch viewed as one sample (brats21_amd/networks/equiunet.py: ConvBnRelu)."""
    g = _golden(golden_dir, "equiunet_w8_16_batchnorm.npz")
    meta = json.loads(str(g["meta"]))
    size = tuple(meta["size"])
    sd = synth.fill_state_dict(unet.equiunet_state_shapes(meta["width"], norm="batch"))
    m = _model(meta["width"], sd, precision, norm="batch").train()
    assert list(m.state_dict().keys()) == meta["keys"]
    x = synth.closed_form_image(meta["batch"], 4, size).cuda()
    t = synth.nested_spheres(meta["batch"], size).cuda()
    out, deeps = m(x)
    err = np.abs(out.detach().cpu().numpy() - g["logits"]).max()
    assert err < LOGIT_ATOL, f"logit max abs err {err}"
    for i, d in enumerate(deeps):
        assert np.abs(d.detach().cpu().numpy()[:, :, ::2, ::2, ::2] - g[f"deep{i}"]).max() < LOGIT_ATOL
    loss = unet.deep_supervision_loss((out, deeps), t)
    assert abs(loss.item() - float(g["loss"])) < 1e-4
    loss.backward()
    names = json.loads(str(g["grad_names"]))
    params = dict(m.named_parameters())
    norms = np.array([float(params[k].grad.double().norm()) for k in names])
    # the reference's f32 CPU gradients are themselves ~2.5e-3 off the float64 truth on this tie-rich closed-form volume (measured:
    # 17 of 61 norms 0.23-0.39 % low); the HIP path is judged against BOTH: 1e-2 of the golden, 1e-3 of the f64 oracle per parameter
    np.testing.assert_allclose(norms, g["grad_norms"], rtol=1e-2, atol=1e-7)
    sd64 = {k: (v.clone().double().requires_grad_(True) if (v.dtype.is_floating_point and "running" not in k) else v) for k, v in sd.items()}
    unet.deep_supervision_loss(unet.equiunet_forward(sd64, x.cpu().double(), norm="batch"), t.cpu().double()).backward()
    worst = max(float((params[k].grad.cpu().double() - sd64[k].grad).norm() / sd64[k].grad.norm()) for k in names)
    assert worst < (5e-3 if precision == "x3" else 1e-3), worst
    bufs = dict(m.named_buffers())
    for k in g.files:
        if k.startswith("buf:"):
            np.testing.assert_allclose(bufs[k[4:]].cpu().numpy(), g[k], atol=2e-6, rtol=2e-5)
    m.eval()
    with torch.no_grad():
        ev = m(x)[0]
    e2 = np.abs(ev.cpu().numpy() - g["eval_logits"]).max()
    assert e2 < LOGIT_ATOL, e2
    print(f"\n--norm batch ({precision}): train logits err {err:.2e}, eval logits err {e2:.2e}, worst gradient rel err vs f64 {worst:.2e}")
    # bf16 storage runs the same path (a width-8 network on 2 x 16^3 amplifies 16-bit rounding through its batch statistics:
    # bounded loosely, the parity claims are the f32 / x3 ones above)
    m.train()
    m.precision = "bf16"
    ob = m(x)[0]
    assert torch.isfinite(ob).all() and float((ob.detach() - out.detach()).abs().max()) < 1.0
