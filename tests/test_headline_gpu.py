"""-m gpu: parity where the headline number lives ("next round" item 1).

* EquiUnet width 48 (BASELINE.json configs[1]) at 1x4x64^3 and 1x4x128^3 against the CPU oracle (oracle/unet.py, the
  reference's CPU arithmetic, fp32) on the bench's own weights (the model's kaiming init, seed 0) and synthetic image:
  - f32 mode (exact-f32 MFMA kernels): logits and every deep head within the north-star bar, 1e-3 abs;
  - bf16 mode (the benchmarked dtype): max / mean / p99.9 deviation printed, hard Dice of sigmoid(logits) > 0.5 against
    the synthetic target within 1e-3 of the oracle's (the north-star bar "Dice within 1e-3 of the CPU reference").
* EquiUnetASSPEvo width 48 at 32^3: logits + per-parameter gradients against the oracle evaluated in float64
  (reference forward: networks/equiunet2021.py:289-333).
* BASELINE.json configs[2] / configs[4] exercised at size: one ASSP-48 2x4x128^3 bf16 training step and one ASSP-64
  conv_fp8="all" step -- finite, bitwise deterministic, loss falls over three steps.
"""
import argparse
import contextlib
import io
import os
import warnings

import pytest
import torch

from oracle import synth, unet

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")
LOGIT_ATOL = 1e-3   # north_star: "logits must match the reference PyTorch CPU path within 1e-3 abs"
DICE_ATOL = 1e-3    # north_star: "Dice within 1e-3 of the CPU reference on identical synthetic volumes"


def _get(model, width, seed=0, act="relu"):
    from brats21_amd import get_model
    torch.manual_seed(seed)
    ns = argparse.Namespace(model=model, width=width, norm="group", act=act, num_classes=3, dropout=0)
    with contextlib.redirect_stdout(io.StringIO()), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return get_model(ns)


def _cpu_threads():
    torch.set_num_threads(min(16, os.cpu_count() or 1))


@pytest.mark.parametrize("size", [64, 128])
def test_equiunet48_vs_oracle_f32_and_bf16(size):
    _cpu_threads()
    m = _get("equiunet", 48)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    m = m.to(DEV).eval()
    s3 = (size,) * 3
    x = synth.random_image(1, 4, s3, seed=1234)   # bench.py's image (rank 0)
    t = synth.nested_spheres(1, s3)
    with torch.no_grad():
        ref, ref_deeps = unet.equiunet_forward(sd, x)
        m.precision = "fp32"
        out, deeps = m(x.to(DEV))
        err = float((out.cpu() - ref).abs().max())
        derr = [float((d.cpu() - r).abs().max()) for d, r in zip(deeps, ref_deeps)]
        print(f"\nEquiUnet-48 @{size}^3 f32: logits max abs err {err:.3e} (|logits| max {float(ref.abs().max()):.2f}); deep heads {derr}")
        assert err < LOGIT_ATOL and max(derr) < LOGIT_ATOL, (err, derr)
        m.precision = "bf16"
        out_b, _ = m(x.to(DEV))
    dev = (out_b.cpu() - ref).abs().flatten()
    p999 = float(torch.quantile(dev[:: max(1, dev.numel() // 4_000_000)], 0.999))
    d_ref, d_b = unet.hard_dice(ref, t), unet.hard_dice(out_b.cpu(), t)
    flips = float(((out_b.cpu() > 0) != (ref > 0)).float().mean())
    print(f"EquiUnet-48 @{size}^3 bf16: max {float(dev.max()):.3e} mean {float(dev.mean()):.3e} p99.9 {p999:.3e}; "
          f"thresholded voxels that differ {flips:.3e}; hard Dice oracle {d_ref.flatten().tolist()} bf16 {d_b.flatten().tolist()}")
    # bf16 against the f32 oracle on |logits| <= 28: measured mean 1.9e-2, p99.9 0.13, max 0.23 at both sizes (deterministic
    # arithmetic: the same on every box) -- bars at ~1.5x, so that a 2x loss of bf16 accuracy fails ("weak" 3)
    assert float(dev.mean()) < 0.03 and p999 < 0.2 and float(dev.max()) < 0.35, (float(dev.mean()), p999, float(dev.max()))
    assert float((d_ref - d_b).abs().max()) <= DICE_ATOL, (d_ref, d_b)


def test_assp48_gradients_vs_f64_oracle():
    """Width-48 channel roles of EquiUnetASSPEvo (24 + 24 concat slices, 24-channel chunks, 96-channel ASPP branches)
    with per-parameter gradients; judged against the oracle in float64 (as test_equiunet_gpu does for EquiUnet)."""
    _cpu_threads()
    m = _get("equiunet_assp_evo", 48)
    g = torch.Generator().manual_seed(3)
    sd = {k: (v.detach().clone() + (0.02 * torch.randn(v.shape, generator=g) if v.dtype.is_floating_point and k.endswith(("gamma", "beta")) else 0))
          for k, v in m.state_dict().items()}
    m.load_state_dict(sd)
    m = m.to(DEV).train()
    m.precision = "fp32"
    size = (32, 32, 32)
    x = synth.random_image(1, 4, size, seed=5)
    t = synth.nested_spheres(1, size)
    sd_ref = {k: (v.clone().double().requires_grad_(True) if v.dtype.is_floating_point else v) for k, v in sd.items()}
    out_ref = unet.assp_evo_forward(sd_ref, x.double())
    loss_ref = unet.deep_supervision_loss(out_ref, t.double())
    loss_ref.backward()
    out, deeps = m(x.to(DEV))
    err = float((out.detach().cpu().double() - out_ref[0].detach()).abs().max())
    loss = unet.deep_supervision_loss((out, deeps), t.to(DEV))
    loss.backward()
    assert err < LOGIT_ATOL, err
    assert abs(loss.item() - loss_ref.item()) < 1e-4
    worst = ("", 0.0)
    for k, p in m.named_parameters():
        if k.endswith(".v"):
            assert p.grad is None  # statically unused
            continue
        ref = sd_ref[k].grad
        rel = float((p.grad.cpu().double() - ref).norm() / (ref.norm() + 1e-30))
        if rel > worst[1]:
            worst = (k, rel)
        assert rel < 5e-3, (k, rel)
    print(f"\nASSP-48 @32^3 f32: logits err {err:.2e}, worst gradient rel err {worst[1]:.2e} ({worst[0]})")
    # bf16: same yardstick as EquiUnet (torch's own CPU bf16 autocast of the oracle vs f64)
    m.zero_grad()
    m.precision = "auto"
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out_b, deeps_b = m(x.to(DEV))
        loss_b = unet.deep_supervision_loss((out_b, deeps_b), t.to(DEV))
    loss_b.backward()
    assert float((out_b.detach().cpu().double() - out_ref[0].detach()).abs().max()) > 1e-4  # (really the bf16 kernels)
    sd_b = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point else v) for k, v in sd.items()}
    with torch.autocast("cpu", dtype=torch.bfloat16):
        out_c = unet.assp_evo_forward(sd_b, x)
    unet.deep_supervision_loss(out_c, t).backward()
    e_hip, e_ref = [], []
    for k, p in m.named_parameters():
        if k.endswith(".v"):
            continue
        ref = sd_ref[k].grad
        e_hip.append(float((p.grad.cpu().double() - ref).norm() / (ref.norm() + 1e-30)))
        e_ref.append(float((sd_b[k].grad.double() - ref).norm() / (ref.norm() + 1e-30)))
    med = lambda v: sorted(v)[len(v) // 2]  # noqa: E731
    print(f"ASSP-48 @32^3 bf16 gradient rel err vs f64: median {med(e_hip):.3f} (torch CPU bf16 autocast: {med(e_ref):.3f}), "
          f"worst {max(e_hip):.3f} ({max(e_ref):.3f})")
    assert med(e_hip) <= 1.5 * med(e_ref) + 0.02 and max(e_hip) <= 1.5 * max(e_ref) + 0.05


@pytest.mark.parametrize("width,fp8", [(48, None), (64, "all")])
def test_assp_full_size_step_is_deterministic_and_learns(width, fp8):
    """BASELINE.json configs[2] (ASSP-48, 2 patches of 4x128^3 per GPU, bf16) and configs[4] (ASSP-64, e4m3 convolutions)."""
    from brats21_amd.engine import TrainStep
    from brats21_amd.optim import Ranger2020
    x = synth.random_image(2, 4, (128, 128, 128), seed=1234).to(DEV)
    t = synth.nested_spheres(2, (128, 128, 128)).to(DEV)
    runs = []
    for _ in range(2):
        m = _get("equiunet_assp_evo", width).to(DEV).train()
        m.conv_fp8 = fp8
        with contextlib.redirect_stdout(io.StringIO()):
            opt = Ranger2020(m.parameters(), lr=1e-3, use_gc=False)
        step = TrainStep(m, opt, amp=True)
        losses = [float(step(x, t).detach()) for _ in range(3)]
        assert all(torch.isfinite(p).all() for p in m.parameters())
        runs.append((losses, torch.cat([p.detach().flatten()[:300] for p in m.parameters()]).clone()))
        del m, opt, step
    print(f"\nASSP-{width} fp8={fp8} 2x4x128^3 losses {runs[0][0]}")
    assert runs[0][0] == runs[1][0] and torch.equal(runs[0][1], runs[1][1])
    assert runs[0][0][2] < runs[0][0][0]


def test_equiunet48_fp16_storage_vs_oracle():
    """model.precision = "fp16" (IEEE half storage: the reference's own autocast dtype, learning/engine.py:304) at 1x4x64^3
    with the bench's weights and image: deviation from the f32 CPU oracle printed beside bf16's (fp16 carries three more
    mantissa bits: expected ~8x closer), hard Dice within the 1e-3 north-star bar, and the torch.autocast(float16) switch
    picks the same kernels."""
    _cpu_threads()
    m = _get("equiunet", 48)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    m = m.to(DEV).eval()
    s3 = (64, 64, 64)
    x = synth.random_image(1, 4, s3, seed=1234)
    t = synth.nested_spheres(1, s3)
    with torch.no_grad():
        ref, _ = unet.equiunet_forward(sd, x)
        outs = {}
        for prec in ("bf16", "fp16"):
            m.precision = prec
            outs[prec] = m(x.to(DEV))[0].float().cpu()
        m.precision = "auto"
        with torch.autocast("cuda", dtype=torch.float16):
            auto16 = m(x.to(DEV))[0].float().cpu()
    assert torch.equal(auto16, outs["fp16"])
    dev = {k: (v - ref).abs() for k, v in outs.items()}
    d_ref = unet.hard_dice(ref, t)
    d16 = unet.hard_dice(outs["fp16"], t)
    flips = float(((outs["fp16"] > 0) != (ref > 0)).float().mean())
    print(f"\nEquiUnet-48 @64^3 vs f32 oracle: fp16 max {float(dev['fp16'].max()):.3e} mean {float(dev['fp16'].mean()):.3e} | "
          f"bf16 max {float(dev['bf16'].max()):.3e} mean {float(dev['bf16'].mean()):.3e}; fp16 thresholded voxels that differ {flips:.3e}; "
          f"hard Dice oracle {d_ref.flatten().tolist()} fp16 {d16.flatten().tolist()}")
    assert torch.isfinite(outs["fp16"]).all()
    assert float(dev["fp16"].mean()) < 0.25 * float(dev["bf16"].mean())   # three more mantissa bits
    assert float(dev["fp16"].mean()) < 0.01 and float(dev["fp16"].max()) < 0.25
    assert float((d_ref - d16).abs().max()) <= DICE_ATOL


def test_fp16_gradscaler_training_loop():
    """The reference's AMP loop (autocast fp16 + GradScaler, learning/engine.py:117-122) through TrainStep(amp_dtype=float16):
    EquiUnet width 8 at 32^3; the first step's unscaled gradients against the oracle's f32 gradients, the scaler stays
    finite, the loss falls, and a forced overflow (huge loss scale) skips the step and halves the scale like the reference."""
    from brats21_amd.engine import TrainStep
    from brats21_amd.optim import Ranger2020
    _cpu_threads()
    m = _get("equiunet", 8)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    m = m.to(DEV).train()
    size = (32, 32, 32)
    x, t = synth.random_image(2, 4, size, seed=7), synth.nested_spheres(2, size)
    sd_ref = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    unet.deep_supervision_loss(unet.equiunet_forward(sd_ref, x), t).backward()
    with contextlib.redirect_stdout(io.StringIO()):
        opt = Ranger2020(m.parameters(), lr=1e-3, use_gc=False)
    step = TrainStep(m, opt, amp=True, amp_dtype=torch.float16)
    assert step.scaler is not None
    xd, td = x.to(DEV), t.to(DEV)
    # gradients of one scaled backward, unscaled by the scaler
    m.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.float16):
        loss0 = step.loss(m(xd), td)
    step.scaler.scale(loss0).backward()
    step.scaler.unscale_(opt)
    rel = []
    for k, p in m.named_parameters():
        g = sd_ref[k].grad
        rel.append(float((p.grad.cpu() - g).norm() / (g.norm() + 1e-12)))
    print(f"\nfp16 + GradScaler: per-parameter gradient rel err vs f32 oracle median {sorted(rel)[len(rel) // 2]:.3e} worst {max(rel):.3e}")
    assert sorted(rel)[len(rel) // 2] < 0.02 and max(rel) < 0.2
    step.scaler.step(opt)
    step.scaler.update()
    losses = [float(step(xd, td).detach()) for _ in range(6)]
    assert all(torch.isfinite(p).all() for p in m.parameters())
    assert losses[-1] < float(loss0.detach())
    # overflow: a loss scale of 2^100 puts inf into the fp16 gradients -> the step is skipped, the scale halves
    before = [p.detach().clone() for p in m.parameters()]
    step.scaler = torch.amp.GradScaler("cuda", init_scale=2.0 ** 100)
    step(xd, td)
    assert all(torch.equal(a, b.detach()) for a, b in zip(before, m.parameters()))
    assert step.scaler.get_scale() == 2.0 ** 99


def test_config4_assp64_fp16_fp8_four_patches():
    """BASELINE.json configs[4] as stated: EquiUnetASSPEvo width 64, fp16 storage + e4m3 MFMA convolutions (conv_fp8 = "all"),
    4 patches of 4x128^3 per GPU, under the reference's GradScaler loop: finite, bitwise deterministic, the loss falls."""
    from brats21_amd.engine import TrainStep
    from brats21_amd.optim import Ranger2020
    x = synth.random_image(4, 4, (128, 128, 128), seed=1234).to(DEV)
    t = synth.nested_spheres(4, (128, 128, 128)).to(DEV)
    runs = []
    for _ in range(2):
        m = _get("equiunet_assp_evo", 64).to(DEV).train()
        m.conv_fp8 = "all"
        with contextlib.redirect_stdout(io.StringIO()):
            opt = Ranger2020(m.parameters(), lr=1e-3, use_gc=False)
        step = TrainStep(m, opt, amp=True, amp_dtype=torch.float16)
        losses = [float(step(x, t).detach()) for _ in range(3)]
        assert all(torch.isfinite(p).all() for p in m.parameters())
        runs.append((losses, torch.cat([p.detach().flatten()[:300] for p in m.parameters()]).clone(), step.scaler.get_scale()))
        del m, opt, step
        torch.cuda.empty_cache()
    print(f"\nASSP-64 fp16 + e4m3, 4x4x128^3: losses {runs[0][0]}, loss scale {runs[0][2]}")
    assert runs[0][0] == runs[1][0] and torch.equal(runs[0][1], runs[1][1])
    assert runs[0][0][2] < runs[0][0][0]


def test_assp48_full_size_patch_vs_oracle():
    """BASELINE.json configs[2]'s model at its patch size: EquiUnetASSPEvo width 48 on 1x4x128^3 (torch-default init, the
    bench's image) against the CPU oracle (oracle/unet.py: networks/equiunet2021.py:289-333 incl. the MONAI stubs) --
    f32 mode: logits and both deep heads within 1e-3; bf16 / fp16 modes: deviation printed, hard Dice within 1e-3."""
    _cpu_threads()
    m = _get("equiunet_assp_evo", 48)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    m = m.to(DEV).eval()
    s3 = (128, 128, 128)
    x = synth.random_image(1, 4, s3, seed=1234)
    t = synth.nested_spheres(1, s3)
    with torch.no_grad():
        ref, ref_deeps = unet.assp_evo_forward(sd, x)
        m.precision = "fp32"
        out, deeps = m(x.to(DEV))
        err = float((out.cpu() - ref).abs().max())
        derr = [float((d.cpu() - r).abs().max()) for d, r in zip(deeps, ref_deeps)]
        print(f"\nASSP-48 @128^3 f32: logits max abs err {err:.3e} (|logits| max {float(ref.abs().max()):.2f}); deep heads {derr}")
        assert err < LOGIT_ATOL and max(derr) < LOGIT_ATOL, (err, derr)
        d_ref = unet.hard_dice(ref, t)
        for prec in ("bf16", "fp16"):
            m.precision = prec
            o = m(x.to(DEV))[0].float().cpu()
            dev = (o - ref).abs()
            d = unet.hard_dice(o, t)
            print(f"ASSP-48 @128^3 {prec}: max {float(dev.max()):.3e} mean {float(dev.mean()):.3e}; hard Dice oracle {d_ref.flatten().tolist()} "
                  f"{prec} {d.flatten().tolist()}")
            assert torch.isfinite(o).all()
            assert float((d_ref - d).abs().max()) <= DICE_ATOL, (prec, d_ref, d)
    m.precision = "auto"


@pytest.mark.parametrize("name,act", [("equiunet", "relu"), ("equiunet_assp_evo", "relu"), ("equiunet", "swish"), ("equiunet", "elu")])
def test_width48_fp16_gradients_vs_f64_oracle(name, act):
    """Per-parameter gradients of both width-48 networks in the fp16 storage mode (loss scaled by 2^14 before backward and
    unscaled afterwards, as the GradScaler does) against the oracle evaluated in float64, 1x4x32^3.  ABSOLUTE bars where the
    bf16 tests can only bound the error relative to torch's own CPU bf16 autocast (bf16: median 1-11 %, worst 17-21 %):
    EquiUnetASSPEvo (smooth EvoNorm / swish units) median < 1 %, worst < 10 % (measured 0.16 % / 4.8 %); EquiUnet
    (GroupNorm + ReLU: a pre-activation that changes sign under the 16-bit rounding flips its gradient mask, so the error
    is set by the mask flips, not by the arithmetic) median < 8 %, worst < 20 % (measured 5.5 % / 12 %).
    The CONTROL for that explanation (round 3 moved EquiUnet's bar from 1.5 % to 8 % on the strength of it): the same
    network, the same GroupNorm forward / backward kernels and fp16 storage, with a SMOOTH activation (--act swish / elu:
    no mask to flip) must reach the ASSP level -- median < 1 %, worst < 10 % -- or the GroupNorm path has an fp16 defect."""
    _cpu_threads()
    m = _get(name, 48, act=act)
    g = torch.Generator().manual_seed(11)
    sd = {k: (v.detach().clone() + (0.02 * torch.randn(v.shape, generator=g) if v.dtype.is_floating_point and k.endswith(("gamma", "beta", "bn.weight", "bn.bias")) else 0))
          for k, v in m.state_dict().items()}
    m.load_state_dict(sd)
    m = m.to(DEV).train()
    size = (32, 32, 32)
    x = synth.random_image(1, 4, size, seed=5)
    t = synth.nested_spheres(1, size)
    sd_ref = {k: (v.clone().double().requires_grad_(True) if v.dtype.is_floating_point else v) for k, v in sd.items()}
    fwd = (lambda sd_, x_: unet.equiunet_forward(sd_, x_, act=act)) if name == "equiunet" else unet.assp_evo_forward
    unet.deep_supervision_loss(fwd(sd_ref, x.double()), t.double()).backward()
    scale = 2.0 ** 14
    m.precision = "fp16"
    out, deeps = m(x.to(DEV))
    (unet.deep_supervision_loss((out, deeps), t.to(DEV)) * scale).backward()
    rel = []
    for k, p in m.named_parameters():
        if k.endswith(".v"):
            continue
        ref = sd_ref[k].grad
        gr = p.grad.cpu().double() / scale
        assert torch.isfinite(gr).all(), k
        rel.append((float((gr - ref).norm() / (ref.norm() + 1e-30)), k))
    rel.sort()
    med, worst = rel[len(rel) // 2][0], rel[-1]
    print(f"\n{name}-48 ({act}) @32^3 fp16 gradients vs f64 oracle: median rel err {med:.3e}, worst {worst[0]:.3e} ({worst[1]})")
    bars = (0.08, 0.20) if (name, act) == ("equiunet", "relu") else (0.01, 0.10)
    assert med < bars[0] and worst[0] < bars[1], (med, worst)
    m.precision = "auto"
