"""-m gpu: EvoNorm-S0 / SE / ASPP kernels and the EquiUnetASSPEvo network against the golden vectors made
from the reference source (under the MONAI stub: parity unpinned at the MONAI boundary, oracle/refshim.py)
and against the CPU oracle."""
import argparse
import json
import os
import warnings

import numpy as np
import pytest
import torch

from oracle import synth, unet

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _to_ndhwc(x, dtype=torch.float32):
    return x.permute(0, 2, 3, 4, 1).contiguous().to(DEV).to(dtype)


def _from_ndhwc(t):
    return t.float().cpu().permute(0, 4, 1, 2, 3).contiguous()


def _model(width, sd, precision="fp32"):
    from brats21_amd import get_model
    m = get_model(argparse.Namespace(model="equiunet_assp_evo", width=width, norm="group", act="relu", num_classes=3, dropout=0))
    m.load_state_dict(sd, strict=True)
    m.precision = precision
    return m.to(DEV)


def test_evonorm_fwd_bwd_matches_reference_golden(golden_dir):
    from brats21_amd import ops
    g = np.load(os.path.join(golden_dir, "ops.npz"))
    x = synth.closed_form_image(1, 16, (12, 12, 12), "opx")
    esd = synth.fill_state_dict({k: (1, 16, 1, 1, 1) for k in ("gamma", "beta", "v", "running_var")})
    gamma, beta = esd["gamma"].reshape(-1).to(DEV), esd["beta"].reshape(-1).to(DEV)
    # statistics through an identity 1x1x1 convolution (the production path: conv epilogue -> finalize)
    w = torch.eye(16).reshape(16, 16, 1, 1, 1).to(DEV)
    xd = _to_ndhwc(x)
    y, stats = ops.conv3d(xd, ops.pack_weights(w, torch.float32, ops.PACK_FWD), 16, 1, 1, want_stats=True)
    torch.testing.assert_close(_from_ndhwc(y), x, atol=1e-6, rtol=0)
    mr, chan = ops.evonorm_finalize(stats, 1, 16, 8, 12 ** 3)
    z, cs = ops.evonorm(y, mr, gamma, beta, 8, want_chansum=True)
    np.testing.assert_allclose(_from_ndhwc(z).numpy(), g["evo_y"], atol=2e-5)
    np.testing.assert_allclose(cs.cpu().numpy()[0], g["evo_y"].sum((0, 2, 3, 4)), rtol=1e-4, atol=1e-2)
    go = synth.closed_form("evo_go", (1, 16, 12, 12, 12))
    dx, dgamma, dbeta, dcb = ops.evonorm_bwd(_to_ndhwc(go), y, mr, gamma, 8, chan=chan)
    np.testing.assert_allclose(dcb.cpu().numpy(), g["evo_dx"].sum((0, 2, 3, 4)), rtol=1e-3, atol=2e-3)  # = conv bias gradient
    np.testing.assert_allclose(_from_ndhwc(dx).numpy(), g["evo_dx"], atol=2e-5)
    np.testing.assert_allclose(dgamma.cpu().numpy(), g["evo_dgamma"].ravel(), rtol=1e-4, atol=1e-3)
    np.testing.assert_allclose(dbeta.cpu().numpy(), g["evo_dbeta"].ravel(), rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_dilated_im2col_conv_and_1x1(dtype):
    """ASPP branches: dilation 4 / 6 via im2col + 1x1 implicit GEMM, and the native 1x1 kernel, vs F.conv3d."""
    from brats21_amd import ops
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 32, 8, 8, 8, generator=g).to(dtype).float()
    for k, dil in ((3, 4), (3, 6), (1, 1)):
        w = (torch.randn(16, 32, k, k, k, generator=g) * 0.05).to(dtype).float()
        b = torch.randn(16, generator=g) * 0.1
        xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
        y_ref = F.conv3d(xr, wr, b, 1, (k - 1) // 2 * dil, dil)
        dy = torch.randn(y_ref.shape, generator=g).to(dtype).float()
        y_ref.backward(dy)
        xd, dyd = _to_ndhwc(x, dtype), _to_ndhwc(dy, dtype)
        if k == 3:
            col = ops.im2col3(xd, dil)
            w1 = w.permute(0, 2, 3, 4, 1).reshape(16, 27 * 32, 1, 1, 1).to(DEV)
        else:
            col, w1 = xd, w.to(DEV)
        y, _ = ops.conv3d(col, ops.pack_weights(w1, dtype, ops.PACK_FWD), 16, 1, 1, bias=b.to(DEV))
        tol = dict(atol=3e-5, rtol=1e-5) if dtype == torch.float32 else dict(atol=4e-2, rtol=2e-2)
        torch.testing.assert_close(_from_ndhwc(y), y_ref.detach(), **tol)
        dcol, _ = ops.conv3d(dyd, ops.pack_weights(w1, dtype, ops.PACK_DGRAD), w1.shape[1], 1, 1)
        dx = ops.col2im3(dcol, 32, dil) if k == 3 else dcol
        tol = dict(atol=5e-5, rtol=1e-5) if dtype == torch.float32 else dict(atol=6e-2, rtol=3e-2)
        torch.testing.assert_close(_from_ndhwc(dx), xr.grad, **tol)
        dw = ops.wgrad_1x1(col, dyd).cpu()
        dw = dw.view(16, 3, 3, 3, 32).permute(0, 4, 1, 2, 3) if k == 3 else dw.view(16, 32, 1, 1, 1)
        tol = dict(atol=2e-4, rtol=1e-4) if dtype == torch.float32 else dict(atol=0.15, rtol=3e-2)
        torch.testing.assert_close(dw, wr.grad, **tol)


def test_assp_f32_matches_reference_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "assp_w16_32.npz"))
    meta = json.loads(str(g["meta"]))
    size = tuple(meta["size"])
    sd = synth.fill_state_dict(unet.assp_evo_state_shapes(meta["width"]))
    m = _model(meta["width"], sd).train()
    x = synth.closed_form_image(1, 4, size).to(DEV)
    t = synth.nested_spheres(1, size).to(DEV)
    out, deeps = m(x)
    assert out.shape == (1, 3, *size) and len(deeps) == 2
    err = np.abs(out.detach().cpu().numpy() - g["logits"]).max()
    assert err < 1e-3, err
    for i, d in enumerate(deeps):
        assert np.abs(d.detach().cpu().numpy()[:, :, ::2, ::2, ::2] - g[f"deep{i}"]).max() < 1e-3
    loss = unet.deep_supervision_loss((out, deeps), t)
    assert abs(loss.item() - float(g["loss"])) < 1e-4
    loss.backward()
    names = json.loads(str(g["grad_names"]))
    params = dict(m.named_parameters())
    assert all(params[k].grad is None for k in params if k.endswith(".v"))  # statically unused
    norms = np.array([float(params[k].grad.double().norm()) for k in names])
    np.testing.assert_allclose(norms, g["grad_norms"], rtol=5e-3, atol=1e-7)
    for k in g.files:
        if k.startswith("grad:"):
            ref = g[k]
            np.testing.assert_allclose(params[k[5:]].grad.cpu().numpy(), ref, atol=5e-3 * max(np.abs(ref).max(), 1e-6), rtol=5e-3)


def test_assp_bf16_and_width48_run():
    sd = synth.fill_state_dict(unet.assp_evo_state_shapes(48))
    m = _model(48, sd, "auto").train()
    size = (16, 16, 16)
    x = synth.random_image(2, 4, size)
    t = synth.nested_spheres(2, size)
    out_ref = unet.assp_evo_forward(sd, x)
    out, deeps = m(x.to(DEV))
    assert float((out.detach().cpu() - out_ref[0]).abs().max()) < 1e-3
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out_b, deeps_b = m(x.to(DEV))
        loss_b = unet.deep_supervision_loss((out_b, deeps_b), t.to(DEV))
    loss_b.backward()
    err = (out_b.detach().cpu() - out_ref[0]).abs()
    assert float(err.mean()) < 0.05 and float(err.max()) < 1.0, (float(err.mean()), float(err.max()))
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for k, p in m.named_parameters() if not k.endswith(".v"))


def test_assp_training_step_is_bitwise_reproducible():
    """EvoNorm / SE / Dice / head reductions add per-block partial sums in a fixed order (no float atomics): two runs
    of the same three bf16 steps give identical losses and parameters."""
    import contextlib
    import io
    from brats21_amd import get_model
    from brats21_amd.engine import TrainStep
    from brats21_amd.optim import Ranger2020
    dev = torch.device("cuda:0")
    ns = argparse.Namespace(model="equiunet_assp_evo", width=16, norm="group", act="relu", num_classes=3, dropout=0)
    x = synth.random_image(2, 4, (32, 32, 32), seed=9).to(dev)
    t = synth.nested_spheres(2, (32, 32, 32)).to(dev)
    runs = []
    for _ in range(2):
        torch.manual_seed(0)
        with contextlib.redirect_stdout(io.StringIO()), warnings.catch_warnings():
            warnings.simplefilter("ignore")
            m = get_model(ns).to(dev).train()
            opt = Ranger2020(m.parameters(), lr=3e-3, use_gc=False)
        step = TrainStep(m, opt, amp=True)
        losses = [float(step(x, t).detach()) for _ in range(3)]
        runs.append((losses, torch.cat([p.detach().flatten()[:500] for p in m.parameters()]).clone()))
    assert runs[0][0] == runs[1][0], (runs[0][0], runs[1][0])
    assert torch.equal(runs[0][1], runs[1][1])
