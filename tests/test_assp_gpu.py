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
import torch.nn.functional as F

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
@pytest.mark.parametrize("size,cin,q", [((8, 8, 8), 32, 16), ((16, 16, 16), 96, 32), ((5, 6, 7), 32, 48)])
def test_aspp_direct_kernels_vs_torch(dtype, size, cin, q):
    """SimpleASPPEVO's branches (networks/equiunet2021.py:179-187; kernel 1 / 3, dilation 1 / 2 / 4 / 6) on the direct
    (gather) convolution -- forward as ONE launch writing four channel slices, input gradient as ONE launch summing
    four terms -- and their weight gradients on the shifted-tap kernel, against F.conv3d / autograd.  The ragged
    5x6x7 volume exercises taps that leave the volume in every direction and partial voxel fragments."""
    from brats21_amd import ops
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(3)
    n = 2
    branches = ((1, 1), (3, 2), (3, 4), (3, 6))
    x = torch.randn(n, cin, *size, generator=g).to(dtype).float()
    ws = [(torch.randn(q, cin, k, k, k, generator=g) * (1.0 / (cin * k ** 3)) ** 0.5).to(dtype).float() for k, _ in branches]
    bs = [torch.randn(q, generator=g) * 0.1 for _ in branches]
    xr = x.clone().requires_grad_(True)
    wr = [w.clone().requires_grad_(True) for w in ws]
    y_ref = torch.cat([F.conv3d(xr, w, b, 1, (k - 1) // 2 * dl, dl) for w, b, (k, dl) in zip(wr, bs, branches)], 1)
    dy = torch.randn(y_ref.shape, generator=g).to(dtype).float()
    y_ref.backward(dy)
    xd, dyd = _to_ndhwc(x, dtype), _to_ndhwc(dy, dtype)
    wd = [w.to(DEV) for w in ws]
    # forward: four jobs, one launch, channel slices of one buffer
    acat = torch.full((n, *size, 4 * q), float("nan"), dtype=dtype, device=DEV)
    jobs = [([(xd, ops.pack_weights_direct(w, dtype, ops.PACK_FWD), k, dl)], b.to(DEV), acat[..., i * q:(i + 1) * q])
            for i, (w, b, (k, dl)) in enumerate(zip(wd, bs, branches))]
    ops.dconv_run(jobs, n, *size, dtype)
    tol = dict(atol=3e-5, rtol=1e-5) if dtype == torch.float32 else dict(atol=4e-2, rtol=2e-2)
    torch.testing.assert_close(_from_ndhwc(acat), y_ref.detach(), **tol)
    # input gradient: one job, four terms summed in the accumulators
    dx = torch.full((n, *size, cin), float("nan"), dtype=dtype, device=DEV)
    terms = [(dyd[..., i * q:(i + 1) * q], ops.pack_weights_direct(w, dtype, ops.PACK_DGRAD), k, dl)
             for i, (w, (k, dl)) in enumerate(zip(wd, branches))]
    ops.dconv_run([(terms, None, dx)], n, *size, dtype)
    tol = dict(atol=1e-4, rtol=1e-5) if dtype == torch.float32 else dict(atol=8e-2, rtol=3e-2)
    torch.testing.assert_close(_from_ndhwc(dx), xr.grad, **tol)
    # weight gradients: shifted-tap kernel (1 tap / 27 taps at any dilation), + bias gradient
    for i, (k, dl) in enumerate(branches):
        dw, db = ops.conv3d_wgrad_shift(xd, dyd[..., i * q:(i + 1) * q], k, dl, want_dbias=True)
        scale = float(wr[i].grad.abs().max())
        tol = dict(atol=2e-5 * scale + 1e-5, rtol=1e-4) if dtype == torch.float32 else dict(atol=2e-2 * scale, rtol=3e-2)
        torch.testing.assert_close(dw.cpu(), wr[i].grad, **tol)
        torch.testing.assert_close(db.cpu(), dy[:, i * q:(i + 1) * q].sum((0, 2, 3, 4)), atol=1e-3 if dtype == torch.float32 else 0.5, rtol=1e-3)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("cin,cout,size", [(48, 24, (16, 16, 32)), (96, 48, (12, 9, 20)), (384, 96, (4, 4, 4)), (16, 8, (8, 8, 8))])
def test_wgrad_1x1_native_vs_torch(dtype, cin, cout, size):
    """Weight gradient of the 1x1x1 convolutions (ConvEvo: bridge / upconv / ASPP conv_k1, networks/equiunet2021.py:
    212-222) on the shifted-tap MFMA kernel (no library GEMM), incl. a channel-slice dy view and ragged tiles."""
    from brats21_amd import ops
    g = torch.Generator().manual_seed(5)
    n = 2
    x = torch.randn(n, cin, *size, generator=g).to(dtype).float()
    dyw = torch.randn(n, cout + 8, *size, generator=g).to(dtype).float()
    xd, dywd = _to_ndhwc(x, dtype), _to_ndhwc(dyw, dtype)
    dyd = dywd[..., 8:]  # a channel-slice view (pitch cout + 8)
    dw, db = ops.conv3d_wgrad_shift(xd, dyd, 1, want_dbias=True)
    ref = torch.einsum("ncdhw,nkdhw->kc", x.double(), dyw[:, 8:].double())
    scale = float(ref.abs().max())
    tol = dict(atol=2e-6 * scale + 1e-5, rtol=1e-4) if dtype == torch.float32 else dict(atol=1e-2 * scale, rtol=2e-2)
    torch.testing.assert_close(dw.cpu().double().view(cout, cin), ref, **tol)
    torch.testing.assert_close(db.cpu().double(), dyw[:, 8:].double().sum((0, 2, 3, 4)), atol=1e-3 if dtype == torch.float32 else 0.5, rtol=1e-3)
    dw2, _ = ops.conv3d_wgrad_shift(xd, dyd, 1)
    assert torch.equal(dw, dw2)  # fixed-order split-K reduction: bitwise reproducible


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


def test_evonorm_bwd_with_folded_se_gradient_map():
    """evonorm_bwd(do, gscale, gadd) = evonorm_bwd(channel_scale(do, gscale, add=gadd)): bitwise in f32 (same arithmetic, no
    intermediate rounding), within bf16 rounding of the intermediate for bf16."""
    from brats21_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(5)
    n, c, size = 2, 48, (6, 8, 16)
    for dt, tol in ((torch.float32, 0.0), (torch.bfloat16, 2e-2)):
        y = torch.randn((n, *size, c), generator=g).to(dev).to(dt)
        do = (torch.randn((n, *size, c), generator=g) * 0.1).to(dev).to(dt)
        mr = torch.stack([torch.randn((n, 8), generator=g) * 0.1, torch.rand((n, 8), generator=g) + 0.5], -1).to(dev).contiguous()
        gamma = (torch.rand(c, generator=g) + 0.5).to(dev)
        gs = (torch.rand((n, c), generator=g) + 0.5).to(dev)
        ga = (torch.randn((n, c), generator=g) * 0.01).to(dev)
        ref = ops.evonorm_bwd(ops.channel_scale(do, gs, add=ga), y, mr, gamma, 8)
        got = ops.evonorm_bwd(do, y, mr, gamma, 8, gscale=gs, gadd=ga)
        for a, b in zip(got[:3], ref[:3]):
            if tol == 0.0:
                assert torch.equal(a, b)
            else:
                torch.testing.assert_close(a.float(), b.float(), atol=tol * float(b.float().abs().max()), rtol=tol)


@pytest.mark.parametrize("n,c,size", [(2, 48, (6, 8, 16)), (3, 96, (4, 4, 8)), (2, 384, (2, 4, 4)), (1, 16, (5, 7, 9))])
def test_evonorm_se_fwd_matches_three_call_composition(n, c, size):
    """brats_evonorm_se_fwd (sum of x*sigmoid(x) -> gate on the reconstructed sum_v z -> out = z * (1 + gate) in the EvoNorm
    pass; z never stored) against evonorm(chansum) -> se_gate -> channel_scale.  f32: the same mathematics in another order;
    bf16: the composition rounds z to bf16 before scaling, the fused call rounds once (closer to the f32 result)."""
    from brats21_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(21 + c)
    ch, vox = c // 2, size[0] * size[1] * size[2]
    for dt, tol in ((torch.float32, 2e-5), (torch.bfloat16, 1.2e-2)):
        y = torch.randn((n, *size, c), generator=g).to(dev).to(dt)
        mr = torch.stack([torch.randn((n, 8), generator=g) * 0.1, torch.rand((n, 8), generator=g) + 0.5], -1).to(dev).contiguous()
        gamma = (torch.rand(c, generator=g) + 0.5).to(dev)
        beta = (torch.randn(c, generator=g) * 0.3).to(dev)
        w1, b1 = (torch.randn((ch, c), generator=g) * 0.3).to(dev), (torch.randn((ch,), generator=g) * 0.2).to(dev)
        w2, b2 = (torch.randn((c, ch), generator=g) * 0.3).to(dev), (torch.randn((c,), generator=g) * 0.2).to(dev)
        z, cs = ops.evonorm(y, mr, gamma, beta, 8, want_chansum=True)
        gate1p, hidden = ops.se_gate(cs, vox, w1, b1, w2, b2)
        ref = ops.channel_scale(z, gate1p)
        out, cs2, gate2, hidden2 = ops.evonorm_se(y, mr, gamma, beta, w1, b1, w2, b2, 8)
        for name, a, b, t in (("chansum", cs2, cs, 2e-5), ("gate1p", gate2, gate1p, 2e-5), ("hidden", hidden2, hidden, 2e-5),
                              ("out", out, ref, tol)):
            scale = float(b.float().abs().max()) + 1e-30
            assert float((a.float() - b.float()).abs().max()) <= t * scale, (name, str(dt))
        again = ops.evonorm_se(y, mr, gamma, beta, w1, b1, w2, b2, 8)
        assert all(torch.equal(a, b) for a, b in zip(again, (out, cs2, gate2, hidden2)))


@pytest.mark.parametrize("n,c,size", [(2, 48, (6, 8, 16)), (3, 96, (4, 4, 8)), (2, 384, (2, 4, 4)), (1, 16, (5, 7, 9))])
def test_evonorm_se_bwd_matches_three_call_composition(n, c, size):
    """brats_evonorm_se_bwd (pass 1 with five raw sums -> SE backward on the sums -> pass 2; csrc/se.hpp) against the
    composition it replaces: channel_dot(do, z) -> se_gate_bwd -> evonorm_bwd(do, gscale = 1 + gate, gadd).  The same
    mathematics with the sums taken in a different order: f32 within 2e-5 of the largest entry; bf16 within one rounding of
    the stored gradient."""
    from brats21_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(11 + c)
    ch, vox = c // 2, size[0] * size[1] * size[2]
    for dt, tol in ((torch.float32, 2e-5), (torch.bfloat16, 1e-2)):
        y = torch.randn((n, *size, c), generator=g).to(dev).to(dt)
        do = (torch.randn((n, *size, c), generator=g) * 0.1).to(dev).to(dt)
        mr = torch.stack([torch.randn((n, 8), generator=g) * 0.1, torch.rand((n, 8), generator=g) + 0.5], -1).to(dev).contiguous()
        gamma = (torch.rand(c, generator=g) + 0.5).to(dev)
        beta = (torch.randn(c, generator=g) * 0.3).to(dev)
        w1, b1 = (torch.randn((ch, c), generator=g) * 0.3).to(dev), (torch.randn((ch,), generator=g) * 0.2).to(dev)
        w2, b2 = (torch.randn((c, ch), generator=g) * 0.3).to(dev), (torch.randn((c,), generator=g) * 0.2).to(dev)
        z, cs = ops.evonorm(y, mr, gamma, beta, 8, want_chansum=True)
        gate1p, hidden = ops.se_gate(cs, vox, w1, b1, w2, b2)
        dgate = ops.channel_dot(do, z)
        gadd, dw1, db1, dw2, db2 = ops.se_gate_bwd(dgate, cs, vox, hidden, gate1p, w1, w2)
        dy, dgamma, dbeta, _ = ops.evonorm_bwd(do, y, mr, gamma, 8, gscale=gate1p, gadd=gadd)
        got = ops.evonorm_se_bwd(do, y, mr, gamma, beta, cs, hidden, gate1p, w1, w2, 8)
        names = ("dy", "dgamma", "dbeta", "dcb", "dw1", "db1", "dw2", "db2")
        for name, a, b in zip(names, got, (dy, dgamma, dbeta, None, dw1, db1, dw2, db2)):
            if b is None:
                assert a is None
                continue
            scale = float(b.float().abs().max()) + 1e-30
            assert float((a.float() - b.float()).abs().max()) <= tol * scale, (name, str(dt))
        again = ops.evonorm_se_bwd(do, y, mr, gamma, beta, cs, hidden, gate1p, w1, w2, 8)
        assert all(torch.equal(a, b) for a, b in zip(again, got) if a is not None)  # bitwise reproducible


@pytest.mark.parametrize("n,c,size,with_avg", [(2, 48, (6, 8, 16), True), (1, 16, (4, 6, 10), True), (2, 96, (4, 4, 4), False)])
def test_evonorm_se_bwd_with_folded_pool_backward(n, c, size, with_avg):
    """brats_evonorm_se_bwd_pool (the block's output gradient = skip gradient + MaxAvgPool backward composed inside the passes
    from the pieces and the arg-max bytes) against maxpool2_bwd + evonorm_se_bwd(dout): f32 bit for bit, 16-bit within the
    rounding of the gradient tensor that is no longer stored."""
    from brats21_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(51 + c)
    ch = c // 2
    for dt, tol in ((torch.float32, 0.0), (torch.bfloat16, 1.5e-2), (torch.float16, 2e-3)):
        y = torch.randn((n, *size, c), generator=g).to(dev).to(dt)
        mr = torch.stack([torch.randn((n, 8), generator=g) * 0.1, torch.rand((n, 8), generator=g) + 0.5], -1).to(dev).contiguous()
        gamma = (torch.rand(c, generator=g) + 0.5).to(dev)
        beta = (torch.randn(c, generator=g) * 0.3).to(dev)
        w1, b1 = (torch.randn((ch, c), generator=g) * 0.3).to(dev), (torch.randn((ch,), generator=g) * 0.2).to(dev)
        w2, b2 = (torch.randn((c, ch), generator=g) * 0.3).to(dev), (torch.randn((c,), generator=g) * 0.2).to(dev)
        out, cs, gate1p, hidden = ops.evonorm_se(y, mr, gamma, beta, w1, b1, w2, b2, 8)
        ops.maxpool2(out, with_avg, want_argmax=True)
        dskip = (torch.randn((n, *size, c), generator=g) * 0.1).to(dev).to(dt)
        dpool = (torch.randn((n, size[0] // 2, size[1] // 2, size[2] // 2, c * (2 if with_avg else 1)), generator=g) * 0.1).to(dev).to(dt)
        do = ops.maxpool2_bwd(out, dpool, dx_skip=dskip, with_avg=with_avg)
        ref = ops.evonorm_se_bwd(do, y, mr, gamma, beta, cs, hidden, gate1p, w1, w2, 8)
        got = ops.evonorm_se_bwd(None, y, mr, gamma, beta, cs, hidden, gate1p, w1, w2, 8, pool=(dskip, dpool, out._pool_argmax, with_avg))
        names = ("dy", "dgamma", "dbeta", "dcb", "dw1", "db1", "dw2", "db2")
        for name, a, b in zip(names, got, ref):
            if b is None:
                assert a is None
                continue
            if tol == 0.0:
                assert torch.equal(a, b), name
            else:
                scale = float(b.float().abs().max()) + 1e-30
                assert float((a.float() - b.float()).abs().max()) <= tol * scale, (name, str(dt))


@pytest.mark.parametrize("n,c,size,k", [(2, 48, (6, 8, 16), 3), (1, 16, (5, 7, 9), 3), (2, 64, (4, 4, 8), 4), (1, 96, (4, 4, 4), 2)])
def test_output_head_on_the_last_block_without_storing_its_output(n, c, size, k):
    """brats_evonorm_head_fwd (the gated EvoNorm output recomputed on load, rounded to the storage type) == head(evonorm_se(y))
    bit for bit, with gate / chansum / hidden from the statistics-only call (apply=False) equal to the full call's."""
    from brats21_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(41 + c)
    ch = c // 2
    for dt in (torch.float32, torch.bfloat16, torch.float16):
        y = torch.randn((n, *size, c), generator=g).to(dev).to(dt)
        mr = torch.stack([torch.randn((n, 8), generator=g) * 0.1, torch.rand((n, 8), generator=g) + 0.5], -1).to(dev).contiguous()
        gamma = (torch.rand(c, generator=g) + 0.5).to(dev)
        beta = (torch.randn(c, generator=g) * 0.3).to(dev)
        w1, b1 = (torch.randn((ch, c), generator=g) * 0.3).to(dev), (torch.randn((ch,), generator=g) * 0.2).to(dev)
        w2, b2 = (torch.randn((c, ch), generator=g) * 0.3).to(dev), (torch.randn((c,), generator=g) * 0.2).to(dev)
        hw = (torch.randn((k, c, 1, 1, 1), generator=g) * 0.2).to(dev)
        hb = (torch.randn((k,), generator=g) * 0.1).to(dev)
        out, cs, gate1p, hidden = ops.evonorm_se(y, mr, gamma, beta, w1, b1, w2, b2, 8)
        ref = ops.head(out, hw, hb, 1)
        none, cs2, gate2, hidden2 = ops.evonorm_se(y, mr, gamma, beta, w1, b1, w2, b2, 8, apply=False)
        assert none is None and torch.equal(cs2, cs) and torch.equal(gate2, gate1p) and torch.equal(hidden2, hidden)
        got = ops.evonorm_head(y, mr, gamma, beta, gate2, hw, hb, 8)
        assert torch.equal(got, ref), str(dt)


@pytest.mark.parametrize("n,c,size", [(2, 48, (6, 8, 16)), (1, 16, (5, 7, 9))])
def test_evonorm_se_bwd_with_folded_output_head(n, c, size):
    """brats_evonorm_se_bwd(dlogits) -- the decoder1 block's backward computing its output gradient W_head^T dlogits on the
    fly and the head's weight / bias gradients out of pass 1 -- against head_bwd on the stored block output followed by
    brats_evonorm_se_bwd(dout)."""
    from brats21_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(31 + c)
    ch = c // 2
    for dt, tol in ((torch.float32, 3e-5), (torch.bfloat16, 1.5e-2)):
        y = torch.randn((n, *size, c), generator=g).to(dev).to(dt)
        mr = torch.stack([torch.randn((n, 8), generator=g) * 0.1, torch.rand((n, 8), generator=g) + 0.5], -1).to(dev).contiguous()
        gamma = (torch.rand(c, generator=g) + 0.5).to(dev)
        beta = (torch.randn(c, generator=g) * 0.3).to(dev)
        w1, b1 = (torch.randn((ch, c), generator=g) * 0.3).to(dev), (torch.randn((ch,), generator=g) * 0.2).to(dev)
        w2, b2 = (torch.randn((c, ch), generator=g) * 0.3).to(dev), (torch.randn((c,), generator=g) * 0.2).to(dev)
        hw = (torch.randn((3, c, 1, 1, 1), generator=g) * 0.2).to(dev)
        dl = (torch.randn((n, 3, *size), generator=g) * 0.1).to(dev)
        out, cs, gate1p, hidden = ops.evonorm_se(y, mr, gamma, beta, w1, b1, w2, b2, 8)
        do, dhw_r, dhb_r = ops.head_bwd(out, hw, dl, 1)
        ref = ops.evonorm_se_bwd(do, y, mr, gamma, beta, cs, hidden, gate1p, w1, w2, 8)
        got = ops.evonorm_se_bwd(None, y, mr, gamma, beta, cs, hidden, gate1p, w1, w2, 8, head=(hw, dl))
        names = ("dy", "dgamma", "dbeta", "dcb", "dw1", "db1", "dw2", "db2", "dhw", "dhb")
        for name, a, b in zip(names, got, ref + (dhw_r, dhb_r)):
            if b is None:
                assert a is None
                continue
            scale = float(b.float().abs().max()) + 1e-30
            assert float((a.float() - b.float()).abs().max()) <= tol * scale, (name, str(dt))
        again = ops.evonorm_se_bwd(None, y, mr, gamma, beta, cs, hidden, gate1p, w1, w2, 8, head=(hw, dl))
        assert all(torch.equal(a, b) for a, b in zip(again, got) if a is not None)  # bitwise reproducible


@pytest.mark.parametrize("n,c", [(2, 48), (1, 16), (4, 384), (3, 96), (19, 48), (9, 192)])  # (> 8 samples: groups of 8)
def test_se_gate_kernels_vs_torch_autograd(n, c):
    """csrc/se.hip (one launch forward, one backward) against the MONAI ResidualSELayer arithmetic in torch f64:
    gate = sigmoid(W2 relu(W1 gap + b1) + b2) with gap = chansum / V, and the gradients of gap, W1, b1, W2, b2
    (networks/equiunet2021.py:204-205)."""
    from brats21_amd import ops
    g = torch.Generator().manual_seed(100 + c)
    ch, vox = c // 2, 4096
    cs = torch.randn((n, c), generator=g) * 300.0
    w1, b1 = torch.randn((ch, c), generator=g) * 0.3, torch.randn((ch,), generator=g) * 0.2
    w2, b2 = torch.randn((c, ch), generator=g) * 0.3, torch.randn((c,), generator=g) * 0.2
    dgate = torch.randn((n, c), generator=g)
    r = [t.double().requires_grad_(True) for t in (cs, w1, b1, w2, b2)]
    gap = r[0] / vox
    gate = torch.sigmoid(F.linear(F.relu(F.linear(gap, r[1], r[2])), r[3], r[4]))
    gate.backward(dgate.double())
    d = [t.to(DEV) for t in (cs, w1, b1, w2, b2)]
    gate1p, hidden = ops.se_gate(d[0], vox, d[1], d[2], d[3], d[4])
    torch.testing.assert_close(gate1p.cpu().double() - 1.0, gate.detach(), atol=2e-6, rtol=1e-5)
    gadd, dw1, db1, dw2, db2 = ops.se_gate_bwd(dgate.to(DEV), d[0], vox, hidden, gate1p, d[1], d[3])
    # gadd = d loss / d gap / V = d loss / d chansum
    for got, ref, name in ((gadd, r[0].grad, "gadd"), (dw1, r[1].grad, "dw1"), (db1, r[2].grad, "db1"), (dw2, r[3].grad, "dw2"),
                           (db2, r[4].grad, "db2")):
        scale = float(ref.abs().max()) + 1e-30
        assert float((got.cpu().double() - ref).abs().max()) <= 2e-5 * scale, name
    # bitwise reproducible
    again = ops.se_gate_bwd(dgate.to(DEV), d[0], vox, hidden, gate1p, d[1], d[3])
    assert all(torch.equal(a, b) for a, b in zip(again, (gadd, dw1, db1, dw2, db2)))


def test_conv_evo_block_matches_reference_golden(golden_dir):
    """ConvEvoBlockCorrected (conv-EvoNorm-conv-EvoNorm-ResidualSE, networks/equiunet2021.py:192-209) through the unit
    program _block_fwd with the native SE gate, against the reference's own output (tests/golden/ops.npz: block_y)."""
    from brats21_amd.networks import equiunet_assp as ea
    gold = np.load(os.path.join(golden_dir, "ops.npz"))
    blk = ea.ConvEvoBlockCorrected(16, 16)
    order = list(blk.state_dict().keys())
    shapes = {k: tuple(v.shape) for k, v in blk.state_dict().items()}
    blk.load_state_dict(synth.fill_state_dict({k: shapes[k] for k in order}))
    blk = blk.to(DEV)

    class _M(torch.nn.Module):
        def __init__(self, b):
            super().__init__()
            self.b = b
            self._grad_sink = None
    x = synth.closed_form_image(1, 16, (12, 12, 12), "opx")
    cx = ea._Ctx(_M(blk), torch.float32)
    y, _ = ea._block_fwd(cx, blk, _to_ndhwc(x))
    np.testing.assert_allclose(_from_ndhwc(y).numpy(), gold["block_y"], atol=1e-4)


def test_dropout_network_matches_oracle_given_the_masks():
    """--dropout p > 0 for EquiUnetASSPEvo (nn.Dropout behind both EvoNorms of every block -- the second one BEFORE the SE layer --
    and behind every ConvEvo's but the ASPP's: networks/equiunet2021.py:200,203,219; :178): the masks of a training forward are
    re-drawn from the model's (seed, step) state and handed to the CPU oracle; logits, loss and every parameter gradient agree
    at the f32 bars.  The fused EvoNorm + SE forms (which would average UN-dropped values) are off; eval mode ignores p."""
    import argparse, contextlib, io, os, warnings
    from brats21_amd import get_model, ops
    dev = torch.device("cuda:0")
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    p, size, width = 0.2, (16, 16, 16), 16
    torch.manual_seed(0)
    with contextlib.redirect_stdout(io.StringIO()), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = get_model(argparse.Namespace(model="equiunet_assp_evo", width=width, norm="group", act="relu", num_classes=3, dropout=p))
    m.precision = "fp32"
    m = m.to(dev).train()
    sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    x, t = synth.random_image(2, 4, size, seed=3), synth.nested_spheres(2, size)
    out, deeps = m(x.to(dev))
    loss = unet.deep_supervision_loss((out, deeps), t.to(dev))
    loss.backward()
    state = m._dropout_state.clone()
    names = {mod: k for k, mod in m.named_modules() if mod in m._unit_ids}
    f = [width * 2 ** i for i in range(4)]
    level = {"encoder1": (1, f[0]), "encoder2": (2, f[1]), "encoder3": (4, f[2]), "encoder4": (8, f[3]), "decoder3": (4, f[2]),
             "decoder2": (2, f[1]), "decoder1": (1, f[0]), "bridge1": (1, f[0] // 2), "bridge2": (2, f[1] // 2), "bridge3": (4, f[2] // 2),
             "upconv3": (8, f[3] // 4), "upconv2": (4, f[2] // 4), "upconv1": (2, f[1] // 4)}
    drop = {}

    def mask(name, uid):
        sc, c = level[name]
        ones = torch.ones(2, *(s // sc for s in size), c, device=dev)
        return ops.dropout(ones, p, state, uid).permute(0, 4, 1, 2, 3).contiguous().cpu()

    for mod, uid in m._unit_ids.items():
        name = names[mod]
        if name.startswith("aspp"):
            continue  # (its dropout probability is pinned to 0 in the reference)
        if name in ("bridge1", "bridge2", "bridge3", "upconv1", "upconv2", "upconv3"):
            drop[name] = mask(name, uid)
        else:
            drop[name + ".2"], drop[name + ".5"] = mask(name, uid), mask(name, uid + 1)
    assert len(drop) == 7 * 2 + 6
    sd_ref = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point else v) for k, v in sd.items()}
    out_ref = unet.assp_evo_forward(sd_ref, x, drop=drop)
    loss_ref = unet.deep_supervision_loss(out_ref, t)
    loss_ref.backward()
    err = float((out.detach().cpu() - out_ref[0].detach()).abs().max())
    worst = ("", 0.0)
    for k, q in m.named_parameters():
        if k.endswith(".v"):
            assert q.grad is None
            continue
        rel = float((q.grad.cpu() - sd_ref[k].grad).norm() / (sd_ref[k].grad.norm() + 1e-12))
        worst = max(worst, (k, rel), key=lambda e: e[1])
    print(f"\nEquiUnetASSPEvo --dropout {p}: logits max abs err vs the oracle given the masks {err:.2e}, loss {loss.item():.6f} vs "
          f"{loss_ref.item():.6f}, worst gradient rel err {worst[1]:.2e} ({worst[0]})")
    assert err < 1e-3 and abs(loss.item() - loss_ref.item()) < 1e-4 and worst[1] < 5e-3
    with torch.no_grad():
        out2 = m(x.to(dev))[0]
        assert int(m._dropout_state[1]) == 2 and not torch.equal(out2, out.detach())
        m.eval()
        e1, e2 = m(x.to(dev))[0], m(x.to(dev))[0]
        assert torch.equal(e1, e2) and int(m._dropout_state[1]) == 2


@pytest.mark.parametrize("precision", ["bf16", "fp16"])
def test_evonorm_backward_statistics_fold(precision):
    """model.fold_bwd_stats for EquiUnetASSPEvo (round 5): the first EvoNorm's backward pass 1 rides in the input-gradient launch of
    the block's second convolution -- brats_conv3d_fwd_bstats on the STORED EvoNorm output z (linear in x * sigmoid(x)) with the
    identity activation, then brats_evonorm_bwd_tiles.  (i) kernel level: ops.evonorm_bwd_tiles on the tile sums == ops.evonorm_bwd
    on the same dz (a gamma with an exact zero and a negative entry included); (ii) network level, width 48: all seven blocks
    take the form, the forward is untouched, the gradients equal the two-pass form's up to the rounding of dz."""
    import argparse, contextlib, copy, io, warnings
    from brats21_amd import get_model, ops
    from brats21_amd import synth as psynth
    from brats21_amd.losses import DiceLoss
    dev = torch.device("cuda:0")
    dt = torch.bfloat16 if precision == "bf16" else torch.float16
    # ---- (i) kernel level
    g = torch.Generator().manual_seed(4)
    n, size, c = 2, (8, 16, 32), 48
    x1 = torch.randn(n, *size, c, generator=g).to(dev).to(dt)              # conv1's raw output
    dy2 = (torch.randn(n, *size, c, generator=g) * 0.1).to(dev).to(dt)     # gradient of conv2's raw output
    w2 = (torch.randn(c, c, 3, 3, 3, generator=g) * 0.05).to(dev)
    gamma = (1.0 + 0.3 * torch.randn(c, generator=g)).to(dev)
    gamma[3], gamma[7] = 0.0, -0.8
    beta = (0.2 * torch.randn(c, generator=g)).to(dev)
    vox = size[0] * size[1] * size[2]
    # EvoNorm's group statistics of x1 (unbiased variance over C/8 x D x H x W, networks/equiunet2021.py:48-52) and its channel sums, in f64
    xf = x1.double()
    mean = xf.view(n, -1, 8, c // 8).mean((1, 3))
    var = xf.view(n, -1, 8, c // 8).var((1, 3), unbiased=True)
    mr = torch.stack([mean, 1.0 / torch.sqrt(var + 1e-5)], -1).float().contiguous()
    chan = torch.stack([xf.sum((1, 2, 3)), (xf * xf).sum((1, 2, 3))], -1).contiguous()
    z1, _ = ops.evonorm(x1, mr, gamma, beta, 8)
    wpk = ops.pack_weights(w2, dt, ops.PACK_DGRAD)
    ss = torch.zeros(n, c, 2, device=dev); ss[..., 0] = 1.0
    assert ops.conv_bstats_ok(dt, 1, c, c, "leakyrelu")
    dz, tiles = ops.conv3d_bstats(dy2, wpk, c, 1, z1, ss, "leakyrelu", slope=1.0)
    dz_plain, _ = ops.conv3d(dy2, wpk, c, 3, 1)
    assert torch.equal(dz, dz_plain)
    a = ops.evonorm_bwd_tiles(tiles, dz, x1, mr, gamma, beta, 8, chan=chan)
    b = ops.evonorm_bwd(dz, x1, mr, gamma, 8, chan=chan)
    for name, u, v, tol in (("dx", a[0].float(), b[0].float(), 2e-2), ("dgamma", a[1], b[1], 1e-2), ("dbeta", a[2], b[2], 1e-2), ("dconvbias", a[3], b[3], 1e-2)):
        e = float((u - v).abs().max() / (v.abs().max() + 1e-12))
        print(f"evonorm_bwd_tiles vs evonorm_bwd ({precision}) {name}: rel-to-max {e:.2e}")
        assert e < tol, (name, e)
    assert bool(torch.isfinite(a[1]).all()) and abs(float(a[1][3] - b[1][3])) <= 1e-2 * float(b[1].abs().max())  # gamma = 0: no division by it
    # ---- (ii) network level
    torch.manual_seed(0)
    with contextlib.redirect_stdout(io.StringIO()), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        base = get_model(argparse.Namespace(model="equiunet_assp_evo", width=48, norm="group", act="relu", num_classes=3, dropout=0)).to(dev).train()
    base.precision = precision
    x = psynth.random_image(2, 4, (32, 32, 32), seed=7, device=dev)
    t = psynth.nested_spheres(2, (32, 32, 32), device=dev)
    crit = DiceLoss().to(dev)
    calls = []
    real = ops.conv3d_bstats

    def counted(*a_, **k_):
        calls.append(1)
        return real(*a_, **k_)

    def run(fold):
        model = copy.deepcopy(base)
        model.fold_bwd_stats = fold
        out, deep = model(x)
        loss = crit(out.float(), t) + sum(crit(d.float(), t) for d in deep)
        loss.backward()
        return out.detach(), [q.grad.detach().clone() if q.grad is not None else None for q in model.parameters()]

    ops.conv3d_bstats = counted
    try:
        o1, g1 = run(True)
        n_fused = len(calls)
        o0, g0 = run(False)
    finally:
        ops.conv3d_bstats = real
    assert n_fused == 7 and len(calls) == 7 and torch.equal(o1, o0)
    # (sums over 65 k signed values with cancellation: the fused form adds the f32 accumulators, the two-pass form the stored 16-bit
    #  dz -- fp16 measured 1.3e-2 of max|grad| on ONE beta vector, everything else below 5e-3)
    tol = 4e-2 if precision == "bf16" else 2.5e-2
    worst = 0.0
    for (name, _), u, v in zip(base.named_parameters(), g1, g0):
        assert (u is None) == (v is None), name
        if u is None:
            continue
        assert bool(torch.isfinite(u).all()), name
        e = float((u - v).abs().max()) / (float(v.abs().max()) + 1e-12)
        worst = max(worst, e)
        assert e <= tol, (name, e)
    print(f"EquiUnetASSPEvo-48 fold_bwd_stats on / off ({precision}): worst gradient difference rel-to-max {worst:.2e} (bar {tol})")


@pytest.mark.parametrize("precision", ["bf16", "fp16"])
@pytest.mark.parametrize("beta_scale,gamma_scale", [(1.0, 1.0), (3.0, 1.0), (3.0, 0.1)])
def test_evonorm_backward_statistics_fold_with_large_beta(precision, beta_scale, gamma_scale):
    """brats_evonorm_bwd_tiles recovers A_g = (S2 - beta * S1) / rstd from sums over the STORED 16-bit z, so the rounding
    of z (2^-9 relative for bf16, 2^-11 for fp16) is amplified by the cancellation when |beta| is large against gamma * num.  The
    reference initialises beta = 0 / gamma = 1 (networks/equiunet2021.py:70-71) and trained ASSP weights stay at |beta| < 0.5; this
    case prices |beta| up to ~3 x randn and a small gamma against the two-pass kernel (which reads x, not z) and holds the measured
    degradation to a stated bound -- the caveat in csrc/norm.hip's header and DESIGN.md quotes these numbers."""
    from brats21_amd import ops
    dev = torch.device("cuda:0")
    dt = torch.bfloat16 if precision == "bf16" else torch.float16
    g = torch.Generator().manual_seed(9)
    n, size, c = 2, (8, 16, 32), 48
    x1 = torch.randn(n, *size, c, generator=g).to(dev).to(dt)
    dy2 = (torch.randn(n, *size, c, generator=g) * 0.1).to(dev).to(dt)
    w2 = (torch.randn(c, c, 3, 3, 3, generator=g) * 0.05).to(dev)
    gamma = (gamma_scale * (1.0 + 0.3 * torch.randn(c, generator=g))).to(dev)
    beta = (beta_scale * torch.randn(c, generator=g)).to(dev)
    xf = x1.double()
    mean = xf.view(n, -1, 8, c // 8).mean((1, 3))
    var = xf.view(n, -1, 8, c // 8).var((1, 3), unbiased=True)
    mr = torch.stack([mean, 1.0 / torch.sqrt(var + 1e-5)], -1).float().contiguous()
    chan = torch.stack([xf.sum((1, 2, 3)), (xf * xf).sum((1, 2, 3))], -1).contiguous()
    z1, _ = ops.evonorm(x1, mr, gamma, beta, 8)
    wpk = ops.pack_weights(w2, dt, ops.PACK_DGRAD)
    ss = torch.zeros(n, c, 2, device=dev); ss[..., 0] = 1.0
    dz, tiles = ops.conv3d_bstats(dy2, wpk, c, 1, z1, ss, "leakyrelu", slope=1.0)
    a = ops.evonorm_bwd_tiles(tiles, dz, x1, mr, gamma, beta, 8, chan=chan)
    b = ops.evonorm_bwd(dz, x1, mr, gamma, 8, chan=chan)
    # bf16 keeps 8 mantissa bits of z, fp16 11: the bound scales with the storage's rounding and with |beta| / |gamma|
    unit = (2.0 ** -8 if precision == "bf16" else 2.0 ** -11) * max(1.0, beta_scale / gamma_scale)
    errs = {}
    for name, u, v in (("dx", a[0].float(), b[0].float()), ("dgamma", a[1], b[1]), ("dbeta", a[2], b[2])):
        errs[name] = float((u - v).abs().max() / (v.abs().max() + 1e-12))
    print(f"\nevonorm_bwd_tiles, {precision}, |beta| ~ {beta_scale}, |gamma| ~ {gamma_scale}: " +
          ", ".join(f"{k} rel-to-max {e:.2e}" for k, e in errs.items()) + f" (unit {unit:.2e})")
    assert errs["dx"] < 2e-2 + 4 * unit and errs["dgamma"] < 1e-2 + 4 * unit and errs["dbeta"] < 1e-2
