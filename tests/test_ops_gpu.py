"""-m gpu: every HIP op through the C ABI against a plain torch fp32 CPU reference of the same op
(the ATen calls the reference's CPU path makes).  f32 kernels: tight tolerances (exact-f32 MFMA);
bf16 kernels: tolerance stated per test (bf16 storage has 8 mantissa bits)."""
import contextlib

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DT = [torch.float32, torch.bfloat16, torch.float16]  # f32 (exact MFMA), bf16, fp16 (the -DBRATS_FP16 twin build of the same kernels)


def _dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch.device("cuda:0")


def _rand(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=g) * scale


def _to_ndhwc(x_ncdhw, dtype, dev, pitch=None, off=0):
    """CPU NCDHW f32 -> device NDHWC (optionally as a channel slice of a wider buffer)."""
    n, c, d, h, w = x_ncdhw.shape
    t = x_ncdhw.permute(0, 2, 3, 4, 1).contiguous().to(dev).to(dtype)
    if pitch is None:
        return t
    buf = torch.full((n, d, h, w, pitch), 7.0, dtype=dtype, device=dev)
    buf[..., off:off + c] = t
    return buf[..., off:off + c]


def _from_ndhwc(t):
    return t.float().cpu().permute(0, 4, 1, 2, 3).contiguous()


def _tol(dtype, f32, bf16):
    return f32 if dtype == torch.float32 else bf16


def _q(x, dtype):
    """Round a CPU f32 tensor through the activation dtype (what the kernel actually sees)."""
    return x.to(dtype).float()


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("cin,cout,dil,size", [
    (4, 8, 1, (8, 8, 8)),       # first layer (padded cin), NF=1 ksplit
    (8, 16, 1, (12, 16, 8)),    # partial tiles in z
    (16, 32, 1, (8, 8, 16)),    # NF=2 ksplit
    (32, 64, 2, (8, 8, 8)),     # dilation 2, NF=2 no-split
    (48, 48, 1, (8, 16, 8)),    # width-48 layer: CK=48 straddling taps, NF=3 ksplit
    (96, 96, 1, (4, 8, 8)),     # two chunks, NF=3 no-split
    (64, 8, 1, (4, 4, 4)),      # tiny volume: tile larger than the volume
])
def test_conv3d_fwd_dgrad_wgrad(dtype, cin, cout, dil, size):
    from brats21_amd import ops
    dev = _dev()
    n = 2
    x = _q(_rand((n, cin, *size), 1), dtype)
    w = _rand((cout, cin, 3, 3, 3), 2, (2.0 / (cin * 27)) ** 0.5)
    wq = _q(w, dtype)
    xr = x.clone().requires_grad_(True)
    wr = wq.clone().requires_grad_(True)
    y_ref = F.conv3d(xr, wr, None, 1, dil, dil)
    dy = _q(_rand(y_ref.shape, 3), dtype)
    y_ref.backward(dy)

    cpad = cin if cin % 8 == 0 else 8
    if dtype == torch.float32 and cin == 4:
        cpad = 4
    xp = torch.zeros((n, cpad, *size))
    xp[:, :cin] = x
    xd = _to_ndhwc(xp, dtype, dev)
    wd = w.to(dev)
    wpk = ops.pack_weights(wd, dtype, ops.PACK_FWD, cin_pad=cpad, dil=dil)
    y, stats = ops.conv3d(xd, wpk, cout, 3, dil, want_stats=True)
    torch.cuda.synchronize()
    yh = _from_ndhwc(y)
    atol = _tol(dtype, 2e-5, 3e-2)
    torch.testing.assert_close(yh, y_ref.detach(), atol=atol, rtol=_tol(dtype, 1e-5, 2e-2))
    # tile statistics: sum and sum of squares per (n, channel) of the f32 result
    s = stats.double().sum(1).cpu()
    ref1 = y_ref.detach().double().sum((2, 3, 4))
    ref2 = (y_ref.detach().double() ** 2).sum((2, 3, 4))
    torch.testing.assert_close(s[..., 0], ref1, atol=_tol(dtype, 1e-3, 2e-2) * y_ref[0, 0].numel() ** 0.5, rtol=1e-3)
    torch.testing.assert_close(s[..., 1], ref2, atol=1e-2, rtol=_tol(dtype, 1e-4, 1e-3))

    # dgrad = same kernel, weights packed transposed + flipped
    dyd = _to_ndhwc(dy, dtype, dev)
    if cin % 8 == 0:
        wpk_d = ops.pack_weights(wd, dtype, ops.PACK_DGRAD, dil=dil)
        dx, _ = ops.conv3d(dyd, wpk_d, cin, 3, dil)
        torch.testing.assert_close(_from_ndhwc(dx), xr.grad, atol=_tol(dtype, 3e-5, 5e-2), rtol=_tol(dtype, 1e-5, 2e-2))
    # wgrad
    dw, _ = ops.conv3d_wgrad(xd, dyd, 3, dil)
    dw = dw[:, :cin].cpu()
    scale = float(wr.grad.abs().max())
    torch.testing.assert_close(dw, wr.grad, atol=_tol(dtype, 2e-5, 4e-3) * max(scale, 1.0), rtol=_tol(dtype, 1e-4, 1e-2))


@pytest.mark.parametrize("dtype", DT)
def test_conv3d_channel_slice_views(dtype):
    """Input read from / output written into channel slices of wider buffers (concat removal)."""
    from brats21_amd import ops
    dev = _dev()
    x = _q(_rand((1, 16, 8, 8, 8), 5), dtype)
    w = _rand((16, 16, 3, 3, 3), 6, 0.07)
    xd = _to_ndhwc(x, dtype, dev, pitch=40, off=8)
    out_buf = torch.full((1, 8, 8, 8, 48), 3.0, dtype=dtype, device=dev)
    wpk = ops.pack_weights(w.to(dev), dtype, ops.PACK_FWD)
    ops.conv3d(xd, wpk, 16, 3, 1, out=out_buf[..., 16:32])
    ref = F.conv3d(x, _q(w, dtype), None, 1, 1)
    torch.testing.assert_close(_from_ndhwc(out_buf[..., 16:32]), ref, atol=_tol(dtype, 2e-5, 3e-2), rtol=_tol(dtype, 1e-5, 2e-2))
    assert float(out_buf[..., :16].float().min()) == 3.0 and float(out_buf[..., 32:].float().max()) == 3.0  # untouched


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("c,act", [(8, "relu"), (48, "relu"), (16, "leakyrelu")])
def test_groupnorm_act_fwd_bwd(dtype, c, act):
    from brats21_amd import ops
    dev = _dev()
    n, size = 2, (8, 8, 8)
    cin = 8
    x = _q(_rand((n, cin, *size), 11), dtype)
    w = _q(_rand((c, cin, 3, 3, 3), 12, 0.1), dtype)
    gamma = 1.0 + 0.2 * _rand((c,), 13)
    beta = 0.1 * _rand((c,), 14)
    xd = _to_ndhwc(x, dtype, dev)
    wpk = ops.pack_weights(w.to(dev), dtype, ops.PACK_FWD)
    y, stats = ops.conv3d(xd, wpk, c, 3, 1, want_stats=True)
    # reference: GN -> act on the conv output *as stored* (bf16-rounded in the bf16 mode, so the
    # activation masks agree), gradient w.r.t. that conv output
    y_ref = (F.conv3d(x, w, None, 1, 1) if dtype == torch.float32 else _from_ndhwc(y)).detach().requires_grad_(True)
    g_r, b_r = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    zn = F.group_norm(y_ref, 8, g_r, b_r, 1e-5)
    z_ref = F.relu(zn) if act == "relu" else F.leaky_relu(zn, 0.01)
    dz = _q(_rand(z_ref.shape, 15), dtype)
    z_ref.backward(dz)
    mr, ss = ops.gn_finalize(stats, n, c, 8, 512, gamma.to(dev), beta.to(dev))
    z = ops.affine_act(y, ss, act)
    torch.testing.assert_close(_from_ndhwc(z), z_ref.detach(), atol=_tol(dtype, 5e-5, 3e-2), rtol=_tol(dtype, 1e-4, 1e-2))
    dy, dgamma, dbeta = ops.gn_act_bwd(_to_ndhwc(dz, dtype, dev), y, ss, mr, gamma.to(dev), 8, act)
    if dtype == torch.float32:
        torch.testing.assert_close(_from_ndhwc(dy), y_ref.grad, atol=1e-4, rtol=1e-3)
    else:
        # a pre-activation within rounding of zero can still flip its mask: allow <= 0.1 % outliers
        bad = ((_from_ndhwc(dy) - y_ref.grad).abs() > 3e-2 + 2e-2 * y_ref.grad.abs()).float().mean()
        assert float(bad) <= 1e-3, float(bad)
    torch.testing.assert_close(dgamma.cpu(), g_r.grad, atol=_tol(dtype, 1e-3, 2.0), rtol=_tol(dtype, 1e-4, 3e-2))
    torch.testing.assert_close(dbeta.cpu(), b_r.grad, atol=_tol(dtype, 1e-3, 2.0), rtol=_tol(dtype, 1e-4, 3e-2))


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("c,act,size", [(48, "relu", (8, 8, 8)), (16, "leakyrelu", (4, 6, 10)), (96, "relu", (4, 4, 4))])
def test_groupnorm_bwd_with_folded_pool_backward(dtype, c, act, size):
    """brats_gn_act_bwd_pool (dz = skip gradient + max-pool backward composed inside the two GroupNorm-backward passes from
    the pieces) against maxpool2_bwd (arg-max bytes) + gn_act_bwd: f32 bit for bit (the composed dz is the same f32 value the
    pooling backward would store), 16-bit within the rounding of the dz tensor that is no longer stored."""
    from brats21_amd import ops
    dev = _dev()
    n, cin, vox = 2, 8, size[0] * size[1] * size[2]
    x = _q(_rand((n, cin, *size), 91), dtype)
    w = _q(_rand((c, cin, 3, 3, 3), 92, 0.1), dtype)
    gamma = 1.0 + 0.2 * _rand((c,), 93)
    beta = 0.1 * _rand((c,), 94)
    y, stats = ops.conv3d(_to_ndhwc(x, dtype, dev), ops.pack_weights(w.to(dev), dtype, ops.PACK_FWD), c, 3, 1, want_stats=True)
    mr, ss = ops.gn_finalize(stats, n, c, 8, vox, gamma.to(dev), beta.to(dev))
    z = ops.affine_act(y, ss, act)
    ops.maxpool2(z, want_argmax=True)
    dskip = _to_ndhwc(_q(_rand((n, c, *size), 95), dtype), dtype, dev)
    dpool = _to_ndhwc(_q(_rand((n, c, size[0] // 2, size[1] // 2, size[2] // 2), 96), dtype), dtype, dev)
    dz = ops.maxpool2_bwd(z, dpool, dx_skip=dskip)
    ref = ops.gn_act_bwd(dz, y, ss, mr, gamma.to(dev), 8, act)
    got = ops.gn_act_bwd_pool(dskip, dpool, z._pool_argmax, y, ss, mr, gamma.to(dev), 8, act)
    for name, a, b in zip(("dy", "dgamma", "dbeta"), got, ref):
        if dtype == torch.float32:
            assert torch.equal(a, b), name
        else:
            scale = float(b.float().abs().max()) + 1e-30
            assert float((a.float() - b.float()).abs().max()) <= 2e-2 * scale, name
    again = ops.gn_act_bwd_pool(dskip, dpool, z._pool_argmax, y, ss, mr, gamma.to(dev), 8, act)
    assert all(torch.equal(a, b) for a, b in zip(again, got))


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("c,act,size,k", [(48, "relu", (8, 8, 8), 3), (16, "leakyrelu", (4, 6, 10), 3), (8, "relu", (3, 5, 7), 4),
                                          (64, "relu", (4, 4, 8), 3), (96, "relu", (4, 4, 4), 2)])
def test_output_head_on_raw_convolution_output_is_bit_identical(dtype, c, act, size, k):
    """brats_gn_head_fwd (GroupNorm + act applied on load, the activation never stored) == head(affine_act(y)) bit for bit,
    in every storage type, on the MFMA form (16-bit, C <= 64) and the generic form."""
    from brats21_amd import ops
    dev = _dev()
    n, vox = 2, size[0] * size[1] * size[2]
    y = _to_ndhwc(_q(_rand((n, c, *size), 71), dtype), dtype, dev)
    ss = torch.stack([1.0 + 0.3 * _rand((n, c), 72), 0.2 * _rand((n, c), 73)], -1).contiguous().to(dev)
    hw = _rand((k, c, 1, 1, 1), 74, 0.2).to(dev)
    hb = _rand((k,), 75, 0.1).to(dev)
    ref = ops.head(ops.affine_act(y, ss, act), hw, hb, 1)
    got = ops.gn_head(y, ss, hw, hb, act)
    assert torch.equal(got, ref)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("c,act,size", [(48, "relu", (8, 8, 8)), (16, "leakyrelu", (4, 6, 10)), (8, "relu", (3, 5, 7))])
def test_groupnorm_bwd_with_folded_output_head(dtype, c, act, size):
    """brats_gn_act_bwd_head (the last layer's GroupNorm backward computing dz = W_head^T dlogits on the fly, the head's
    weight / bias gradients out of the same passes) against torch autograd of conv1x1(act(GN(y))) on the conv output as
    stored, and against the two calls it replaces (head_bwd + gn_act_bwd)."""
    from brats21_amd import ops
    dev = _dev()
    n, cin, vox = 2, 8, size[0] * size[1] * size[2]
    x = _q(_rand((n, cin, *size), 61), dtype)
    w = _q(_rand((c, cin, 3, 3, 3), 62, 0.1), dtype)
    gamma = 1.0 + 0.2 * _rand((c,), 63)
    beta = 0.1 * _rand((c,), 64)
    hw = _rand((3, c, 1, 1, 1), 65, 0.2)
    dl = _rand((n, 3, *size), 66)
    y, stats = ops.conv3d(_to_ndhwc(x, dtype, dev), ops.pack_weights(w.to(dev), dtype, ops.PACK_FWD), c, 3, 1, want_stats=True)
    mr, ss = ops.gn_finalize(stats, n, c, 8, vox, gamma.to(dev), beta.to(dev))
    z = ops.affine_act(y, ss, act)
    # torch autograd on the stored conv output (so the activation masks agree)
    y_ref = _from_ndhwc(y).detach().requires_grad_(True)
    g_r, b_r, hw_r = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True), hw.clone().requires_grad_(True)
    zn = F.group_norm(y_ref, 8, g_r, b_r, 1e-5)
    z_ref = F.relu(zn) if act == "relu" else F.leaky_relu(zn, 0.01)
    hb_r = torch.zeros(3, requires_grad=True)
    F.conv3d(z_ref, hw_r, hb_r).backward(dl)
    dy, dgamma, dbeta, dhw, dhb = ops.gn_act_bwd_head(dl.to(dev), hw.to(dev), y, ss, mr, gamma.to(dev), 8, act)
    bad = ((_from_ndhwc(dy) - y_ref.grad).abs() > _tol(dtype, 1e-4, 3e-2) + _tol(dtype, 1e-3, 2e-2) * y_ref.grad.abs()).float().mean()
    assert float(bad) <= (0.0 if dtype == torch.float32 else 1e-3), float(bad)
    torch.testing.assert_close(dgamma.cpu(), g_r.grad, atol=_tol(dtype, 1e-3, 2.0), rtol=_tol(dtype, 1e-4, 3e-2))
    torch.testing.assert_close(dbeta.cpu(), b_r.grad, atol=_tol(dtype, 1e-3, 2.0), rtol=_tol(dtype, 1e-4, 3e-2))
    torch.testing.assert_close(dhb.cpu(), hb_r.grad, atol=1e-3, rtol=1e-4)
    # the head's weight gradient reads z: in the 16-bit modes torch saw the f32 z, the kernels the same f32 z (not the stored one)
    torch.testing.assert_close(dhw.cpu(), hw_r.grad, atol=_tol(dtype, 1e-3, 5e-2), rtol=_tol(dtype, 1e-4, 1e-2))
    # the two-call path
    dz2, dhw2, dhb2 = ops.head_bwd(z, hw.to(dev), dl.to(dev), 1)
    dy2, dgamma2, dbeta2 = ops.gn_act_bwd(dz2, y, ss, mr, gamma.to(dev), 8, act)
    t = _tol(dtype, 2e-5, 2e-2)
    for name, a, b in (("dy", dy, dy2), ("dgamma", dgamma, dgamma2), ("dbeta", dbeta, dbeta2), ("dhw", dhw, dhw2), ("dhb", dhb, dhb2)):
        scale = float(b.float().abs().max()) + 1e-30
        assert float((a.float() - b.float()).abs().max()) <= t * scale, name
    again = ops.gn_act_bwd_head(dl.to(dev), hw.to(dev), y, ss, mr, gamma.to(dev), 8, act)
    assert all(torch.equal(a, b) for a, b in zip(again, (dy, dgamma, dbeta, dhw, dhb)))  # bitwise reproducible


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("c,act,size,with_avg", [(48, "relu", (8, 8, 8), False), (16, "leakyrelu", (4, 6, 10), False),
                                                  (8, "relu", (2, 6, 14), True), (96, "relu", (4, 4, 4), False)])
def test_affine_act_with_fused_maxpool_is_bit_identical(dtype, c, act, size, with_avg):
    """brats_affine_act_pool_fwd == brats_affine_act_fwd followed by brats_maxpool2_fwd, bit for bit (z, pooled, |max|)."""
    from brats21_amd import ops
    dev = _dev()
    n = 2
    y = _to_ndhwc(_q(_rand((n, c, *size), 81), dtype), dtype, dev)
    ss = torch.stack([1.0 + 0.3 * _rand((n, c), 82), 0.2 * _rand((n, c), 83)], -1).contiguous().to(dev)
    a_ref = torch.zeros(1, device=dev)
    z_ref = ops.affine_act(y, ss, act, amax=a_ref)
    p_ref = ops.maxpool2(z_ref, with_avg)
    a_got = torch.zeros(1, device=dev)
    z, pooled = ops.affine_act_pool(y, ss, act, amax=a_got, with_avg=with_avg, want_argmax=True)
    assert torch.equal(z, z_ref) and torch.equal(pooled, p_ref) and torch.equal(a_got, a_ref)
    # backward from the recorded arg-max bytes == backward that recomputes the arg-max from the window
    dy = _to_ndhwc(_q(_rand((n, c * (2 if with_avg else 1), size[0] // 2, size[1] // 2, size[2] // 2), 84), dtype), dtype, dev)
    skip = _to_ndhwc(_q(_rand((n, c, *size), 85), dtype), dtype, dev)
    for sk in (None, skip):
        ref = ops.maxpool2_bwd(z_ref, dy, dx_skip=sk, with_avg=with_avg)
        got = ops.maxpool2_bwd(z, dy, dx_skip=sk, with_avg=with_avg)  # (z carries _pool_argmax)
        assert torch.equal(got, ref)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("with_avg", [False, True])
def test_pool_fwd_bwd(dtype, with_avg):
    from brats21_amd import ops
    dev = _dev()
    x = _q(_rand((2, 16, 8, 12, 8), 21), dtype)
    x[:, :, :2] = x[:, :, :2].clamp(min=0)  # exact ties (zeros) like after ReLU
    xr = x.clone().requires_grad_(True)
    y_ref = F.max_pool3d(xr, 2, 2)
    if with_avg:
        y_ref = torch.cat([y_ref, F.avg_pool3d(xr, 2)], 1)
    dy = _q(_rand(y_ref.shape, 22), dtype)
    y_ref.backward(dy)
    skip = _q(_rand(x.shape, 23), dtype)
    xd = _to_ndhwc(x, dtype, dev, pitch=32, off=0)
    y = ops.maxpool2(xd, with_avg)
    torch.testing.assert_close(_from_ndhwc(y), y_ref.detach(), atol=_tol(dtype, 1e-6, 1e-2), rtol=0)
    dx = ops.maxpool2_bwd(xd, _to_ndhwc(dy, dtype, dev), dx_skip=_to_ndhwc(skip, dtype, dev, pitch=32, off=16), with_avg=with_avg)
    torch.testing.assert_close(_from_ndhwc(dx), xr.grad + skip, atol=_tol(dtype, 1e-6, 3e-2), rtol=_tol(dtype, 0, 1e-2))
    # the forward that records the arg-max bytes + the backward that reads them: bit-identical to the pair above
    xd2 = xd.clone()
    y2 = ops.maxpool2(xd2, with_avg, want_argmax=True)
    assert torch.equal(y2, y) and xd2._pool_argmax.dtype == torch.uint8
    dx2 = ops.maxpool2_bwd(xd2, _to_ndhwc(dy, dtype, dev), dx_skip=_to_ndhwc(skip, dtype, dev, pitch=32, off=16), with_avg=with_avg)
    assert torch.equal(dx2, dx)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("size,scale", [((4, 4, 4), 2), ((2, 6, 4), 2), ((4, 4, 4), 4), ((3, 5, 6), 2), ((8, 10, 12), 2)])
def test_upsample_fwd_bwd(dtype, size, scale):
    from brats21_amd import ops
    dev = _dev()
    x = _q(_rand((2, 8, *size), 31), dtype)
    xr = x.clone().requires_grad_(True)
    y_ref = F.interpolate(xr, scale_factor=scale, mode="trilinear", align_corners=True)
    dy = _q(_rand(y_ref.shape, 32), dtype)
    y_ref.backward(dy)
    y = ops.upsample(_to_ndhwc(x, dtype, dev), scale)
    torch.testing.assert_close(_from_ndhwc(y), y_ref.detach(), atol=_tol(dtype, 1e-5, 2e-2), rtol=_tol(dtype, 1e-5, 1e-2))
    dx = ops.upsample_bwd(_to_ndhwc(dy, dtype, dev, pitch=24, off=8), scale)
    torch.testing.assert_close(_from_ndhwc(dx), xr.grad, atol=_tol(dtype, 2e-5, 8e-2), rtol=_tol(dtype, 1e-5, 3e-2))


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("c,scale", [(8, 1), (16, 2), (48, 8)])
def test_head_fwd_bwd(dtype, c, scale):
    from brats21_amd import ops
    dev = _dev()
    size = (4, 4, 4)
    x = _q(_rand((2, c, *size), 41), dtype)
    w = _rand((3, c, 1, 1, 1), 42, 0.2)
    b = _rand((3,), 43, 0.1)
    xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    o_ref = F.conv3d(xr, wr, br)
    if scale > 1:
        o_ref = F.interpolate(o_ref, scale_factor=scale, mode="trilinear", align_corners=True)
    do = _rand(o_ref.shape, 44)
    o_ref.backward(do)
    xd = _to_ndhwc(x, dtype, dev)
    o = ops.head(xd, w.to(dev), b.to(dev), scale)
    torch.testing.assert_close(o.cpu(), o_ref.detach(), atol=1e-5, rtol=1e-5)
    dx, dw, db = ops.head_bwd(xd, w.to(dev), do.to(dev), scale)
    torch.testing.assert_close(_from_ndhwc(dx), xr.grad, atol=_tol(dtype, 1e-4, 2e-2), rtol=_tol(dtype, 1e-4, 1e-2))
    torch.testing.assert_close(dw.cpu(), wr.grad, atol=1e-3, rtol=1e-4)
    torch.testing.assert_close(db.cpu(), br.grad, atol=1e-3, rtol=1e-4)


def test_layout_roundtrip():
    from brats21_amd import ops
    dev = _dev()
    x = _rand((2, 4, 8, 8, 8), 51).to(dev)
    for dtype, cpad in ((torch.float32, 4), (torch.bfloat16, 8)):
        t = ops.ncdhw_to_ndhwc(x, dtype, cpad)
        assert t.shape == (2, 8, 8, 8, cpad)
        back = ops.ndhwc_to_ncdhw(t[..., :4])
        torch.testing.assert_close(back, x.to(dtype).float(), atol=0, rtol=0)
        if cpad > 4:
            assert float(t[..., 4:].float().abs().max()) == 0.0


@pytest.mark.parametrize("jaccard", [False, True])
def test_fused_dice_matches_torch_dice(jaccard):
    """Fused HIP Dice passes == the PyTorch Dice of brats21_amd.losses == the oracle's dice_loss."""
    from brats21_amd.losses import DiceLoss, deep_supervision_loss, fused_deep_supervision_dice
    from oracle import unet
    dev = _dev()
    g = torch.Generator().manual_seed(5)
    heads = [torch.randn(2, 3, 8, 12, 10, generator=g).to(dev).requires_grad_(True) for _ in range(3)]
    t = (torch.rand(2, 3, 8, 12, 10, generator=g) > 0.6).float().to(dev)
    ref, _ = deep_supervision_loss(DiceLoss(jaccard=jaccard), (heads[0], heads[1:]), t)
    gref = torch.autograd.grad(ref, heads)
    loss = fused_deep_supervision_dice((heads[0], heads[1:]), t, jaccard=jaccard)
    gfused = torch.autograd.grad(loss, heads)
    assert abs(loss.item() - ref.item()) < 1e-6
    cpu = torch.stack([unet.dice_loss(h.detach().cpu(), t.cpu(), jaccard) for h in heads]).mean()
    assert abs(loss.item() - cpu.item()) < 1e-6
    for a, b in zip(gfused, gref):
        torch.testing.assert_close(a, b, atol=1e-9, rtol=1e-4)


@pytest.mark.parametrize("cin,cin2,cout,n,size", [
    (48, 0, 48, 2, (32, 64, 64)),
    (48, 48, 48, 3, (34, 62, 66)),     # two sources, ragged in z / y / x
    (96, 0, 96, 2, (32, 32, 64)),      # 2 x 2 channel blocks
    (8, 0, 48, 2, (64, 64, 64)),       # the first layer: one 16-channel ci block, 8 real channels
    (64, 0, 64, 2, (32, 32, 64)),      # width 64: 64 co x 32 ci blocks (LDS-DMA form only)
    (32, 32, 128, 3, (18, 30, 66)),    # ... over a 32 | 32 concat, ragged in z / y / x
    (8, 0, 64, 2, (64, 64, 64)),       # the first layer of a width-64 network: 64 co x 16 ci blocks (round 5; 160-byte dY voxel stride)
    (8, 0, 128, 3, (18, 30, 66)),      # ... two co blocks, ragged in z / y / x
])
def test_conv3d_wgrad_alltaps_kernel_matches_tapplane_kernel(cin, cin2, cout, n, size):
    """The all-taps wgrad kernel (one 8-wave workgroup per CU, X tile with z halo staged once) against the tap-plane
    kernel on the same inputs (f32 split-K sums in a different order: 1e-5 relative) and against torch autograd."""
    from brats21_amd import _lib, ops
    dev = _dev()
    dt = torch.bfloat16
    g = torch.Generator().manual_seed(13)
    x = torch.randn((n, *size, cin), generator=g).to(dev).to(dt)
    x2 = torch.randn((n, *size, cin2), generator=g).to(dev).to(dt) if cin2 else None
    dy = (torch.randn((n, *size, cout), generator=g) * 0.1).to(dev).to(dt)
    lib = _lib.lib()
    res = {}
    for mode in (0, 1):
        old = lib.brats_conv3d_set_wgrad_alltaps(mode)
        try:
            dw, db = ops.conv3d_wgrad(x, dy, 3, 1, want_dbias=True, x2=x2)
            torch.cuda.synchronize()
        finally:
            lib.brats_conv3d_set_wgrad_alltaps(old)
        res[mode] = (dw, db)
    scale = float(res[0][0].abs().max())
    assert scale > 0
    assert float((res[0][0] - res[1][0]).abs().max()) <= 2e-5 * scale
    assert torch.equal(res[0][1], res[1][1])
    # torch autograd on sample 0 only would not see the other samples' contribution: use a small sub-problem instead
    xs = x[:1, :12, :12, :16] if x2 is None else torch.cat([x[:1, :12, :12, :16], x2[:1, :12, :12, :16]], -1)
    dys = dy[:1, :12, :12, :16]
    old = lib.brats_conv3d_set_wgrad_alltaps(1)
    try:
        dws, _ = ops.conv3d_wgrad(xs.contiguous(), dys.contiguous(), 3, 1)  # too few tiles: falls back to the tap-plane kernel
    finally:
        lib.brats_conv3d_set_wgrad_alltaps(old)
    w = torch.zeros((cout, cin + cin2, 3, 3, 3), requires_grad=True)
    F.conv3d(xs.float().cpu().permute(0, 4, 1, 2, 3), w, None, 1, 1).backward(dys.float().cpu().permute(0, 4, 1, 2, 3))
    torch.testing.assert_close(dws.cpu(), w.grad, atol=4e-3 * float(w.grad.abs().max()), rtol=1e-2)


def test_conv3d_wgrad_alltaps_vs_torch_full_volume():
    """All-taps kernel against torch autograd on a volume large enough to select it (boundary tiles included)."""
    from brats21_amd import _lib, ops
    dev = _dev()
    dt = torch.bfloat16
    g = torch.Generator().manual_seed(17)
    n, size = 2, (32, 60, 66)
    x = torch.randn((n, *size, 48), generator=g).to(dt)
    dy = (torch.randn((n, *size, 48), generator=g) * 0.1).to(dt)
    lib = _lib.lib()
    old = lib.brats_conv3d_set_wgrad_alltaps(1)
    try:
        dw, _ = ops.conv3d_wgrad(x.to(dev), dy.to(dev), 3, 1)
        torch.cuda.synchronize()
    finally:
        lib.brats_conv3d_set_wgrad_alltaps(old)
    w = torch.zeros((48, 48, 3, 3, 3), requires_grad=True)
    torch.set_num_threads(16)
    F.conv3d(x.float().permute(0, 4, 1, 2, 3), w, None, 1, 1).backward(dy.float().permute(0, 4, 1, 2, 3))
    torch.testing.assert_close(dw.cpu(), w.grad, atol=4e-3 * float(w.grad.abs().max()), rtol=1e-2)


def test_conv3d_full_size_layer_vs_torch():
    """One 48->48 layer at the bench's full spatial size (1 x 128^3): forward, dgrad and weight gradient of the bf16
    kernels (tile igemm + all-taps wgrad, boundary tiles on every face) against torch's CPU f32 convolution."""
    from brats21_amd import ops
    dev = _dev()
    dt = torch.bfloat16
    g = torch.Generator().manual_seed(23)
    s = 128
    x = torch.relu(torch.randn((1, 48, s, s, s), generator=g)).to(dt).float()
    w = (torch.randn((48, 48, 3, 3, 3), generator=g) * (2.0 / (48 * 27)) ** 0.5).to(dt).float()
    dy = (torch.randn((1, 48, s, s, s), generator=g) * 0.1).to(dt).float()
    torch.set_num_threads(16)
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    y_ref = F.conv3d(xr, wr, None, 1, 1)
    y_ref.backward(dy)
    xd, dyd = _to_ndhwc(x, dt, dev), _to_ndhwc(dy, dt, dev)
    y, stats = ops.conv3d(xd, ops.pack_weights(w.to(dev), dt, ops.PACK_FWD), 48, 3, 1, want_stats=True)
    dx, _ = ops.conv3d(dyd, ops.pack_weights(w.to(dev), dt, ops.PACK_DGRAD), 48, 3, 1)
    dw, _ = ops.conv3d_wgrad(xd, dyd, 3, 1)
    torch.cuda.synchronize()
    torch.testing.assert_close(_from_ndhwc(y), y_ref.detach(), atol=3e-2, rtol=2e-2)     # bf16 output rounding
    torch.testing.assert_close(_from_ndhwc(dx), xr.grad, atol=2e-2, rtol=2e-2)
    scale = float(wr.grad.abs().max())
    torch.testing.assert_close(dw.cpu(), wr.grad, atol=2e-3 * scale, rtol=1e-2)           # f32 accumulation of 2M products
    # epilogue statistics = per-channel sum / sum of squares of the f32 result
    st = stats.double().sum(1).cpu()[0]
    ref1, ref2 = y_ref.detach().double().sum((0, 2, 3, 4)), (y_ref.detach().double() ** 2).sum((0, 2, 3, 4))
    torch.testing.assert_close(st[:, 0], ref1, atol=2e-2 * s ** 1.5, rtol=1e-3)
    torch.testing.assert_close(st[:, 1], ref2, atol=1e-2, rtol=1e-3)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("act", ["elu", "swish", "mish"])
def test_groupnorm_other_activations_fwd_bwd(dtype, act):
    """--act elu | swish | mish (src/arguments_train.py:49-50) through the fused norm + activation kernels against
    torch autograd of GroupNorm(8) followed by the same activation."""
    from brats21_amd import ops
    dev = _dev()
    c, size = 16, (6, 8, 10)
    fn = {"elu": F.elu, "swish": lambda t: t * torch.sigmoid(t), "mish": F.mish}[act]
    y = _q(_rand((2, c, *size), 31, 1.5), dtype)
    gamma, beta = _rand((c,), 32, 0.5) + 1.0, _rand((c,), 33, 0.3)
    dz = _q(_rand((2, c, *size), 34), dtype)
    yr, gr, br = y.clone().requires_grad_(True), gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    z_ref = fn(F.group_norm(yr, 8, gr, br, 1e-5))
    z_ref.backward(dz)
    yd = _to_ndhwc(y, dtype, dev)
    # statistics as the conv epilogue would deliver them: one "tile" per sample
    st = torch.stack([y.double().sum((2, 3, 4)), (y.double() ** 2).sum((2, 3, 4))], -1).float().to(dev)[:, None]
    vox = size[0] * size[1] * size[2]
    mean_rstd, scale_shift = ops.gn_finalize(st.contiguous(), 2, c, 8, vox, gamma.to(dev), beta.to(dev))
    z = ops.affine_act(yd, scale_shift, act)
    torch.testing.assert_close(_from_ndhwc(z), z_ref.detach(), atol=_tol(dtype, 2e-5, 3e-2), rtol=_tol(dtype, 1e-5, 2e-2))
    dy, dgamma, dbeta = ops.gn_act_bwd(_to_ndhwc(dz, dtype, dev), yd, scale_shift, mean_rstd, gamma.to(dev), 8, act)
    torch.testing.assert_close(_from_ndhwc(dy), yr.grad, atol=_tol(dtype, 5e-5, 5e-2), rtol=_tol(dtype, 1e-4, 5e-2))
    torch.testing.assert_close(dgamma.cpu(), gr.grad, atol=_tol(dtype, 2e-3, 0.15), rtol=_tol(dtype, 1e-4, 2e-2))
    torch.testing.assert_close(dbeta.cpu(), br.grad, atol=_tol(dtype, 2e-3, 0.15), rtol=_tol(dtype, 1e-4, 2e-2))


@pytest.mark.parametrize("cin,cin2,cout,n,size", [
    (48, 0, 48, 2, (32, 32, 32)),     # interior tiles
    (48, 48, 48, 1, (12, 20, 40)),    # two-source input, ragged tiles in z, y (20 = 2.5 tiles of 8 rows) and x
    (96, 0, 144, 1, (8, 12, 16)),     # three cout blocks, four 24-channel chunks, y = 1.5 tiles
])
@pytest.mark.parametrize("mode", [1])
def test_conv3d_vs8_kernel_matches_tile_kernel(cin, cin2, cout, n, size, mode):
    """The 4x8x16-tile kernel (24-channel chunks, conv_igemm_vs8.hpp) against the 4x4x16-tile kernel (48-channel chunks):
    same products in f32, a different summation order over K, so the bf16 outputs differ by at most one rounding step
    of the result; the tile statistics (f32, taken before that rounding) agree to 1e-5 relative."""
    from brats21_amd import ops, _lib
    dev = _dev()
    dt = torch.bfloat16
    x = _to_ndhwc(_rand((n, cin) + size, 31), dt, dev)
    x2 = _to_ndhwc(_rand((n, cin2) + size, 32), dt, dev) if cin2 else None
    w = _rand((cout, cin + cin2, 3, 3, 3), 33, 0.05).to(dev)
    b = _rand((cout,), 34, 0.1).to(dev)
    lib = _lib.lib()
    old = ops.set_vs8(0)
    try:
        wpk = ops.pack_weights(w, dt, ops.PACK_FWD, c1=cin if cin2 else None)
        y0, s0 = ops.conv3d(x, wpk, cout, 3, 1, bias=b, want_stats=True, x2=x2)
        ops.set_vs8(mode)
        assert ops.conv_chunk(dt, 3, 1, cin, cin2, cout) == 24
        wpk8 = ops.pack_weights(w, dt, ops.PACK_FWD, c1=cin if cin2 else None)
        assert wpk8.numel() != wpk.numel() or not torch.equal(wpk8, wpk)  # really the other layout
        y1, s1 = ops.conv3d(x, wpk8, cout, 3, 1, bias=b, want_stats=True, x2=x2)
    finally:
        ops.set_vs8(old)
    err = (y0.float() - y1.float()).abs().max().item()
    assert err <= y0.float().abs().max().item() * 2 ** -7, err
    t0, t1 = s0.sum(1), s1.sum(1)   # [n, cout, 2]: per-tile entries may be dealt differently only at ragged edges
    assert torch.allclose(t0, t1, rtol=1e-5, atol=1e-3)
    assert torch.allclose(s0, s1, rtol=1e-4, atol=1e-3)  # and each 4x4x16 entry holds the same voxels


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("cin,cin2,cout,dil,size", [
    (384, 0, 384, 1, (16, 16, 16)),   # encoder4 / its input gradient: 256 workgroups of 48 couts
    (384, 0, 384, 2, (16, 16, 16)),   # bottom (dilation 2)
    (384, 384, 192, 1, (16, 16, 16)), # bottom_2 over the virtual concat [down4 | bottom]: 128 workgroups
    (192, 0, 96, 1, (10, 12, 20)),    # ragged tiles on every axis
])
def test_conv3d_eight_wave_k_parity_form_matches_four_wave_form(dtype, cin, cin2, cout, dil, size):
    """conv_igemm_kernel's KP form (round 6: grids of at most one workgroup per CU run EIGHT waves -- two K-parity teams on one LDS
    tile and one output tile, team 1's partial sums through LDS) against the 4-wave form of the same launch at the networks'
    16^3-level shapes: same packed weights, same products in f32; the sum over K is taken as two partial sums, so the 16-bit
    outputs agree to one rounding step of the result and the tile statistics to 1e-5; and against torch's CPU f32 convolution.
    The backward-statistics form (the input gradient of a block's second convolution) takes the same switch."""
    from brats21_amd import ops
    dev = _dev()
    n = 2
    x = _q(_rand((n, cin) + size, 71), dtype)
    x2 = _q(_rand((n, cin2) + size, 72), dtype) if cin2 else None
    w = _q(_rand((cout, cin + cin2, 3, 3, 3), 73, (2.0 / ((cin + cin2) * 27)) ** 0.5), dtype)
    b = _rand((cout,), 74, 0.1)
    xd, x2d = _to_ndhwc(x, dtype, dev), (_to_ndhwc(x2, dtype, dev) if cin2 else None)
    wpk = ops.pack_weights(w.to(dev), dtype, ops.PACK_FWD, dil=dil, c1=cin if cin2 else None)
    assert ops.conv_chunk(dtype, 3, dil, cin, cin2, cout) == 48
    res = {}
    old = ops.set_kp(0)
    try:
        for mode in (0, 1):
            ops.set_kp(mode)
            res[mode] = ops.conv3d(xd, wpk, cout, 3, dil, bias=b.to(dev), want_stats=True, x2=x2d)
            if not cin2 and ops.conv_bstats_ok(dtype, dil, cin, cout, "relu"):
                # the same launch as an input gradient with the backward statistics of a GroupNorm + ReLU unit in its epilogue
                by = _to_ndhwc(_q(_rand((n, cout) + size, 75), dtype), dtype, dev)
                ss = torch.stack([1.0 + 0.1 * _rand((n, cout), 76), 0.1 * _rand((n, cout), 77)], -1).contiguous().to(dev)
                res[mode] += ops.conv3d_bstats(xd, wpk, cout, dil, by, ss, "relu")
    finally:
        ops.set_kp(old)
    y0, s0, y1, s1 = res[0][0], res[0][1], res[1][0], res[1][1]
    ulp = 2.0 ** -7 if dtype == torch.bfloat16 else 2.0 ** -10
    assert float((y0.float() - y1.float()).abs().max()) <= float(y0.float().abs().max()) * ulp
    assert not torch.equal(s0, s1) or torch.equal(y0, y1)  # (really two code paths: the f32 sums differ in their last bits)
    assert torch.allclose(s0, s1, rtol=1e-5, atol=1e-5 * float(s0.abs().max()))
    if len(res[0]) == 4:
        assert float((res[0][2].float() - res[1][2].float()).abs().max()) <= float(res[0][2].float().abs().max()) * ulp
        assert torch.allclose(res[0][3], res[1][3], rtol=1e-4, atol=1e-5 * float(res[0][3].abs().max()))
    torch.set_num_threads(16)
    ref = F.conv3d(torch.cat([x, x2], 1) if cin2 else x, w, b, 1, dil, dil)
    torch.testing.assert_close(_from_ndhwc(y1), ref, atol=_tol(dtype, 2e-5, 3e-2), rtol=_tol(dtype, 1e-5, 2e-2))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("n,size", [(1, (64, 64, 64)), (2, (32, 48, 80))])
def test_conv3d_first_layer_weight_gradient_vs_torch(dtype, n, size):
    """The first layer's weight gradient (8 padded input channels -> 48) on the LDS-DMA all-taps form (round 6: two X buffers of
    16-channel rows of which only the real 8-channel piece is fetched, the next tile's loads behind the MFMA phase) against torch
    autograd on the CPU; the gradient columns of the four padding channels are exactly zero (the padding channels are zero)."""
    from brats21_amd import ops
    dev = _dev()
    x = _q(_rand((n, 4) + size, 81), dtype)
    dy = _q(_rand((n, 48) + size, 82, 0.1), dtype)
    wr = torch.zeros(48, 4, 3, 3, 3, requires_grad=True)
    torch.set_num_threads(16)
    F.conv3d(x, wr, None, 1, 1).backward(dy)
    xin = torch.zeros(n, *size, 8, dtype=dtype, device=dev)
    xin[..., :4] = _to_ndhwc(x, dtype, dev)
    dw, _ = ops.conv3d_wgrad(xin, _to_ndhwc(dy, dtype, dev), 3, 1)
    assert tuple(dw.shape) == (48, 8, 3, 3, 3)
    assert float(dw[:, 4:].abs().max()) == 0.0
    ref = wr.grad
    err = float((dw[:, :4].cpu() - ref).abs().max())
    assert err <= 2e-3 * float(ref.abs().max()), (err, float(ref.abs().max()))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("n,size,with_bias", [(2, (32, 64, 64), True), (1, (64, 64, 128), False), (1, (36, 52, 240), True)])
def test_conv3d_first_layer_kernel(dtype, n, size, with_bias):
    """The first layer (4 modalities padded to 8 channels -> 48, networks/equiunet2020.py:424) on the persistent kernel of
    csrc/conv_igemm_first.hpp (weights in registers for the whole launch, output tile through LDS into 1 KB contiguous stores):
    (i) bit-identical outputs to the 4x8x16-tile kernel, which the same call takes when the output is a channel slice of a wider
    buffer (same packed weights, same K order, f32 accumulation from the bias), tile statistics equal to 1e-5; every face of
    the volume is a boundary tile, the third case has 13 and 15 tiles per row (odd counts for the persistent walk);
    (ii) against torch's CPU f32 convolution."""
    from brats21_amd import ops
    dev = _dev()
    x = _q(_rand((n, 4) + size, 61), dtype)
    w = _q(_rand((48, 4, 3, 3, 3), 62, (2.0 / (4 * 27)) ** 0.5), dtype)
    b = _rand((48,), 63, 0.1) if with_bias else None
    xin = torch.zeros(n, *size, 8, dtype=dtype, device=dev)      # channels 4..7 are the zero padding the networks add
    xin[..., :4] = _to_ndhwc(x, dtype, dev)
    w8 = torch.zeros(48, 8, 3, 3, 3)
    w8[:, :4] = w
    wpk = ops.pack_weights(w8.to(dev), dtype, ops.PACK_FWD)
    bd = b.to(dev) if with_bias else None
    y_new, s_new = ops.conv3d(xin, wpk, 48, 3, 1, bias=bd, want_stats=True)              # dense output: the new kernel
    wide = torch.zeros(n, *size, 96, dtype=dtype, device=dev)
    y_old, s_old = ops.conv3d(xin, wpk, 48, 3, 1, bias=bd, want_stats=True, out=wide[..., 16:64])  # slice: the tile kernel
    assert torch.equal(y_new, y_old.contiguous())
    assert torch.allclose(s_new, s_old, rtol=1e-5, atol=1e-4)
    torch.set_num_threads(16)
    ref = F.conv3d(x, w, b, 1, 1)
    torch.testing.assert_close(_from_ndhwc(y_new), ref, atol=_tol(dtype, 2e-5, 3e-2), rtol=_tol(dtype, 1e-5, 2e-2))


@pytest.mark.parametrize("mode", [1])
@pytest.mark.parametrize("cin,cin2,cout,n,size,pitch", [
    (48, 48, 48, 1, (9, 21, 37), None),   # two-source, ragged in z, y and x (every boundary mask of the halo staging)
    (48, 0, 48, 2, (16, 16, 32), 64),     # channel-slice input views (pitch 64 > 48 channels), batch 2
    (32, 64, 48, 1, (4, 8, 16), None),    # sources of different widths (c1 = 32: 24 does not divide it -> the 4x4x16-tile kernel)
])
def test_conv3d_cout48_kernels_vs_torch(cin, cin2, cout, n, size, pitch, mode):
    """The Cout = 48 kernels of the 128^3 level straight against torch's CPU f32 convolution: forward with
    bias + tile statistics, two-source (virtual concat) input, ragged volumes, channel-slice views."""
    from brats21_amd import ops
    dev = _dev()
    dt = torch.bfloat16
    vs8 = not (cin % 24 or cin2 % 24)
    x = _q(_rand((n, cin) + size, 41), dt)
    x2 = _q(_rand((n, cin2) + size, 42), dt) if cin2 else None
    w = _q(_rand((cout, cin + cin2, 3, 3, 3), 43, 0.05), dt)
    b = _rand((cout,), 44, 0.1)
    torch.set_num_threads(16)
    y_ref = F.conv3d(torch.cat([x, x2], 1) if cin2 else x, w, b, 1, 1)
    xd = _to_ndhwc(x, dt, dev, pitch, 8 if pitch else 0)
    x2d = _to_ndhwc(x2, dt, dev) if cin2 else None
    old = ops.set_vs8(mode)
    try:
        assert ops.conv_chunk(dt, 3, 1, cin, cin2, cout) == (24 if vs8 else 32)
        wpk = ops.pack_weights(w.to(dev), dt, ops.PACK_FWD, c1=cin if cin2 else None)
        y, stats = ops.conv3d(xd, wpk, cout, 3, 1, bias=b.to(dev), want_stats=True, x2=x2d)
        torch.cuda.synchronize()
    finally:
        ops.set_vs8(old)
    torch.testing.assert_close(_from_ndhwc(y), y_ref, atol=2e-2, rtol=2e-2)
    st = stats.double().sum(1).cpu()
    ref1, ref2 = y_ref.double().sum((2, 3, 4)), (y_ref.double() ** 2).sum((2, 3, 4))
    torch.testing.assert_close(st[..., 0], ref1, atol=1e-2 * float(ref2.max()) ** 0.5, rtol=1e-3)
    torch.testing.assert_close(st[..., 1], ref2, atol=1e-3, rtol=1e-3)


def _pack_layout_oracle(wn, mode, cin_off, kdim, rows, ck, bf):
    """numpy statement of the fragment layout out[chunk][ms][row16][lane][e] (csrc/conv_igemm.hpp) as f32 values."""
    import numpy as np
    taps = wn.shape[2]
    rows16 = (rows + 15) // 16
    epl = 8 if bf else 4
    units = taps * (ck // 8) if bf else taps * ck
    ms_n = (units + 3) // 4 if bf else (units // 4 + 3) // 4
    # A[row][kc][tap]: the GEMM's left operand (rows zero-padded to whole 16-row fragments)
    A = np.zeros((rows16 * 16, kdim, taps), dtype=np.float32)
    if mode == 0:
        A[:rows] = wn[:, cin_off:cin_off + kdim]
    else:
        A[:rows] = wn[:, cin_off:cin_off + rows, ::-1].transpose(1, 0, 2)
    chunk, ms, ft, lane, e = np.meshgrid(np.arange(kdim // ck), np.arange(ms_n), np.arange(rows16), np.arange(64), np.arange(epl),
                                         indexing="ij")
    q, r = lane >> 4, lane & 15
    if bf:
        g = 4 * ms + q
        tap, kc = g // (ck // 8), chunk * ck + (g % (ck // 8)) * 8 + e
    else:
        g = 4 * (4 * ms + e) + q
        tap, kc = g // ck, chunk * ck + g % ck
    valid = g < units
    return np.where(valid, A[ft * 16 + r, kc, np.minimum(tap, taps - 1)], np.float32(0))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32, "x3"])
@pytest.mark.parametrize("cout,cin,k,mode,cin_off,cin_cnt", [
    (48, 48, 3, 0, 0, None),     # forward layout, 48-channel chunk (or 24 / 16 by the kernel switch)
    (48, 96, 3, 1, 0, None),     # input-gradient layout: rows = input channels, taps flipped
    (40, 32, 3, 0, 0, None),     # rows not a multiple of 16 (zero-padded fragment rows)
    (96, 64, 3, 1, 16, 32),      # a channel slice of the weight tensor (two-source layers pack their halves separately)
    (24, 48, 1, 0, 0, None),     # 1x1x1
    (8, 8, 3, 0, 0, None),       # 8-channel chunk (first layer after padding)
    (24, 8, 3, 1, 0, None),      # input-gradient layout with 8 rows: the second 8-row half of the fragment is padding
    (384, 192, 3, 0, 0, None),   # the widest layers of EquiUnet-48 (8 / 4 chunks x 24 / 12 row groups)
    (192, 384, 3, 1, 192, 192),
    (16, 10, 3, 0, 0, 8),        # runs that do not start on 16-byte boundaries (the element-wise staging path)
    (16, 10, 3, 1, 0, None),
    (16, 6, 3, 1, 0, None),
])
def test_pack_weights_matches_layout_oracle(dtype, cout, cin, k, mode, cin_off, cin_cnt):
    """brats_conv3d_pack_weights (one workgroup per (K chunk, 16-row group, 8-row half): contiguous runs of the torch
    layout staged through LDS) against a numpy statement of the fragment layout, bit for bit; "x3": the split-precision
    layout out[chunk][ms][row16][hi | lo][lane][e] with hi = rn_fp16(w), lo = rn_fp16(w - hi)."""
    import numpy as np
    from brats21_amd import ops
    dev = _dev()
    w = _rand((cout, cin, k, k, k), 71)
    x3 = dtype == "x3"
    if x3 and k != 3:
        pytest.skip("split precision: 3x3x3 only")
    cnt = cin - cin_off if cin_cnt is None else cin_cnt
    kdim, rows = (cnt, cout) if mode == 0 else (cout, cnt)
    pdt = ops.X3F if x3 else dtype
    packed = ops._pack_weights(w.to(dev), pdt, mode, None, cin_off, cin_cnt, 1, None)
    ck = ops.conv_chunk(pdt, k, 1, kdim, 0, rows)
    torch.cuda.synchronize()
    ref = torch.from_numpy(_pack_layout_oracle(w.numpy().reshape(cout, cin, k ** 3), mode, cin_off, kdim, rows, ck, dtype != torch.float32))
    if x3:
        hi = ref.to(torch.float16)
        lo = (ref - hi.float()).to(torch.float16)
        want = torch.stack([hi, lo], dim=3)  # [chunk][ms][row16][hi | lo][lane][e]
        got = packed.view(torch.float16).cpu().reshape(want.shape)
    else:
        want = ref.to(dtype)
        got = packed.view(dtype).cpu().reshape(want.shape)
    assert torch.equal(got.view(torch.int16 if dtype != torch.float32 else torch.int32), want.view(torch.int16 if dtype != torch.float32 else torch.int32))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("c2,c1,dil,n,size,act", [
    (48, 48, 1, 2, (8, 16, 32), "relu"),        # the 4x8x16-tile kernel (24-channel chunks), whole tiles
    (48, 48, 1, 1, (6, 10, 20), "leakyrelu"),   # ragged volume: masked statistics, partial tiles in every direction
    (96, 96, 1, 1, (8, 8, 32), "relu"),         # 4x4x16 tile, 96-row workgroups
    (96, 192, 1, 1, (4, 8, 16), "relu"),        # small grid: the y-split roles
    (384, 384, 2, 1, (4, 4, 16), "relu"),       # the dilated bottom block
    (48, 96, 1, 1, (4, 12, 16), "relu"),
])
def test_conv3d_bstats_matches_two_pass_groupnorm_backward(dtype, c2, c1, dil, n, size, act):
    """brats_conv3d_fwd_bstats + brats_gn_act_bwd_tiles (GroupNorm backward's first pass taken from the accumulators of the
    input-gradient convolution) against brats_conv3d_fwd + brats_gn_act_bwd on the same data: the same dz bit for bit; the
    sums differ only by where dz was rounded to 16 bits (the two-pass form sums the stored values, the fused form the f32
    accumulators) and by f64 against f32 partial sums."""
    from brats21_amd import ops
    dev = _dev()
    d, h, w = size
    groups = 8
    assert ops.conv_bstats_ok(dtype, dil, c2, c1, act)
    y1 = _rand((n, d, h, w, c1), 5).to(dev).to(dtype)
    gamma, beta = (1.0 + 0.3 * _rand((c1,), 6)).to(dev), (0.2 * _rand((c1,), 7)).to(dev)
    yf = y1.float().reshape(n, -1, groups, c1 // groups)
    mean = yf.mean((1, 3))
    rstd = (yf.var((1, 3), unbiased=False) + 1e-5).rsqrt()
    mean_rstd = torch.stack([mean, rstd], -1).contiguous()                                   # [n][g][2]
    sc = rstd.repeat_interleave(c1 // groups, 1) * gamma
    scale_shift = torch.stack([sc, beta - mean.repeat_interleave(c1 // groups, 1) * sc], -1).contiguous()  # [n][c][2]
    dy2 = (_rand((n, d, h, w, c2), 8) * 0.05).to(dev).to(dtype)
    w2 = (_rand((c2, c1, 3, 3, 3), 9) * 0.05).to(dev)
    wpk = ops.pack_weights(w2, dtype, ops.PACK_DGRAD, dil=dil)
    dz_a, _ = ops.conv3d(dy2, wpk, c1, 3, dil)
    dy_a, dg_a, db_a = ops.gn_act_bwd(dz_a, y1, scale_shift, mean_rstd, gamma, groups, act)
    dz_b, tiles = ops.conv3d_bstats(dy2, wpk, c1, dil, y1, scale_shift, act)
    dy_b, dg_b, db_b = ops.gn_act_bwd_tiles(tiles, dz_b, y1, scale_shift, mean_rstd, gamma, groups, act)
    torch.cuda.synchronize()
    assert torch.equal(dz_a, dz_b)
    # f64 statements of the sums: from the stored 16-bit dz (what the two-pass form adds up) and from the unrounded dz (what
    # the fused form sees in its accumulators; taken here from the exact-f32 kernel on the same 16-bit operands)
    w2r = w2.to(dtype).float()
    dz32, _ = ops.conv3d(dy2.float(), ops.pack_weights(w2r, torch.float32, ops.PACK_DGRAD, dil=dil), c1, 3, dil)
    pre = y1.double() * scale_shift[..., 0].double()[:, None, None, None, :] + scale_shift[..., 1].double()[:, None, None, None, :]
    mask = torch.where(pre > 0, 1.0, 0.0 if act == "relu" else 0.01)
    xhat = ((y1.double().reshape(n, -1, groups, c1 // groups) - mean.double()[:, None, :, None]) * rstd.double()[:, None, :, None]).reshape(pre.shape)

    def sums(dz):
        u = dz.double() * mask
        return u.sum((0, 1, 2, 3)), (u * xhat).sum((0, 1, 2, 3))

    db_st, dg_st = sums(dz_a)
    db_ex, dg_ex = sums(dz32)
    scale = float(dg_ex.abs().max()) + float((dz32.double().abs() * xhat.abs()).sum((0, 1, 2, 3)).max()) * 1e-3
    assert float((dg_a.double() - dg_st).abs().max()) <= 2e-5 * scale and float((db_a.double() - db_st).abs().max()) <= 2e-5 * scale
    assert float((dg_b.double() - dg_ex).abs().max()) <= 2e-5 * scale, float((dg_b.double() - dg_ex).abs().max()) / scale
    assert float((db_b.double() - db_ex).abs().max()) <= 2e-5 * scale, float((db_b.double() - db_ex).abs().max()) / scale
    # the two forms against each other: the rounding of dz to 16 bits
    eps16 = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    torch.testing.assert_close(dg_b, dg_a, atol=eps16 * scale, rtol=0)
    torch.testing.assert_close(db_b, db_a, atol=eps16 * scale, rtol=0)
    dscale = float(dy_a.float().abs().max())
    torch.testing.assert_close(dy_b.float(), dy_a.float(), atol=4 * eps16 * dscale, rtol=0)
