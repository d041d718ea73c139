"""-m gpu: the e4m3 convolution path (BASELINE.json configs[4]) against a torch emulation of the same quantisation:
x and w are rounded to OCP e4m3 with the kernel's power-of-two scales (torch.float8_e4m3fn casts, round-to-nearest-even)
and convolved in f64 on the CPU.  What remains is f32 accumulation order + the bf16 rounding of the output, so the
tolerances below are those of a bf16 kernel -- the fp8 rounding itself is reproduced exactly, not tolerated.
Deviation of the fp8 path from the bf16 path (the price of e4m3) is measured and bounded separately."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch.device("cuda:0")


def _rand(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=g) * scale


def _scale_from_amax(amax):
    """The kernel's rule (conv_igemm_f8.hpp f8_scale_from_amax): amax / scale lands in [128, 256)."""
    if amax == 0:
        return 2.0 ** -126
    return 2.0 ** (math.floor(math.log2(amax)) - 7)


def _q8(x, scale):
    return (x / scale).to(torch.float8_e4m3fn).float() * scale


def _q8_rows(w):
    """per-output-row power-of-two scales, w: [rows, ...]"""
    out = torch.empty_like(w)
    for r in range(w.shape[0]):
        out[r] = _q8(w[r], _scale_from_amax(float(w[r].abs().max())))
    return out


def _ndhwc(x, dev):
    return x.permute(0, 2, 3, 4, 1).contiguous().to(dev).to(torch.bfloat16)


def _ncdhw(t):
    return t.float().cpu().permute(0, 4, 1, 2, 3).contiguous()


@pytest.mark.parametrize("cin,cin2,cout,dil,size", [
    (48, 0, 48, 1, (8, 8, 16)),     # the width-48 layer: y-split roles, 3 units per tap
    (48, 48, 48, 1, (12, 8, 20)),   # two-source input (decoder), ragged tiles in z and x
    (48, 0, 96, 1, (8, 8, 32)),     # 96 couts: small grid -> y-split with 2 cout blocks
    (96, 0, 192, 2, (8, 8, 16)),    # dilation 2, two chunks
    (16, 0, 32, 1, (8, 8, 16)),     # CK = 16, NF = 2 y-split
    (32, 32, 64, 1, (4, 8, 16)),    # CK = 32 (padded LDS stride), NF = 2 cout-split
    (64, 0, 16, 1, (4, 4, 4)),      # NF = 1, tile larger than the volume
])
def test_conv3d_f8_matches_quantised_reference(cin, cin2, cout, dil, size):
    from brats21_amd import ops
    from brats21_amd._lib import PACK_FWD
    dev = _dev()
    n = 2
    x = _rand((n, cin + cin2) + size, 1)
    x[:, :, 0, 0, 0] *= 6.0  # an outlier sets the scale
    x = x.to(torch.bfloat16).float()
    w = _rand((cout, cin + cin2, 3, 3, 3), 2, 0.05)
    b = _rand((cout,), 3, 0.1)
    x1 = _ndhwc(x[:, :cin], dev)
    x2 = _ndhwc(x[:, cin:], dev) if cin2 else None
    wpk = ops.pack_weights_f8(w.to(dev), PACK_FWD, c1=cin if cin2 else None)
    y, stats = ops.conv3d_f8(x1, wpk, cout, dil, bias=b.to(dev), want_stats=True, x2=x2)
    torch.cuda.synchronize()
    xs = _scale_from_amax(float(x.abs().max()))
    ref = F.conv3d(_q8(x, xs).double(), _q8_rows(w).double(), b.double(), padding=dil, dilation=dil).float()
    got = _ncdhw(y)
    err = (got - ref).abs().max().item()
    bound = ref.abs().max().item() * 2 ** -8 + 1e-3   # bf16 output rounding (half an ulp of the largest value) + f32 order
    assert err <= bound, f"max |err| {err} > {bound}"
    # the tile statistics are taken from the f32 result before the bf16 rounding: compare sums tightly
    s = stats.float().cpu().sum(1)  # [n, cout, 2]
    ref_s1 = ref.sum((2, 3, 4))
    ref_s2 = (ref * ref).sum((2, 3, 4))
    assert torch.allclose(s[..., 0], ref_s1, rtol=1e-4, atol=1e-2)
    assert torch.allclose(s[..., 1], ref_s2, rtol=1e-4, atol=1e-2)


def test_conv3d_f8_given_amax_static_scale_and_dgrad_split():
    from brats21_amd import ops
    from brats21_amd._lib import PACK_DGRAD
    dev = _dev()
    n, cin, c1, cout, size = 1, 96, 48, 48, (8, 8, 16)
    dy = _rand((n, cout) + size, 5, 1e-3).to(torch.bfloat16).float()   # gradient-sized values: the scale must adapt
    w = _rand((cout, cin, 3, 3, 3), 6, 0.05)
    dyd = _ndhwc(dy, dev)
    wpk = ops.pack_weights_f8(w.to(dev), PACK_DGRAD)
    amax = ops.absmax(dyd)
    assert float(amax.item()) == float(dy.abs().max())
    (dx1, dx2), _ = ops.conv3d_f8(dyd, wpk, cin, 1, split=c1, amax=amax)
    xs = _scale_from_amax(float(dy.abs().max()))
    # dgrad = conv of dy with the transposed, flipped kernel; rows of the dgrad GEMM are input channels
    wt = w.permute(1, 0, 2, 3, 4).flip(2, 3, 4).contiguous()
    ref = F.conv3d(_q8(dy, xs).double(), _q8_rows(wt).double(), None, padding=1).float()
    got = torch.cat([_ncdhw(dx1), _ncdhw(dx2)], 1)
    err = (got - ref).abs().max().item()
    assert err <= ref.abs().max().item() * 2 ** -8 + 1e-7, err
    # static scale: same result when the given power of two equals the dynamic one
    (sx1, sx2), _ = ops.conv3d_f8(dyd, wpk, cin, 1, split=c1, xscale=xs)
    assert torch.equal(sx1, dx1) and torch.equal(sx2, dx2)


def test_absmax_side_outputs_of_norm_kernels():
    from brats21_amd import ops
    dev = _dev()
    n, c, size = 2, 48, (8, 8, 16)
    y = _ndhwc(_rand((n, c) + size, 7, 2.0), dev)
    ss = torch.stack([_rand((n, c), 8).abs() + 0.5, _rand((n, c), 9)], -1).to(dev).contiguous()
    for act in ("relu", "leakyrelu", "swish"):
        amax = torch.zeros(1, device=dev)
        z = ops.affine_act(y, ss, act, amax=amax)
        assert float(amax.item()) == float(z.float().abs().max().item()), act
    # GroupNorm backward: |max| of dy
    stats_in = y.float()
    mean = stats_in.reshape(n, -1, 8, c // 8).mean((1, 3))
    var = stats_in.reshape(n, -1, 8, c // 8).var((1, 3), unbiased=False)
    mean_rstd = torch.stack([mean, (var + 1e-5).rsqrt()], -1).contiguous()
    gamma = (_rand((c,), 10).abs() + 0.5).to(dev)
    beta = _rand((c,), 11).to(dev)
    rs = mean_rstd[..., 1].repeat_interleave(c // 8, 1) * gamma
    ss2 = torch.stack([rs, beta - mean_rstd[..., 0].repeat_interleave(c // 8, 1) * rs], -1).contiguous()
    dz = _ndhwc(_rand((n, c) + size, 12, 1e-2), dev)
    amax = torch.zeros(1, device=dev)
    dy, _, _ = ops.gn_act_bwd(dz, y, ss2, mean_rstd, gamma, 8, "relu", amax=amax)
    assert float(amax.item()) == float(dy.float().abs().max().item())
    ref_dy, _, _ = ops.gn_act_bwd(dz, y, ss2, mean_rstd, gamma, 8, "relu")
    assert torch.equal(dy, ref_dy)
