"""Synthetic BraTS-shaped data for benchmarks: image = i.i.d. N(0,1) zeroed outside a
centred ellipsoid (matches the non-zero z-score normalisation of utils/transforms.py:364-385),
seed 1234+rank; target = three nested spheres WT > TC > ET as {0,1} float channels."""
import torch


def _grid(size, device):
    d, h, w = size
    zz = torch.linspace(-1, 1, d, device=device).view(d, 1, 1)
    yy = torch.linspace(-1, 1, h, device=device).view(1, h, 1)
    xx = torch.linspace(-1, 1, w, device=device).view(1, 1, w)
    return zz, yy, xx


def random_image(n, c, size, seed=1234, device="cpu"):
    g = torch.Generator(device="cpu").manual_seed(seed)
    x = torch.randn((n, c) + tuple(size), generator=g).to(device)
    zz, yy, xx = _grid(size, device)
    mask = ((zz / 0.95) ** 2 + (yy / 0.9) ** 2 + (xx / 0.85) ** 2 <= 1.0).float()
    return x * mask


def nested_spheres(n, size, device="cpu"):
    zz, yy, xx = _grid(size, device)
    r2 = (zz - 0.1) ** 2 + (yy + 0.05) ** 2 + (xx - 0.15) ** 2
    t = torch.stack([(r2 <= r * r).float() for r in (0.6, 0.4, 0.25)], 0)
    return t[None].repeat(n, 1, 1, 1, 1).contiguous()
