"""ORACLE helper (test infrastructure): RNG-free closed-form weights / inputs, so that this
container (where the golden vectors are made from the reference) and the GPU box (which never sees
/root/reference) regenerate bit-identical tensors without committing them."""
import zlib

import numpy as np
import torch


def _phase(name):
    return (zlib.crc32(name.encode()) % 10007) * 1e-3


def closed_form(name, shape, scale=1.0, offset=0.0, freq=0.7368):
    n = int(np.prod(shape))
    i = np.arange(n, dtype=np.float64)
    v = np.sin(i * freq + _phase(name)) + 0.5 * np.sin(i * 0.1931 * freq + 2.0 * _phase(name))
    return torch.from_numpy((offset + scale * v).astype(np.float32).reshape(shape))


def fill_state_dict(shapes):
    """Deterministic, well-conditioned values for every key of an {key: shape} map.
    conv / linear weights ~ N(0, 2/fan_in)-like magnitude so activations stay O(1);
    norm scales near 1; biases small; EvoNorm v / running_var = 1 (their init)."""
    sd = {}
    for k, shp in shapes.items():
        leaf = k.rsplit(".", 1)[-1]
        if k.endswith("bn.running_var"):      # nn.BatchNorm3d buffers (--norm batch): positive, not all ones
            sd[k] = closed_form(k, shp, 0.2, 1.0)
        elif k.endswith("bn.running_mean"):
            sd[k] = closed_form(k, shp, 0.05, 0.0)
        elif leaf == "num_batches_tracked":
            sd[k] = torch.tensor(0, dtype=torch.long)
        elif leaf in ("v", "running_var"):
            sd[k] = torch.ones(shp)
        elif leaf == "gamma" or k.endswith("bn.weight"):
            sd[k] = closed_form(k, shp, 0.15, 1.0)
        elif leaf in ("beta", "bias"):
            sd[k] = closed_form(k, shp, 0.08, 0.0)
        elif leaf == "weight":
            fan_in = int(np.prod(shp[1:]))
            sd[k] = closed_form(k, shp, float(np.sqrt(2.0 / fan_in)) * 1.2, 0.0)
        else:
            raise KeyError(k)
    return sd


def closed_form_image(n, c, size, tag="img"):
    """[n, c, D, H, W] fp32, zero outside a centred ellipsoid (brain-like support)."""
    d, h, w = size
    x = closed_form(tag, (n, c, d, h, w), 0.9, 0.0, freq=1.2345)
    zz, yy, xx = np.meshgrid(np.linspace(-1, 1, d), np.linspace(-1, 1, h), np.linspace(-1, 1, w), indexing="ij")
    mask = torch.from_numpy(((zz / 0.95) ** 2 + (yy / 0.9) ** 2 + (xx / 0.85) ** 2 <= 1.0).astype(np.float32))
    return x * mask


def nested_spheres(n, size):
    """[n, 3, D, H, W] {0,1} float target: WT > TC > ET nested spheres."""
    d, h, w = size
    zz, yy, xx = np.meshgrid(np.linspace(-1, 1, d), np.linspace(-1, 1, h), np.linspace(-1, 1, w), indexing="ij")
    r2 = (zz - 0.1) ** 2 + (yy + 0.05) ** 2 + (xx - 0.15) ** 2
    t = np.stack([(r2 <= r * r).astype(np.float32) for r in (0.6, 0.4, 0.25)], 0)
    return torch.from_numpy(t)[None].repeat(n, 1, 1, 1, 1).contiguous()


def random_image(n, c, size, seed=1234):
    """Bench input of SURVEY.md 8(d): N(0,1) zeroed outside a centred ellipsoid, seed 1234+rank."""
    g = torch.Generator().manual_seed(seed)
    d, h, w = size
    x = torch.randn(n, c, d, h, w, generator=g)
    zz, yy, xx = np.meshgrid(np.linspace(-1, 1, d), np.linspace(-1, 1, h), np.linspace(-1, 1, w), indexing="ij")
    mask = torch.from_numpy(((zz / 0.95) ** 2 + (yy / 0.9) ** 2 + (xx / 0.85) ** 2 <= 1.0).astype(np.float32))
    return x * mask
