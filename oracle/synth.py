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
        elif leaf == "estbn_moving_speed":   # EstBN's buffer: zeros(1), never set by the reference (networks/factory.py:160)
            sd[k] = torch.zeros(shp)
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


def tumour_phantom(n, size, seed, contrast=1.0):
    """A BraTS-shaped volume whose label is a FUNCTION OF THE IMAGE (tests/test_trained_gpu.py trains on it, so that the
    network's logits sit near the decision surface because it learned the target, not because the weights are random):
    4 modalities of i.i.d. N(0,1) noise inside the brain ellipsoid of ``random_image`` plus, per modality, intensity offsets
    inside three nested ellipsoids WT > TC > ET whose centre, radii and aspect are drawn per sample from ``seed`` (the way
    z-scored FLAIR / T1 / T1ce / T2 behave over oedema, core and enhancing tumour).  Returns (image [n,4,D,H,W] f32,
    target [n,3,D,H,W] {0,1} f32 in the reference's channel order TC, WT, ET, utils/transforms.py:155-166).
    ``contrast`` scales the offsets (the training uses 1.0; a test volume at 0.5 puts a large share of the voxels near the
    decision surface).  CPU generator only: bit-identical here and on the GPU box."""
    g = torch.Generator().manual_seed(int(seed))
    d, h, w = size
    zz, yy, xx = np.meshgrid(np.linspace(-1, 1, d), np.linspace(-1, 1, h), np.linspace(-1, 1, w), indexing="ij")
    brain = torch.from_numpy(((zz / 0.95) ** 2 + (yy / 0.9) ** 2 + (xx / 0.85) ** 2 <= 1.0).astype(np.float32))
    zz, yy, xx = (torch.from_numpy(a.astype(np.float32)) for a in (zz, yy, xx))
    # offsets per modality (rows) in WT-only, TC-only, ET regions (columns)
    amp = float(contrast) * torch.tensor([[1.6, 0.9, 0.6], [-0.3, -1.2, -0.8], [0.2, 0.5, 2.0], [1.1, 1.5, 0.9]])
    img = torch.randn(n, 4, d, h, w, generator=g)
    tgt = torch.zeros(n, 3, d, h, w)
    for i in range(n):
        u = torch.rand(8, generator=g)
        c = (u[:3] - 0.5) * 0.7
        r = 0.34 + 0.2 * float(u[3])
        asp = 0.8 + 0.4 * u[4:7]
        q = ((zz - c[0]) / asp[0]) ** 2 + ((yy - c[1]) / asp[1]) ** 2 + ((xx - c[2]) / asp[2]) ** 2
        wt, tc, et = (q <= r * r).float() * brain, (q <= (0.68 * r) ** 2).float() * brain, (q <= (0.42 * r) ** 2).float() * brain
        for m in range(4):
            img[i, m] += amp[m, 0] * (wt - tc) + amp[m, 1] * (tc - et) + amp[m, 2] * et
        img[i] *= brain
        tgt[i, 0], tgt[i, 1], tgt[i, 2] = tc, wt, et
    return img, tgt
