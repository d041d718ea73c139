"""ttach-style test-time augmentation, API of the reference's tta/base.py (Compose :103-136,
Transformer :78-100), executed on the GPU: every transformer's image / mask pipeline is folded into
ONE signed axis permutation and run by a single gather kernel (brats_spatial_signed_perm)."""
import itertools

import torch

from .. import _lib


class SignedPerm:
    """dst axis a (0..2 = D,H,W) <- src axis perm[a], reversed when flip[a]."""

    def __init__(self, perm=(0, 1, 2), flip=(False, False, False)):
        self.perm, self.flip = tuple(perm), tuple(bool(f) for f in flip)

    def then(self, other):
        """apply self first, then other."""
        perm = tuple(self.perm[other.perm[a]] for a in range(3))
        flip = tuple(other.flip[a] ^ self.flip[other.perm[a]] for a in range(3))
        return SignedPerm(perm, flip)

    @property
    def is_identity(self):
        return self.perm == (0, 1, 2) and not any(self.flip)

    def out_shape(self, shape):
        return tuple(shape[:2]) + tuple(shape[2 + self.perm[a]] for a in range(3))

    def apply(self, x, out=None, mode=0):
        """mode 0: out = T(x); 1: out += T(x); 2: out += sigmoid(T(x)).  x: [N, C, D, H, W] f32 CUDA."""
        if not x.is_cuda:
            raise _lib.BratsHipError("brats21_amd.tta runs on the GPU only")
        x = x.contiguous().float()
        n, c, s0, s1, s2 = x.shape
        if out is None:
            out = torch.empty(self.out_shape(x.shape), dtype=torch.float32, device=x.device)
            assert mode == 0
        _lib.check(_lib.lib().brats_spatial_signed_perm(
            x.data_ptr(), out.data_ptr(), n * c, s0, s1, s2, *self.perm, *[int(f) for f in self.flip], mode,
            torch.cuda.current_stream().cuda_stream), "spatial_signed_perm")
        return out


class BaseTransform:
    identity_param = None

    def __init__(self, name, params):
        self.params = params
        self.pname = name

    def aug(self, **param):      # SignedPerm applied to the image
        raise NotImplementedError

    def deaug(self, **param):    # SignedPerm applied to the mask
        raise NotImplementedError

    # reference-compatible eager entry points (tta/base.py:27-37)
    def apply_aug_image(self, image, **param):
        return self.aug(**param).apply(image)

    def apply_deaug_mask(self, mask, **param):
        return self.deaug(**param).apply(mask)

    def apply_deaug_label(self, label, **param):
        return label


class Transformer:
    """tta/base.py:78-100; additionally exposes the fused on-GPU accumulation of sigmoid(deaug(mask))."""

    def __init__(self, aug, deaug):
        self.aug_perm, self.deaug_perm = aug, deaug

    def augment_image(self, image):
        return self.aug_perm.apply(image)

    def deaugment_mask(self, mask):
        return self.deaug_perm.apply(mask)

    def deaugment_label(self, label):
        return label

    def accumulate_probability(self, logits, acc):
        """acc += sigmoid(deaugment_mask(logits)), one fused pass (learning/engine.py:239-249 on the GPU)."""
        return self.deaug_perm.apply(logits, out=acc, mode=2)


class Compose:
    """Cartesian product of the transforms' parameters in itertools.product order (tta/base.py:110-117)."""

    def __init__(self, transforms):
        self.aug_transforms = transforms
        self.aug_transform_parameters = list(itertools.product(*[t.params for t in self.aug_transforms]))
        self.deaug_transforms = transforms[::-1]
        self.deaug_transform_parameters = [p[::-1] for p in self.aug_transform_parameters]

    def __iter__(self):
        for aug_params, deaug_params in zip(self.aug_transform_parameters, self.deaug_transform_parameters):
            aug = SignedPerm()
            for t, p in zip(self.aug_transforms, aug_params):
                aug = aug.then(t.aug(**{t.pname: p}))
            deaug = SignedPerm()
            for t, p in zip(self.deaug_transforms, deaug_params):
                deaug = deaug.then(t.deaug(**{t.pname: p}))
            yield Transformer(aug, deaug)

    def __len__(self):
        return len(self.aug_transform_parameters)
