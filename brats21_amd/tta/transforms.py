"""The reference's dual transforms (tta/transforms.py) expressed as signed permutations of the spatial
axes (0,1,2) = tensor dims (2,3,4) of an NCDHW tensor."""
from .base import BaseTransform, SignedPerm


def _permute(order):
    """torch.permute(0, 1, *order) with order given in tensor dims 2..4 -> SignedPerm."""
    return SignedPerm(tuple(o - 2 for o in order))


def _rot90(k):
    """torch.rot90(x, k, (2, 3)) (tta/transforms.py:165-167): dims (2,3) = spatial axes (0,1)."""
    k %= 4
    if k == 0:
        return SignedPerm()
    if k == 2:
        return SignedPerm((0, 1, 2), (True, True, False))
    if k == 1:  # rot90 = flip(dim 3) then transpose(2, 3):  out[i][j] = in[j][n1-1-i]
        return SignedPerm((1, 0, 2), (True, False, False))
    return SignedPerm((1, 0, 2), (False, True, False))  # k == 3: out[i][j] = in[n0-1-j][i]


class OnAxes(BaseTransform):
    """tta/transforms.py:16-49"""
    identity_param = "zxy"

    def __init__(self, axes):
        super().__init__("axe", axes)
        assert all(a in ["xyz", "yzx", "zxy"] for a in axes), "axes need to be 'xyz', 'yzx', 'zxy'"

    def aug(self, axe="zxy"):
        return {"zxy": SignedPerm(), "xyz": _permute((3, 4, 2)), "yzx": _permute((4, 2, 3))}[axe]

    def deaug(self, axe="zxy"):
        return {"zxy": SignedPerm(), "xyz": _permute((4, 2, 3)), "yzx": _permute((3, 4, 2))}[axe]


class HorizontalFlip(BaseTransform):
    """flip(3), tta/transforms.py:52-70"""
    identity_param = False

    def __init__(self):
        super().__init__("apply", [False, True])

    def aug(self, apply=False):
        return SignedPerm((0, 1, 2), (False, bool(apply), False))

    deaug = aug


class VerticalFlip(BaseTransform):
    """flip(2), tta/transforms.py:73-91"""
    identity_param = False

    def __init__(self):
        super().__init__("apply", [False, True])

    def aug(self, apply=False):
        return SignedPerm((0, 1, 2), (bool(apply), False, False))

    deaug = aug


class Rotate90(BaseTransform):
    """tta/transforms.py:149-173"""
    identity_param = 0

    def __init__(self, angles):
        if self.identity_param not in angles:
            angles = [self.identity_param] + list(angles)
        super().__init__("angle", angles)

    def aug(self, angle=0):
        k = angle // 90 if angle >= 0 else (angle + 360) // 90
        return _rot90(k)

    def deaug(self, angle=0):
        return self.aug(-angle)
