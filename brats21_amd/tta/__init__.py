from .base import Compose  # noqa: F401
from .transforms import HorizontalFlip, VerticalFlip, Rotate90, OnAxes  # noqa: F401


def get_tta_transforms():
    """src/definer.py:647-658: OnAxes(['zxy', 'xyz']) x HorizontalFlip x Rotate90([0, 90, 180, 270]) -> 16 transformers
    (the reference's published evaluation runs all 16 on the whole volume, learning/engine.py:424-440)."""
    return Compose([OnAxes(["zxy", "xyz"]), HorizontalFlip(), Rotate90(angles=[0, 90, 180, 270])])


def flip8():
    """The 8 flips over all subsets of the three spatial axes (BASELINE.json configs[3]: "8-flip TTA"): image and mask
    pipelines are the same involution; shape-preserving, so sliding-window patch graphs are reused by every pass."""
    import itertools
    from .base import SignedPerm, Transformer
    return [Transformer(SignedPerm((0, 1, 2), f), SignedPerm((0, 1, 2), f)) for f in itertools.product([False, True], repeat=3)]
