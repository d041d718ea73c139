from .base import Compose  # noqa: F401
from .transforms import HorizontalFlip, VerticalFlip, Rotate90, OnAxes  # noqa: F401
