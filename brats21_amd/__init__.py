"""brats21_amd -- MI355X-native (gfx950) hot path of Alxaline/BraTS21: the EquiUnet / EquiUnetASSPEvo
3D U-Nets, sliding-window + TTA inference and data-parallel training, on hand-written HIP kernels
behind a C ABI (include/brats_hip.h, brats21_amd/libbrats_hip.so).  No CPU fallback by design."""
from ._lib import BratsHipError, LIB_PATH  # noqa: F401
from .definer import get_model  # noqa: F401

__version__ = "0.1.0"
