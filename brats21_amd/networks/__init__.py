from .equiunet import EquiUnet  # noqa: F401
