"""rayzath_amd — MI355X (gfx950) path-tracing backend for RayZath: the HIPGPU engine.

Only the hot path of BASELINE.json's north_star lives here: the C-ABI
library (csrc/, include/hiprz.h), and the Python host side that mirrors the reference's
backend interface.  Importing this package never touches the GPU and never loads the
oracle.
"""
from . import _abi  # noqa: F401
from ._lib import HiprzError, load as load_library  # noqa: F401

__all__ = ["HiprzError", "load_library"]
