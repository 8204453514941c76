"""Loader of the in-tree C-ABI library (rayzath_amd/csrc/libhiprz.so).

There is no fallback: if the library is missing this raises, so nothing can silently run
on a CPU path.  Build it with `python -c "import __graft_entry__ as g; g.build()"` or
`make -C rayzath_amd/csrc`.
"""
import ctypes
import os

from . import _abi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("HIPRZ_LIB") or os.path.join(_HERE, "csrc", "libhiprz.so")  # HIPRZ_LIB: A/B builds (tools/ab_variants.sh)
_lib = None


class HiprzError(RuntimeError):
    """Raised for any non-zero return of the C-ABI (RayZath::Exception on the C++ side)."""

    def __init__(self, code, message):
        super().__init__(f"hiprz error {code}: {message}")
        self.code = code


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it first (python -c 'import __graft_entry__ as g; g.build()'). "
                "rayzath_amd has no CPU fallback.")
        _lib = _abi.bind(ctypes.CDLL(LIB_PATH))
    return _lib
