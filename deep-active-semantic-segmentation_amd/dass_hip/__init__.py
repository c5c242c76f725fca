"""dass_hip: ctypes binding + autograd glue over libdass_hip.so (HIP/CDNA4 kernels for gfx950).

Importing this package loads the shared library and FAILS LOUDLY if it has not been built --
there is no CPU or eager-PyTorch fallback anywhere in the product path.
"""
from . import ops  # noqa: F401
from ._lib import LIB_PATH, PROTOTYPES, lib  # noqa: F401
from .ops import compute_dtype, set_compute_dtype  # noqa: F401
