"""pvio_amd -- MI355X-native back-end for PVIO's bundle adjustment and KLT hot path.

Layout: `csrc/` hand-written HIP (gfx950) + the C-ABI (include/pvio_hip.h); `host/` C++ mirror of the
reference seams (pvio::BundleAdjustor, pvio::Image); this Python package is only the test/bench harness
around the C-ABI (ctypes), never the product path.
"""
from . import capi  # noqa: F401
from .problem import BAProblem, BAState, BASummary  # noqa: F401
