"""MI355X-native dense RGB-D edge-alignment engine (hot path of mpkuse/rgbd_odometry).

The product is the C-ABI shared library ``lib/libdvo_amd.so`` (HIP kernels for gfx950 +
``include/dvo_amd.h``).  This package only holds the ctypes binding used by the tests,
``bench.py`` and the multi-GPU sharding helpers; there is no Python or CPU compute path.
"""
from .capi import DvoContext, DvoError, DvoParams, load_library, library_path  # noqa: F401
from .synth import SynthScene  # noqa: F401

__all__ = ["DvoContext", "DvoError", "DvoParams", "load_library", "library_path", "SynthScene"]
