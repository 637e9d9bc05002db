"""ctypes wrapper of lib/libdvo_synth.so (seeded synthetic RGB-D edge scenes)."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None


def _load():
    global _lib
    if _lib is not None:
        return _lib
    path = os.path.join(_HERE, "lib", "libdvo_synth.so")
    if not os.path.exists(path):
        import subprocess
        subprocess.run(["make", "-C", os.path.join(_HERE, "csrc"), os.path.join("..", "lib", "libdvo_synth.so")], check=False,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    if not os.path.exists(path):
        raise FileNotFoundError(f"{path} not found: run __graft_entry__.build() or make -C rgbd_odometry_amd/csrc")
    lib = C.CDLL(path)
    lib.dvo_synth_create.restype = C.c_void_p
    lib.dvo_synth_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_uint64]
    lib.dvo_synth_create_ex.restype = C.c_void_p
    lib.dvo_synth_create_ex.argtypes = [C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_int, C.c_double]
    lib.dvo_synth_destroy.argtypes = [C.c_void_p]
    for n in ("rows", "cols"):
        getattr(lib, "dvo_synth_" + n).argtypes = [C.c_void_p, C.c_int]
        getattr(lib, "dvo_synth_" + n).restype = C.c_int
    for n in ("ref_edge", "now_edge"):
        getattr(lib, "dvo_synth_" + n).argtypes = [C.c_void_p, C.c_int]
        getattr(lib, "dvo_synth_" + n).restype = C.POINTER(C.c_int32)
    for n in ("ref_depth", "now_dt", "now_gx", "now_gy"):
        getattr(lib, "dvo_synth_" + n).argtypes = [C.c_void_p, C.c_int]
        getattr(lib, "dvo_synth_" + n).restype = C.POINTER(C.c_float)
    lib.dvo_synth_intrinsics.argtypes = [C.c_void_p, C.c_void_p]
    lib.dvo_synth_true_pose.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    _lib = lib
    return lib


class SynthLevel:
    __slots__ = ("rows", "cols", "ref_edge", "ref_depth", "now_edge", "now_dt", "now_gx", "now_gy")


class SynthScene:
    """One synthetic frame pair: per-level ref edge mask + depth and now DT + gradients.

    All images are flat column-major arrays (``a[yy + xx*rows]``), the layout of the
    reference's Eigen::MatrixXf members.
    """

    def __init__(self, W: int, H: int, n_levels: int, seed: int, n_seg: int = 0, x_frac: float = 1.0):
        """n_seg / x_frac: a SPARSE scene -- n_seg segments drawn inside the columns [0, x_frac W) only (default: SURVEY.md 8d's scene)"""
        lib = _load()
        h = lib.dvo_synth_create_ex(W, H, n_levels, seed, int(n_seg), float(x_frac)) if (n_seg > 0 or x_frac < 1.0) else lib.dvo_synth_create(W, H, n_levels, seed)
        if not h:
            raise ValueError("dvo_synth_create failed (bad size/levels)")
        try:
            self.W, self.H, self.n_levels, self.seed = W, H, n_levels, seed
            k = np.zeros(4, np.float32)
            lib.dvo_synth_intrinsics(h, k.ctypes.data)
            self.fx, self.fy, self.cx, self.cy = (float(x) for x in k)
            Rt = np.zeros((3, 3), order="F")
            tt = np.zeros(3)
            lib.dvo_synth_true_pose(h, Rt.ctypes.data, tt.ctypes.data)
            self.R_true, self.t_true = np.array(Rt), tt
            self.levels = []
            for l in range(n_levels):
                L = SynthLevel()
                L.rows, L.cols = lib.dvo_synth_rows(h, l), lib.dvo_synth_cols(h, l)
                n = L.rows * L.cols
                for name in ("ref_edge", "ref_depth", "now_edge", "now_dt", "now_gx", "now_gy"):
                    p = getattr(lib, "dvo_synth_" + name)(h, l)
                    setattr(L, name, np.ctypeslib.as_array(p, shape=(n,)).copy())
                self.levels.append(L)
        finally:
            lib.dvo_synth_destroy(h)

    @property
    def intrinsics(self):
        return (self.fx, self.fy, self.cx, self.cy)
