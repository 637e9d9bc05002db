"""ctypes binding of include/dvo_amd.h (one Python method per C entry point).

Fails loudly when the HIP library has not been built or when no HIP device is present:
there is no CPU fallback anywhere in this package.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

import numpy as np

DVO_MAX_LEVELS = 8
DVO_NUM_ACC = 29
DVO_OK, DVO_ERR_INVALID, DVO_ERR_NO_DEVICE, DVO_ERR_HIP, DVO_ERR_STATE, DVO_ERR_NOMEM = range(6)
DVO_FLAG_FINAL_OUTPUTS = 1
DVO_FLAG_IDENTITY_START = 2
DVO_FLAG_NORMAL_MATRIX = 4

_HERE = os.path.dirname(os.path.abspath(__file__))

#: every symbol include/dvo_amd.h declares (checked by tests/test_capi_symbols.py)
C_ABI_SYMBOLS = [
    "dvo_params_default", "dvo_create", "dvo_create_batch", "dvo_destroy", "dvo_last_error",
    "dvo_num_pairs", "dvo_set_stream", "dvo_use_own_stream", "dvo_synchronize", "dvo_set_keep_warm", "dvo_set_keep_warm2", "dvo_set_intrinsics",
    "dvo_set_ref_level", "dvo_set_ref_level_pair", "dvo_set_now_level", "dvo_set_now_level_pair",
    "dvo_set_ref_level_device", "dvo_set_now_level_device", "dvo_set_ref_level_from_images",
    "dvo_run_iterations", "dvo_run_iterations_pair", "dvo_align_pyramid", "dvo_align_batch",
    "dvo_set_poses", "dvo_align_batch_enqueue", "dvo_get_poses", "dvo_get_level_report",
    "dvo_get_final_outputs", "dvo_get_level_normal_matrix", "dvo_eval_points", "dvo_accumulate", "dvo_device_se3_exp",
    "dvo_device_se3_log", "dvo_device_rotationize", "dvo_algorithmic_bytes", "dvo_point_iterations",
    "dvo_debug_stamps", "dvo_get_level_texel_mode", "dvo_get_level_exact_fallback", "dvo_get_level_energy_sweeps", "dvo_get_level_points4", "dvo_get_level_ranks_in_lds", "dvo_now_prepare", "dvo_set_direct_compact", "dvo_host_alloc_mapped", "dvo_host_free_mapped", "dvo_get_now_compact_info", "dvo_get_now_compact_partial", "dvo_get_last_launch_shape", "dvo_replicate_pairs", "dvo_set_now_level_from_edges", "dvo_get_now_level", "dvo_iter_begin", "dvo_iter_accumulate", "dvo_iter_update", "dvo_iter_end",
    "dvo_align_pyramid_wide", "dvo_tiled_attach", "dvo_tiled_detach", "dvo_align_pyramid_tiled", "dvo_tiled_shard", "dvo_tiled_graph_replayed", "dvo_wide_packed_levels", "dvo_wide_team_levels",
    "dvo_get_ref_level", "dvo_frames_reserve", "dvo_frames_upload_pyramids", "dvo_frames_upload_cameras", "dvo_frames_set_undistort",
    "dvo_photo_params_default", "dvo_photo_configure", "dvo_photo_set_ref", "dvo_photo_align", "dvo_photo_get_jacobian", "dvo_frames_as_now",
    "dvo_frames_as_ref", "dvo_frame_get_level", "dvo_frames_num_levels",
]

DVO_PIX_U8, DVO_PIX_U16, DVO_PIX_F32 = 0, 1, 2
DVO_LAYOUT_COL_MAJOR, DVO_LAYOUT_ROW_MAJOR = 0, 1
DVO_UPLOAD_ASYNC = 1
DVO_UPLOAD_DEPTH_RAW = 2
DVO_UPLOAD_DIRECT = 4
DVO_UPLOAD_DEVICE = 8
DVO_UPLOAD_MAPPED = 16


class DvoImage(C.Structure):
    """Mirror of ``struct dvo_image``."""
    _fields_ = [("data", C.c_void_p), ("rows", C.c_int), ("cols", C.c_int), ("dtype", C.c_int), ("layout", C.c_int)]


class DvoParams(C.Structure):
    """Mirror of ``struct dvo_params`` (defaults = the literals of SolveDVO.cpp:21-33,653-773)."""
    _fields_ = [
        ("beta", C.c_double), ("precond_rot", C.c_double), ("reg_lambda", C.c_double),
        ("step_a", C.c_double), ("step_b", C.c_double),
        ("step_decay_after", C.c_int), ("step_decay_offset", C.c_int),
        ("trust_radius", C.c_float), ("psi_norm_stop", C.c_float),
        ("enable_rotationize", C.c_int), ("enable_l2_reg", C.c_int), ("interpolate_dt", C.c_int),
        ("block_threads", C.c_int), ("points_in_flight", C.c_int), ("engine_variant", C.c_int),
        ("lds_point_bytes", C.c_int), ("debug_alias_mod", C.c_int),
        ("canny_threshold1", C.c_int), ("canny_threshold2", C.c_int), ("team_size", C.c_int),
    ]


class DvoPhotoParams(C.Structure):
    """Mirror of ``struct dvo_photo_params`` (defaults = the constants of RGBDOdometry.cpp:32-34, :545, :556)."""
    _fields_ = [("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double),
                ("gradient_threshold", C.c_int), ("max_jacobian_size", C.c_int), ("min_required_pts", C.c_int),
                ("iterations", C.c_int), ("eps_norm_stop", C.c_double), ("fixed", C.c_int), ("reserved", C.c_int)]


class RcclComm:
    """A raw RCCL communicator made through ctypes (tests / tools of the C-driven tiled mode; a C++ node creates its
    ncclComm_t itself).  unique_id: 128 bytes from RcclComm.unique_id() of rank 0, distributed by the launcher."""
    RCCL = "/opt/rocm/lib/librccl.so.1"

    class _Id(C.Structure):
        _fields_ = [("internal", C.c_char * 128)]

    @classmethod
    def _lib(cls):
        import torch  # noqa: F401  (one HIP runtime per process, see load_library)
        lib = C.CDLL(cls.RCCL, mode=C.RTLD_GLOBAL)
        lib.ncclGetUniqueId.argtypes = [C.POINTER(cls._Id)]
        lib.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, cls._Id, C.c_int]
        lib.ncclCommDestroy.argtypes = [C.c_void_p]
        return lib

    @classmethod
    def unique_id(cls) -> bytes:
        uid = cls._Id()
        rc = cls._lib().ncclGetUniqueId(C.byref(uid))
        if rc != 0:
            raise RuntimeError("ncclGetUniqueId failed: %d" % rc)
        return bytes(uid)

    def __init__(self, unique_id: bytes, rank: int, world: int):
        self.lib = self._lib()
        uid = self._Id.from_buffer_copy(unique_id)
        self.comm = C.c_void_p()
        rc = self.lib.ncclCommInitRank(C.byref(self.comm), world, uid, rank)
        if rc != 0:
            raise RuntimeError("ncclCommInitRank failed: %d" % rc)
        self.comm = self.comm.value
        self.rank, self.world = rank, world

    def count(self) -> int:
        """ncclCommCount: the number of ranks RCCL itself says this communicator has"""
        n = C.c_int(0)
        self.lib.ncclCommCount.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        rc = self.lib.ncclCommCount(C.c_void_p(self.comm), C.byref(n))
        if rc != 0:
            raise RuntimeError("ncclCommCount failed: %d" % rc)
        return n.value

    def close(self):
        if self.comm:
            self.lib.ncclCommDestroy(C.c_void_p(self.comm))
            self.comm = None


def _mapped_ok(flags: int, passed, given):
    """DVO_UPLOAD_MAPPED hands the GPU the caller's (pinned, mapped) buffer itself: a silent dtype / layout conversion here would
    hand it a pageable copy instead"""
    if flags & DVO_UPLOAD_MAPPED and not (isinstance(given, np.ndarray) and np.shares_memory(passed, given)):
        raise ValueError("DVO_UPLOAD_MAPPED needs the images as contiguous numpy views of pinned memory in their final dtype "
                         "(uint8 BGR / grey, float32 or uint16 depth): this one would have been copied")


class MappedHostArray:
    """A numpy array over pinned host memory the GPU can address (dvo_host_alloc_mapped): what DVO_UPLOAD_MAPPED wants, for
    callers without torch.  Keep the object alive while the array is in use; free() or garbage collection releases it."""

    def __init__(self, shape, dtype):
        lib = load_library()
        self._lib = lib
        self.nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
        self._p = lib.dvo_host_alloc_mapped(self.nbytes)
        if not self._p:
            raise MemoryError("dvo_host_alloc_mapped(%d) failed" % self.nbytes)
        buf = (C.c_ubyte * self.nbytes).from_address(self._p)
        self.array = np.frombuffer(buf, dtype=dtype).reshape(shape)

    def free(self):
        if self._p:
            self.array = None
            self._lib.dvo_host_free_mapped(C.c_void_p(self._p))
            self._p = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class DvoError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"dvo error {code}: {msg}")
        self.code = code


def library_path() -> str:
    # DVO_LIB_VARIANT (e.g. "_w4") selects an experiment build made by `make -C csrc variants`
    return os.path.join(_HERE, "lib", "libdvo_amd%s.so" % os.environ.get("DVO_LIB_VARIANT", ""))


_lib = None


def _build_native():
    """make -C rgbd_odometry_amd/csrc: builds lib/libdvo_amd.so (gfx950), lib/libdvo_synth.so and the helper binaries.

    Serialised across processes with a file lock (N ranks of one torch.distributed launch on a fresh checkout would
    otherwise all run make at once and could dlopen a half-written library): whoever gets the lock builds, the others
    wait and find the library present.  A failed build raises with make's own output."""
    import fcntl
    import subprocess
    csrc = os.path.join(_HERE, "csrc")
    with open(os.path.join(csrc, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if os.path.exists(library_path()):
                return
            r = subprocess.run(["make", "-C", csrc], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
            if r.returncode != 0:
                raise RuntimeError("building the HIP extension failed (make -C %s):\n%s" % (csrc, r.stdout[-4000:]))
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def load_library() -> C.CDLL:
    """Load lib/libdvo_amd.so; raise if it is missing (build with __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        return _lib
    # One HIP runtime per process: PyTorch's wheel bundles its own libamdhip64 (SONAME
    # libamdhip64.so.7, same as /opt/rocm's).  Importing torch FIRST makes the dynamic loader bind
    # libdvo_amd.so to that already-loaded copy, so torch streams / device pointers / RCCL and this
    # library share one runtime.  Loading in the other order would put two runtimes in the process.
    if os.environ.get("DVO_NO_TORCH", "0") != "1":
        import torch  # noqa: F401
    path = library_path()
    if not os.path.exists(path) and not os.environ.get("DVO_LIB_VARIANT"):
        _build_native()           # a fresh checkout: compile in-tree (hipcc), never fall back to anything else
    if not os.path.exists(path):
        raise FileNotFoundError(
            f"{path} not found: build the HIP extension first "
            "(python -c 'import __graft_entry__ as g; g.build()' or make -C rgbd_odometry_amd/csrc)")
    lib = C.CDLL(path)
    vp, ip, fp, dp = C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_float), C.POINTER(C.c_double)
    i, f = C.c_int, C.c_float
    sig = {
        "dvo_params_default": [C.POINTER(DvoParams)],
        "dvo_create": [C.POINTER(DvoParams), C.POINTER(vp)],
        "dvo_create_batch": [C.POINTER(DvoParams), i, C.POINTER(vp)],
        "dvo_destroy": [vp],
        "dvo_num_pairs": [vp],
        "dvo_set_stream": [vp, vp],
        "dvo_use_own_stream": [vp],
        "dvo_synchronize": [vp],
        "dvo_set_keep_warm": [vp, i],
        "dvo_set_keep_warm2": [vp, i, i],
        "dvo_set_intrinsics": [vp, f, f, f, f],
        "dvo_set_ref_level": [vp, i, vp, i],
        "dvo_set_ref_level_pair": [vp, i, i, vp, i],
        "dvo_set_now_level": [vp, i, vp, vp, vp, i, i],
        "dvo_set_now_level_pair": [vp, i, i, vp, vp, vp, i, i],
        "dvo_set_ref_level_device": [vp, i, i, vp, i],
        "dvo_set_now_level_device": [vp, i, i, vp, vp, vp, i, i],
        "dvo_set_ref_level_from_images": [vp, i, i, vp, vp, i, i, vp, vp, i, ip],
        "dvo_run_iterations": [vp, i, i, vp, vp, vp, vp, vp, ip, fp],
        "dvo_run_iterations_pair": [vp, i, i, i, vp, vp, vp, vp, vp, ip, fp],
        "dvo_align_pyramid": [vp, i, ip, i, vp, vp],
        "dvo_align_batch": [vp, i, i, i, ip, i, vp, vp],
        "dvo_set_poses": [vp, i, i, vp, vp],
        "dvo_align_batch_enqueue": [vp, i, i, i, ip, i],
        "dvo_get_poses": [vp, i, i, vp, vp],
        "dvo_get_level_report": [vp, i, i, vp, i, ip, fp],
        "dvo_get_final_outputs": [vp, i, vp, vp, i, ip],
        "dvo_get_level_normal_matrix": [vp, i, i, i, vp],
        "dvo_eval_points": [vp, i, i, vp, vp, vp, vp, vp, vp, vp],
        "dvo_accumulate": [vp, i, i, vp, vp, vp],
        "dvo_device_se3_exp": [vp, vp, vp, vp],
        "dvo_device_se3_log": [vp, vp, vp, vp],
        "dvo_device_rotationize": [vp, vp],
        "dvo_debug_stamps": [vp, i, vp],
        "dvo_get_level_texel_mode": [vp, i, i, ip],
        "dvo_get_level_exact_fallback": [vp, i, i, ip],
        "dvo_get_level_energy_sweeps": [vp, i, i, ip],
        "dvo_get_level_points4": [vp, i, i, ip],
        "dvo_get_level_ranks_in_lds": [vp, i, i, ip],
        "dvo_now_prepare": [vp, i, i],
        "dvo_get_last_launch_shape": [vp, ip, ip, ip],
        "dvo_get_now_compact_info": [vp, i, i, ip],
        "dvo_get_now_compact_partial": [vp, i, i, ip],
        "dvo_set_direct_compact": [vp, i],
        "dvo_host_alloc_mapped": [C.c_size_t],
        "dvo_host_free_mapped": [vp],
        "dvo_replicate_pairs": [vp, i, i, i],
        "dvo_set_now_level_from_edges": [vp, i, i, vp, i, i],
        "dvo_get_now_level": [vp, i, i, vp, vp, vp],
        "dvo_iter_begin": [vp, i, i, i, vp, vp],
        "dvo_iter_accumulate": [vp, i, i, i, i, vp],
        "dvo_iter_update": [vp, i, i, i, i, vp],
        "dvo_iter_end": [vp, i, i, vp, vp, vp, ip, fp],
        "dvo_align_pyramid_wide": [vp, i, i, ip, i, vp, vp],
        "dvo_frames_set_undistort": [vp, i, i, vp, vp],
        "dvo_photo_params_default": [C.POINTER(DvoPhotoParams)],
        "dvo_photo_configure": [vp, C.POINTER(DvoPhotoParams)],
        "dvo_photo_set_ref": [vp, i, i, ip],
        "dvo_photo_align": [vp, i, ip, i, vp, vp, ip],
        "dvo_photo_get_jacobian": [vp, i, vp, vp, vp, i, vp, ip],
        "dvo_tiled_attach": [vp, vp, i, i, C.c_char_p],
        "dvo_tiled_detach": [vp],
        "dvo_align_pyramid_tiled": [vp, i, i, ip, i, vp, vp],
        "dvo_tiled_shard": [vp, i, i, ip, ip],
        "dvo_tiled_graph_replayed": [vp, ip],
        "dvo_wide_packed_levels": [vp, ip, ip],
        "dvo_wide_team_levels": [vp, ip],
        "dvo_get_ref_level": [vp, i, i, vp, i, ip],
        "dvo_frames_reserve": [vp, i],
        "dvo_frames_upload_pyramids": [vp, i, i, i, C.POINTER(DvoImage), C.POINTER(DvoImage), i, i],
        "dvo_frames_upload_cameras": [vp, i, i, C.POINTER(vp), C.POINTER(vp), i, i, i, i, i, i],
        "dvo_frames_as_now": [vp, i, i, i],
        "dvo_frames_as_ref": [vp, i, i, i, ip],
        "dvo_frame_get_level": [vp, i, i, ip, ip, vp, vp, vp, ip],
        "dvo_frames_num_levels": [vp],
        "dvo_algorithmic_bytes": [vp, i, i, ip, i, C.POINTER(C.c_uint64)],
        "dvo_point_iterations": [vp, i, i, ip, C.POINTER(C.c_uint64)],
    }
    for name, args in sig.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = C.c_int
    lib.dvo_last_error.argtypes = [vp]
    lib.dvo_last_error.restype = C.c_char_p
    lib.dvo_host_alloc_mapped.restype = C.c_void_p
    lib.dvo_host_free_mapped.restype = None
    _lib = lib
    return lib


def _f32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


def _ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _iters(iters: Sequence[int]):
    arr = (C.c_int * len(iters))(*[int(x) for x in iters])
    return arr


class DvoContext:
    """One engine context = ``dvo_ctx`` (n_pairs frame pairs resident in HBM)."""

    def __init__(self, n_pairs: int = 1, params: Optional[DvoParams] = None, **overrides):
        self.lib = load_library()
        p = DvoParams()
        self.lib.dvo_params_default(C.byref(p))
        if params is not None:
            p = params
        for k, v in overrides.items():
            setattr(p, k, v)
        self.params = p
        self._h = C.c_void_p()
        rc = self.lib.dvo_create_batch(C.byref(p), int(n_pairs), C.byref(self._h))
        if rc != DVO_OK:
            raise DvoError(rc, (self.lib.dvo_last_error(None) or b"").decode())
        self.n_pairs = int(n_pairs)
        self._N = {}      # (pair, level) -> N
        self._dims = {}   # level -> (rows, cols)
        self._frame_keep = []   # host buffers borrowed by DVO_UPLOAD_ASYNC uploads, released at the next synchronisation

    # -- plumbing -----------------------------------------------------------
    def _chk(self, rc: int):
        if rc != DVO_OK:
            raise DvoError(rc, (self.lib.dvo_last_error(self._h) or b"").decode())

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self.lib.dvo_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def set_stream(self, hip_stream: int):
        self._chk(self.lib.dvo_set_stream(self._h, C.c_void_p(hip_stream)))

    def use_own_stream(self):
        self._chk(self.lib.dvo_use_own_stream(self._h))

    def set_keep_warm(self, period_us: int):
        """launch a no-op kernel every period_us microseconds from a thread of the context (0 = stop): holds the GPU's active
        clocks between the frames of a single camera stream"""
        self._chk(self.lib.dvo_set_keep_warm(self._h, int(period_us)))

    def set_keep_warm2(self, busy_us: int, pause_us: int = 0):
        """one wave busy for busy_us microseconds, pause_us apart (0 = back to back), from a thread of the context; busy_us = 0 stops"""
        self._chk(self.lib.dvo_set_keep_warm2(self._h, int(busy_us), int(pause_us)))

    def synchronize(self):
        self._chk(self.lib.dvo_synchronize(self._h))
        self._frame_keep.clear()

    # -- inputs -------------------------------------------------------------
    def set_intrinsics(self, fx, fy, cx, cy):
        self._chk(self.lib.dvo_set_intrinsics(self._h, fx, fy, cx, cy))

    def set_ref_level(self, level: int, xyz, pair: int = 0):
        xyz = _f32(xyz).reshape(-1)
        n = xyz.size // 3
        self._chk(self.lib.dvo_set_ref_level_pair(self._h, pair, level, _ptr(xyz), n))
        self._N[(pair, level)] = n

    def set_now_level(self, level: int, dt, gx, gy, rows: int, cols: int, pair: int = 0):
        dt, gx, gy = _f32(dt).reshape(-1), _f32(gx).reshape(-1), _f32(gy).reshape(-1)
        assert dt.size == gx.size == gy.size == rows * cols
        self._chk(self.lib.dvo_set_now_level_pair(self._h, pair, level, _ptr(dt), _ptr(gx), _ptr(gy), rows, cols))
        self._dims[level] = (rows, cols)

    def set_ref_level_device(self, level: int, d_xyz_ptr: int, n: int, pair: int = 0):
        self._chk(self.lib.dvo_set_ref_level_device(self._h, pair, level, C.c_void_p(d_xyz_ptr), n))
        self._N[(pair, level)] = n

    def set_now_level_device(self, level: int, d_dt: int, d_gx: int, d_gy: int, rows: int, cols: int, pair: int = 0):
        self._chk(self.lib.dvo_set_now_level_device(self._h, pair, level, C.c_void_p(d_dt), C.c_void_p(d_gx),
                                                    C.c_void_p(d_gy), rows, cols))
        self._dims[level] = (rows, cols)

    def set_now_level_from_edges(self, level: int, edge, rows: int, cols: int, pair: int = 0):
        """computeDistTransfrmOfNow after Canny, on the GPU: uint8 edge mask -> resident now level"""
        edge = np.ascontiguousarray(edge, dtype=np.uint8).reshape(-1)
        assert edge.size == rows * cols
        self._chk(self.lib.dvo_set_now_level_from_edges(self._h, pair, level, _ptr(edge), rows, cols))
        self._dims[level] = (rows, cols)

    def get_now_level(self, level: int, pair: int = 0):
        rows, cols = self._dims[level]
        dt, gx, gy = (np.zeros(rows * cols, np.float32) for _ in range(3))
        self._chk(self.lib.dvo_get_now_level(self._h, pair, level, _ptr(dt), _ptr(gx), _ptr(gy)))
        return dt, gx, gy

    def get_ref_level(self, level: int, pair: int = 0):
        n = C.c_int()
        self._chk(self.lib.dvo_get_ref_level(self._h, pair, level, None, 0, C.byref(n)))
        xyz = np.zeros(3 * n.value, np.float32)
        self._chk(self.lib.dvo_get_ref_level(self._h, pair, level, _ptr(xyz), n.value, C.byref(n)))
        return xyz.reshape(-1, 3)

    def replicate_pairs(self, n_src: int, dst_first: int = 0, dst_count: Optional[int] = None):
        """slot p <- device copy of pair (p - dst_first) % n_src, for every level that is set"""
        dst_count = self.n_pairs - dst_first if dst_count is None else dst_count
        self._chk(self.lib.dvo_replicate_pairs(self._h, n_src, dst_first, dst_count))
        for (p, l), n in list(self._N.items()):
            if p < n_src:
                for q in range(dst_first, dst_first + dst_count):
                    if (q - dst_first) % n_src == p:
                        self._N[(q, l)] = n

    def set_ref_level_from_images(self, level: int, edge, depth_mm, rows: int, cols: int, pair: int = 0):
        """selectedPts + enlistRefEdgePts on the GPU; returns (xyz[N,3], uv[N,2])."""
        edge = np.ascontiguousarray(edge, dtype=np.int32).reshape(-1)
        depth = _f32(depth_mm).reshape(-1)
        cap = rows * cols
        xyz = np.zeros(3 * cap, np.float32)
        uv = np.zeros(2 * cap, np.float32)
        n = C.c_int(0)
        self._chk(self.lib.dvo_set_ref_level_from_images(self._h, pair, level, _ptr(edge), _ptr(depth), rows, cols,
                                                         _ptr(xyz), _ptr(uv), cap, C.byref(n)))
        self._N[(pair, level)] = n.value
        return xyz[:3 * n.value].reshape(-1, 3).copy(), uv[:2 * n.value].reshape(-1, 2).copy()

    # -- hot path -----------------------------------------------------------
    def run_iterations(self, level: int, max_iters: int, R, t, pair: int = 0, want_final: bool = True):
        """SolveDVO::runIterations.  Returns dict(R, t, energy, final_eps, final_reproj, best_idx, visible_ratio)."""
        R = np.array(R, dtype=np.float64, order="F").copy(order="F")
        t = np.array(t, dtype=np.float64).copy()
        n = self._N[(pair, level)]
        energy = np.zeros(max_iters, np.float32)
        feps = np.zeros(n, np.float32) if want_final else None
        frep = np.zeros(3 * n, np.float32) if want_final else None
        best = C.c_int(-2)
        ratio = C.c_float(0)
        self._chk(self.lib.dvo_run_iterations_pair(self._h, pair, level, max_iters, _ptr(R), _ptr(t), _ptr(energy),
                                                   _ptr(feps), _ptr(frep), C.byref(best), C.byref(ratio)))
        return dict(R=R, t=t, energy=energy, final_eps=feps,
                    final_reproj=None if frep is None else frep.reshape(-1, 3),
                    best_idx=best.value, visible_ratio=ratio.value)

    def align_batch(self, iters: Sequence[int], R, t, first_pair: int = 0, n_pairs: Optional[int] = None,
                    flags: int = 0):
        """Level schedule of SolveDVO::loop for a range of pairs.  R: [n,3,3] (math layout), t: [n,3]."""
        n_pairs = self.n_pairs - first_pair if n_pairs is None else n_pairs
        Rm = np.asarray(R, dtype=np.float64).reshape(n_pairs, 3, 3)
        Rc = np.ascontiguousarray(np.transpose(Rm, (0, 2, 1)))   # column-major per pair
        tc = np.ascontiguousarray(np.asarray(t, dtype=np.float64).reshape(n_pairs, 3))
        self._chk(self.lib.dvo_align_batch(self._h, first_pair, n_pairs, len(iters), _iters(iters), flags,
                                           _ptr(Rc), _ptr(tc)))
        return np.transpose(Rc, (0, 2, 1)).copy(), tc

    def set_poses(self, R, t, first_pair: int = 0):
        Rm = np.asarray(R, dtype=np.float64).reshape(-1, 3, 3)
        n = Rm.shape[0]
        Rc = np.ascontiguousarray(np.transpose(Rm, (0, 2, 1)))
        tc = np.ascontiguousarray(np.asarray(t, dtype=np.float64).reshape(n, 3))
        self._chk(self.lib.dvo_set_poses(self._h, first_pair, n, _ptr(Rc), _ptr(tc)))

    def enqueue(self, iters: Sequence[int], first_pair: int = 0, n_pairs: Optional[int] = None, flags: int = 0):
        n_pairs = self.n_pairs - first_pair if n_pairs is None else n_pairs
        self._chk(self.lib.dvo_align_batch_enqueue(self._h, first_pair, n_pairs, len(iters), _iters(iters), flags))

    def get_poses(self, first_pair: int = 0, n_pairs: Optional[int] = None):
        n_pairs = self.n_pairs - first_pair if n_pairs is None else n_pairs
        Rc = np.zeros((n_pairs, 3, 3))
        tc = np.zeros((n_pairs, 3))
        self._chk(self.lib.dvo_get_poses(self._h, first_pair, n_pairs, _ptr(Rc), _ptr(tc)))
        self._frame_keep.clear()          # dvo_get_poses synchronises the context stream
        return np.transpose(Rc, (0, 2, 1)).copy(), tc

    def level_report(self, pair: int, level: int, n_energy: int):
        energy = np.zeros(n_energy, np.float32)
        best = C.c_int(-2)
        ratio = C.c_float(0)
        self._chk(self.lib.dvo_get_level_report(self._h, pair, level, _ptr(energy), n_energy, C.byref(best),
                                                C.byref(ratio)))
        return energy, best.value, ratio.value

    def final_outputs(self, pair: int, capacity: int):
        feps = np.zeros(capacity, np.float32)
        frep = np.zeros(3 * capacity, np.float32)
        n = C.c_int(0)
        self._chk(self.lib.dvo_get_final_outputs(self._h, pair, _ptr(feps), _ptr(frep), capacity, C.byref(n)))
        return feps[:n.value].copy(), frep[:3 * n.value].reshape(-1, 3).copy()

    # -- frames in: rows f1 + f2 (Canny, distance transform, point extraction, pyramid on the GPU) -----
    def frames_reserve(self, n_slots: int):
        self._chk(self.lib.dvo_frames_reserve(self._h, n_slots))

    @staticmethod
    def _image(a, kind: str, layout: int):
        """kind 'grey': uint8 or float32; 'depth': uint16 (mm) or float32 (mm).  `a` is a 2-D (rows, cols) array
        for ROW_MAJOR, or the same 2-D array given in Fortran order / a transposed buffer for COL_MAJOR."""
        a = np.asarray(a)
        rows, cols = a.shape
        if a.dtype == np.uint8 and kind == "grey":
            dt = DVO_PIX_U8
        elif a.dtype == np.uint16 and kind == "depth":
            dt = DVO_PIX_U16
        else:
            a = a.astype(np.float32, copy=False)
            dt = DVO_PIX_F32
        buf = np.ascontiguousarray(a) if layout == DVO_LAYOUT_ROW_MAJOR else np.ascontiguousarray(a.T)
        return DvoImage(buf.ctypes.data, rows, cols, dt, layout), buf

    def frames_upload_pyramids(self, frames, first_slot: int = 0, layout: int = DVO_LAYOUT_ROW_MAJOR, flags: int = 0,
                               now_first_pair: int = -1):
        """frames: list of frames; a frame = list over levels of (grey, depth) or (grey, None); 2-D (rows, cols) arrays."""
        count, nl = len(frames), len(frames[0])
        keep, G, D = [], (DvoImage * (count * nl))(), (DvoImage * (count * nl))()
        have_depth = frames[0][0][1] is not None
        for f, fr in enumerate(frames):
            for l, (g, d) in enumerate(fr):
                G[f * nl + l], b = self._image(g, "grey", layout); keep.append(b)
                _mapped_ok(flags, b, g)
                if have_depth:
                    D[f * nl + l], b = self._image(d, "depth", layout); keep.append(b)
                    _mapped_ok(flags, b, d)
        self._chk(self.lib.dvo_frames_upload_pyramids(self._h, first_slot, count, nl, G, D if have_depth else None,
                                                      now_first_pair, flags))
        if flags & DVO_UPLOAD_ASYNC:
            self._frame_keep.append(keep)
        for l, (g, _) in enumerate(frames[0]):
            self._dims[l] = tuple(np.asarray(g).shape)

    def frames_upload_cameras(self, bgr_list, depth_list=None, n_levels: int = 4, first_shift: int = 1,
                              first_slot: int = 0, flags: int = 0, now_first_pair: int = -1):
        """bgr_list: list of (rows, cols, 3) uint8 BGR images; depth_list: list of (rows, cols) float32 metres or None"""
        count = len(bgr_list)
        bl = [np.ascontiguousarray(b, dtype=np.uint8) for b in bgr_list]
        for b, src in zip(bl, bgr_list):
            _mapped_ok(flags, b, src)
        rows, cols = bl[0].shape[:2]
        B = (C.c_void_p * count)(*[b.ctypes.data for b in bl])
        Dp, dl = None, None
        if depth_list is not None:
            dl = [np.ascontiguousarray(d, dtype=np.float32) for d in depth_list]
            for d, src in zip(dl, depth_list):
                _mapped_ok(flags, d, src)
            Dp = (C.c_void_p * count)(*[d.ctypes.data for d in dl])
        self._chk(self.lib.dvo_frames_upload_cameras(self._h, first_slot, count, B, Dp, rows, cols, n_levels, first_shift,
                                                     now_first_pair, flags))
        if now_first_pair >= 0:
            self._note_dims(first_slot)
        if flags & DVO_UPLOAD_ASYNC:
            self._frame_keep.append((bl, dl))

    def frames_upload_cameras_device(self, bgr_ptrs, depth_ptrs, rows: int, cols: int, n_levels: int = 4, first_shift: int = 1,
                                     first_slot: int = 0, flags: int = 0, now_first_pair: int = -1):
        """camera frames by address: in this GPU's memory (DVO_UPLOAD_DEVICE, the default) or, with flags | DVO_UPLOAD_MAPPED, in
        pinned host memory the GPU addresses (pulled over PCIe by a kernel).  Lists of addresses (ints) of (rows, cols, 3) uint8
        BGR images and, or None, (rows, cols) float32 depth images"""
        count = len(bgr_ptrs)
        # a caller that feeds the same buffers every step (a decoder's ring) passes prepared tables: pointer_table(addresses)
        B = bgr_ptrs if isinstance(bgr_ptrs, C.Array) else (C.c_void_p * count)(*[int(p) for p in bgr_ptrs])
        Dp = None if depth_ptrs is None else (depth_ptrs if isinstance(depth_ptrs, C.Array) else (C.c_void_p * count)(*[int(p) for p in depth_ptrs]))
        if not flags & DVO_UPLOAD_MAPPED:
            flags |= DVO_UPLOAD_DEVICE
        self._chk(self.lib.dvo_frames_upload_cameras(self._h, first_slot, count, B, Dp, rows, cols, n_levels, first_shift,
                                                     now_first_pair, flags))
        if now_first_pair >= 0:
            self._note_dims(first_slot)

    @staticmethod
    def pointer_table(addresses):
        """a list of addresses as the C array frames_upload_cameras_device takes (built once, passed every step)"""
        return (C.c_void_p * len(addresses))(*[int(p) for p in addresses])

    def frames_as_now(self, first_slot: int = 0, first_pair: int = 0, count: int = 1):
        self._chk(self.lib.dvo_frames_as_now(self._h, first_slot, first_pair, count))
        self._note_dims(first_slot)

    def _note_dims(self, first_slot: int):
        for l in range(self.lib.dvo_frames_num_levels(self._h)):
            r, c_ = C.c_int(), C.c_int()
            self._chk(self.lib.dvo_frame_get_level(self._h, first_slot, l, C.byref(r), C.byref(c_), None, None, None, None))
            self._dims[l] = (r.value, c_.value)

    def frames_as_ref(self, first_slot: int = 0, first_pair: int = 0, count: int = 1):
        nl = self.lib.dvo_frames_num_levels(self._h)
        N = (C.c_int * (count * max(nl, 1)))()
        self._chk(self.lib.dvo_frames_as_ref(self._h, first_slot, first_pair, count, N))
        N = np.array(N[:], dtype=np.int64).reshape(count, max(nl, 1))
        for i in range(count):
            for l in range(nl):
                self._N[(first_pair + i, l)] = int(N[i, l])
        return N

    def frame_level(self, slot: int, level: int, want_depth: bool = True):
        """resident (grey u8, depth mm f32 or None, edge u8 0/255, n_edges) of a stored level, as 2-D (rows, cols) arrays"""
        r, c_ = C.c_int(), C.c_int()
        self._chk(self.lib.dvo_frame_get_level(self._h, slot, level, C.byref(r), C.byref(c_), None, None, None, None))
        rows, cols = r.value, c_.value
        grey, edge = np.zeros(rows * cols, np.uint8), np.zeros(rows * cols, np.uint8)
        depth = np.zeros(rows * cols, np.float32) if want_depth else None
        ne = C.c_int()
        self._chk(self.lib.dvo_frame_get_level(self._h, slot, level, None, None, _ptr(grey), _ptr(depth), _ptr(edge), C.byref(ne)))
        cm = lambda a: None if a is None else a.reshape(cols, rows).T.copy()      # column-major buffer -> (rows, cols)
        return cm(grey), cm(depth), cm(edge), ne.value

    # -- host-driven iteration (large frames / multi-GPU tiled mode) ---------
    def n_points(self, level: int, pair: int = 0) -> int:
        return self._N[(pair, level)]

    def iter_begin(self, level: int, max_iters: int, R, t, pair: int = 0):
        R = np.array(R, dtype=np.float64, order="F")
        t = np.array(t, dtype=np.float64)
        self._chk(self.lib.dvo_iter_begin(self._h, pair, level, max_iters, _ptr(R), _ptr(t)))
        self._iter_max = max_iters

    def iter_accumulate(self, level: int, first: int, count: int, d_acc32_ptr: int, pair: int = 0):
        self._chk(self.lib.dvo_iter_accumulate(self._h, pair, level, first, count, C.c_void_p(d_acc32_ptr)))

    def iter_update(self, level: int, itr: int, n_total: int, d_acc32_ptr: int, pair: int = 0):
        self._chk(self.lib.dvo_iter_update(self._h, pair, level, itr, n_total, C.c_void_p(d_acc32_ptr)))

    def iter_end(self, level: int, pair: int = 0):
        R, t = np.zeros((3, 3), order="F"), np.zeros(3)
        energy = np.zeros(self._iter_max, np.float32)
        best, ratio = C.c_int(-2), C.c_float(0)
        self._chk(self.lib.dvo_iter_end(self._h, pair, level, _ptr(R), _ptr(t), _ptr(energy), C.byref(best),
                                        C.byref(ratio)))
        return dict(R=R, t=t, energy=energy, best_idx=best.value, visible_ratio=ratio.value)

    def align_pyramid_wide(self, iters: Sequence[int], R, t, pair: int = 0, flags: int = 0):
        """level schedule with every iteration spread over all CUs (large single frames), driven from C"""
        R = np.array(R, dtype=np.float64, order="F").copy(order="F")
        t = np.array(t, dtype=np.float64).copy()
        self._chk(self.lib.dvo_align_pyramid_wide(self._h, pair, len(iters), _iters(iters), flags, _ptr(R), _ptr(t)))
        return R, t

    def frames_set_undistort(self, rows: int, cols: int, K4=None, D5=None):
        """cv::undistort with the camera-info calibration on every camera frame uploaded from now on; None, None = off"""
        if K4 is None and D5 is None:
            self._chk(self.lib.dvo_frames_set_undistort(self._h, rows, cols, None, None))
            return
        K = np.asarray(K4, np.float64).copy(); D = np.asarray(D5, np.float64).copy()
        assert K.size == 4 and D.size == 5
        self._chk(self.lib.dvo_frames_set_undistort(self._h, rows, cols, _ptr(K), _ptr(D)))

    # -- photometric Gauss-Newton (RGBDOdometry's engine) ------------------------
    def photo_configure(self, K, fixed: bool = False, **overrides):
        p = DvoPhotoParams()
        self.lib.dvo_photo_params_default(C.byref(p))
        p.fx, p.fy, p.cx, p.cy = (float(k) for k in K)
        p.fixed = int(fixed)
        for k, v in overrides.items():
            setattr(p, k, v)
        self._chk(self.lib.dvo_photo_configure(self._h, C.byref(p)))
        self._photo_iters = p.iterations

    def photo_set_ref(self, slot: int, first_level: int = 1):
        n = (C.c_int * DVO_MAX_LEVELS)()
        self._chk(self.lib.dvo_photo_set_ref(self._h, slot, first_level, n))
        return list(n)

    def photo_align(self, now_slot: int, T, levels=(3, 2)):
        T = np.array(T, dtype=np.float64, order="C").copy()
        lv = (C.c_int * len(levels))(*levels)
        norms = np.zeros((len(levels), getattr(self, "_photo_iters", 3)), np.float64)
        upd = (C.c_int * len(levels))()
        self._chk(self.lib.dvo_photo_align(self._h, now_slot, lv, len(levels), _ptr(T), _ptr(norms), upd))
        return T, norms, list(upd)

    def photo_jacobian(self, level: int, capacity: int = 50000):
        J = np.zeros((capacity, 6), np.float64); si = np.zeros(capacity, np.int32); sj = np.zeros(capacity, np.int32)
        A = np.zeros((6, 6), np.float64); n = C.c_int(0)
        self._chk(self.lib.dvo_photo_get_jacobian(self._h, level, _ptr(J), _ptr(si), _ptr(sj), capacity, _ptr(A), C.byref(n)))
        return dict(J=J[:n.value].copy(), sel_i=si[:n.value].copy(), sel_j=sj[:n.value].copy(), A=A, n=n.value)

    # -- tiled mode driven from C (RCCL) ---------------------------------------
    def tiled_attach(self, comm, rank: int, world: int, rccl_library: Optional[str] = None):
        """comm: the rank's ncclComm_t (integer / c_void_p), e.g. RcclComm(...).comm"""
        lib = rccl_library.encode() if rccl_library else None
        self._chk(self.lib.dvo_tiled_attach(self._h, C.c_void_p(int(comm)), rank, world, lib))

    def tiled_detach(self):
        self._chk(self.lib.dvo_tiled_detach(self._h))

    def align_pyramid_tiled(self, iters: Sequence[int], R, t, pair: int = 0, flags: int = 0):
        """dvo_align_pyramid_tiled: this rank's share of every iteration + ncclAllReduce of the 32 sums + identical update"""
        R = np.array(R, dtype=np.float64, order="F").copy(order="F")
        t = np.array(t, dtype=np.float64).copy()
        self._chk(self.lib.dvo_align_pyramid_tiled(self._h, pair, len(iters), _iters(iters), flags, _ptr(R), _ptr(t)))
        return R, t

    def tiled_graph_replayed(self) -> bool:
        """the last align_pyramid_tiled replayed its captured graph (kernel + ncclAllReduce per iteration, no host work between)"""
        g = C.c_int(0)
        self._chk(self.lib.dvo_tiled_graph_replayed(self._h, C.byref(g)))
        return bool(g.value)

    def wide_packed_levels(self) -> int:
        """bit l set: level l of the last enqueued align_pyramid_wide / _tiled schedule ran the packed step kernel (compact list)"""
        g = C.c_int(0)
        self._chk(self.lib.dvo_wide_packed_levels(self._h, C.byref(g), None))
        return g.value

    def wide_team_levels(self) -> int:
        """bit l set: level l of the last align_pyramid_wide ran inside the fused kernel's team launch (coarse levels of a large frame)"""
        g = C.c_int(0)
        self._chk(self.lib.dvo_wide_team_levels(self._h, C.byref(g)))
        return g.value

    def wide_solo_levels(self) -> int:
        """bit l set: level l of that schedule ran as ONE launch of one workgroup for all its iterations (small levels)"""
        g, s = C.c_int(0), C.c_int(0)
        self._chk(self.lib.dvo_wide_packed_levels(self._h, C.byref(g), C.byref(s)))
        return s.value

    def tiled_shard(self, level: int, pair: int = 0):
        """(first, count): the index range of `level`'s reference list this rank works on"""
        f, n = C.c_int(0), C.c_int(0)
        self._chk(self.lib.dvo_tiled_shard(self._h, pair, level, C.byref(f), C.byref(n)))
        return f.value, n.value

    # -- inspection -----------------------------------------------------------
    def eval_points(self, level: int, R, t, pair: int = 0):
        n = self._N[(pair, level)]
        R = np.array(R, dtype=np.float64, order="F")
        t = np.array(t, dtype=np.float64)
        rep, J = np.zeros(3 * n, np.float32), np.zeros(6 * n, np.float32)
        eps, w, vis = np.zeros(n, np.float32), np.zeros(n, np.float32), np.zeros(n, np.int32)
        self._chk(self.lib.dvo_eval_points(self._h, pair, level, _ptr(R), _ptr(t), _ptr(rep), _ptr(J), _ptr(eps),
                                           _ptr(w), _ptr(vis)))
        return dict(reproj=rep.reshape(-1, 3), J=J.reshape(-1, 6), eps=eps, w=w, visible=vis)

    def accumulate(self, level: int, R, t, pair: int = 0) -> np.ndarray:
        R = np.array(R, dtype=np.float64, order="F")
        t = np.array(t, dtype=np.float64)
        acc = np.zeros(DVO_NUM_ACC)
        self._chk(self.lib.dvo_accumulate(self._h, pair, level, _ptr(R), _ptr(t), _ptr(acc)))
        return acc

    def se3_exp(self, psi):
        psi = np.array(psi, dtype=np.float64)
        R = np.zeros((3, 3), order="F")
        t = np.zeros(3)
        self._chk(self.lib.dvo_device_se3_exp(self._h, _ptr(psi), _ptr(R), _ptr(t)))
        return R, t

    def se3_log(self, R, t):
        R = np.array(R, dtype=np.float64, order="F")
        t = np.array(t, dtype=np.float64)
        psi = np.zeros(6)
        self._chk(self.lib.dvo_device_se3_log(self._h, _ptr(R), _ptr(t), _ptr(psi)))
        return psi

    def rotationize(self, R):
        R = np.array(R, dtype=np.float64, order="F").copy(order="F")
        self._chk(self.lib.dvo_device_rotationize(self._h, _ptr(R)))
        return R

    def debug_stamps(self, pair: int = 0) -> np.ndarray:
        out = np.zeros(64, np.uint64)
        self._chk(self.lib.dvo_debug_stamps(self._h, pair, _ptr(out)))
        return out.reshape(8, 8)

    def level_normal_matrix(self, pair: int, level: int, itr: int = -1) -> np.ndarray:
        """H = sum w J^T J (6x6) of iterate `itr` (default: the best) of the last align made with DVO_FLAG_NORMAL_MATRIX"""
        H = np.zeros(36, np.float64)
        self._chk(self.lib.dvo_get_level_normal_matrix(self._h, pair, level, itr, _ptr(H)))
        return H.reshape(6, 6)

    def level_texel_mode(self, pair: int, level: int) -> int:
        """0 = 16-byte texels gathered from HBM/L2, 1 = the level's texels staged in LDS, 2 = the compact form, -1 = not run"""
        m = C.c_int(-2)
        self._chk(self.lib.dvo_get_level_texel_mode(self._h, pair, level, C.byref(m)))
        return m.value

    def level_energy_sweeps(self, pair: int, level: int) -> int:
        """iterations of that level whose energy came from the exact sweep of the residuals (the certificate of the fast sum failed,
        or engine_variant = 5)"""
        m = C.c_int(0)
        self._chk(self.lib.dvo_get_level_energy_sweeps(self._h, pair, level, C.byref(m)))
        return m.value

    def level_exact_fallback(self, pair: int, level: int) -> bool:
        """True if a wave of the packed kernel took its literal-division fallback at that level of the last launch"""
        m = C.c_int(0)
        self._chk(self.lib.dvo_get_level_exact_fallback(self._h, pair, level, C.byref(m)))
        return bool(m.value)

    def level_points4(self, pair: int, level: int) -> bool:
        """True if that level's reference points were read in their 4-byte form during the last launch"""
        m = C.c_int(0)
        self._chk(self.lib.dvo_get_level_points4(self._h, pair, level, C.byref(m)))
        return bool(m.value)

    def level_ranks_in_lds(self, pair: int, level: int) -> bool:
        """True if the last fused launch looked that level's ranks up in an LDS copy of the whole level (round 5)"""
        m = C.c_int(0)
        self._chk(self.lib.dvo_get_level_ranks_in_lds(self._h, pair, level, C.byref(m)))
        return bool(m.value)

    def last_launch_shape(self):
        """(threads per workgroup, workgroups per pair, packed kernel?) of the last fused launch"""
        b, t, k = C.c_int(0), C.c_int(0), C.c_int(0)
        self._chk(self.lib.dvo_get_last_launch_shape(self._h, C.byref(b), C.byref(t), C.byref(k)))
        return b.value, t.value, bool(k.value)

    def now_prepare(self, first_pair: int = 0, count: Optional[int] = None):
        """build the compact (4 bytes per pixel) form of the resident now levels of these pairs now"""
        count = self.n_pairs - first_pair if count is None else count
        self._chk(self.lib.dvo_now_prepare(self._h, first_pair, count))

    def set_direct_compact(self, on: bool = True):
        """float now levels (set_now_level*) are turned into the compact form at installation (off by default)"""
        self._chk(self.lib.dvo_set_direct_compact(self._h, 1 if on else 0))

    def now_compact_info(self, pair: int, level: int) -> int:
        """> 0 palette size of the compact form, 0 not built, < 0 the builder's reason for keeping the 16-byte form"""
        m = C.c_int(0)
        self._chk(self.lib.dvo_get_now_compact_info(self._h, pair, level, C.byref(m)))
        return m.value

    # -- measurement ----------------------------------------------------------
    def now_compact_partial(self, pair: int, level: int) -> bool:
        """True if that compact form is PARTIAL (round 5): the image's lowest ranks are in the compact form, the pixels it cannot express
        (too many distinct distances, 512 px or more from every edge, a rank step beyond +-127) are looked up in its 16-byte texels"""
        m = C.c_int(0)
        self._chk(self.lib.dvo_get_now_compact_partial(self._h, pair, level, C.byref(m)))
        return bool(m.value)

    def algorithmic_bytes(self, iters: Sequence[int], pair: int = 0, flags: int = 0) -> int:
        b = C.c_uint64(0)
        self._chk(self.lib.dvo_algorithmic_bytes(self._h, pair, len(iters), _iters(iters), flags, C.byref(b)))
        return b.value

    def point_iterations(self, iters: Sequence[int], pair: int = 0) -> int:
        b = C.c_uint64(0)
        self._chk(self.lib.dvo_point_iterations(self._h, pair, len(iters), _iters(iters), C.byref(b)))
        return b.value
