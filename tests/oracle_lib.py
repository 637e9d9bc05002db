"""ctypes wrapper of the CPU oracle (oracle/build/libdvo_oracle.so).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.environ.get("DVO_ORACLE_SO") or os.path.join(ORACLE_DIR, "build", "libdvo_oracle.so")      # DVO_ORACLE_SO: a sanitizer build of the oracle (CPU suite only)


class OracleParams(C.Structure):
    _fields_ = [
        ("beta", C.c_double), ("precond_rot", C.c_double), ("reg_lambda", C.c_double),
        ("step_a", C.c_double), ("step_b", C.c_double),
        ("step_decay_after", C.c_int), ("step_decay_offset", C.c_int),
        ("trust_radius", C.c_float), ("psi_norm_stop", C.c_float),
        ("enable_rotationize", C.c_int), ("enable_l2_reg", C.c_int), ("interpolate_dt", C.c_int),
    ]


class IterTrace(C.Structure):
    _fields_ = [
        ("g", C.c_double * 6), ("H", C.c_double * 21), ("sum_eps2", C.c_double),
        ("psi", C.c_double * 6), ("R", C.c_double * 9), ("t", C.c_double * 3),
        ("energy", C.c_float), ("n_visible", C.c_int), ("broke", C.c_int),
    ]


class OracleState(C.Structure):
    _fields_ = [
        ("R", C.c_double * 9), ("t", C.c_double * 3), ("d", C.c_double * 6),
        ("bestR", C.c_double * 9), ("bestT", C.c_double * 3),
        ("bestE", C.c_float), ("bestRatio", C.c_float), ("bestItr", C.c_int), ("stop", C.c_int),
    ]


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR])


_lib = None


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Oracle:
    def __init__(self, lib):
        self.lib = lib

    def default_params(self) -> OracleParams:
        p = OracleParams()
        self.lib.dvo_oracle_params_default(C.byref(p))
        return p

    def weight(self, r: float) -> float:
        return self.lib.dvo_oracle_weight(C.c_float(r))

    def enlist_ref_points(self, level, edge, depth_mm, rows, cols, K):
        edge = np.ascontiguousarray(edge, dtype=np.int32)
        depth = _f32(depth_mm)
        cap = rows * cols
        xyz = np.zeros(3 * cap, np.float32)
        uv = np.zeros(2 * cap, np.float32)
        n = self.lib.dvo_oracle_enlist_ref_points(level, _p(edge), _p(depth), rows, cols,
                                                  *[C.c_float(k) for k in K], _p(xyz), _p(uv), cap)
        assert n >= 0
        return xyz[:3 * n].reshape(-1, 3).copy(), uv[:2 * n].reshape(-1, 2).copy()

    def eval_points(self, level, xyz, dt, gx, gy, rows, cols, K, R, t, params=None):
        """R (3x3 math layout, double) and t are cast to float exactly like SolveDVO.cpp:673-674."""
        xyz = _f32(xyz).reshape(-1)
        n = xyz.size // 3
        Rf = np.asfortranarray(np.asarray(R, dtype=np.float64)).astype(np.float32, order="F")
        tf = np.asarray(t, dtype=np.float64).astype(np.float32)
        rep, J = np.zeros(3 * n, np.float32), np.zeros(6 * n, np.float32)
        eps, w, vis = np.zeros(n, np.float32), np.zeros(n, np.float32), np.zeros(n, np.int32)
        self.lib.dvo_oracle_eval_points(C.byref(params) if params is not None else None, level, _p(xyz), n,
                                        _p(_f32(dt)), _p(_f32(gx)), _p(_f32(gy)), rows, cols,
                                        *[C.c_float(k) for k in K], _p(Rf), _p(tf),
                                        _p(rep), _p(J), _p(eps), _p(w), _p(vis))
        return dict(reproj=rep.reshape(-1, 3), J=J.reshape(-1, 6), eps=eps, w=w, visible=vis)

    def run_iterations(self, level, max_iters, xyz, dt, gx, gy, rows, cols, K, R, t, params=None, trace=False):
        xyz = _f32(xyz).reshape(-1)
        n = xyz.size // 3
        R = np.array(R, dtype=np.float64, order="F").copy(order="F")
        t = np.array(t, dtype=np.float64).copy()
        energy = np.zeros(max_iters, np.float32)
        feps, frep = np.zeros(n, np.float32), np.zeros(3 * n, np.float32)
        best, ratio = C.c_int(-2), C.c_float(0)
        tr = (IterTrace * max_iters)() if trace else None
        nit = self.lib.dvo_oracle_run_iterations(
            C.byref(params) if params is not None else None, level, max_iters, _p(xyz), n,
            _p(_f32(dt)), _p(_f32(gx)), _p(_f32(gy)), rows, cols, *[C.c_float(k) for k in K],
            _p(R), _p(t), _p(energy), _p(feps), _p(frep), C.byref(best), C.byref(ratio), tr)
        out = dict(R=R, t=t, energy=energy, final_eps=feps, final_reproj=frep.reshape(-1, 3),
                   best_idx=best.value, visible_ratio=ratio.value, iters_run=nit)
        if trace:
            out["trace"] = [dict(g=np.array(x.g), H=np.array(x.H), sum_eps2=x.sum_eps2, psi=np.array(x.psi),
                                 R=np.array(x.R).reshape(3, 3, order="F"), t=np.array(x.t), energy=x.energy,
                                 n_visible=x.n_visible, broke=x.broke) for x in tr[:nit]]
        return out

    def accumulate(self, level, xyz, first, n, dt, gx, gy, rows, cols, K, R, t, params=None):
        """29 accumulators over points [first, first+n) at the float cast of (R, t)."""
        xyz = _f32(xyz).reshape(-1)
        Rf = np.asfortranarray(np.asarray(R, dtype=np.float64)).astype(np.float32, order="F")
        tf = np.asarray(t, dtype=np.float64).astype(np.float32)
        acc = np.zeros(29)
        self.lib.dvo_oracle_accumulate(C.byref(params) if params is not None else C.byref(self.default_params()),
                                       level, _p(xyz), first, n, _p(_f32(dt)), _p(_f32(gx)), _p(_f32(gy)), rows, cols,
                                       *[C.c_float(k) for k in K], _p(Rf), _p(tf), _p(acc))
        return acc

    def accumulate32(self, level, xyz, first, n, dt, gx, gy, rows, cols, K, R, t, params=None):
        """the 29 accumulators + the three limbs of the exact sum of eps^2 in [29..31] (they add exactly over shards)"""
        xyz = _f32(xyz).reshape(-1)
        Rf = np.asfortranarray(np.asarray(R, dtype=np.float64)).astype(np.float32, order="F")
        tf = np.asarray(t, dtype=np.float64).astype(np.float32)
        acc = np.zeros(32)
        self.lib.dvo_oracle_accumulate32(C.byref(params) if params is not None else C.byref(self.default_params()),
                                         level, _p(xyz), first, n, _p(_f32(dt)), _p(_f32(gx)), _p(_f32(gy)), rows, cols,
                                         *[C.c_float(k) for k in K], _p(Rf), _p(tf), _p(acc))
        return acc

    def e2_limbs(self, eps):
        eps = _f32(eps).reshape(-1)
        limbs = np.zeros(3)
        self.lib.dvo_oracle_e2_limbs(_p(eps), eps.size, _p(limbs))
        return limbs

    def e2_from_limbs(self, limbs, fallback=float("nan")):
        limbs = np.ascontiguousarray(limbs, dtype=np.float64)
        return float(self.lib.dvo_oracle_e2_from_limbs(_p(limbs), C.c_double(fallback)))

    def state_begin(self, R, t) -> OracleState:
        st = OracleState()
        R = np.array(R, dtype=np.float64, order="F")
        t = np.array(t, dtype=np.float64)
        self.lib.dvo_oracle_state_begin(C.byref(st), _p(R), _p(t))
        return st

    def state_update(self, st: OracleState, itr, N, g6, sum_eps2, n_vis, params=None):
        g6 = np.array(g6, dtype=np.float64)
        e = C.c_float(0)
        psi = np.zeros(6)
        p = params if params is not None else self.default_params()
        broke = self.lib.dvo_oracle_state_update(C.byref(p), C.byref(st), itr, N, _p(g6), C.c_double(sum_eps2),
                                                 int(n_vis), C.byref(e), _p(psi))
        return e.value, bool(broke), psi

    def state_finish(self, st: OracleState, params=None):
        R, t = np.zeros((3, 3), order="F"), np.zeros(3)
        p = params if params is not None else self.default_params()
        self.lib.dvo_oracle_state_finish(C.byref(p), C.byref(st), _p(R), _p(t))
        return R, t

    @staticmethod
    def state_pose(st: OracleState):
        return np.array(st.R).reshape(3, 3, order="F"), np.array(st.t)

    def align_pyramid(self, iters, levels, K, R, t, params=None):
        """levels: list of dict(xyz, dt, gx, gy, rows, cols).  Coarse-to-fine schedule of SolveDVO::loop."""
        R = np.array(R, dtype=np.float64, order="F").copy(order="F")
        t = np.array(t, dtype=np.float64).copy()
        reports = {}
        last = None
        for l in range(len(iters) - 1, -1, -1):
            if iters[l] <= 0:
                continue
            L = levels[l]
            r = self.run_iterations(l, iters[l], L["xyz"], L["dt"], L["gx"], L["gy"], L["rows"], L["cols"], K, R, t,
                                    params=params)
            R, t = r["R"], r["t"]
            reports[l] = r
            last = l
        return dict(R=R, t=t, levels=reports, last_level=last)

    def align_batch_omp(self, iters, scenes_levels, K, n_pairs, n_threads=0, params=None):
        """n_pairs alignments from the identity, OpenMP over pairs (pair i aligns scene i % len(scenes)).
        scenes_levels: list (scenes) of lists (levels) of dict(xyz, dt, gx, gy, rows, cols).  Returns dict(R, t, seconds, threads,
        thread_seconds, thread_pairs)."""
        ns, nl = len(scenes_levels), len(iters)
        keep = []

        def arr(key, dtype=np.float32):
            out = (C.c_void_p * (ns * nl))()
            for s, lv in enumerate(scenes_levels):
                for l in range(nl):
                    a = np.ascontiguousarray(lv[l][key], dtype=dtype)
                    keep.append(a)
                    out[s * nl + l] = a.ctypes.data
            return out
        xyz, dt, gx, gy = arr("xyz"), arr("dt"), arr("gx"), arr("gy")
        N = np.array([len(np.asarray(lv[l]["xyz"]).reshape(-1)) // 3 for lv in scenes_levels for l in range(nl)], np.int32)
        rows = np.array([lv[l]["rows"] for lv in scenes_levels for l in range(nl)], np.int32)
        cols = np.array([lv[l]["cols"] for lv in scenes_levels for l in range(nl)], np.int32)
        it = np.array(iters, np.int32)
        nt = int(n_threads) if n_threads else (len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1))
        R, t = np.zeros((n_pairs, 9)), np.zeros((n_pairs, 3))
        secs = C.c_double(0.0)
        tsec, tpairs = np.zeros(nt), np.zeros(nt, np.int32)
        p = params if params is not None else self.default_params()
        self.lib.dvo_oracle_align_batch_omp.restype = C.c_int
        used = self.lib.dvo_oracle_align_batch_omp(C.byref(p), int(n_pairs), ns, nl, _p(it), xyz, _p(N), dt, gx, gy, _p(rows), _p(cols),
                                                   C.c_float(K[0]), C.c_float(K[1]), C.c_float(K[2]), C.c_float(K[3]), nt, _p(R), _p(t),
                                                   C.byref(secs), _p(tsec), _p(tpairs))
        return dict(R=R.reshape(n_pairs, 3, 3).transpose(0, 2, 1), t=t, seconds=secs.value, threads=used, thread_seconds=tsec[:used],
                    thread_pairs=tpairs[:used])

    def now_level_from_edges(self, edge, rows, cols):
        edge = np.ascontiguousarray(edge, dtype=np.uint8)
        dt, gx, gy = (np.zeros(rows * cols, np.float32) for _ in range(3))
        self.lib.dvo_oracle_now_level_from_edges(_p(edge), rows, cols, _p(dt), _p(gx), _p(gy))
        return dt, gx, gy

    # ---- rows f1/f2 (dvo_oracle_frames.cpp); images ROW-major numpy arrays (rows, cols) ----
    def canny(self, grey, t1=150.0, t2=100.0, stages=False):
        grey = np.ascontiguousarray(grey, dtype=np.uint8)
        rows, cols = grey.shape
        dst = np.zeros((rows, cols), np.uint8)
        if not stages:
            self.lib.dvo_oracle_canny(_p(grey), rows, cols, t1, t2, _p(dst))
            return dst
        mag = np.zeros((rows, cols), np.int32)
        cand = np.zeros((rows, cols), np.uint8)
        self.lib.dvo_oracle_canny_stages(_p(grey), rows, cols, t1, t2, _p(mag), _p(cand), _p(dst))
        return dst, mag, cand

    def sobel3(self, grey):
        grey = np.ascontiguousarray(grey, dtype=np.uint8)
        rows, cols = grey.shape
        dx, dy = np.zeros((rows, cols), np.int16), np.zeros((rows, cols), np.int16)
        self.lib.dvo_oracle_sobel3(_p(grey), rows, cols, _p(dx), _p(dy))
        return dx, dy

    def bgr2gray(self, bgr):
        bgr = np.ascontiguousarray(bgr, dtype=np.uint8)
        rows, cols, _ = bgr.shape
        out = np.zeros((rows, cols), np.uint8)
        self.lib.dvo_oracle_bgr2gray(_p(bgr), rows * cols, _p(out))
        return out

    def resize_nn(self, img, scale):
        img = np.ascontiguousarray(img)
        rows, cols = img.shape[:2]
        es = img.dtype.itemsize * (img.shape[2] if img.ndim == 3 else 1)
        dr, dc = C.c_int(), C.c_int()
        self.lib.dvo_oracle_resize_nn_size(rows, cols, scale, C.byref(dr), C.byref(dc))
        out = np.zeros((dr.value, dc.value) + img.shape[2:], img.dtype)
        self.lib.dvo_oracle_resize_nn(_p(img), rows, cols, es, scale, _p(out))
        return out

    def depth_m_to_mm16(self, depth_m):
        d = _f32(depth_m)
        out = np.zeros(d.shape, np.uint16)
        self.lib.dvo_oracle_depth_m_to_mm16(_p(d), d.size, _p(out))
        return out

    def undistort_bgr8(self, bgr, K4, D5):
        """cv::undistort of a rows x cols x 3 uint8 image (camTopic2PublisherPyD.cpp:88-107)"""
        src = np.ascontiguousarray(bgr, dtype=np.uint8)
        rows, cols = src.shape[:2]
        out = np.zeros_like(src)
        K, D = np.asarray(K4, np.float64).copy(), np.asarray(D5, np.float64).copy()
        self.lib.dvo_oracle_undistort_bgr8(_p(src), rows, cols, _p(K), _p(D), _p(out))
        return out

    def undistort_u16(self, img, K4, D5):
        src = np.ascontiguousarray(img, dtype=np.uint16)
        rows, cols = src.shape
        out = np.zeros_like(src)
        K, D = np.asarray(K4, np.float64).copy(), np.asarray(D5, np.float64).copy()
        self.lib.dvo_oracle_undistort_u16(_p(src), rows, cols, _p(K), _p(D), _p(out))
        return out

    def build_pyramid(self, bgr, depth_m, n_levels=4, first_shift=1, undistort=None):
        """camTopic2PublisherPyD.cpp:73-77,322-347: per level (mono8 row-major, mono16 row-major); level i is decimated by
        2^(first_shift+i) from full resolution.  undistort = (K4, D5): the camera-info topic was received, so both images
        go through cv::undistort first (:306-308); None: undistort is a copy (:90-95)."""
        d16 = self.depth_m_to_mm16(depth_m)
        if undistort is not None:
            bgr = self.undistort_bgr8(bgr, *undistort)
            d16 = self.undistort_u16(d16.reshape(np.asarray(depth_m).shape), *undistort)
        out = []
        for i in range(n_levels):
            s = 0.5 ** (first_shift + i)
            out.append((self.bgr2gray(self.resize_nn(bgr, s)), self.resize_nn(d16, s)))
        return out

    def now_level_from_grey(self, grey_rm, t1=150.0, t2=100.0):
        """computeDistTransfrmOfNow (SolveDVO.cpp:1740-1799) of one level from the mono8 image (row-major):
        column-major dt, gx, gy and the column-major edge map (0/255)."""
        rows, cols = grey_rm.shape
        edge_rm = self.canny(grey_rm, t1, t2)
        edge_cm = np.ascontiguousarray(edge_rm.T).ravel()            # cv2eigen: (yy,xx) at yy + xx*rows
        dt, gx, gy = self.now_level_from_edges(edge_cm, rows, cols)
        return dt, gx, gy, edge_cm

    def ref_level_from_grey(self, level, grey_rm, depth16_rm, K, t1=150.0, t2=100.0):
        """computeDistTransfrmOfRef's edge map (:1700-1712) + selectedPts + enlistRefEdgePts of one level;
        depth is the mono16 image after the node's 0 -> 1 step (:514), widened to float (cv2eigen)."""
        rows, cols = grey_rm.shape
        edge_cm = np.ascontiguousarray(self.canny(grey_rm, t1, t2).T).ravel().astype(np.int32)
        d = np.ascontiguousarray(depth16_rm).astype(np.uint16).copy()
        d[d == 0] = 1
        depth_cm = np.ascontiguousarray(d.T).ravel().astype(np.float32)
        xyz, uv = self.enlist_ref_points(level, edge_cm, depth_cm, rows, cols, K)
        return xyz, uv, edge_cm

    # ---- photometric Gauss-Newton (RGBDOdometry, src/RGBDOdometry.cpp:363-746) ----
    def photo_jacobian(self, grey, depth16, level, K, fixed=False, grad_threshold=5.0, capacity=50000):
        g = np.ascontiguousarray(grey, np.uint8); d = np.ascontiguousarray(depth16, np.uint16)
        rows, cols = g.shape
        J = np.zeros((capacity, 6), np.float64); si = np.zeros(capacity, np.int32); sj = np.zeros(capacity, np.int32)
        A = np.zeros((6, 6), np.float64)
        n = self.lib.dvo_oracle_photo_jacobian(_p(g), _p(d), rows, cols, level, *[C.c_double(k) for k in K], int(fixed),
                                               C.c_double(grad_threshold), capacity, _p(J), _p(si), _p(sj), _p(A))
        if n < 0:
            raise RuntimeError("more than %d selected pixels (RGBDOdometry.cpp:464 asserts)" % capacity)
        return dict(J=J[:n].copy(), sel_i=si[:n].copy(), sel_j=sj[:n].copy(), A=A, n=n)

    def photo_epsilon(self, grey_ref, depth_ref, grey_now, level, K, jac, T, fixed=False):
        gr = np.ascontiguousarray(grey_ref, np.uint8); dr = np.ascontiguousarray(depth_ref, np.uint16)
        gn = np.ascontiguousarray(grey_now, np.uint8)
        rows, cols = gr.shape
        T = np.ascontiguousarray(T, np.float64)
        eps = np.zeros(max(jac["n"], 1), np.float64)
        self.lib.dvo_oracle_photo_epsilon.restype = C.c_double
        nrm = self.lib.dvo_oracle_photo_epsilon(_p(gr), _p(dr), _p(gn), rows, cols, level, *[C.c_double(k) for k in K], int(fixed),
                                                _p(jac["sel_i"]), _p(jac["sel_j"]), jac["n"], _p(T), _p(eps))
        return eps[:jac["n"]], nrm

    def photo_gauss_newton(self, grey_ref, depth_ref, grey_now, level, K, jac, T, fixed=False, max_iters=3, eps_stop=200.0):
        gr = np.ascontiguousarray(grey_ref, np.uint8); dr = np.ascontiguousarray(depth_ref, np.uint16)
        gn = np.ascontiguousarray(grey_now, np.uint8)
        rows, cols = gr.shape
        T = np.array(T, np.float64, order="C").copy()
        norms = np.zeros(max_iters, np.float64)
        J = np.ascontiguousarray(jac["J"], np.float64)
        nu = self.lib.dvo_oracle_photo_gauss_newton(_p(gr), _p(dr), _p(gn), rows, cols, level, *[C.c_double(k) for k in K], int(fixed),
                                                    _p(J), _p(jac["sel_i"]), _p(jac["sel_j"]), jac["n"], _p(np.ascontiguousarray(jac["A"])),
                                                    max_iters, C.c_double(eps_stop), _p(T), _p(norms))
        return T, norms, nu

    def photo_exponential_map(self, psi, fixed=False):
        out = np.zeros((4, 4), np.float64)
        self.lib.dvo_oracle_photo_exponential_map(_p(np.asarray(psi, np.float64).copy()), int(fixed), _p(out))
        return out

    def photo_solve6(self, A, b):
        x = np.zeros(6, np.float64)
        self.lib.dvo_oracle_photo_solve6(_p(np.ascontiguousarray(A, np.float64)), _p(np.asarray(b, np.float64).copy()), _p(x))
        return x

    def photo_track(self, ref_pyr, now_pyr, K, T0=None, fixed=False):
        """eventLoop's per-frame work (:162-163): gaussNewtonIterations(3, T); gaussNewtonIterations(2, T) -- pyramids are lists of
        (grey u8, depth u16) row-major for levels 0..3 (setRefFrame / setNowFrame, :296-357)"""
        T = np.eye(4) if T0 is None else np.array(T0, np.float64)
        rep = {}
        for level in (3, 2):
            jac = self.photo_jacobian(ref_pyr[level][0], ref_pyr[level][1], level, K, fixed)
            T, norms, nu = self.photo_gauss_newton(ref_pyr[level][0], ref_pyr[level][1], now_pyr[level][0], level, K, jac, T, fixed)
            rep[level] = dict(norms=norms, updates=nu, n=jac["n"])
        return T, rep

    def se3_exp(self, psi):
        psi = np.array(psi, dtype=np.float64)
        R, t = np.zeros((3, 3), order="F"), np.zeros(3)
        self.lib.dvo_oracle_se3_exp(_p(psi), _p(R), _p(t))
        return R, t

    def se3_log(self, R, t):
        R = np.array(R, dtype=np.float64, order="F")
        t = np.array(t, dtype=np.float64)
        psi = np.zeros(6)
        self.lib.dvo_oracle_se3_log(_p(R), _p(t), _p(psi))
        return psi

    def rotationize(self, R):
        R = np.array(R, dtype=np.float64, order="F").copy(order="F")
        self.lib.dvo_oracle_rotationize(_p(R))
        return R

    def svd3(self, A):
        A = np.array(A, dtype=np.float64, order="F")
        U, V, S = np.zeros((3, 3), order="F"), np.zeros((3, 3), order="F"), np.zeros(3)
        self.lib.dvo_oracle_svd3(_p(A), _p(U), _p(S), _p(V))
        return U, S, V

    def interpolate(self, F, rows, cols, ry, rx):
        return self.lib.dvo_oracle_interpolate(_p(_f32(F)), rows, cols, C.c_float(ry), C.c_float(rx))


def load() -> Oracle:
    global _lib
    if _lib is None:
        srcs = [os.path.join(ORACLE_DIR, f) for f in ("dvo_oracle.cpp", "dvo_oracle_frames.cpp", "dvo_oracle.h")]
        if not os.path.exists(ORACLE_SO) or os.path.getmtime(ORACLE_SO) < max(os.path.getmtime(f) for f in srcs):
            build()
        lib = C.CDLL(ORACLE_SO)
        lib.dvo_oracle_weight.restype = C.c_float
        lib.dvo_oracle_weight.argtypes = [C.c_float]
        lib.dvo_oracle_interpolate.restype = C.c_float
        lib.dvo_oracle_interpolate.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float]
        lib.dvo_oracle_enlist_ref_points.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int] + \
            [C.c_float] * 4 + [C.c_void_p, C.c_void_p, C.c_int]
        lib.dvo_oracle_eval_points.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                               C.c_void_p, C.c_int, C.c_int] + [C.c_float] * 4 + [C.c_void_p] * 7
        lib.dvo_oracle_eval_points.restype = None
        lib.dvo_oracle_run_iterations.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p,
                                                  C.c_void_p, C.c_void_p, C.c_int, C.c_int] + [C.c_float] * 4 + \
            [C.c_void_p] * 5 + [C.POINTER(C.c_int), C.POINTER(C.c_float), C.c_void_p]
        lib.dvo_oracle_accumulate.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                              C.c_void_p, C.c_int, C.c_int] + [C.c_float] * 4 + [C.c_void_p] * 3
        lib.dvo_oracle_accumulate.restype = None
        lib.dvo_oracle_accumulate32.argtypes = lib.dvo_oracle_accumulate.argtypes
        lib.dvo_oracle_accumulate32.restype = None
        lib.dvo_oracle_e2_limbs.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        lib.dvo_oracle_e2_limbs.restype = None
        lib.dvo_oracle_e2_from_limbs.argtypes = [C.c_void_p, C.c_double]
        lib.dvo_oracle_e2_from_limbs.restype = C.c_double
        lib.dvo_oracle_state_begin.argtypes = [C.c_void_p] * 3
        lib.dvo_oracle_state_begin.restype = None
        lib.dvo_oracle_state_update.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_double,
                                                C.c_int, C.POINTER(C.c_float), C.c_void_p]
        lib.dvo_oracle_state_update.restype = C.c_int
        lib.dvo_oracle_state_finish.argtypes = [C.c_void_p] * 4
        lib.dvo_oracle_state_finish.restype = None
        lib.dvo_oracle_now_level_from_edges.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.dvo_oracle_now_level_from_edges.restype = None
        lib.dvo_oracle_sobel3.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        lib.dvo_oracle_sobel3.restype = None
        lib.dvo_oracle_canny.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_double, C.c_void_p]
        lib.dvo_oracle_canny.restype = None
        lib.dvo_oracle_canny_stages.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_double] + [C.c_void_p] * 3
        lib.dvo_oracle_canny_stages.restype = None
        lib.dvo_oracle_bgr2gray.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
        lib.dvo_oracle_bgr2gray.restype = None
        lib.dvo_oracle_resize_nn_size.argtypes = [C.c_int, C.c_int, C.c_double, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        lib.dvo_oracle_resize_nn_size.restype = None
        lib.dvo_oracle_resize_nn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_void_p]
        lib.dvo_oracle_resize_nn.restype = None
        lib.dvo_oracle_depth_m_to_mm16.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
        lib.dvo_oracle_depth_m_to_mm16.restype = None
        lib.dvo_oracle_undistort_bgr8.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.dvo_oracle_undistort_bgr8.restype = None
        lib.dvo_oracle_undistort_u16.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.dvo_oracle_undistort_u16.restype = None
        for n in ("se3_exp", "se3_log"):
            getattr(lib, "dvo_oracle_" + n).argtypes = [C.c_void_p] * 3
        lib.dvo_oracle_rotationize.argtypes = [C.c_void_p]
        lib.dvo_oracle_svd3.argtypes = [C.c_void_p] * 4
        _lib = Oracle(lib)
    return _lib


def scene_levels(scene, oracle: Oracle):
    """Per-level hot-path inputs of a SynthScene: ref points via the oracle's enlistRefEdgePts."""
    out = []
    for l, L in enumerate(scene.levels):
        xyz, uv = oracle.enlist_ref_points(l, L.ref_edge, L.ref_depth, L.rows, L.cols, scene.intrinsics)
        out.append(dict(xyz=xyz, uv=uv, dt=L.now_dt, gx=L.now_gx, gy=L.now_gy, rows=L.rows, cols=L.cols))
    return out


def rot_angle(Ra, Rb) -> float:
    M = np.asarray(Ra).T @ np.asarray(Rb)
    return float(np.arccos(np.clip((np.trace(M) - 1.0) / 2.0, -1.0, 1.0)))
