"""Pins of the oracle's third-party restatements against INDEPENDENT implementations that exist in this image (scipy).

The reference cannot be built here and ships no vectors (parity unpinned); what can be pinned is every
piece of third-party arithmetic the oracle restates from a published algorithm:
  Sophus SE3d::exp / log   (call sites SolveDVO.cpp:736-739, :905-907)   <- scipy.spatial.transform.Rotation + closed-form V
  Eigen JacobiSVD U*V^T    (rotationize, SolveDVO.cpp:1269-1282)         <- scipy.linalg.polar
  cv::distanceTransform(CV_DIST_L2, CV_DIST_MASK_PRECISE) (:1771)        <- scipy.ndimage.distance_transform_edt
  cv::normalize(NORM_MINMAX) float semantics (:1774)                      <- numpy float32 arithmetic, stated explicitly
and the oracle's own scalar restatement of the per-point path against the reference's MATRIX form written out in numpy:
  computeJacobianOfNowFrame + getReprojectedEpsilons (SolveDVO.cpp:306-462) as 3xN / 2x3 / 3x6 matrix products in float64
"""
import numpy as np
import pytest
from scipy.linalg import polar
from scipy.ndimage import distance_transform_edt
from scipy.spatial.transform import Rotation


def _hat(w):
    return np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]], float)


def _V(w):
    th = np.linalg.norm(w)
    W = _hat(w)
    if th < 1e-8:
        return np.eye(3) + 0.5 * W + W @ W / 6.0
    return np.eye(3) + (1 - np.cos(th)) / th**2 * W + (th - np.sin(th)) / th**3 * (W @ W)


def test_se3_exp_against_scipy(oracle):
    rng = np.random.default_rng(0)
    for k in range(300):
        scale = [1e-9, 1e-4, 0.003, 0.3, 2.0, 3.1][k % 6]
        psi = rng.standard_normal(6) * scale
        n = np.linalg.norm(psi[3:])
        if n > 3.1:
            psi[3:] *= 3.1 / n
        R, t = oracle.se3_exp(psi)
        Rs = Rotation.from_rotvec(psi[3:]).as_matrix()
        ts = _V(psi[3:]) @ psi[:3]
        assert np.abs(np.asarray(R) - Rs).max() <= 5e-15, (k, psi)
        assert np.abs(np.asarray(t) - ts).max() <= 1e-14 * max(1.0, np.abs(ts).max()), (k, psi)


def test_se3_log_against_scipy(oracle):
    rng = np.random.default_rng(1)
    for k in range(300):
        scale = [1e-9, 1e-4, 0.003, 0.3, 2.0, 3.0][k % 6]
        w = rng.standard_normal(3) * scale
        n = np.linalg.norm(w)
        if n > 3.0:
            w *= 3.0 / n
        R = Rotation.from_rotvec(w).as_matrix()
        t = rng.standard_normal(3) * (1.0 if k % 2 else 1e-3)
        psi = np.asarray(oracle.se3_log(R, t))
        ws = Rotation.from_matrix(R).as_rotvec()
        ups = np.linalg.solve(_V(ws), t)
        assert np.abs(psi[3:] - ws).max() <= 1e-12 * max(1.0, np.linalg.norm(ws)), (k, w)       # atan2-based vs scipy's path
        assert np.abs(psi[:3] - ups).max() <= 1e-11 * max(1.0, np.abs(ups).max()), (k, w)


def test_rotationize_against_scipy_polar(oracle):
    rng = np.random.default_rng(2)
    for k in range(200):
        R0 = Rotation.from_rotvec(rng.standard_normal(3)).as_matrix()
        noise = [1e-15, 1e-9, 1e-4, 0.05, 0.4][k % 5]
        A = R0 + rng.standard_normal((3, 3)) * noise
        if np.linalg.det(A) <= 0.05:
            continue
        U, _ = polar(A)                         # A = U P, U = the orthogonal polar factor = U_svd V_svd^T
        got = np.asarray(oracle.rotationize(A))
        assert np.abs(got - U).max() <= 1e-13, (k, noise)
        assert abs(np.linalg.det(got) - 1.0) <= 1e-13


@pytest.mark.parametrize("shape", [(23, 31), (1, 40), (37, 1), (64, 64), (120, 160)])
def test_distance_transform_against_scipy(oracle, shape):
    rows, cols = shape
    rng = np.random.default_rng(rows * 1000 + cols)
    for density in (0.002, 0.05, 0.5):
        E = rng.random((rows, cols)) < density
        E[rng.integers(0, rows), rng.integers(0, cols)] = True
        edge = np.asfortranarray(E.astype(np.uint8)).reshape(-1, order="F")
        dt, gx, gy = oracle.now_level_from_edges(edge, rows, cols)
        raw = distance_transform_edt(~E).astype(np.float32)        # exact Euclidean distance to the nearest edge pixel
        mx = float(raw.max())
        scale_f = np.float32(255.0 * (1.0 / mx)) if mx > 0 else np.float32(0.0)
        want = raw * scale_f + np.float32(0.0)                       # OpenCV 2.4 normalize -> convertTo in float
        assert np.array_equal(dt.reshape(rows, cols, order="F"), want), (shape, density)


def _to_se_3(w):
    """SolveDVO::to_se_3 (SolveDVO.cpp:1104-1114): the hat matrix"""
    return np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]], float)


def test_per_point_path_against_the_matrix_form(oracle):
    """The oracle evaluates a point in a simplified scalar form (structural zeros dropped, to_se_3 folded into three cross
    products).  Here the same quantities come from the reference's own formulation, line by line as matrices
    (:328-345 transform / de-homogenise / project, :379-406 G * A1 * A2 with A2 = [-cR^T | to_se_3(cR^T * p)], :446 nearest
    look-up, :1047-1053 weight) in float64 -- an independent derivation, so agreement is to float32 rounding, not bit for bit:
    the quirks (Q1: A1 is evaluated at the DE-HOMOGENISED point, Q2: cR^T applied a second time) must be reproduced to agree
    at all."""
    from rgbd_odometry_amd import SynthScene
    sc = SynthScene(320, 240, 3, 17)
    fx, fy, cx, cy = sc.intrinsics
    rng = np.random.default_rng(3)
    for level in (0, 2):
        L = oracle_lib_levels(sc, oracle)[level]
        rows, cols = L["rows"], L["cols"]
        dt = np.asarray(L["dt"], np.float64).reshape(cols, rows).T           # (yy, xx) at yy + xx*rows
        gx = np.asarray(L["gx"], np.float64).reshape(cols, rows).T
        gy = np.asarray(L["gy"], np.float64).reshape(cols, rows).T
        P = np.asarray(L["xyz"], np.float64).T                               # 3 x N
        for trial in range(3):
            psi = rng.standard_normal(6) * [0.002, 0.01, 0.05][trial]           # (at the identity every point lands on a pixel corner)
            R, t = oracle.se3_exp(psi)
            R = np.asarray(R, np.float32).astype(np.float64)                  # the reference casts the pose to float (:673-674)
            t = np.asarray(t, np.float32).astype(np.float64)
            got = oracle.eval_points(level, L["xyz"], L["dt"], L["gx"], L["gy"], rows, cols, sc.intrinsics, R, t)
            s = 2.0 ** (-level)
            K = np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1]], float)
            S = np.diag([s, s, 1.0])
            Pt = R.T @ (P - t[:, None])                                      # :330
            Pt = Pt / Pt[2]                                                  # :339-341 (all three rows: Z becomes 1)
            rep = S @ K @ Pt                                                 # :344
            n_checked = 0
            for i in range(P.shape[1]):
                u, v = rep[0, i], rep[1, i]
                inside = (0 <= u < cols) and (0 <= v < rows)
                if abs(u - round(u)) < 1e-3 or abs(v - round(v)) < 1e-3 or min(u, v) < 1e-3 or cols - u < 1e-3 or rows - v < 1e-3:
                    continue                                                  # float32 vs float64 may land on another pixel
                assert bool(got["visible"][i]) == inside, (level, trial, i, u, v)
                assert abs(got["reproj"][i, 0] - u) <= 2e-3 and abs(got["reproj"][i, 1] - v) <= 2e-3
                if not inside:
                    continue
                xx, yy = int(u), int(v)
                X, Y, Z = Pt[:, i]                                           # Q1: the de-homogenised point, Z == 1
                G = np.array([[gx[yy, xx], gy[yy, xx]]])
                A1 = np.array([[s * fx / Z, 0, -s * fx * X / (Z * Z)], [0, s * fy / Z, -s * fy * Y / (Z * Z)]])
                A2 = np.hstack([-R.T, _to_se_3(R.T @ Pt[:, i])])             # Q2: cR^T once more (:399)
                J = (G @ A1 @ A2).ravel()
                scale = max(1.0, np.abs(J).max())
                assert np.abs(got["J"][i] - J).max() <= 2e-4 * scale, (level, trial, i, got["J"][i], J)
                assert got["eps"][i] == np.float32(dt[yy, xx])
                w = 6.0 / (6.0 + (float(np.float32(dt[yy, xx])) ** 2) / .25)
                assert abs(got["w"][i] - w) <= 1e-6
                n_checked += 1
            assert n_checked > 0.5 * P.shape[1] * (0.5 if trial == 2 else 0.8), (level, trial, n_checked)


def oracle_lib_levels(sc, oracle):
    import oracle_lib
    return oracle_lib.scene_levels(sc, oracle)
