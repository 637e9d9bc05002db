"""CPU properties of the photometric Gauss-Newton oracle (oracle/dvo_oracle_photo.cpp; RGBDOdometry.cpp:363-746)."""
import numpy as np
from scipy.spatial.transform import Rotation

import frame_gen

K = (525.0, 525.0, 319.5, 239.5)


def _pyr(oracle, bgr, d16):
    return [(oracle.bgr2gray(oracle.resize_nn(bgr, 0.5 ** l)), oracle.resize_nn(d16, 0.5 ** l)) for l in range(4)]


def test_solver_and_exponential_map(oracle):
    rng = np.random.default_rng(0)
    for _ in range(50):
        A = rng.standard_normal((40, 6)); A = A.T @ A
        b = rng.standard_normal(6)
        np.testing.assert_allclose(oracle.photo_solve6(A, b), np.linalg.solve(A, b), rtol=1e-9, atol=1e-12)
    # rank-deficient: colPivHouseholderQr's basic solution (free components zero) still satisfies A x = b for consistent b
    A = np.zeros((6, 6)); A[:3, :3] = np.diag([4.0, 2.0, 1.0])
    x = oracle.photo_solve6(A, np.array([4.0, 2.0, 1.0, 0, 0, 0]))
    np.testing.assert_allclose(x, [1, 1, 1, 0, 0, 0], atol=1e-14)
    psi = np.array([1.0, -2.0, 0.5, 0.01, -0.02, 0.03])
    E = oracle.photo_exponential_map(psi)
    np.testing.assert_allclose(E[:3, :3], Rotation.from_rotvec(psi[3:]).as_matrix(), atol=1e-15)
    assert np.array_equal(E[3], [0, 0, 0, 1])
    # defect D7 (:727-731): a pure translation is dropped unless fixed
    assert np.array_equal(oracle.photo_exponential_map([1, 2, 3, 0, 0, 0]), np.eye(4))
    np.testing.assert_allclose(oracle.photo_exponential_map([1, 2, 3, 0, 0, 0], fixed=True)[:3, 3], [1, 2, 3])


def test_jacobian_definition_and_defects(oracle):
    bgr, depth = frame_gen.camera_frame(3, 480, 640)
    d16 = np.clip(np.nan_to_num(np.round(depth * 1000.0), nan=0.0, posinf=65535, neginf=0), 1, 65535).astype(np.uint16)
    grey, dep = _pyr(oracle, bgr, d16)[2]
    jac = oracle.photo_jacobian(grey, dep, 2, K)
    rows, cols = grey.shape
    g = grey.astype(np.float64)
    gx = np.empty_like(g); gx[:, :-1] = g[:, 1:] - g[:, :-1]; gx[:, -1] = g[:, -2] - g[:, -1]          # [0 -1 1], reflect-101
    gy = np.empty_like(g); gy[:-1] = g[1:] - g[:-1]; gy[-1] = g[-2] - g[-1]
    sel = np.argwhere((gx >= 5).T)                     # column-major scan: (j, i) pairs in order
    assert jac["n"] == len(sel) and np.array_equal(jac["sel_j"], sel[:, 0]) and np.array_equal(jac["sel_i"], sel[:, 1])
    i, j = jac["sel_i"], jac["sel_j"]
    Z = dep[i, j].astype(np.float64)
    fx, fy, cx, cy = K
    np.testing.assert_allclose(jac["J"][:, 0], fx * fx / Z, rtol=1e-15)                               # D1 reproduced
    np.testing.assert_allclose(jac["J"][:, 1], fy * gy[i, j] / Z, rtol=1e-15)
    np.testing.assert_allclose(jac["A"], jac["J"].T @ jac["J"], rtol=1e-12)
    fixed = oracle.photo_jacobian(grey, dep, 2, K, fixed=True)
    s = 0.25
    np.testing.assert_allclose(fixed["J"][:, 0], (fx * s) * gx[i, j] / Z, rtol=1e-15)                 # D1 + D4 corrected


def test_identical_frames_stop_at_once(oracle):
    bgr, depth = frame_gen.camera_frame(4, 480, 640)
    d16 = np.clip(np.nan_to_num(np.round(depth * 1000.0), nan=0.0, posinf=65535, neginf=0), 1, 65535).astype(np.uint16)
    pyr = _pyr(oracle, bgr, d16)
    T, rep = oracle.photo_track(pyr, pyr, K)
    assert np.array_equal(T, np.eye(4))
    for l in (3, 2):
        assert rep[l]["updates"] == 0 and rep[l]["norms"][0] < 200 and rep[l]["norms"][1] == -1      # :556
