"""CPU tests of the oracle: the algebraic properties the cited reference lines imply.

The reference ships no tests or vectors, so these properties (plus tests/golden) are what pins the oracle.
"""
import numpy as np
import pytest

import oracle_lib
from oracle_lib import rot_angle


@pytest.fixture(scope="module")
def scene(oracle):
    from rgbd_odometry_amd import SynthScene
    sc = SynthScene(320, 240, 4, 0)
    return sc, oracle_lib.scene_levels(sc, oracle)


def test_weight_function(oracle):
    """getWeightOf (SolveDVO.cpp:1047-1053): w(0)=1, w(.5)=6/7, even, double arithmetic narrowed to float"""
    assert oracle.weight(0.0) == 1.0
    assert oracle.weight(0.5) == np.float32(6.0 / 7.0)
    for r in (0.1, 3.7, 254.9, 1e-20, 1e19):
        assert oracle.weight(r) == oracle.weight(-r)
        r2 = np.float32(r) * np.float32(r)
        assert oracle.weight(r) == np.float32(6.0 / (6.0 + float(r2) / 0.25))


def test_identity_pose_reprojects_onto_own_pixel(scene, oracle):
    """enlistRefEdgePts (:244-250) inverts the projection (:334-345): u ~ xx, v ~ yy at identity"""
    sc, lv = scene
    for l, L in enumerate(lv):
        r = oracle.eval_points(l, L["xyz"], L["dt"], L["gx"], L["gy"], L["rows"], L["cols"], sc.intrinsics,
                               np.eye(3), np.zeros(3))
        assert np.abs(r["reproj"][:, 0] - L["uv"][:, 0]).max() < 2e-3
        assert np.abs(r["reproj"][:, 1] - L["uv"][:, 1]).max() < 2e-3
        # row 2 of `reprojections` is z*(1/z): 1 or 1-2^-24 (quirk Q1 reads it back as Z)
        assert set(np.unique(r["reproj"][:, 2])) <= {np.float32(1.0), np.float32(1.0) - np.float32(2.0 ** -24)}
        # points on the first row/column can round to u or v slightly below 0 -> skipped (:371)
        interior = (L["uv"] > 0).all(axis=1)
        assert r["visible"][interior].all()


def test_enlist_order_is_column_major(scene):
    sc, lv = scene
    uv = lv[0]["uv"]
    key = uv[:, 0].astype(np.int64) * 100000 + uv[:, 1].astype(np.int64)    # xx outer, yy inner (:237-239)
    assert np.all(np.diff(key) > 0)
    assert (lv[0]["xyz"][:, 2] > 0.1).all()                                   # depth > 100 mm, in metres (:248, :1251)


def test_ref_equals_now_gives_zero_energy(oracle):
    """DT is 0 on edges (:1705-1709): aligning a frame to itself has eps == 0 at identity"""
    from rgbd_odometry_amd import SynthScene
    sc = SynthScene(160, 120, 2, 4)
    L = sc.levels[0]
    xyz, uv = oracle.enlist_ref_points(0, L.ref_edge, L.ref_depth, L.rows, L.cols, sc.intrinsics)
    # a "now" DT that is zero on the reference edge pixels.  u = (s*fx)*(x/z) + (s*cx)*zn lands within
    # ~1e-4 of the integer xx on either side, and floor() then picks xx or xx-1 (:446), so the zero
    # set is dilated by one pixel; with the undilated mask about half of the residuals are non-zero --
    # the reference has the same float artefact.
    m = (L.ref_edge > 0).reshape(L.rows, L.cols, order="F")
    d = m.copy()
    for dy in (-1, 0, 1):
        for dx in (-1, 0, 1):
            d |= np.roll(np.roll(m, dy, axis=0), dx, axis=1)
    dt = np.where(d, 0.0, 7.0).astype(np.float32).reshape(-1, order="F")
    z = np.zeros_like(dt)
    r = oracle.run_iterations(0, 3, xyz, dt, z, z, L.rows, L.cols, sc.intrinsics, np.eye(3), np.zeros(3))
    assert r["energy"][0] == 0.0
    assert r["best_idx"] >= 0 and r["visible_ratio"] > 0.99      # border points may round just outside


def test_se3_exp_log_roundtrip_and_rotationize(oracle):
    rng = np.random.default_rng(0)
    for k in range(50):
        psi = rng.standard_normal(6) * [1e-9, 1e-3, 0.3, 1.0][k % 4]
        n = np.linalg.norm(psi[3:])
        if n > 3.0:
            psi[3:] *= 3.0 / n
        R, t = oracle.se3_exp(psi)
        assert np.allclose(R.T @ R, np.eye(3), atol=1e-14) and abs(np.linalg.det(R) - 1) < 1e-13
        assert np.allclose(oracle.se3_log(R, t), psi, atol=1e-10)
        A = R + 1e-2 * rng.standard_normal((3, 3))
        P = oracle.rotationize(A)
        u, s, vt = np.linalg.svd(A)
        assert np.allclose(P, u @ vt, atol=1e-12)                 # R = U V^T  (:1271-1280)
        assert np.allclose(oracle.rotationize(P), P, atol=1e-14)  # idempotent on SO(3)
        U, S, V = oracle.svd3(A)
        assert np.allclose(U @ np.diag(S) @ V.T, A, atol=1e-13)


def test_hat_matrix_is_cross_product(oracle):
    """to_se_3(w) v = w x v (:1104-1114): d/dpsi of exp at 0 along a rotation axis"""
    w = np.array([0.3, -0.2, 0.5])
    v = np.array([1.0, 2.0, -1.5])
    eps = 1e-7
    R, _ = oracle.se3_exp(np.concatenate([np.zeros(3), eps * w]))
    assert np.allclose((R @ v - v) / eps, np.cross(w, v), atol=1e-6)


def test_step_schedule_and_trust_region(scene, oracle):
    """stepLength = 0.09/(itr>5 ? itr-4 : 1) (:773); |psi| clamped to (double)0.003f (:835-837)"""
    sc, lv = scene
    L = lv[3]
    r = oracle.run_iterations(3, 12, L["xyz"], L["dt"], L["gx"], L["gy"], L["rows"], L["cols"], sc.intrinsics,
                              np.eye(3), np.zeros(3), trace=True)
    radius = float(np.float32(0.003))
    for tr in r["trace"]:
        n = np.linalg.norm(tr["psi"])
        assert n <= radius * (1 + 1e-12)
    # with gradients this large every step is clamped onto the trust-region sphere
    assert all(abs(np.linalg.norm(tr["psi"]) - radius) < 1e-12 for tr in r["trace"])
    # compose rule cT += cR*xT; cR = cR*xR (:916-917)
    prevR, prevt = np.eye(3), np.zeros(3)
    for tr in r["trace"]:
        xR, xT = oracle.se3_exp(tr["psi"])
        assert np.allclose(tr["R"], prevR @ xR, atol=1e-12)
        assert np.allclose(tr["t"], prevt + prevR @ xT, atol=1e-14)
        prevR, prevt = tr["R"], tr["t"]


def test_state_machine_equals_run_iterations(scene, oracle):
    """the explicit state machine (used to check host-driven / multi-GPU loops) is run_iterations"""
    sc, lv = scene
    L = lv[2]
    ref = oracle.run_iterations(2, 15, L["xyz"], L["dt"], L["gx"], L["gy"], L["rows"], L["cols"], sc.intrinsics,
                                np.eye(3), np.zeros(3))
    st = oracle.state_begin(np.eye(3), np.zeros(3))
    N = len(L["xyz"])
    energies = []
    for itr in range(15):
        R, t = oracle.state_pose(st)
        acc = oracle.accumulate(2, L["xyz"], 0, N, L["dt"], L["gx"], L["gy"], L["rows"], L["cols"], sc.intrinsics, R, t)
        e, broke, _ = oracle.state_update(st, itr, N, acc[21:27], acc[27], acc[28])
        energies.append(e)
        if broke:
            break
    R, t = oracle.state_finish(st)
    assert np.array_equal(np.array(energies, np.float32), ref["energy"][:len(energies)])
    assert np.array_equal(R, ref["R"]) and np.array_equal(t, ref["t"]) and st.bestItr == ref["best_idx"]


def test_best_iterate_uses_less_or_equal(oracle):
    """`<=` at :696: among equal energies the LATEST iterate wins; energies after an early exit stay 0 (:634)"""
    from rgbd_odometry_amd import SynthScene
    sc = SynthScene(160, 120, 2, 7)
    lv = oracle_lib.scene_levels(sc, oracle)
    L = lv[1]
    flat = np.full_like(L["dt"], 3.0)          # constant DT, zero gradient: energy identical every iteration
    z = np.zeros_like(flat)
    keep = ((L["uv"] > 8).all(axis=1) & (L["uv"][:, 0] < L["cols"] - 8) & (L["uv"][:, 1] < L["rows"] - 8))
    xyz_in = L["xyz"][keep]                    # interior points stay visible while the pose drifts
    # start off identity so that the regulariser (0.05 * log(pose)/|log(pose)|, :734-743,:796) keeps psi != 0
    R0, t0 = oracle.se3_exp(np.array([1e-3, 0, 0, 0, 1e-3, 0]))
    r = oracle.run_iterations(1, 6, xyz_in, flat, z, z, L["rows"], L["cols"], sc.intrinsics, R0, t0)
    assert r["iters_run"] == 6 and np.all(r["energy"] == r["energy"][0])
    assert r["best_idx"] == 5
    # no regulariser + zero gradient -> psi = 0 -> break in iteration 0, later energies remain 0
    p = oracle.default_params()
    p.enable_l2_reg = 0
    r = oracle.run_iterations(1, 6, L["xyz"], flat, z, z, L["rows"], L["cols"], sc.intrinsics, np.eye(3), np.zeros(3),
                              params=p)
    assert r["iters_run"] == 1 and r["energy"][0] > 0 and np.all(r["energy"][1:] == 0) and r["best_idx"] == 0


def test_visibility_bounds_and_invisible_points(scene, oracle):
    """Q3/Q4: half-open bounds, NaN never visible, invisible points contribute eps = 0 and J = 0"""
    sc, lv = scene
    L = lv[1]
    xyz = np.array([[0.0, 0.0, 1.0], [100.0, 0.0, 1.0], [0.0, 0.0, 0.0], [0.1, 0.1, -1.0]], np.float32)
    r = oracle.eval_points(1, xyz, L["dt"], L["gx"], L["gy"], L["rows"], L["cols"], sc.intrinsics, np.eye(3), np.zeros(3))
    assert list(r["visible"]) == [1, 0, 0, 1] or list(r["visible"])[:3] == [1, 0, 0]
    assert r["eps"][1] == 0 and r["w"][1] == 0 and not r["J"][1].any()
    assert np.isnan(r["reproj"][2]).any()
    # there is no z > 0 test in the reference: a point behind the camera that projects inside is used
    u, v = r["reproj"][3, :2]
    assert bool(r["visible"][3]) == (0 <= u < L["cols"] and 0 <= v < L["rows"])


def test_interpolate_restatement(oracle):
    """SolveDVO::interpolate (:1285-1308): sqrt-of-weighted-squares 'bilinear'; exact at integer coordinates"""
    F = (np.arange(20, dtype=np.float32).reshape(4, 5, order="F") + 1).reshape(-1, order="F")
    assert oracle.interpolate(F, 4, 5, 2.0, 3.0) == F[2 + 3 * 4]
    a, b = F[1 + 1 * 4], F[1 + 2 * 4]
    want = np.float32(np.sqrt(np.float32(0.5) * a * a + np.float32(0.5) * b * b))
    assert abs(oracle.interpolate(F, 4, 5, 1.0, 1.5) - want) < 1e-6


def test_oracle_converges_towards_true_motion(scene, oracle):
    sc, lv = scene
    r = oracle.align_pyramid([50, 50, 50, 50], lv, sc.intrinsics, np.eye(3), np.zeros(3))
    assert rot_angle(sc.R_true, r["R"]) < 0.25 * rot_angle(sc.R_true, np.eye(3))
    assert np.linalg.norm(r["t"] - sc.t_true) < 0.35 * np.linalg.norm(sc.t_true)


def test_now_level_from_edges_brute_force(oracle):
    """exact EDT of the oracle (Felzenszwalb envelopes) against an O(n^2) brute force, and against the
    scene generator's independent implementation (Meijster, integers)"""
    rng = np.random.default_rng(5)
    rows, cols = 23, 31
    edge = (rng.random(rows * cols) < 0.05).astype(np.uint8)
    edge[7] = 1
    dt, gx, gy = oracle.now_level_from_edges(edge, rows, cols)
    E = edge.reshape(rows, cols, order="F")
    ys, xs = np.nonzero(E)
    Y, X = np.mgrid[0:rows, 0:cols]
    d2 = ((Y[..., None] - ys) ** 2 + (X[..., None] - xs) ** 2).min(axis=-1)
    raw = np.sqrt(d2.astype(np.float64)).astype(np.float32)
    # cv::normalize(0,255,NORM_MINMAX), :1774, in OpenCV 2.4's arithmetic: scale in double, the 32F->32F conversion in float
    scale_f = np.float32(255.0 * (1.0 / float(raw.max())))
    want = raw * scale_f + np.float32(0.0)
    got = dt.reshape(rows, cols, order="F")
    assert got.dtype == np.float32 and want.dtype == np.float32
    assert np.array_equal(got, want)
    assert got.min() == 0.0 and abs(float(got.max()) - 255.0) <= 255.0 * 2.0 ** -23
    # ... which is NOT the double-precision evaluation (src-min)*(255/(max-min)) on every pixel
    dbl = ((raw.astype(np.float64) - 0.0) * (255.0 / float(raw.max()))).astype(np.float32)
    assert np.abs(got - dbl).max() <= 255.0 * 2.0 ** -22
    # central differences with reflect-101 (:1077-1090): zero at the first/last column/row
    GX = gx.reshape(rows, cols, order="F")
    assert np.all(GX[:, 0] == 0) and np.all(GX[:, -1] == 0)
    assert np.array_equal(GX[:, 1:-1], np.float32(0.5) * got[:, 2:] - np.float32(0.5) * got[:, :-2])
    from rgbd_odometry_amd import SynthScene
    sc = SynthScene(320, 240, 3, 9)
    for L in sc.levels:
        a, b, c = oracle.now_level_from_edges((L.now_edge > 0).astype(np.uint8), L.rows, L.cols)
        assert np.array_equal(a, L.now_dt) and np.array_equal(b, L.now_gx) and np.array_equal(c, L.now_gy)


def test_openmp_batch_leg_gives_the_single_thread_results(oracle):
    """bench.py's `cpu_baseline_openmp` leg (OpenMP over independent pairs, oracle/dvo_oracle_batch.cpp):
    every pair of the batch comes out with the bits of the single-threaded schedule on the same scene"""
    from rgbd_odometry_amd import SynthScene
    import oracle_lib
    scs = [SynthScene(160, 120, 3, 5 + i) for i in range(2)]
    lvs = [oracle_lib.scene_levels(sc, oracle) for sc in scs]
    iters = [4, 0, 4]
    r = oracle.align_batch_omp(iters, lvs, scs[0].intrinsics, 7, n_threads=3)
    assert r["threads"] >= 1 and int(r["thread_pairs"].sum()) == 7 and r["seconds"] > 0
    for i in range(7):
        ref = oracle.align_pyramid(iters, lvs[i % 2], scs[0].intrinsics, np.eye(3), np.zeros(3))
        assert np.array_equal(ref["R"], r["R"][i]) and np.array_equal(ref["t"], r["t"][i]), i
