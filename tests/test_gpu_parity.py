"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle.

Bars (BASELINE.json north_star): float32 per-point quantities bit-equal; energies and best
index bit-equal; final pose within 1e-5 rad / 1e-4 m of the oracle (in practice ~1e-15: the only
non-bit-identical ingredients are double-precision libm vs ocml sin/cos/atan and the order of
the double-precision sums, both ~1e-16 relative, far below the float cast at SolveDVO.cpp:673-674).
"""
import os

import numpy as np
import pytest

import oracle_lib
from oracle_lib import rot_angle

pytestmark = pytest.mark.gpu

ROT_TOL = 1e-5      # rad  (north_star)
TRANS_TOL = 1e-4    # m    (north_star)


def _ctx_for(scene, levels, n_pairs=1, **kw):
    from rgbd_odometry_amd import DvoContext
    ctx = DvoContext(n_pairs, **kw)
    ctx.set_intrinsics(*scene.intrinsics)
    for p in range(n_pairs):
        for l, L in enumerate(levels):
            ctx.set_ref_level(l, L["xyz"], pair=p)
            ctx.set_now_level(l, L["dt"], L["gx"], L["gy"], L["rows"], L["cols"], pair=p)
    return ctx


def _same(a, b):
    """value equality for float arrays, NaN == NaN, -0 == +0"""
    a, b = np.asarray(a), np.asarray(b)
    return np.array_equal(a, b, equal_nan=True)


@pytest.fixture(scope="module")
def scene320(oracle):
    from rgbd_odometry_amd import SynthScene
    sc = SynthScene(320, 240, 4, 0)
    return sc, oracle_lib.scene_levels(sc, oracle)


@pytest.fixture(scope="module")
def ctx320(scene320):
    sc, lv = scene320
    ctx = _ctx_for(sc, lv)
    yield ctx
    ctx.close()


def _poses(scene, oracle):
    """identity, the true pose, and a few perturbed poses (incl. points leaving the image)"""
    rng = np.random.default_rng(42)
    out = [(np.eye(3), np.zeros(3)), (scene.R_true, scene.t_true)]
    for scale in (0.01, 0.05, 0.3):
        psi = rng.standard_normal(6) * scale
        out.append(oracle.se3_exp(psi))
    return out


def test_per_point_bit_equal(scene320, ctx320, oracle):
    sc, lv = scene320
    for R, t in _poses(sc, oracle):
        for l, L in enumerate(lv):
            ref = oracle.eval_points(l, L["xyz"], L["dt"], L["gx"], L["gy"], L["rows"], L["cols"], sc.intrinsics, R, t)
            got = ctx320.eval_points(l, R, t)
            assert np.array_equal(ref["visible"], got["visible"]), (l, "visible")
            # reprojections of ALL points (visible or not) -- SolveDVO.cpp:345
            assert _same(ref["reproj"], got["reproj"]), (l, "reproj")
            for key in ("eps", "w", "J"):
                assert _same(ref[key], got[key]), (l, key, np.abs(ref[key] - got[key]).max())
            assert ref["visible"].sum() > 0


def test_accumulators(scene320, ctx320, oracle):
    sc, lv = scene320
    for R, t in _poses(sc, oracle)[:4]:
        for l, L in enumerate(lv):
            r = oracle.run_iterations(l, 1, L["xyz"], L["dt"], L["gx"], L["gy"], L["rows"], L["cols"], sc.intrinsics,
                                      R, t, trace=True)
            tr = r["trace"][0]
            acc = ctx320.accumulate(l, R, t)
            assert int(acc[28]) == tr["n_visible"]
            assert acc[27] == tr["sum_eps2"]      # round 6: both sides hold the correctly rounded EXACT sum of eps^2 -- no order, no tolerance
            np.testing.assert_allclose(acc[21:27], tr["g"], rtol=1e-11, atol=1e-9 * np.abs(tr["g"]).max())
            # H: exact float x float products summed in double on both sides; only the summation order differs
            np.testing.assert_allclose(acc[:21], tr["H"], rtol=1e-12, atol=1e-12 * np.abs(tr["H"]).max())


@pytest.mark.parametrize("lists", ["3xN-lists", "3xN-lists-foreign", "engine-lists"])
def test_wide_schedule_one_launch_per_iteration_with_and_without_normal_matrix(scene320, oracle, lists):
    """dvo_align_pyramid_wide (the single-GPU form of the tiled schedule, same tiled_step_kernel: the update of an iteration at
    the head of the next launch, the sums added by the last workgroup): energies / best index / ratio / final outputs bit-equal to
    the oracle, skipped levels included; with DVO_FLAG_NORMAL_MATRIX the same launches also form H per iterate"""
    from rgbd_odometry_amd import DvoContext
    from rgbd_odometry_amd.capi import DVO_FLAG_NORMAL_MATRIX, DVO_FLAG_FINAL_OUTPUTS
    sc, lv = scene320
    iters = [6, 0, 5, 4]
    # round 6: a caller's 3xN list that IS an enlistRefEdgePts list for the context's intrinsics (every point verifies bit for bit) gets the
    # compact twin like the engine's own lists; a foreign list -- one point moved by an ulp is enough -- keeps the one-point-per-lane route
    engine_lists = lists != "3xN-lists-foreign"
    if lists == "3xN-lists-foreign":
        lv = [dict(L) for L in lv]
        for L in lv:
            xyz = np.array(L["xyz"], np.float32).copy()
            xyz.reshape(-1)[3 * (len(xyz) // 2)] = np.nextafter(xyz.reshape(-1)[3 * (len(xyz) // 2)], np.float32(10), dtype=np.float32)
            L["xyz"] = xyz
    with DvoContext(1) as ctx:
        ctx.set_intrinsics(*sc.intrinsics)
        for l, L in enumerate(lv):
            if lists == "engine-lists":   # lists built by the engine's own kernels have the compact twin: the packed step kernel (round 5), with H too
                ctx.set_ref_level_from_images(l, sc.levels[l].ref_edge, sc.levels[l].ref_depth, L["rows"], L["cols"])
            else:
                ctx.set_ref_level(l, L["xyz"])
            ctx.set_now_level(l, L["dt"], L["gx"], L["gy"], L["rows"], L["cols"])
        ref = oracle.align_pyramid(iters, lv, sc.intrinsics, np.eye(3), np.zeros(3))
        for flags in (DVO_FLAG_FINAL_OUTPUTS, DVO_FLAG_FINAL_OUTPUTS | DVO_FLAG_NORMAL_MATRIX):
            for rep in range(2):                        # the second call replays the captured graph
                R, t = ctx.align_pyramid_wide(iters, np.eye(3), np.zeros(3), flags=flags)
                if not os.environ.get("DVO_TILED_PACKED") and not os.environ.get("DVO_TILED_SOLO_MAX"):
                    used = sum(1 << l for l in range(4) if iters[l] > 0)
                    assert ctx.wide_packed_levels() == (used if engine_lists else 0)
                    # small levels run as one launch -- unless H is asked for (its per-iterate sums come from the step launches)
                    assert ctx.wide_solo_levels() == (used if engine_lists and not (flags & DVO_FLAG_NORMAL_MATRIX) else 0)
                for l, rep_l in ref["levels"].items():
                    e, b, ratio = ctx.level_report(0, l, iters[l])
                    assert np.array_equal(e, rep_l["energy"]) and b == rep_l["best_idx"] and ratio == rep_l["visible_ratio"], (flags, l)
                assert rot_angle(ref["R"], R) <= 1e-5 and np.linalg.norm(ref["t"] - t) <= 1e-4
                last = ref["levels"][ref["last_level"]]
                fe, fr = ctx.final_outputs(0, len(last["final_eps"]))
                assert np.array_equal(fe, last["final_eps"]) and np.array_equal(fr, last["final_reproj"], equal_nan=True)
        Rc, tc = np.eye(3), np.zeros(3)
        for l in (3, 2, 0):
            L = lv[l]
            r = oracle.run_iterations(l, iters[l], L["xyz"], L["dt"], L["gx"], L["gy"], L["rows"], L["cols"], sc.intrinsics, Rc, tc, trace=True)
            for itr, tr in enumerate(r["trace"]):
                H = ctx.level_normal_matrix(0, l, itr)
                want = np.zeros((6, 6)); k = 0
                for i in range(6):
                    for j in range(i, 6):
                        want[i, j] = want[j, i] = tr["H"][k]; k += 1
                np.testing.assert_allclose(H, want, rtol=1e-12, atol=1e-12 * np.abs(want).max())
            Rc, tc = r["R"], r["t"]


def test_normal_matrix_of_every_iterate(scene320, oracle):
    """DVO_FLAG_NORMAL_MATRIX: the fused launch also keeps H = sum w J^T J (the 21 of the "21+6" accumulators) per
    iterate, in double; poses and energies are those of the plain launch"""
    from rgbd_odometry_amd import DvoContext
    from rgbd_odometry_amd.capi import DVO_FLAG_NORMAL_MATRIX
    sc, lv = scene320
    iters = [6, 0, 5, 4]
    ref = oracle.align_pyramid(iters, lv, sc.intrinsics, np.eye(3), np.zeros(3))
    # the reference's 3 x N lists (one-point-per-lane kernel); the engine's own lists (round 5: the packed kernel carries H too, on
    # 16-byte texels here); the same with now levels written natively in the compact form (its throughput look-up path)
    # round 6: the reference's 3 x N lists verify as enlistRefEdgePts lists and get the compact twin too ("lists"); the one-point-per-lane
    # kernel is asked for by engine_variant = 1 ("lists-one-point")
    for compact, want_blk in ((False, 512), ("lists-one-point", 512), (True, 512), ("native", 512), ("native", 256)):     # 256: the throughput shape of large batches
        one_point = compact == "lists-one-point"
        if one_point:
            compact = False
        with DvoContext(1, block_threads=(256 if want_blk == 256 else 0), **({"engine_variant": 1} if one_point else {})) as ctx:
            ctx.set_intrinsics(*sc.intrinsics)
            for l, L in enumerate(lv):
                if compact:
                    S = sc.levels[l]
                    ctx.set_ref_level_from_images(l, S.ref_edge, S.ref_depth, S.rows, S.cols)
                else:
                    ctx.set_ref_level(l, L["xyz"])
                if compact == "native":
                    ctx.set_now_level_from_edges(l, (np.asarray(sc.levels[l].now_edge) != 0).astype(np.uint8) * 255, L["rows"], L["cols"])
                else:
                    ctx.set_now_level(l, L["dt"], L["gx"], L["gy"], L["rows"], L["cols"])
            R, t = ctx.align_batch(iters, np.eye(3)[None], np.zeros((1, 3)), flags=DVO_FLAG_NORMAL_MATRIX)
            blk, g, packed = ctx.last_launch_shape()
            assert bool(packed) == (not one_point) and blk == want_blk and g == 1, (compact, one_point, blk, g, packed)
            if compact == "native":
                assert [ctx.level_texel_mode(0, l) for l in (0, 2, 3)] == [2, 2, 2]
            for l, rep in ref["levels"].items():
                e, b, ratio = ctx.level_report(0, l, iters[l])
                assert np.array_equal(e, rep["energy"]) and b == rep["best_idx"] and ratio == rep["visible_ratio"], (compact, l)
            Rp, tp = ctx.align_batch(iters, np.eye(3)[None], np.zeros((1, 3)))
            # same policy, same per-point bits; another kernel may add the double sums in another order (~1e-16)
            assert np.abs(R - Rp).max() <= 1e-12 and np.abs(t - tp).max() <= 1e-12
            ctx.align_batch(iters, np.eye(3)[None], np.zeros((1, 3)), flags=DVO_FLAG_NORMAL_MATRIX)
            Rc, tc = np.eye(3), np.zeros(3)
            for l in (3, 2, 0):
                L = lv[l]
                r = oracle.run_iterations(l, iters[l], L["xyz"], L["dt"], L["gx"], L["gy"], L["rows"], L["cols"],
                                          sc.intrinsics, Rc, tc, trace=True)
                for itr, tr in enumerate(r["trace"]):
                    H = ctx.level_normal_matrix(0, l, itr)
                    want = np.zeros((6, 6)); k = 0
                    for i in range(6):
                        for j in range(i, 6):
                            want[i, j] = want[j, i] = tr["H"][k]; k += 1
                    np.testing.assert_allclose(H, want, rtol=1e-12, atol=1e-12 * np.abs(want).max())
                Hb = ctx.level_normal_matrix(0, l)          # best iterate
                assert np.array_equal(Hb, ctx.level_normal_matrix(0, l, r["best_idx"]))
                assert np.all(np.linalg.eigvalsh(Hb) > -1e-9 * np.abs(Hb).max())      # positive semi-definite
                Rc, tc = r["R"], r["t"]


def test_device_se3_matches_oracle(ctx320, oracle):
    rng = np.random.default_rng(1)
    for k in range(60):
        scale = [1e-12, 3e-3, 0.3, 1.2][k % 4]
        psi = rng.standard_normal(6) * scale
        n = np.linalg.norm(psi[3:])
        if n > 3.0:
            psi[3:] *= 3.0 / n
        Ro, to = oracle.se3_exp(psi)
        Rg, tg = ctx320.se3_exp(psi)
        np.testing.assert_allclose(Rg, Ro, atol=1e-14)
        np.testing.assert_allclose(tg, to, atol=1e-14 * max(1.0, scale))
        back = ctx320.se3_log(Rg, tg)
        np.testing.assert_allclose(back, oracle.se3_log(Ro, to), atol=1e-11)
        np.testing.assert_allclose(back, psi, atol=1e-10)
        A = Ro + 1e-3 * rng.standard_normal((3, 3))
        np.testing.assert_allclose(ctx320.rotationize(A), oracle.rotationize(A), atol=1e-13)


@pytest.mark.parametrize("level,iters", [(3, 50), (2, 50), (0, 20)])
def test_run_iterations_single_level(scene320, ctx320, oracle, level, iters):
    sc, lv = scene320
    L = lv[level]
    ref = oracle.run_iterations(level, iters, L["xyz"], L["dt"], L["gx"], L["gy"], L["rows"], L["cols"],
                                sc.intrinsics, np.eye(3), np.zeros(3))
    got = ctx320.run_iterations(level, iters, np.eye(3), np.zeros(3))
    assert np.array_equal(ref["energy"], got["energy"]), (ref["energy"], got["energy"])
    assert ref["best_idx"] == got["best_idx"]
    assert ref["visible_ratio"] == got["visible_ratio"]
    assert rot_angle(ref["R"], got["R"]) <= ROT_TOL
    assert np.linalg.norm(ref["t"] - got["t"]) <= TRANS_TOL
    assert _same(ref["final_eps"], got["final_eps"])
    assert _same(ref["final_reproj"], got["final_reproj"])
    # tighter, informational: double-side agreement
    assert np.abs(ref["R"] - got["R"]).max() < 1e-9
    assert np.abs(ref["t"] - got["t"]).max() < 1e-9


def _check_pyramid(sc, lv, ctx, oracle, iters, pair=0, R0=None, t0=None):
    from rgbd_odometry_amd.capi import DVO_FLAG_FINAL_OUTPUTS
    R0 = np.eye(3) if R0 is None else R0
    t0 = np.zeros(3) if t0 is None else t0
    ref = oracle.align_pyramid(iters, lv, sc.intrinsics, R0, t0)
    R, t = ctx.align_batch(iters, R0[None], t0[None], first_pair=pair, n_pairs=1, flags=DVO_FLAG_FINAL_OUTPUTS)
    for l, rep in ref["levels"].items():
        e, b, ratio = ctx.level_report(pair, l, iters[l])
        assert np.array_equal(e, rep["energy"]), (l, e, rep["energy"])
        assert b == rep["best_idx"], l
        assert ratio == rep["visible_ratio"], l
    assert rot_angle(ref["R"], R[0]) <= ROT_TOL
    assert np.linalg.norm(ref["t"] - t[0]) <= TRANS_TOL
    last = ref["levels"][ref["last_level"]]
    feps, frep = ctx.final_outputs(pair, len(last["final_eps"]))
    assert _same(feps, last["final_eps"])
    assert _same(frep, last["final_reproj"])
    return ref, R[0], t[0]


def test_align_pyramid_reference_default(scene320, ctx320, oracle):
    """the reference's shipped configuration: 320x240, 4 levels, 50 iterations each (SolveDVO.cpp:30-33)"""
    sc, lv = scene320
    ref, R, t = _check_pyramid(sc, lv, ctx320, oracle, [50, 50, 50, 50])
    # and it actually converges toward the true motion
    assert rot_angle(sc.R_true, R) < rot_angle(sc.R_true, np.eye(3))


def test_align_pyramid_skipped_levels(scene320, ctx320, oracle):
    sc, lv = scene320
    _check_pyramid(sc, lv, ctx320, oracle, [0, 7, 0, 13])


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_align_pyramid_c2_640x480(oracle, seed):
    """BASELINE config 2: 640x480, 4 levels, 10 iterations per level"""
    from rgbd_odometry_amd import SynthScene
    sc = SynthScene(640, 480, 4, seed)
    lv = oracle_lib.scene_levels(sc, oracle)
    ctx = _ctx_for(sc, lv)
    try:
        _check_pyramid(sc, lv, ctx, oracle, [10, 10, 10, 10])
    finally:
        ctx.close()


def test_warm_start_from_previous_pose(scene320, ctx320, oracle):
    """pose is carried from frame to frame (SolveDVO.cpp:2102): start from a non-identity pose"""
    sc, lv = scene320
    R0, t0 = oracle.se3_exp(np.array([0.01, -0.005, 0.008, 0.004, -0.01, 0.006]))
    _check_pyramid(sc, lv, ctx320, oracle, [5, 5, 5, 5], R0=np.array(R0), t0=t0)


@pytest.mark.parametrize("kw", [
    dict(block_threads=256), dict(block_threads=1024),
    dict(points_in_flight=2), dict(points_in_flight=4), dict(block_threads=256, points_in_flight=4),
    dict(lds_point_bytes=-1),                 # every point read from HBM
    dict(lds_point_bytes=12 * 1024),          # 1024 points in LDS, the rest from HBM (both passes run)
    dict(block_threads=256, lds_point_bytes=12 * 700),   # budget smaller than one round -> 0 resident
])
def test_engine_tuning_variants(scene320, oracle, kw):
    """every tuning knob must leave the results bit-identical to the oracle"""
    sc, lv = scene320
    ctx = _ctx_for(sc, lv, **kw)
    try:
        _check_pyramid(sc, lv, ctx, oracle, [10, 10, 10, 10])
    finally:
        ctx.close()


def test_batch_of_independent_pairs(oracle):
    """BASELINE config 4 in miniature: distinct pairs in one launch, each equal to its own oracle run"""
    from rgbd_odometry_amd import DvoContext, SynthScene
    n = 6
    scenes = [SynthScene(320, 240, 4, 1000 + i) for i in range(n)]
    lvs = [oracle_lib.scene_levels(s, oracle) for s in scenes]
    ctx = DvoContext(n)
    try:
        ctx.set_intrinsics(*scenes[0].intrinsics)
        for p in range(n):
            for l, L in enumerate(lvs[p]):
                ctx.set_ref_level(l, L["xyz"], pair=p)
                ctx.set_now_level(l, L["dt"], L["gx"], L["gy"], L["rows"], L["cols"], pair=p)
        iters = [10, 10, 10, 10]
        R, t = ctx.align_batch(iters, np.tile(np.eye(3), (n, 1, 1)), np.zeros((n, 3)))
        for p in range(n):
            ref = oracle.align_pyramid(iters, lvs[p], scenes[p].intrinsics, np.eye(3), np.zeros(3))
            assert rot_angle(ref["R"], R[p]) <= ROT_TOL
            assert np.linalg.norm(ref["t"] - t[p]) <= TRANS_TOL
            for l, rep in ref["levels"].items():
                e, b, ratio = ctx.level_report(p, l, iters[l])
                assert np.array_equal(e, rep["energy"]) and b == rep["best_idx"] and ratio == rep["visible_ratio"]
        # run-to-run determinism: same launch again gives the same bits
        R2, t2 = ctx.align_batch(iters, np.tile(np.eye(3), (n, 1, 1)), np.zeros((n, 3)))
        assert np.array_equal(R, R2) and np.array_equal(t, t2)
    finally:
        ctx.close()


def test_enlist_ref_points_on_gpu(scene320, oracle):
    """selectedPts + enlistRefEdgePts (SolveDVO.cpp:1230-1264, :224-264): same points, same order"""
    from rgbd_odometry_amd import DvoContext
    sc, lv = scene320
    ctx = DvoContext(1)
    try:
        ctx.set_intrinsics(*sc.intrinsics)
        for l, L in enumerate(sc.levels):
            depth = L.ref_depth.copy()
            depth[::7] = 50.0          # some pixels fail the depth > 100 test
            xyz_o, uv_o = oracle.enlist_ref_points(l, L.ref_edge, depth, L.rows, L.cols, sc.intrinsics)
            xyz_g, uv_g = ctx.set_ref_level_from_images(l, L.ref_edge, depth, L.rows, L.cols)
            assert xyz_g.shape == xyz_o.shape
            assert np.array_equal(xyz_g, xyz_o) and np.array_equal(uv_g, uv_o)
    finally:
        ctx.close()


def test_edge_cases(oracle):
    """single point, all points invisible, point behind the camera, early termination"""
    from rgbd_odometry_amd import DvoContext, SynthScene
    sc = SynthScene(320, 240, 4, 3)
    lv = oracle_lib.scene_levels(sc, oracle)
    L = lv[2]
    K = sc.intrinsics
    cases = {
        "single": L["xyz"][:1],
        "ragged_65": L["xyz"][:65],
        "behind_camera": np.concatenate([L["xyz"][:40], L["xyz"][:40] * np.array([1, 1, -1], np.float32)]),
        "all_invisible": (L["xyz"][:50] + np.array([100.0, 0, 0], np.float32)),
        "z_zero": np.concatenate([L["xyz"][:10], np.array([[0.1, 0.1, 0.0]], np.float32)]),
    }
    ctx = DvoContext(1)
    try:
        ctx.set_intrinsics(*K)
        ctx.set_now_level(2, L["dt"], L["gx"], L["gy"], L["rows"], L["cols"])
        for name, xyz in cases.items():
            ctx.set_ref_level(2, xyz)
            ref = oracle.run_iterations(2, 12, xyz, L["dt"], L["gx"], L["gy"], L["rows"], L["cols"], K,
                                        np.eye(3), np.zeros(3))
            got = ctx.run_iterations(2, 12, np.eye(3), np.zeros(3))
            assert np.array_equal(ref["energy"], got["energy"]), name
            assert ref["best_idx"] == got["best_idx"], name
            assert _same(ref["final_eps"], got["final_eps"]), name
            assert _same(ref["final_reproj"], got["final_reproj"]), name
            assert rot_angle(ref["R"], got["R"]) <= ROT_TOL and np.linalg.norm(ref["t"] - got["t"]) <= TRANS_TOL, name
        # early termination: zero gradient + zero regulariser -> |psi| < 1e-7 at itr 0 (SolveDVO.cpp:872)
        ctx2 = DvoContext(1, enable_l2_reg=0)
        try:
            ctx2.set_intrinsics(*K)
            flat = np.zeros_like(L["dt"])
            ctx2.set_now_level(2, flat, flat, flat, L["rows"], L["cols"])
            ctx2.set_ref_level(2, L["xyz"])
            p = oracle.default_params()
            p.enable_l2_reg = 0
            ref = oracle.run_iterations(2, 9, L["xyz"], flat, flat, flat, L["rows"], L["cols"], K, np.eye(3),
                                        np.zeros(3), params=p)
            got = ctx2.run_iterations(2, 9, np.eye(3), np.zeros(3))
            assert ref["iters_run"] == 1
            assert np.array_equal(ref["energy"], got["energy"]) and got["best_idx"] == ref["best_idx"] == 0
        finally:
            ctx2.close()
    finally:
        ctx.close()


def test_errors_are_loud():
    from rgbd_odometry_amd import DvoContext, DvoError
    ctx = DvoContext(2)
    try:
        with pytest.raises(DvoError):
            ctx.align_batch([5], np.eye(3)[None], np.zeros((1, 3)), first_pair=0, n_pairs=1)   # nothing set
        ctx.set_intrinsics(500, 500, 160, 120)
        with pytest.raises(DvoError):
            ctx.set_ref_level(0, np.zeros((0, 3), np.float32))
        with pytest.raises(DvoError):
            ctx.set_ref_level(9, np.zeros((4, 3), np.float32))
        with pytest.raises(DvoError):
            ctx.set_ref_level(0, np.zeros((4, 3), np.float32), pair=2)
    finally:
        ctx.close()


def test_host_driven_iteration_matches_oracle(scene320, oracle):
    """dvo_iter_* (large-frame / tiled-mode path, grid over all CUs) == runIterations of the oracle"""
    import torch
    from rgbd_odometry_amd.distributed import HipTiledEngine, TiledAligner
    sc, lv = scene320
    ctx = _ctx_for(sc, lv)
    try:
        iters = [12, 12, 12, 12]
        res = TiledAligner(HipTiledEngine(ctx)).align(iters, np.eye(3), np.zeros(3))
        torch.cuda.synchronize()
        ref = oracle.align_pyramid(iters, lv, sc.intrinsics, np.eye(3), np.zeros(3))
        for l, rep in ref["levels"].items():
            got = res["levels"][l]
            assert np.array_equal(got["energy"], rep["energy"]), l
            assert got["best_idx"] == rep["best_idx"] and got["visible_ratio"] == rep["visible_ratio"]
        assert rot_angle(ref["R"], res["R"]) <= ROT_TOL and np.linalg.norm(ref["t"] - res["t"]) <= TRANS_TOL
        assert np.abs(ref["R"] - res["R"]).max() < 1e-9
    finally:
        ctx.close()


def test_point_shards_sum_to_the_full_frame(scene320, oracle):
    """tiled mode on one GPU: the sums of point shards, added like an all-reduce would -- in ANY order -- drive the same alignment
    (this is what every rank computes after ncclAllReduce of the 32 doubles).  Round 6: the energy comes from the three limbs of the
    exact sum of eps^2 (slots 29..31), so 1, 3 and 7 shards, added forwards or backwards, give the same energies bit for bit"""
    import torch
    from rgbd_odometry_amd.distributed import HipTiledEngine, shard_range
    sc, lv = scene320
    ctx = _ctx_for(sc, lv)
    try:
        eng = HipTiledEngine(ctx)
        level, iters = 1, 10
        n_total = eng.n_points(level)
        L = lv[level]
        ref = oracle.run_iterations(level, iters, L["xyz"], L["dt"], L["gx"], L["gy"], L["rows"], L["cols"],
                                    sc.intrinsics, np.eye(3), np.zeros(3))
        for world, backwards in ((3, False), (1, False), (7, True)):
            parts = [eng.new_acc() for _ in range(world)]
            eng.iter_begin(level, iters, np.eye(3), np.zeros(3))
            for itr in range(iters):
                for r in range(world):
                    f, c = shard_range(n_total, r, world)
                    eng.iter_accumulate(level, f, c, parts[r].data_ptr())
                total = torch.zeros_like(parts[0])
                for r in (range(world - 1, -1, -1) if backwards else range(world)):
                    total = total + parts[r]
                limbs = total[29:32].cpu().numpy()
                assert all(float(v).is_integer() and 0 <= v < 2.0 ** 53 for v in limbs), limbs
                eng.iter_update(level, itr, n_total, total.data_ptr())
                torch.cuda.synchronize()          # `total` must outlive the update kernel
            got = eng.iter_end(level)
            assert np.array_equal(got["energy"], ref["energy"]) and got["best_idx"] == ref["best_idx"], world
            assert rot_angle(ref["R"], got["R"]) <= ROT_TOL and np.linalg.norm(ref["t"] - got["t"]) <= TRANS_TOL
    finally:
        ctx.close()


def test_replicate_pairs(oracle):
    """dvo_replicate_pairs: slot p is a faithful device copy of pair p % n_src"""
    from rgbd_odometry_amd import DvoContext, SynthScene
    n, n_src = 7, 3
    scenes = [SynthScene(160, 120, 3, 50 + i) for i in range(n_src)]
    lvs = [oracle_lib.scene_levels(s, oracle) for s in scenes]
    ctx = DvoContext(n)
    try:
        ctx.set_intrinsics(*scenes[0].intrinsics)
        for p in range(n_src):
            for l, L in enumerate(lvs[p]):
                ctx.set_ref_level(l, L["xyz"], pair=p)
                ctx.set_now_level(l, L["dt"], L["gx"], L["gy"], L["rows"], L["cols"], pair=p)
        ctx.replicate_pairs(n_src)
        iters = [6, 6, 6]
        R, t = ctx.align_batch(iters, np.tile(np.eye(3), (n, 1, 1)), np.zeros((n, 3)))
        for p in range(n):
            assert np.array_equal(R[p], R[p % n_src]) and np.array_equal(t[p], t[p % n_src])
            ref = oracle.align_pyramid(iters, lvs[p % n_src], scenes[0].intrinsics, np.eye(3), np.zeros(3))
            assert rot_angle(ref["R"], R[p]) <= ROT_TOL and np.linalg.norm(ref["t"] - t[p]) <= TRANS_TOL
    finally:
        ctx.close()


def test_align_pyramid_wide_matches_oracle(scene320, oracle):
    """dvo_align_pyramid_wide: the host-driven schedule enqueued from C"""
    sc, lv = scene320
    ctx = _ctx_for(sc, lv)
    try:
        iters = [9, 0, 9, 9]
        R, t = ctx.align_pyramid_wide(iters, np.eye(3), np.zeros(3))
        assert ctx.wide_packed_levels() == 0b1101       # round 6: these 3 x N lists verify as enlistRefEdgePts lists: compact twin, packed step kernel
        ref = oracle.align_pyramid(iters, lv, sc.intrinsics, np.eye(3), np.zeros(3))
        for l, rep in ref["levels"].items():
            e, b, ratio = ctx.level_report(0, l, iters[l])
            assert np.array_equal(e, rep["energy"]) and b == rep["best_idx"] and ratio == rep["visible_ratio"]
        assert rot_angle(ref["R"], R) <= ROT_TOL and np.linalg.norm(ref["t"] - t) <= TRANS_TOL
    finally:
        ctx.close()


def test_config3_1920x1080_five_levels(oracle):
    """BASELINE configs[2]: 1920x1080, 5-level pyramid (level sizes by cvRound, 67.5 -> 68), fused kernel and wide path"""
    from rgbd_odometry_amd import SynthScene
    sc = SynthScene(1920, 1080, 5, 0)
    assert [(L.rows, L.cols) for L in sc.levels] == [(1080, 1920), (540, 960), (270, 480), (135, 240), (68, 120)]
    lv = oracle_lib.scene_levels(sc, oracle)
    ctx = _ctx_for(sc, lv)
    try:
        iters = [4, 4, 4, 4, 4]
        ref, R, t = _check_pyramid(sc, lv, ctx, oracle, iters)
        Rw, tw = ctx.align_pyramid_wide(iters, np.eye(3), np.zeros(3))
        for l, rep in ref["levels"].items():
            e, b, ratio = ctx.level_report(0, l, iters[l])
            assert np.array_equal(e, rep["energy"]) and b == rep["best_idx"] and ratio == rep["visible_ratio"]
        assert rot_angle(ref["R"], Rw) <= ROT_TOL and np.linalg.norm(ref["t"] - tw) <= TRANS_TOL
    finally:
        ctx.close()


def test_large_frame_wide_path(oracle):
    """a 2048x1536 frame (N0 ~ 170 k points, beyond one workgroup's LDS budget many times over)"""
    from rgbd_odometry_amd import SynthScene
    sc = SynthScene(2048, 1536, 3, 7)
    lv = oracle_lib.scene_levels(sc, oracle)
    ctx = _ctx_for(sc, lv)
    try:
        iters = [3, 3, 3]
        ref = oracle.align_pyramid(iters, lv, sc.intrinsics, np.eye(3), np.zeros(3))
        Rw, tw = ctx.align_pyramid_wide(iters, np.eye(3), np.zeros(3))
        for l, rep in ref["levels"].items():
            e, b, ratio = ctx.level_report(0, l, iters[l])
            assert np.array_equal(e, rep["energy"]) and b == rep["best_idx"] and ratio == rep["visible_ratio"]
        assert rot_angle(ref["R"], Rw) <= ROT_TOL and np.linalg.norm(ref["t"] - tw) <= TRANS_TOL
        # the fused one-workgroup kernel must agree too (points mostly streamed from HBM)
        Rf, tf = ctx.align_batch(iters, np.eye(3)[None], np.zeros((1, 3)))
        assert rot_angle(ref["R"], Rf[0]) <= ROT_TOL and np.linalg.norm(ref["t"] - tf[0]) <= TRANS_TOL
    finally:
        ctx.close()


@pytest.mark.parametrize("W,H,nl,seed", [(320, 240, 4, 0), (640, 480, 4, 1), (1000, 700, 2, 2)])
def test_now_level_from_edges_on_gpu(oracle, W, H, nl, seed):
    """f1: exact EDT -> normalise -> gradients -> texels on the device == the oracle's restatement, bit for bit"""
    from rgbd_odometry_amd import DvoContext, SynthScene
    sc = SynthScene(W, H, nl, seed)
    ctx = DvoContext(2)
    try:
        for l, L in enumerate(sc.levels):
            edge = (L.now_edge > 0).astype(np.uint8)
            dt, gx, gy = oracle.now_level_from_edges(edge, L.rows, L.cols)
            ctx.set_now_level_from_edges(l, edge, L.rows, L.cols, pair=1)
            gdt, ggx, ggy = ctx.get_now_level(l, pair=1)
            assert np.array_equal(gdt, dt), (l, np.abs(gdt - dt).max())
            assert np.array_equal(ggx, gx) and np.array_equal(ggy, gy), l
            # the planar upload path round-trips too
            ctx.set_now_level(l, dt, gx, gy, L.rows, L.cols, pair=0)
            a, b, c = ctx.get_now_level(l, pair=0)
            assert np.array_equal(a, dt) and np.array_equal(b, gx) and np.array_equal(c, gy)
    finally:
        ctx.close()


def test_now_level_from_edges_edge_cases(oracle):
    from rgbd_odometry_amd import DvoContext, DvoError
    ctx = DvoContext(1)
    try:
        rng = np.random.default_rng(3)
        for rows, cols in [(1, 70), (65, 1), (64, 64), (130, 3), (37, 129)]:
            edge = (rng.random(rows * cols) < 0.02).astype(np.uint8)
            edge[rng.integers(rows * cols)] = 1
            dt, gx, gy = oracle.now_level_from_edges(edge, rows, cols)
            ctx.set_now_level_from_edges(0, edge, rows, cols)
            gdt, ggx, ggy = ctx.get_now_level(0)
            assert np.array_equal(gdt, dt) and np.array_equal(ggx, gx) and np.array_equal(ggy, gy), (rows, cols)
        with pytest.raises(DvoError):
            ctx.set_now_level_from_edges(0, np.zeros(64, np.uint8), 8, 8)      # no edge pixel at all
    finally:
        ctx.close()


def test_interpolate_distance_transform_flag(scene320, oracle):
    """row A10: __INTERPOLATE_DISTANCE_TRANSFORM (SolveDVO.h:97, :443-444): eps from SolveDVO::interpolate"""
    sc, lv = scene320
    p = oracle.default_params()
    p.interpolate_dt = 1
    ctx = _ctx_for(sc, lv, interpolate_dt=1)
    try:
        R0, t0 = oracle.se3_exp(np.array([0.004, -0.003, 0.002, 0.003, -0.004, 0.002]))
        for l, L in enumerate(lv):
            ref = oracle.eval_points(l, L["xyz"], L["dt"], L["gx"], L["gy"], L["rows"], L["cols"], sc.intrinsics, R0, t0,
                                     params=p)
            got = ctx.eval_points(l, R0, t0)
            for key in ("eps", "w", "J", "reproj"):
                assert _same(ref[key], got[key]), (l, key)
        iters = [6, 6, 6, 6]
        ref = oracle.align_pyramid(iters, lv, sc.intrinsics, np.eye(3), np.zeros(3), params=p)
        plain = oracle.align_pyramid(iters, lv, sc.intrinsics, np.eye(3), np.zeros(3))
        assert not np.array_equal(ref["levels"][0]["energy"], plain["levels"][0]["energy"])   # the flag matters
        from rgbd_odometry_amd.capi import DVO_FLAG_FINAL_OUTPUTS
        R, t = ctx.align_batch(iters, np.eye(3)[None], np.zeros((1, 3)), flags=DVO_FLAG_FINAL_OUTPUTS)
        for l, rep in ref["levels"].items():
            e, b, ratio = ctx.level_report(0, l, iters[l])
            assert np.array_equal(e, rep["energy"]) and b == rep["best_idx"], l
        assert rot_angle(ref["R"], R[0]) <= ROT_TOL and np.linalg.norm(ref["t"] - t[0]) <= TRANS_TOL
        feps, _ = ctx.final_outputs(0, len(lv[0]["xyz"]))
        assert _same(feps, ref["levels"][0]["final_eps"])
        Rw, tw = ctx.align_pyramid_wide(iters, np.eye(3), np.zeros(3))
        assert rot_angle(ref["R"], Rw) <= ROT_TOL and np.linalg.norm(ref["t"] - tw) <= TRANS_TOL
    finally:
        ctx.close()


def test_compact_point_lists_give_the_same_bits(oracle):
    """lists built by the engine's enlist kernels are read in their 8-byte form {xx | yy << 16, Z} (packed kernel, a team
    of workgroups for a single pair); the same points handed over as a 3 x N float list take the 12-byte path of the
    one-point-per-lane kernel: bit-identical energies, indices, ratios and final outputs; poses to ~1e-16 (the double
    sums are added in another order)"""
    from rgbd_odometry_amd import DvoContext, SynthScene
    from rgbd_odometry_amd.capi import DVO_FLAG_FINAL_OUTPUTS
    for (W, H, nl, it, seed) in ((640, 480, 4, 10, 5), (1000, 700, 3, 6, 6)):      # the second one exceeds the LDS budget
        sc = SynthScene(W, H, nl, seed)
        iters = [it] * nl
        with DvoContext(1) as a, DvoContext(1) as b:
            for ctx in (a, b):
                ctx.set_intrinsics(*sc.intrinsics)
            for l, L in enumerate(sc.levels):
                xyz, _ = a.set_ref_level_from_images(l, L.ref_edge, L.ref_depth, L.rows, L.cols)     # compact available
                b.set_ref_level(l, xyz)                                                              # float list only
                for ctx in (a, b):
                    ctx.set_now_level(l, L.now_dt, L.now_gx, L.now_gy, L.rows, L.cols)
            Ra, ta = a.align_batch(iters, np.eye(3)[None], np.zeros((1, 3)), flags=DVO_FLAG_FINAL_OUTPUTS)
            Rb, tb = b.align_batch(iters, np.eye(3)[None], np.zeros((1, 3)), flags=DVO_FLAG_FINAL_OUTPUTS)
            assert np.abs(Ra - Rb).max() <= 1e-12 and np.abs(ta - tb).max() <= 1e-12
            for l in range(nl):
                ea, eb = a.level_report(0, l, it), b.level_report(0, l, it)
                assert np.array_equal(ea[0], eb[0]) and ea[1:] == eb[1:]
            fa, fb = a.final_outputs(0, a.n_points(0)), b.final_outputs(0, b.n_points(0))
            assert np.array_equal(fa[0], fb[0]) and np.array_equal(fa[1], fb[1])
            # new intrinsics invalidate the short form (it is expanded with K at run time); the float list is unaffected
            a.set_intrinsics(sc.intrinsics[0] * 1.01, sc.intrinsics[1], sc.intrinsics[2], sc.intrinsics[3])
            b.set_intrinsics(sc.intrinsics[0] * 1.01, sc.intrinsics[1], sc.intrinsics[2], sc.intrinsics[3])
            Ra, ta = a.align_batch(iters, np.eye(3)[None], np.zeros((1, 3)))
            Rb, tb = b.align_batch(iters, np.eye(3)[None], np.zeros((1, 3)))
            assert np.abs(Ra - Rb).max() <= 1e-12 and np.abs(ta - tb).max() <= 1e-12


def test_wide_path_graph_is_rebuilt_when_inputs_change(oracle):
    """dvo_align_pyramid_wide replays a captured hipGraph while its signature is unchanged; new point lists, new now
    levels or another schedule must rebuild it (results checked against the fused kernel each time)"""
    from rgbd_odometry_amd import DvoContext, SynthScene
    scenes = [SynthScene(320, 240, 4, s) for s in (11, 12)]
    with DvoContext(1) as ctx:
        ctx.set_intrinsics(*scenes[0].intrinsics)
        for rep, (sc, iters) in enumerate([(scenes[0], [6, 6, 6, 6]), (scenes[0], [6, 6, 6, 6]), (scenes[1], [6, 6, 6, 6]),
                                           (scenes[1], [4, 0, 5, 3]), (scenes[0], [4, 0, 5, 3])]):
            for l, L in enumerate(sc.levels):
                ctx.set_ref_level_from_images(l, L.ref_edge, L.ref_depth, L.rows, L.cols)
                ctx.set_now_level(l, L.now_dt, L.now_gx, L.now_gy, L.rows, L.cols)
            Rw, tw = ctx.align_pyramid_wide(iters, np.eye(3), np.zeros(3))
            Rf, tf = ctx.align_batch(iters, np.eye(3)[None], np.zeros((1, 3)))
            assert rot_angle(Rf[0], Rw) <= 1e-12 and np.linalg.norm(tf[0] - tw) <= 1e-12, rep
            lv = oracle_lib.scene_levels(sc, oracle)
            want = oracle.align_pyramid(iters, lv, sc.intrinsics, np.eye(3), np.zeros(3))
            assert rot_angle(want["R"], Rw) <= 1e-5 and np.linalg.norm(want["t"] - tw) <= 1e-4
