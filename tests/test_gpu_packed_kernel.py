"""GPU parity of the packed two-points-per-lane fused kernel (rgbd_odometry_amd/csrc/dvo_fused.hip) against the CPU oracle.

The packed kernel runs whenever every reference list of a launch has the engine's compact form, i.e. was built by the
enlist kernels (dvo_set_ref_level_from_images / dvo_frames_as_ref) -- the same entry points the bench uses.  Bars as in
test_gpu_parity.py: energies / best index / visible ratio / final outputs bit-equal, pose within 1e-5 rad / 1e-4 m.
"""
import numpy as np
import pytest

import oracle_lib
from oracle_lib import rot_angle

pytestmark = pytest.mark.gpu

ROT_TOL, TRANS_TOL = 1e-5, 1e-4


def _modes(ctx, n=4):
    """texel modes of the last launch; under DVO_COMPACT_NOW=eager (the whole suite through the compact now form,
    tests/test_gpu_compact_now.py covers it by default) a gathered level reads mode 2 instead of 0"""
    import os
    m = [ctx.level_texel_mode(0, l) for l in range(n)]
    if os.environ.get("DVO_COMPACT_NOW") == "eager":
        return None             # every level reads 2 then (the compact form comes before LDS staging): nothing to tell apart
    return m


def _same(a, b):
    return np.array_equal(np.asarray(a), np.asarray(b), equal_nan=True)


def _load(ctx, sc, pair=0):
    """reference lists through the GPU's own enlistRefEdgePts (compact form available), now levels as planar floats"""
    lists = []
    for l, L in enumerate(sc.levels):
        xyz, _ = ctx.set_ref_level_from_images(l, L.ref_edge, L.ref_depth, L.rows, L.cols, pair=pair)
        ctx.set_now_level(l, L.now_dt, L.now_gx, L.now_gy, L.rows, L.cols, pair=pair)
        lists.append(xyz)
    return lists


def _check(ctx, oracle, sc, lv, iters, pair=0, R0=None, t0=None):
    from rgbd_odometry_amd.capi import DVO_FLAG_FINAL_OUTPUTS
    R0 = np.eye(3) if R0 is None else R0
    t0 = np.zeros(3) if t0 is None else t0
    ref = oracle.align_pyramid(iters, lv, sc.intrinsics, R0, t0)
    R, t = ctx.align_batch(iters, R0[None], t0[None], first_pair=pair, n_pairs=1, flags=DVO_FLAG_FINAL_OUTPUTS)
    for l, rep in ref["levels"].items():
        e, b, ratio = ctx.level_report(pair, l, iters[l])
        assert np.array_equal(e, rep["energy"]), (l, e, rep["energy"])
        assert b == rep["best_idx"] and ratio == rep["visible_ratio"], l
    assert rot_angle(ref["R"], R[0]) <= ROT_TOL and np.linalg.norm(ref["t"] - t[0]) <= TRANS_TOL
    last = ref["levels"][ref["last_level"]]
    feps, frep = ctx.final_outputs(pair, len(last["final_eps"]))
    assert _same(feps, last["final_eps"]) and _same(frep, last["final_reproj"])
    return ref


@pytest.mark.parametrize("kw", [
    dict(),                                          # auto: a single pair -> a team of 8 workgroups
    dict(team_size=1),                               # one 512-thread workgroup, coarse levels staged into LDS
    dict(engine_variant=2),                          # never stage texels into LDS
    dict(engine_variant=1),                          # the one-point-per-lane kernel on the same compact lists
    dict(engine_variant=3),                          # every wave through the literal-division fallback of the packed kernel
    dict(engine_variant=3, team_size=1), dict(engine_variant=2, team_size=1),
    dict(engine_variant=5), dict(engine_variant=5, team_size=1),      # every energy from the exact sweep (16-byte texels here)
    dict(block_threads=256), dict(block_threads=1024),
    dict(lds_point_bytes=-1),                        # every point streamed from HBM
    dict(lds_point_bytes=16 * 1024),                 # 2048 points resident, the rest streamed (both passes run)
    dict(block_threads=256, lds_point_bytes=3000),   # budget below one round -> nothing resident
])
def test_packed_kernel_variants_640x480(oracle, kw):
    """C2 (640x480, 4 levels, 10 iterations) through every launch shape of the packed kernel"""
    from rgbd_odometry_amd import DvoContext, SynthScene
    sc = SynthScene(640, 480, 4, 3)
    lv = oracle_lib.scene_levels(sc, oracle)
    with DvoContext(1, **kw) as ctx:
        ctx.set_intrinsics(*sc.intrinsics)
        _load(ctx, sc)
        _check(ctx, oracle, sc, lv, [10, 10, 10, 10])
        modes = _modes(ctx)
        variant = kw.get("engine_variant", 0)
        team_on = variant != 1 and kw.get("team_size", 0) != 1 and kw.get("block_threads", 0) in (0, 512)
        if variant != 1:            # engine_variant = 3 really runs the literal-division code, nothing else does here
            assert [ctx.level_exact_fallback(0, l) for l in range(4)] == [variant == 3] * 4
            assert [ctx.level_energy_sweeps(0, l) for l in range(4)] == [10 if variant == 5 else 0] * 4
        if variant == 1 or modes is None:
            pass                                      # the other kernel: modes untouched
        elif team_on or variant == 2 or kw.get("lds_point_bytes", 0) != 0:
            assert modes == [0, 0, 0, 0], modes       # teams read their texels through L2; no LDS budget / staging switched off
        elif kw.get("block_threads") == 256:
            assert modes[0] == 0 and modes[1] == 0, modes     # 77 KB per workgroup: level 3 (80x60 texels = 77 KB) does not fit beside its points
        else:
            assert modes == [0, 0, 0, 1], modes       # level 3 (80x60) lives in LDS, texels and points


def test_lds_staged_levels_reference_default(oracle):
    """the reference's own configuration (320x240 first level, 50 iterations): levels 2 and 3 are staged into LDS"""
    from rgbd_odometry_amd import DvoContext, SynthScene
    sc = SynthScene(320, 240, 4, 9)
    lv = oracle_lib.scene_levels(sc, oracle)
    with DvoContext(1, block_threads=512, team_size=1) as ctx:
        ctx.set_intrinsics(*sc.intrinsics)
        _load(ctx, sc)
        _check(ctx, oracle, sc, lv, [50, 50, 50, 50])
        assert _modes(ctx) in ([0, 0, 1, 1], None)
        # warm start + skipped level
        R0, t0 = oracle.se3_exp(np.array([0.01, -0.005, 0.008, 0.004, -0.01, 0.006]))
        _check(ctx, oracle, sc, lv, [7, 0, 9, 3], R0=np.array(R0), t0=t0)


def _random_frame(rng, rows, cols, k):
    """edge map, depth and an arbitrary (NOT derived) now level"""
    edge = (rng.random((cols, rows)) < rng.choice([0.02, 0.2, 0.6])).astype(np.int32).reshape(-1) * 255
    depth = rng.choice([0.0, 50.0, 100.0, 101.0, 400.0, 2500.0, 65535.0], size=rows * cols,
                       p=[0.05, 0.05, 0.05, 0.15, 0.3, 0.35, 0.05]).astype(np.float32)
    if k % 2:
        depth = np.where(depth > 100, depth + rng.integers(0, 900, rows * cols), depth).astype(np.float32)
    kind = k % 4
    n = rows * cols
    if kind == 0:
        dt = rng.uniform(0, 255, n); gx = rng.normal(0, 3, n); gy = rng.normal(0, 3, n)
    elif kind == 1:
        dt = rng.normal(0, 1e3, n); gx = rng.normal(0, 1e4, n); gy = rng.normal(0, 1e-4, n)
    elif kind == 2:
        dt = np.abs(rng.normal(0, 1e-18, n)); gx = rng.normal(0, 1e18, n); gy = np.zeros(n)
    else:
        dt = rng.integers(0, 4, n).astype(float); gx = rng.integers(-2, 3, n).astype(float); gy = gx[::-1].copy()
    return edge, depth, dt.astype(np.float32), gx.astype(np.float32), gy.astype(np.float32)


@pytest.mark.parametrize("seed", range(4))
def test_packed_kernel_random_stress(oracle, seed):
    """arbitrary images at the boundary (random edges / depths / DT / gradients, all image sizes incl. tiny ones that are
    staged into LDS), random start poses that throw points out of view and behind the camera"""
    from rgbd_odometry_amd import DvoContext
    rng = np.random.default_rng(777 + seed)
    with DvoContext(1) as ctx:
        for k in range(8):
            rows, cols = int(rng.integers(4, 130)), int(rng.integers(4, 170))
            level = int(rng.integers(0, 4))
            s = 2.0 ** (-level)
            fx, fy = (float(x) for x in rng.uniform(0.6, 1.6, 2) * cols / s)
            cx, cy = float(rng.uniform(0.3, 0.7) * cols / s), float(rng.uniform(0.3, 0.7) * rows / s)
            K = tuple(float(np.float32(v)) for v in (fx, fy, cx, cy))
            edge, depth, dt, gx, gy = _random_frame(rng, rows, cols, k)
            ctx.set_intrinsics(*K)
            xyz, _ = ctx.set_ref_level_from_images(level, edge, depth, rows, cols)
            if xyz.size == 0:
                continue
            ctx.set_now_level(level, dt, gx, gy, rows, cols)
            for scale in (0.0, 0.02, 0.5):
                R0, t0 = oracle.se3_exp(rng.standard_normal(6) * scale)
                ref = oracle.run_iterations(level, 6, xyz, dt, gx, gy, rows, cols, K, R0, t0)
                got = ctx.run_iterations(level, 6, R0, t0)
                assert _same(ref["energy"], got["energy"]), (seed, k, scale, ref["energy"], got["energy"])
                assert ref["best_idx"] == got["best_idx"] and ref["visible_ratio"] == got["visible_ratio"], (seed, k, scale)
                assert _same(ref["final_eps"], got["final_eps"]) and _same(ref["final_reproj"], got["final_reproj"]), (seed, k, scale)
                if np.all(np.isfinite(ref["R"])) and np.all(np.isfinite(got["R"])):
                    assert rot_angle(ref["R"], got["R"]) <= ROT_TOL and np.linalg.norm(ref["t"] - got["t"]) <= TRANS_TOL


def test_degenerate_depth_takes_the_exact_fallback(oracle):
    """a start pose whose translation EQUALS a reference point puts that point at z = 0 exactly: the fast reciprocal is out
    of its proven range, the wave redoes its share with the literal IEEE divisions -- same bits as the oracle"""
    from rgbd_odometry_amd import DvoContext, SynthScene
    sc = SynthScene(320, 240, 3, 21)
    lv = oracle_lib.scene_levels(sc, oracle)
    with DvoContext(1) as ctx:
        ctx.set_intrinsics(*sc.intrinsics)
        lists = _load(ctx, sc)
        for level in (0, 2):
            xyz = np.asarray(lists[level]).reshape(-1, 3)
            for idx in (0, len(xyz) // 2, len(xyz) - 1):
                t0 = xyz[idx].astype(np.float64)              # P - cT == 0 for this point, float-exactly
                L = lv[level]
                ref = oracle.run_iterations(level, 5, L["xyz"], L["dt"], L["gx"], L["gy"], L["rows"], L["cols"],
                                            sc.intrinsics, np.eye(3), t0)
                got = ctx.run_iterations(level, 5, np.eye(3), t0)
                assert ctx.level_exact_fallback(0, level)          # the fallback is what ran, not the fast path by luck
                assert _same(ref["energy"], got["energy"]), (level, idx, ref["energy"], got["energy"])
                assert ref["best_idx"] == got["best_idx"] and ref["visible_ratio"] == got["visible_ratio"]
                assert _same(ref["final_eps"], got["final_eps"]) and _same(ref["final_reproj"], got["final_reproj"])


def test_overflowing_invisible_point_takes_the_exact_fallback(oracle):
    """a point whose z is tiny but INSIDE the range of the fast reciprocal (1e-37) gets coordinates so large that
    s*fx*x/z overflows: the point is not visible (u = inf), but in the packed loop -- which keeps the coordinates of invisible
    lanes -- 0 * inf = NaN would reach the lane's sums.  The per-iteration finiteness test sends such a wave through the
    literal scalar code, which skips the point like the reference (:371): same bits as the oracle, finite energies."""
    from rgbd_odometry_amd import DvoContext, SynthScene
    sc = SynthScene(320, 240, 3, 21)
    lv = oracle_lib.scene_levels(sc, oracle)
    with DvoContext(1) as ctx:
        ctx.set_intrinsics(*sc.intrinsics)
        lists = _load(ctx, sc)
        level = 0
        xyz = np.asarray(lists[level]).reshape(-1, 3)
        for idx in (1, len(xyz) // 3):
            t0 = xyz[idx].astype(np.float64)
            t0[0] = float(np.float32(t0[0]) - np.float32(0.25))        # d0 = 0.25 (float-exactly), d1 = d2 = 0
            R0 = np.eye(3)
            R0[0, 2] = 2e-37                                            # p2 = cR(0,2) * d0 = 5e-38: tiny, yet a normal float; x/z = 5e36, s*fx*x/z = inf
            L = lv[level]
            ref = oracle.run_iterations(level, 5, L["xyz"], L["dt"], L["gx"], L["gy"], L["rows"], L["cols"], sc.intrinsics, R0, t0)
            got = ctx.run_iterations(level, 5, R0, t0)
            assert np.all(np.isfinite(ref["energy"])) and np.all(np.isfinite(got["energy"]))
            assert ctx.level_exact_fallback(0, level)
            assert _same(ref["energy"], got["energy"]), (idx, ref["energy"], got["energy"])
            assert ref["best_idx"] == got["best_idx"] and ref["visible_ratio"] == got["visible_ratio"]
            assert _same(ref["final_eps"], got["final_eps"]) and _same(ref["final_reproj"], got["final_reproj"])


def test_final_outputs_refused_after_the_reference_list_was_replaced(oracle):
    """the packed kernel stores finalEpsilons / finalReprojections in the order of its compact point list and the
    getter permutes them with that list's index.  Once the pair's reference list has been rewritten (a key-frame switch, a new
    dvo_set_ref_level*) that index describes other points: the getter must refuse, not hand out scrambled values; other pairs
    and a fresh alignment are unaffected."""
    from rgbd_odometry_amd import DvoContext, SynthScene
    from rgbd_odometry_amd.capi import DVO_FLAG_FINAL_OUTPUTS, DvoError, DVO_ERR_STATE
    a, b = SynthScene(320, 240, 3, 31), SynthScene(320, 240, 3, 32)
    iters = [5, 5, 5]
    with DvoContext(2) as ctx:
        ctx.set_intrinsics(*a.intrinsics)
        _load(ctx, a, pair=0)
        _load(ctx, b, pair=1)
        ref_a = _check(ctx, oracle, a, oracle_lib.scene_levels(a, oracle), iters, pair=0)
        n0 = len(ref_a["levels"][0]["final_eps"])
        ctx.align_batch(iters, np.tile(np.eye(3), (2, 1, 1)), np.zeros((2, 3)), flags=DVO_FLAG_FINAL_OUTPUTS)
        feps1, _ = ctx.final_outputs(1, 1 << 17)
        # pair 0 gets another reference list (scene b's) at the finest level AFTER the alignment
        L = b.levels[0]
        ctx.set_ref_level_from_images(0, L.ref_edge, L.ref_depth, L.rows, L.cols, pair=0)
        with pytest.raises(DvoError) as ei:
            ctx.final_outputs(0, n0)
        assert ei.value.code == DVO_ERR_STATE and "replaced" in str(ei.value)
        feps1b, _ = ctx.final_outputs(1, 1 << 17)               # the other pair's outputs are still there
        assert np.array_equal(feps1, feps1b)
        # a fresh alignment of pair 0 (ref of b against now of a: nothing to compare with, but it must come back whole)
        ctx.align_batch(iters, np.eye(3)[None], np.zeros((1, 3)), first_pair=0, n_pairs=1, flags=DVO_FLAG_FINAL_OUTPUTS)
        fe, fr = ctx.final_outputs(0, 1 << 17)
        assert len(fe) == ctx._N[(0, 0)] and np.all(np.isfinite(fe)) and np.all(fr[:, 2] > 0.99)


def test_config4_batch_of_distinct_640x480_pairs(oracle):
    """BASELINE configs[3] at its per-GPU share: 32 DISTINCT 640x480 pairs in one launch (compact lists, packed kernel),
    every pair checked against its own oracle run"""
    from rgbd_odometry_amd import DvoContext, SynthScene
    n = 32
    scenes = [SynthScene(640, 480, 4, 1000 + i) for i in range(n)]
    iters = [10, 10, 10, 10]
    with DvoContext(n) as ctx:
        ctx.set_intrinsics(*scenes[0].intrinsics)
        for p, sc in enumerate(scenes):
            _load(ctx, sc, pair=p)
        R, t = ctx.align_batch(iters, np.tile(np.eye(3), (n, 1, 1)), np.zeros((n, 3)))
        worst_r = worst_t = 0.0
        for p, sc in enumerate(scenes):
            lv = oracle_lib.scene_levels(sc, oracle)
            ref = oracle.align_pyramid(iters, lv, sc.intrinsics, np.eye(3), np.zeros(3))
            for l, rep in ref["levels"].items():
                e, b, ratio = ctx.level_report(p, l, iters[l])
                assert np.array_equal(e, rep["energy"]) and b == rep["best_idx"] and ratio == rep["visible_ratio"], (p, l)
            worst_r = max(worst_r, rot_angle(ref["R"], R[p]))
            worst_t = max(worst_t, float(np.linalg.norm(ref["t"] - t[p])))
        assert worst_r <= ROT_TOL and worst_t <= TRANS_TOL
        R2, t2 = ctx.align_batch(iters, np.tile(np.eye(3), (n, 1, 1)), np.zeros((n, 3)))      # run-to-run determinism
        assert np.array_equal(R, R2) and np.array_equal(t, t2)


@pytest.mark.parametrize("n_pairs,team", [(1, 0), (1, 2), (3, 4), (5, 8), (9, 16), (12, 0), (32, 0), (8, 16), (2, 32)])
def test_team_mode_small_batches(oracle, n_pairs, team):
    """small batches: G workgroups share each pair (contiguous shares of every level's points, sums exchanged through L2,
    identical update on every member) -- same bits as the oracle, for every team size and for pair counts that are not
    multiples of the 8 XCDs"""
    from rgbd_odometry_amd import DvoContext, SynthScene
    from rgbd_odometry_amd.capi import DVO_FLAG_FINAL_OUTPUTS
    scenes = [SynthScene(640, 480, 4, 2000 + (i % 3)) for i in range(min(n_pairs, 3))]
    lvs = [oracle_lib.scene_levels(s, oracle) for s in scenes]
    iters = [10, 10, 10, 10]
    refs = [oracle.align_pyramid(iters, lv, s.intrinsics, np.eye(3), np.zeros(3)) for s, lv in zip(scenes, lvs)]
    with DvoContext(n_pairs, team_size=team) as ctx:
        ctx.set_intrinsics(*scenes[0].intrinsics)
        for p in range(n_pairs):
            _load(ctx, scenes[p % len(scenes)], pair=p)
        R, t = ctx.align_batch(iters, np.tile(np.eye(3), (n_pairs, 1, 1)), np.zeros((n_pairs, 3)), flags=DVO_FLAG_FINAL_OUTPUTS)
        for p in range(n_pairs):
            ref = refs[p % len(scenes)]
            for l, rep in ref["levels"].items():
                e, b, ratio = ctx.level_report(p, l, iters[l])
                assert np.array_equal(e, rep["energy"]) and b == rep["best_idx"] and ratio == rep["visible_ratio"], (p, l)
            assert rot_angle(ref["R"], R[p]) <= ROT_TOL and np.linalg.norm(ref["t"] - t[p]) <= TRANS_TOL
            last = ref["levels"][ref["last_level"]]
            feps, frep = ctx.final_outputs(p, len(last["final_eps"]))
            assert _same(feps, last["final_eps"]) and _same(frep, last["final_reproj"]), p
        # the one-workgroup-per-pair launch: the double sums are added in another order, so ~1e-16, not bit-identical
        with DvoContext(n_pairs, team_size=1) as solo:
            solo.set_intrinsics(*scenes[0].intrinsics)
            for p in range(n_pairs):
                _load(solo, scenes[p % len(scenes)], pair=p)
            Rs, ts = solo.align_batch(iters, np.tile(np.eye(3), (n_pairs, 1, 1)), np.zeros((n_pairs, 3)))
            assert np.abs(R - Rs).max() <= 1e-12 and np.abs(t - ts).max() <= 1e-12


def test_team_size_that_cannot_be_resident_is_refused():
    from rgbd_odometry_amd import DvoContext, DvoError, SynthScene
    sc = SynthScene(160, 120, 2, 1)
    with DvoContext(64, team_size=16) as ctx:
        ctx.set_intrinsics(*sc.intrinsics)
        for p in range(64):
            _load(ctx, sc, pair=p)
        with pytest.raises(DvoError):
            ctx.align_batch([2, 2], np.tile(np.eye(3), (64, 1, 1)), np.zeros((64, 3)))


def test_large_batch_launch_order(oracle):
    """more pairs than compute units: the engine starts the pairs with the most point-iterations first (longest-processing-time
    order, csrc/dvo_capi.cpp) -- a pure permutation of which workgroup aligns which pair: every pair still gets the result of
    ITS scene, with and without the compact now form"""
    from rgbd_odometry_amd import DvoContext, SynthScene
    n = 300
    scenes = [SynthScene(160, 120, 2, 40 + i) for i in range(5)]             # different point counts
    iters = [6, 6]
    refs = [oracle.align_pyramid(iters, oracle_lib.scene_levels(s, oracle), s.intrinsics, np.eye(3), np.zeros(3)) for s in scenes]
    assert len({len(r["levels"][0]["final_eps"]) for r in refs}) > 1
    with DvoContext(n) as ctx:
        ctx.set_intrinsics(*scenes[0].intrinsics)
        for p, sc in enumerate(scenes):
            _load(ctx, sc, pair=p)
        ctx.replicate_pairs(5)
        for prepared in (False, True):
            if prepared:
                ctx.now_prepare()
            R, t = ctx.align_batch(iters, np.tile(np.eye(3), (n, 1, 1)), np.zeros((n, 3)))
            assert ctx.last_launch_shape()[1] == 1                            # no team: one workgroup per pair
            for p in range(n):
                ref = refs[p % 5]
                for l, rep in ref["levels"].items():
                    e, b, ratio = ctx.level_report(p, l, iters[l])
                    assert np.array_equal(e, rep["energy"]) and b == rep["best_idx"] and ratio == rep["visible_ratio"], (prepared, p, l)
                assert rot_angle(ref["R"], R[p]) <= ROT_TOL and np.linalg.norm(ref["t"] - t[p]) <= TRANS_TOL, (prepared, p)
