"""GPU parity of the compact ("P4", 4 bytes per pixel) form of resident now levels (rgbd_odometry_amd/csrc/dvo_palette.h,
dvo_palette.hip, TEX_P4 in dvo_fused.hip) against the CPU oracle.

The compact form replaces the reference's three float images of a now level (distance transform SolveDVO.cpp:1768-1795, its
imageGradient :1063-1098, the weight :1047-1053) by palette ranks; the builder verifies per pixel that it decodes to the same
four floats, so every result must stay bit-identical to the oracle: energies / best index / visible ratio / final outputs
bit-equal, pose within 1e-5 rad / 1e-4 m -- the bars of test_gpu_parity.py.
"""
import numpy as np
import pytest

import oracle_lib
from oracle_lib import rot_angle

pytestmark = pytest.mark.gpu

ROT_TOL, TRANS_TOL = 1e-5, 1e-4


def _same(a, b):
    return np.array_equal(np.asarray(a), np.asarray(b), equal_nan=True)


def _load(ctx, sc, pair=0):
    for l, L in enumerate(sc.levels):
        ctx.set_ref_level_from_images(l, L.ref_edge, L.ref_depth, L.rows, L.cols, pair=pair)
        ctx.set_now_level(l, L.now_dt, L.now_gx, L.now_gy, L.rows, L.cols, pair=pair)


def _check(ctx, ref, iters, pair=0, R0=None, t0=None):
    from rgbd_odometry_amd.capi import DVO_FLAG_FINAL_OUTPUTS
    R0 = np.eye(3) if R0 is None else R0
    t0 = np.zeros(3) if t0 is None else t0
    R, t = ctx.align_batch(iters, R0[None], t0[None], first_pair=pair, n_pairs=1, flags=DVO_FLAG_FINAL_OUTPUTS)
    for l, rep in ref["levels"].items():
        e, b, ratio = ctx.level_report(pair, l, iters[l])
        assert np.array_equal(e, rep["energy"]), (l, e, rep["energy"])
        assert b == rep["best_idx"] and ratio == rep["visible_ratio"], l
    assert rot_angle(ref["R"], R[0]) <= ROT_TOL and np.linalg.norm(ref["t"] - t[0]) <= TRANS_TOL
    last = ref["levels"][ref["last_level"]]
    feps, frep = ctx.final_outputs(pair, len(last["final_eps"]))
    assert _same(feps, last["final_eps"]) and _same(frep, last["final_reproj"])
    return R[0], t[0]


@pytest.mark.parametrize("kw", [
    dict(),                                          # auto: a single pair -> a team of 8 workgroups, every level compact
    dict(team_size=1),                               # one 512-thread workgroup (the compact form comes before LDS-staged texels)
    dict(engine_variant=2, team_size=1),             # no LDS staging: all four levels through the compact form
    dict(block_threads=256), dict(block_threads=1024),
    dict(lds_point_bytes=16 * 1024, team_size=1),    # palette + 1 k points resident, the rest streamed (both passes)
    dict(engine_variant=3, team_size=1),             # every wave redone by the literal-division fallback, reading the compact form
    dict(engine_variant=3),                          # the same in a team
    dict(engine_variant=5, team_size=1),             # every energy from the exact sweep of the residuals (round 6: the energy without an order)
    dict(engine_variant=5),                          # the same in a team: the limbs go through the team's exchange
    dict(engine_variant=5, block_threads=256), dict(engine_variant=5, lds_point_bytes=16 * 1024, team_size=1),
])
def test_compact_now_640x480(oracle, kw):
    """C2 (640x480, 4 levels, 10 iterations): prepared now levels through every launch shape of the packed kernel"""
    from rgbd_odometry_amd import DvoContext, SynthScene
    sc = SynthScene(640, 480, 4, 3)
    lv = oracle_lib.scene_levels(sc, oracle)
    iters = [10, 10, 10, 10]
    ref = oracle.align_pyramid(iters, lv, sc.intrinsics, np.eye(3), np.zeros(3))
    with DvoContext(1, **kw) as ctx:
        ctx.set_intrinsics(*sc.intrinsics)
        _load(ctx, sc)
        ctx.now_prepare()
        sizes = [ctx.now_compact_info(0, l) for l in range(4)]
        assert all(0 < s <= 4096 for s in sizes), sizes
        assert sizes[0] > sizes[1] > sizes[2] > sizes[3]          # coarser levels have fewer distinct distances
        R1, t1 = _check(ctx, ref, iters)
        modes = [ctx.level_texel_mode(0, l) for l in range(4)]
        assert modes == [2, 2, 2, 2], modes                       # a level that has a compact form is read through it
        # engine_variant = 3 really runs the literal-division code (it used to be a silent no-op), nothing else does
        assert [ctx.level_exact_fallback(0, l) for l in range(4)] == [kw.get("engine_variant", 0) == 3] * 4
        # engine_variant = 5: every iteration's energy came from the exact limbs; otherwise the certificate of the fast sum holds
        # (it fails once in ~20 000 iterations of such levels: not on this scene)
        sweeps = [ctx.level_energy_sweeps(0, l) for l in range(4)]
        assert sweeps == ([10] * 4 if kw.get("engine_variant", 0) == 5 else [0] * 4), sweeps
    if kw.get("engine_variant", 0) == 0:
        # the same alignment with the compact form switched off: same kernel otherwise -> the very same bits
        with DvoContext(1, **{**kw, "engine_variant": 4}) as ctx2:
            ctx2.set_intrinsics(*sc.intrinsics)
            _load(ctx2, sc)
            ctx2.now_prepare()
            R2, t2 = _check(ctx2, ref, iters)
            assert 2 not in [ctx2.level_texel_mode(0, l) for l in range(4)]
            assert np.array_equal(R1, R2) and np.array_equal(t1, t2)


@pytest.mark.parametrize("epoch0", [None, "4294967000"])
@pytest.mark.parametrize("variant", [0, 5])
def test_team_launches_in_a_row_keep_their_exchange_tags_apart(oracle, epoch0, variant, monkeypatch):
    """round 6: the team records are no longer zeroed between launches -- a record is recognised by its tag, the count of exchanges,
    which runs on from launch to launch.  A launch must therefore reserve as many tags as it can use (with engine_variant 5 every
    iteration makes a second exchange for the limbs of its exact energy), and the 32-bit count must survive its wrap
    (DVO_TEAM_EPOCH0 starts it just below): six alignments in a row, teams of different sizes, every one equal to the oracle's"""
    from rgbd_odometry_amd import DvoContext, SynthScene
    if epoch0:
        monkeypatch.setenv("DVO_TEAM_EPOCH0", epoch0)
    sc = SynthScene(640, 480, 4, 3)
    lv = oracle_lib.scene_levels(sc, oracle)
    iters = [10, 10, 10, 10]
    ref = oracle.align_pyramid(iters, lv, sc.intrinsics, np.eye(3), np.zeros(3))
    ref2 = oracle.align_pyramid([3, 0, 4, 2], lv, sc.intrinsics, np.eye(3), np.zeros(3))
    with DvoContext(1, engine_variant=variant) as ctx:
        ctx.set_intrinsics(*sc.intrinsics)
        _load(ctx, sc)
        ctx.now_prepare()
        for rep in range(6):
            if rep % 3 == 2:
                _check(ctx, ref2, [3, 0, 4, 2])
            else:
                _check(ctx, ref, iters)
            assert ctx.last_launch_shape()[1] > 1                  # a team launch


@pytest.mark.parametrize("kw", [dict(), dict(team_size=1), dict(block_threads=256), dict(engine_variant=3, team_size=1),
                                dict(lds_point_bytes=16 * 1024, team_size=1), dict(engine_variant=1)])
def test_native_compact_now_levels_from_edges_640x480(oracle, kw):
    """round 3: now levels produced by the engine's own distance transform (dvo_set_now_level_from_edges) exist in the compact
    form ONLY -- ranks straight from the integer squared distances, no dvo_now_prepare, no 16-byte texels written.  C2 through
    the packed kernel's launch shapes; the paths that read 16-byte texels (one-point-per-lane kernel, a launch whose LDS cannot
    hold every palette) get them decoded from the compact form on demand.  Same bits as the oracle and as a context with the
    compact form switched off."""
    from rgbd_odometry_amd import DvoContext, SynthScene
    sc = SynthScene(640, 480, 4, 3)
    lv = oracle_lib.scene_levels(sc, oracle)
    iters = [10, 10, 10, 10]
    ref = oracle.align_pyramid(iters, lv, sc.intrinsics, np.eye(3), np.zeros(3))

    def load(ctx):
        ctx.set_intrinsics(*sc.intrinsics)
        for l, L in enumerate(sc.levels):
            ctx.set_ref_level_from_images(l, L.ref_edge, L.ref_depth, L.rows, L.cols)
            ctx.set_now_level_from_edges(l, (np.asarray(L.now_edge) != 0).astype(np.uint8) * 255, L.rows, L.cols)
    with DvoContext(1, **kw) as ctx:
        load(ctx)
        sizes = [ctx.now_compact_info(0, l) for l in range(4)]
        assert all(0 < s < 8192 for s in sizes) and sizes[0] > sizes[1] > sizes[2] > sizes[3], sizes
        for l, L in enumerate(sc.levels):
            # round 6 (dvo_edt_band.h): the palette lists every squared distance an image CAN have up to this image's maximum -- the
            # sums of two squares -- plus the two fixed entries (zero sentinel, NaN); rounds 3-5 listed the distances present
            import os
            if os.environ.get("DVO_EDT_FUSED") == "0":         # the three-pass stage: one palette entry per distinct squared distance
                assert sizes[l] == len(np.unique(L.now_dt)), (l, sizes[l])
                continue
            d2max = int(round(float(np.max(np.asarray(L.now_dt, np.float64) / np.min(L.now_dt[L.now_dt > 0]))) ** 2))
            sos = {a * a + b * b for a in range(200) for b in range(200)}
            assert sizes[l] == 2 + sum(1 for v in sos if v <= d2max), (l, sizes[l], d2max)
            assert sizes[l] >= len(np.unique(L.now_dt))
        R1, t1 = _check(ctx, ref, iters)
        if kw.get("engine_variant", 0) != 1:
            assert [ctx.level_texel_mode(0, l) for l in range(4)] == [2, 2, 2, 2], kw
            # 4-byte reference points: only for lists beyond what the LDS holds as 8-byte points -- with 16 KB of LDS that is
            # the finest levels here (the same bits come out: the builder validated every point against the 8-byte list)
            import os
            p4 = [ctx.level_points4(0, l) for l in range(4)]
            if os.environ.get("DVO_POINTS4") == "off":
                assert p4 == [False] * 4, (kw, p4)
            elif kw.get("block_threads") == 256:       # half a CU's LDS: level 0 (15.9 k points) is beyond its 8.8 k 8-byte points
                assert p4 == [True, False, False, False], (kw, p4)
            elif "lds_point_bytes" not in kw:          # a team, or one 512-thread workgroup with the whole LDS: every list fits
                assert p4 == [False] * 4, (kw, p4)
            else:
                assert p4[0] and p4[1] and not p4[3], (kw, p4)
        # the planar images decoded from the compact form are the scene generator's (== the oracle's), bit for bit
        for l, L in enumerate(sc.levels):
            dt, gx, gy = ctx.get_now_level(l)
            assert _same(dt, L.now_dt) and _same(gx, L.now_gx) and _same(gy, L.now_gy), l
        R1b, t1b = _check(ctx, ref, iters)                      # and the compact form is still what the packed kernel reads
        assert np.array_equal(R1, R1b) and np.array_equal(t1, t1b)
    with DvoContext(1, **{**kw, "engine_variant": 4}) as ctx2:  # compact form off: the same stage writes 16-byte texels
        load(ctx2)
        assert [ctx2.now_compact_info(0, l) for l in range(4)] == [0, 0, 0, 0]
        R2, t2 = _check(ctx2, ref, iters)
        assert 2 not in [ctx2.level_texel_mode(0, l) for l in range(4)]
        if kw.get("engine_variant", 0) in (0, 3):
            assert np.array_equal(R1, R2) and np.array_equal(t1, t2)


def test_config3_1920x1080_native_compact_and_4_byte_points(oracle):
    """BASELINE configs[2] at its full schedule (1920x1080, 5 levels, 10 iterations per level) through the throughput path:
    natively produced compact now levels, one workgroup per pair, and -- the finest lists being many times what the LDS holds
    (130 k / 67 k points against 19 k) -- reference points streamed in their 4-byte form.  Oracle parity as everywhere;
    the same alignment with the 4-byte form's levels forced back to 8-byte points must give the very same bits."""
    import os
    from rgbd_odometry_amd import DvoContext, SynthScene
    sc = SynthScene(1920, 1080, 5, 0)
    lv = oracle_lib.scene_levels(sc, oracle)
    iters = [10, 10, 10, 10, 10]
    ref = oracle.align_pyramid(iters, lv, sc.intrinsics, np.eye(3), np.zeros(3))
    with DvoContext(2, team_size=1) as ctx:
        ctx.set_intrinsics(*sc.intrinsics)
        for l, L in enumerate(sc.levels):
            ctx.set_ref_level_from_images(l, L.ref_edge, L.ref_depth, L.rows, L.cols)
            ctx.set_now_level_from_edges(l, (np.asarray(L.now_edge) != 0).astype(np.uint8) * 255, L.rows, L.cols)
        ctx.replicate_pairs(1)
        for pair in (0, 1):
            R1, t1 = _check(ctx, ref, iters, pair=pair)
            assert [ctx.level_texel_mode(pair, l) for l in range(5)] == [2] * 5
            if not os.environ.get("DVO_POINTS4"):
                p4 = [ctx.level_points4(pair, l) for l in range(5)]     # lists beyond what the LDS holds as 8-byte points (round 5: from 1x on)
                assert p4[0] and p4[1] and not p4[3] and not p4[4], p4


def test_native_compact_replicated_batch_and_overwrite(oracle):
    """native compact now levels in a batch: replicated slots carry their own copy of the compact form; a slot overwritten by
    caller-supplied float images falls back to 16-byte texels, one overwritten from edges again gets a fresh compact form"""
    from rgbd_odometry_amd import DvoContext, SynthScene
    n = 10
    scenes = [SynthScene(320, 240, 3, 4000 + i) for i in range(3)]
    iters = [6, 6, 6]
    with DvoContext(n) as ctx:
        ctx.set_intrinsics(*scenes[0].intrinsics)
        for p, sc in enumerate(scenes):
            for l, L in enumerate(sc.levels):
                ctx.set_ref_level_from_images(l, L.ref_edge, L.ref_depth, L.rows, L.cols, pair=p)
                ctx.set_now_level_from_edges(l, (np.asarray(L.now_edge) != 0).astype(np.uint8) * 255, L.rows, L.cols, pair=p)
        ctx.replicate_pairs(3)
        assert all(ctx.now_compact_info(p, l) == ctx.now_compact_info(p % 3, l) > 0 for p in range(n) for l in range(3))
        # slot 4 (a copy of scene 1) gets scene 2's now frame as float images; slot 5 (scene 2) gets scene 0's from edges
        for l, L in enumerate(scenes[2].levels):
            ctx.set_now_level(l, L.now_dt, L.now_gx, L.now_gy, L.rows, L.cols, pair=4)
        for l, L in enumerate(scenes[0].levels):
            ctx.set_now_level_from_edges(l, (np.asarray(L.now_edge) != 0).astype(np.uint8) * 255, L.rows, L.cols, pair=5)
        assert [ctx.now_compact_info(4, l) for l in range(3)] == [0, 0, 0]
        R, t = ctx.align_batch(iters, np.tile(np.eye(3), (n, 1, 1)), np.zeros((n, 3)))
        for p in range(n):
            src, now = p % 3, {4: 2, 5: 0}.get(p, p % 3)
            lv = oracle_lib.scene_levels(scenes[src], oracle)
            for l, L in enumerate(scenes[now].levels):
                lv[l].update(dt=L.now_dt, gx=L.now_gx, gy=L.now_gy)
            ref = oracle.align_pyramid(iters, lv, scenes[src].intrinsics, np.eye(3), np.zeros(3))
            for l, rep in ref["levels"].items():
                e, b, ratio = ctx.level_report(p, l, iters[l])
                assert np.array_equal(e, rep["energy"]) and b == rep["best_idx"] and ratio == rep["visible_ratio"], (p, l)
            assert rot_angle(ref["R"], R[p]) <= ROT_TOL and np.linalg.norm(ref["t"] - t[p]) <= TRANS_TOL
            assert ctx.level_texel_mode(p, 0) == (0 if p == 4 else 2), p


def test_native_builder_writes_a_partial_form_for_images_it_cannot_hold(oracle):
    """an image the compact form cannot hold completely -- here: pixels 512 or more pixels from every edge (rounds 3-4 refused it,
    -7) -- gets a PARTIAL compact form (round 5, dvo_palette.h): the pixels it cannot express carry the rank of a NaN palette entry,
    the same launch writes the image's 16-byte texels, and a wave whose look-up meets the NaN redoes its share of the iteration on
    those (this image has two edge pixels: nearly every look-up is far from them).  Results are the oracle's all the same."""
    from rgbd_odometry_amd import DvoContext
    rows, cols = 40, 700
    edge = np.zeros(rows * cols, np.uint8)
    edge[5 + 3 * rows] = 255                                    # one edge pixel near the left border: distances up to ~696
    edge[7 + 20 * rows] = 255
    dt, gx, gy = oracle.now_level_from_edges(edge, rows, cols)
    rng = np.random.default_rng(3)
    ref_edge = (rng.random(rows * cols) < 0.3).astype(np.int32) * 255
    depth = rng.uniform(400, 3000, rows * cols).astype(np.float32)
    K = (600.0, 600.0, 350.0, 20.0)
    with DvoContext(1, team_size=1) as ctx:
        ctx.set_intrinsics(*K)
        xyz, _ = ctx.set_ref_level_from_images(0, ref_edge, depth, rows, cols)
        ctx.set_now_level_from_edges(0, edge, rows, cols)
        assert ctx.now_compact_info(0, 0) > 0 and ctx.now_compact_partial(0, 0)
        d2, g2, h2 = ctx.get_now_level(0)
        assert _same(d2, dt) and _same(g2, gx) and _same(h2, gy)
        ref = oracle.run_iterations(0, 6, xyz, dt, gx, gy, rows, cols, K, np.eye(3), np.zeros(3))
        got = ctx.run_iterations(0, 6, np.eye(3), np.zeros(3))
        assert ctx.level_texel_mode(0, 0) == 2 and ctx.level_exact_fallback(0, 0)      # the compact form, and waves that met the NaN rank
        assert _same(ref["energy"], got["energy"])
        assert _same(ref["final_eps"], got["final_eps"]) and _same(ref["final_reproj"], got["final_reproj"])
        # an image with no edge pixel at all is refused by the entry point (the distance transform is undefined)
        with pytest.raises(Exception):
            ctx.set_now_level_from_edges(0, np.zeros(rows * cols, np.uint8), rows, cols)


def test_replicating_a_source_with_real_texels_carries_them(oracle):
    """a source whose image the compact form cannot hold completely (a partial form since round 5; refused before) has 16-byte
    texels as its complete form; replicating it must copy them (and, on a sparse texel slab, map memory behind the destinations') --
    beside a second source that has the compact form only.  Both kinds of destination then give their source's oracle results.  (tests/test_gpu_capacity.py runs this
    file again with DVO_TEX_SLAB=sparse.)"""
    from rgbd_odometry_amd import DvoContext
    rows, cols = 40, 700
    far = np.zeros(rows * cols, np.uint8)                       # pixels 512 or more from every edge: a partial form + real texels (round 5)
    far[5 + 3 * rows] = 255
    far[7 + 20 * rows] = 255
    rng = np.random.default_rng(11)
    dense = (rng.random(rows * cols) < 0.05).astype(np.uint8) * 255      # holds a compact form
    ref_edge = (rng.random(rows * cols) < 0.3).astype(np.int32) * 255
    depth = rng.uniform(400, 3000, rows * cols).astype(np.float32)
    K = (600.0, 600.0, 350.0, 20.0)
    B = 6
    with DvoContext(B, team_size=1) as ctx:
        ctx.set_intrinsics(*K)
        xyz = None
        for p, e in enumerate((far, dense)):
            xyz, _ = ctx.set_ref_level_from_images(0, ref_edge, depth, rows, cols, pair=p)
            ctx.set_now_level_from_edges(0, e, rows, cols, pair=p)
        assert ctx.now_compact_partial(0, 0) and ctx.now_compact_info(1, 0) > 0 and not ctx.now_compact_partial(1, 0)
        ctx.replicate_pairs(2)
        assert [ctx.now_compact_info(p, 0) for p in range(B)] == [ctx.now_compact_info(p % 2, 0) for p in range(B)]
        refs = []
        for e in (far, dense):
            dt, gx, gy = oracle.now_level_from_edges(e, rows, cols)
            refs.append((dt, gx, gy, oracle.run_iterations(0, 6, xyz, dt, gx, gy, rows, cols, K, np.eye(3), np.zeros(3))))
        for p in range(B):
            dt, gx, gy, ref = refs[p % 2]
            d2, g2, h2 = ctx.get_now_level(0, pair=p)
            assert _same(d2, dt) and _same(g2, gx) and _same(h2, gy), p
        ctx.enqueue([6], flags=3)                               # identity start + final outputs: all pairs in one launch
        ctx.synchronize()
        for p in range(B):
            e, b, ratio = ctx.level_report(p, 0, 6)
            ref = refs[p % 2][3]
            assert _same(e, ref["energy"]) and b == ref["best_idx"], p
            assert (ctx.level_texel_mode(p, 0) & 3) == 2, p                 # the compact form: complete (odd pairs) or partial (even pairs)
            assert ctx.now_compact_partial(p, 0) == (p % 2 == 0), p


def test_compact_now_is_built_for_a_level_that_keeps_being_aligned(oracle):
    """policy (DVO_COMPACT_NOW_AFTER = 16 in dvo_amd.h): a now level keeps the 16-byte form for its first 16 alignments, the
    17th builds the compact form; writing the level again makes it stale (and the results follow the new image)"""
    import os
    from rgbd_odometry_amd import DvoContext, SynthScene
    if os.environ.get("DVO_COMPACT_NOW"):
        pytest.skip("policy overridden by DVO_COMPACT_NOW")
    AFTER = 16
    sc, sc2 = SynthScene(160, 120, 2, 5), SynthScene(160, 120, 2, 6)
    iters = [4, 4]
    ref = oracle.align_pyramid(iters, oracle_lib.scene_levels(sc, oracle), sc.intrinsics, np.eye(3), np.zeros(3))
    with DvoContext(1, team_size=1, engine_variant=2) as ctx:
        ctx.set_intrinsics(*sc.intrinsics)
        _load(ctx, sc)
        for k in range(AFTER):
            _check(ctx, ref, iters)
            assert [ctx.level_texel_mode(0, l) for l in range(2)] == [0, 0], k
            assert [ctx.now_compact_info(0, l) for l in range(2)] == [0, 0], k
        _check(ctx, ref, iters)
        assert [ctx.level_texel_mode(0, l) for l in range(2)] == [2, 2]
        assert all(ctx.now_compact_info(0, l) > 0 for l in range(2))
        # a new now frame in the same slot (same reference): compact form stale -> 16-byte form, results of the new image
        for l, L in enumerate(sc2.levels):
            ctx.set_now_level(l, L.now_dt, L.now_gx, L.now_gy, L.rows, L.cols)
        lv_mixed = oracle_lib.scene_levels(sc, oracle)
        for l, L in enumerate(sc2.levels):
            lv_mixed[l].update(dt=L.now_dt, gx=L.now_gx, gy=L.now_gy)
        ref2 = oracle.align_pyramid(iters, lv_mixed, sc.intrinsics, np.eye(3), np.zeros(3))
        assert [ctx.now_compact_info(0, l) for l in range(2)] == [0, 0]
        _check(ctx, ref2, iters)
        assert [ctx.level_texel_mode(0, l) for l in range(2)] == [0, 0]
        ctx.now_prepare()                                  # on request: now
        _check(ctx, ref2, iters)
        assert [ctx.level_texel_mode(0, l) for l in range(2)] == [2, 2]


@pytest.mark.parametrize("rows,cols", [(7, 9), (12, 8), (13, 5), (61, 83), (6, 4), (240, 322), (2, 2), (5, 2)])
def test_compact_now_odd_sizes_from_edges(oracle, rows, cols):
    """now levels built by the engine from an edge map (exact distance transform + gradient), sizes that are not multiples
    of the 6 x 4 interior of a line, single-tile images, reflect-101 borders on every side"""
    from rgbd_odometry_amd import DvoContext
    rng = np.random.default_rng(rows * 1000 + cols)
    edge = (rng.random((cols, rows)) < 0.15).astype(np.uint8).reshape(-1) * 255
    edge[int(rng.integers(0, rows * cols))] = 255
    dt, gx, gy = oracle.now_level_from_edges(edge, rows, cols)
    ref_edge = (rng.random(rows * cols) < 0.5).astype(np.int32) * 255
    ref_edge[0] = 255
    depth = rng.uniform(400, 3000, rows * cols).astype(np.float32)
    K = (float(np.float32(0.9 * cols)), float(np.float32(0.9 * cols)), float(np.float32(cols / 2)), float(np.float32(rows / 2)))
    with DvoContext(1, team_size=1, engine_variant=2) as ctx:
        ctx.set_intrinsics(*K)
        xyz, _ = ctx.set_ref_level_from_images(0, ref_edge, depth, rows, cols)
        ctx.set_now_level_from_edges(0, edge, rows, cols)     # native: the compact form is what this writes (no dvo_now_prepare)
        n = ctx.now_compact_info(0, 0)
        assert n > 0, n
        import os
        if os.environ.get("DVO_EDT_FUSED") == "0":
            assert n == len(np.unique(dt)), (n, len(np.unique(dt)))       # the three-pass stage: the distances present
        else:
            assert n >= len(np.unique(dt)) + 2, (n, len(np.unique(dt)))   # round 6: every sum of two squares up to the maximum + the two fixed entries
        for scale in (0.0, 0.01, 0.2):
            R0, t0 = oracle.se3_exp(rng.standard_normal(6) * scale)
            ref = oracle.run_iterations(0, 8, xyz, dt, gx, gy, rows, cols, K, R0, t0)
            got = ctx.run_iterations(0, 8, R0, t0)
            assert ctx.level_texel_mode(0, 0) == 2
            assert _same(ref["energy"], got["energy"]), (scale, ref["energy"], got["energy"])
            assert ref["best_idx"] == got["best_idx"] and ref["visible_ratio"] == got["visible_ratio"]
            assert _same(ref["final_eps"], got["final_eps"]) and _same(ref["final_reproj"], got["final_reproj"])


def test_compact_now_refused_keeps_the_16_byte_form(oracle):
    """images the compact form cannot represent: the builder says why, the alignment reads the 16-byte texels, results are
    those of the oracle all the same"""
    from rgbd_odometry_amd import DvoContext, SynthScene
    sc = SynthScene(160, 120, 1, 11)
    L = sc.levels[0]
    rows, cols, n = L.rows, L.cols, L.rows * L.cols
    rng = np.random.default_rng(5)
    dt, gx, gy = (np.asarray(a, np.float32).copy() for a in (L.now_dt, L.now_gx, L.now_gy))
    cases = {}
    g = gx.copy(); g[n // 2] = np.nextafter(g[n // 2], np.float32(1e9)); cases[-4] = (dt, g, gy)                 # one gradient off by 1 ulp
    g = gy.copy(); g[7] = -g[7] if g[7] != 0 else np.float32(1.0); cases["-4y"] = (dt, gx, g)                    # wrong sign in gy
    cases[-2] = (rng.uniform(0, 255, n).astype(np.float32), gx, gy)                                             # > 4096 distinct values
    d = dt.copy(); d[3] = -1.0; cases[-1] = (d, gx, gy)                                                         # negative "distance"
    d = dt.copy(); d[5] = np.nan; cases["-1n"] = (d, gx, gy)
    with DvoContext(1, team_size=1, engine_variant=2) as ctx:
        ctx.set_intrinsics(*sc.intrinsics)
        xyz, _ = ctx.set_ref_level_from_images(0, L.ref_edge, L.ref_depth, rows, cols)
        for why, (a, b, c_) in cases.items():
            ctx.set_now_level(0, a, b, c_, rows, cols)
            ctx.now_prepare()
            code = ctx.now_compact_info(0, 0)
            assert code == int(str(why)[:2]), (why, code)
            ref = oracle.run_iterations(0, 6, xyz, a, b, c_, rows, cols, sc.intrinsics, np.eye(3), np.zeros(3))
            got = ctx.run_iterations(0, 6, np.eye(3), np.zeros(3))
            assert ctx.level_texel_mode(0, 0) == 0
            assert _same(ref["energy"], got["energy"]), (why, ref["energy"], got["energy"])
            assert _same(ref["final_eps"], got["final_eps"]) and _same(ref["final_reproj"], got["final_reproj"])
        # and the derived image again: accepted
        ctx.set_now_level(0, dt, gx, gy, rows, cols)
        ctx.now_prepare()
        assert ctx.now_compact_info(0, 0) > 0


def test_compact_now_mixed_batch(oracle):
    """one launch whose pairs are partly compact (derived now levels) and partly not (arbitrary gradients): per pair and level
    the kernel takes what exists; replicated pairs get their own compact form"""
    from rgbd_odometry_amd import DvoContext, SynthScene
    n = 12
    scenes = [SynthScene(320, 240, 3, 3000 + i) for i in range(4)]
    iters = [8, 8, 8]
    rng = np.random.default_rng(1)
    with DvoContext(n, engine_variant=2) as ctx:
        ctx.set_intrinsics(*scenes[0].intrinsics)
        lvs = []
        for p, sc in enumerate(scenes):
            lv = oracle_lib.scene_levels(sc, oracle)
            for l, L in enumerate(sc.levels):
                ctx.set_ref_level_from_images(l, L.ref_edge, L.ref_depth, L.rows, L.cols, pair=p)
                gx = np.asarray(L.now_gx, np.float32).copy()
                if p % 2:                                     # odd pairs: gradient perturbed -> no compact form
                    gx += rng.normal(0, 0.01, gx.shape).astype(np.float32)
                    lv[l]["gx"] = gx
                ctx.set_now_level(l, L.now_dt, gx, L.now_gy, L.rows, L.cols, pair=p)
            lvs.append(lv)
        ctx.replicate_pairs(4)
        ctx.now_prepare()
        for p in range(n):
            assert (ctx.now_compact_info(p, 0) > 0) == (p % 2 == 0), (p, ctx.now_compact_info(p, 0))
        R, t = ctx.align_batch(iters, np.tile(np.eye(3), (n, 1, 1)), np.zeros((n, 3)))
        for p in range(n):
            ref = oracle.align_pyramid(iters, lvs[p % 4], scenes[p % 4].intrinsics, np.eye(3), np.zeros(3))
            for l, rep in ref["levels"].items():
                e, b, ratio = ctx.level_report(p, l, iters[l])
                assert np.array_equal(e, rep["energy"]) and b == rep["best_idx"] and ratio == rep["visible_ratio"], (p, l)
            assert rot_angle(ref["R"], R[p]) <= ROT_TOL and np.linalg.norm(ref["t"] - t[p]) <= TRANS_TOL
            assert ctx.level_texel_mode(p, 0) == (2 if p % 2 == 0 else 0)


def test_compact_now_builder_survives_images_of_all_distinct_values(oracle):
    """a 640x480 level whose every pixel has another value: every chunk of the collect pass alone exceeds the palette limit --
    the builder must say "too many" (-2), not spin on a full hash set"""
    from rgbd_odometry_amd import DvoContext
    rows, cols = 480, 640
    rng = np.random.default_rng(11)
    dt = rng.permutation(rows * cols).astype(np.float32) * np.float32(1.0 / 4096.0)
    g = np.zeros(rows * cols, np.float32)
    edge = (rng.random(rows * cols) < 0.02).astype(np.int32) * 255
    depth = rng.uniform(500, 3000, rows * cols).astype(np.float32)
    K = (525.0, 525.0, 319.5, 239.5)
    for n_pairs in (1, 3):                     # different chunkings of the image
        with DvoContext(n_pairs, team_size=1) as ctx:
            ctx.set_intrinsics(*K)
            for p in range(n_pairs):
                xyz, _ = ctx.set_ref_level_from_images(0, edge, depth, rows, cols, pair=p)
                ctx.set_now_level(0, dt, g, g, rows, cols, pair=p)
            ctx.now_prepare()
            assert [ctx.now_compact_info(p, 0) for p in range(n_pairs)] == [-2] * n_pairs
            ref = oracle.run_iterations(0, 3, xyz, dt, g, g, rows, cols, K, np.eye(3), np.zeros(3))
            got = ctx.run_iterations(0, 3, np.eye(3), np.zeros(3))
            assert ctx.level_texel_mode(0, 0) == 0
            assert _same(ref["energy"], got["energy"])


def _image_gradient(dt, rows, cols):
    """imageGradient (SolveDVO.cpp:1063-1098) of a column-major float32 image: 0.5 * central differences, reflect-101 border"""
    a = np.asarray(dt, np.float32).reshape(cols, rows).T                 # (rows, cols)
    px = np.pad(a, ((0, 0), (1, 1)), mode="reflect")
    py = np.pad(a, ((1, 1), (0, 0)), mode="reflect")
    gx = (np.float32(0.5) * (px[:, 2:] - px[:, :-2])).astype(np.float32)
    gy = (np.float32(0.5) * (py[2:, :] - py[:-2, :])).astype(np.float32)
    return np.ascontiguousarray(gx.T).ravel(), np.ascontiguousarray(gy.T).ravel()


@pytest.mark.parametrize("kw", [dict(), dict(team_size=1), dict(block_threads=256)], ids=["auto", "no-teams", "256"])
def test_float_images_go_straight_to_the_compact_form(oracle, kw):
    """round 3 (VERDICT r2 next #1b): dvo_set_now_level with the reference's three float images -- a normalised exact distance
    transform and its imageGradient -- installs the compact form at once (d2 recovered from DT, verified bit for bit; the
    gradients compared with what the kernel decodes); the very first alignment reads 4-byte words, results are the oracle's"""
    from rgbd_odometry_amd import DvoContext, SynthScene
    sc = SynthScene(640, 480, 4, 31)
    iters = [10, 10, 10, 10]
    ref = oracle.align_pyramid(iters, oracle_lib.scene_levels(sc, oracle), sc.intrinsics, np.eye(3), np.zeros(3))
    with DvoContext(1, **kw) as ctx:
        ctx.set_intrinsics(*sc.intrinsics)
        ctx.set_direct_compact(True)
        _load(ctx, sc)
        assert all(ctx.now_compact_info(0, l) > 0 for l in range(4))
        for l, L in enumerate(sc.levels):
            for got, want in zip(ctx.get_now_level(l), (L.now_dt, L.now_gx, L.now_gy)):
                assert np.array_equal(got, np.asarray(want, np.float32).ravel())
        _check(ctx, ref, iters)
        assert [ctx.level_texel_mode(0, l) for l in range(4)] == [2, 2, 2, 2]
        # the native builder decodes to the same image (round 6: its palette lists every sum of two squares up to the maximum and
        # two fixed entries, the direct build the distances present)
        sizes = [ctx.now_compact_info(0, l) for l in range(4)]
        for l, L in enumerate(sc.levels):
            ctx.set_now_level_from_edges(l, (np.asarray(L.now_edge) != 0).astype(np.uint8) * 255, L.rows, L.cols)
        assert all(a >= b for a, b in zip([ctx.now_compact_info(0, l) for l in range(4)], sizes))
        for l, L in enumerate(sc.levels):
            for got, want in zip(ctx.get_now_level(l), (L.now_dt, L.now_gx, L.now_gy)):
                assert np.array_equal(got, np.asarray(want, np.float32).ravel())
        _check(ctx, ref, iters)


def test_float_images_direct_build_refusals_and_other_scales(oracle):
    """what the direct build accepts and refuses: any positive scale of an exact distance transform goes in (the unit is read
    off the image); a gradient that is not imageGradient(DT) is refused with -4, a DT that is no exact transform with -1; refused
    pairs keep their 16-byte texels; every result is the oracle's on the images as given"""
    from rgbd_odometry_amd import DvoContext, SynthScene
    sc = SynthScene(320, 240, 2, 77)
    iters = [8, 8]

    def run(ctx, images):
        lv = oracle_lib.scene_levels(sc, oracle)
        for l, L in enumerate(sc.levels):
            ctx.set_ref_level_from_images(l, L.ref_edge, L.ref_depth, L.rows, L.cols)
            dt, gx, gy = images(l, L)
            ctx.set_now_level(l, dt, gx, gy, L.rows, L.cols)
            lv[l].update(dt=dt, gx=gx, gy=gy)
        ref = oracle.align_pyramid(iters, lv, sc.intrinsics, np.eye(3), np.zeros(3))
        info = [ctx.now_compact_info(0, l) for l in range(2)]
        _check(ctx, ref, iters)
        return info, [ctx.level_texel_mode(0, l) for l in range(2)]

    def rescaled(l, L):                                          # DT normalised to [0, 1] instead of [0, 255]
        dt = (np.asarray(L.now_dt, np.float32) * np.float32(1.0 / 256)).astype(np.float32)       # a power of two: still sqrt(d2) * s exactly
        return (dt,) + _image_gradient(dt, L.rows, L.cols)

    def bad_gradient(l, L):
        gx = np.array(L.now_gx, np.float32).ravel().copy()
        gx[gx.size // 2] += np.float32(0.25)
        return np.asarray(L.now_dt, np.float32), gx, np.asarray(L.now_gy, np.float32)

    def inexact_dt(l, L):                                        # one value moved by an ulp: no integer d2 reproduces it
        dt = np.array(L.now_dt, np.float32).ravel().copy()
        k = int(np.argmax(dt > 10))
        dt[k] = np.nextafter(dt[k], np.float32(1e9), dtype=np.float32)
        return (dt,) + _image_gradient(dt, L.rows, L.cols)

    with DvoContext(1, team_size=1) as ctx:
        ctx.set_intrinsics(*sc.intrinsics)
        ctx.set_direct_compact(True)
        info, modes = run(ctx, rescaled)
        assert all(i > 0 for i in info) and modes == [2, 2]
        info, modes = run(ctx, bad_gradient)
        assert info == [-4, -4] and modes == [0, 0]
        info, modes = run(ctx, inexact_dt)
        assert info == [-1, -1] and modes == [0, 0]
        info, modes = run(ctx, lambda l, L: (L.now_dt, L.now_gx, L.now_gy))
        assert all(i > 0 for i in info) and modes == [2, 2]


def test_float_images_in_device_memory_direct_build(oracle):
    """dvo_set_now_level_device: the same build from device pointers, no host round trip"""
    import torch
    from rgbd_odometry_amd import DvoContext, SynthScene
    sc = SynthScene(320, 240, 3, 12)
    iters = [6, 6, 6]
    ref = oracle.align_pyramid(iters, oracle_lib.scene_levels(sc, oracle), sc.intrinsics, np.eye(3), np.zeros(3))
    with DvoContext(1) as ctx:
        ctx.set_intrinsics(*sc.intrinsics)
        ctx.set_direct_compact(True)
        keep = []
        for l, L in enumerate(sc.levels):
            ctx.set_ref_level_from_images(l, L.ref_edge, L.ref_depth, L.rows, L.cols)
            t = [torch.from_numpy(np.ascontiguousarray(np.asarray(a, np.float32).ravel())).cuda() for a in (L.now_dt, L.now_gx, L.now_gy)]
            keep.append(t)
            ctx.set_now_level_device(l, t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(), L.rows, L.cols)
        ctx.synchronize()
        assert all(ctx.now_compact_info(0, l) > 0 for l in range(3))
        _check(ctx, ref, iters)
        assert [ctx.level_texel_mode(0, l) for l in range(3)] == [2, 2, 2]


def test_launch_shape_follows_the_number_of_pairs(oracle):
    """policy (dvo_capi.cpp, enqueue): with compact now levels, up to one pair per CU a pair gets one 512-thread workgroup;
    as soon as there are more pairs than CUs, 256-thread workgroups (two per CU, all resident in one round); results are the
    oracle's in both shapes"""
    import torch
    from rgbd_odometry_amd import DvoContext, SynthScene
    n_cu = torch.cuda.get_device_properties(0).multi_processor_count
    sc = SynthScene(640, 480, 2, 21)              # a finest list that does not fit half a CU's LDS (shorter lists take 256 threads earlier)
    iters = [5, 5]
    ref = oracle.align_pyramid(iters, oracle_lib.scene_levels(sc, oracle), sc.intrinsics, np.eye(3), np.zeros(3))
    for n, want_block in ((n_cu, 512), (n_cu + 1, 256)):
        with DvoContext(n) as ctx:
            ctx.set_intrinsics(*sc.intrinsics)
            for l, L in enumerate(sc.levels):
                ctx.set_ref_level_from_images(l, L.ref_edge, L.ref_depth, L.rows, L.cols)
                ctx.set_now_level_from_edges(l, (np.asarray(L.now_edge) != 0).astype(np.uint8) * 255, L.rows, L.cols)
            ctx.replicate_pairs(1)
            R, t = ctx.align_batch(iters, np.tile(np.eye(3), (n, 1, 1)), np.zeros((n, 3)))
            block, team, packed = ctx.last_launch_shape()
            assert (block, team, packed) == (want_block, 1, True), (n, block, team, packed)
            for p in (0, n - 1):
                assert rot_angle(ref["R"], R[p]) <= ROT_TOL and np.linalg.norm(ref["t"] - t[p]) <= TRANS_TOL
                for l, rep in ref["levels"].items():
                    e, b, ratio = ctx.level_report(p, l, iters[l])
                    assert np.array_equal(e, rep["energy"]) and b == rep["best_idx"] and ratio == rep["visible_ratio"]
