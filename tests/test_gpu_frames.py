"""Rows f1 + f2 on the GPU (frames in: pyramid, Canny, distance transform, point extraction) vs the oracle,
bit for bit, through the C ABI (dvo_frames_*)."""
import numpy as np
import pytest

import frame_gen
import oracle_lib
from oracle_lib import rot_angle

pytestmark = pytest.mark.gpu

K640 = (525.0, 525.0, 319.5, 239.5)


def _ctx(n_pairs=1, **kw):
    from rgbd_odometry_amd import DvoContext
    return DvoContext(n_pairs, **kw)


def _cm(a_rm):
    """row-major 2-D array -> flat column-major buffer (cv2eigen)"""
    return np.ascontiguousarray(np.asarray(a_rm).T).ravel()


@pytest.mark.parametrize("shape,first_shift,levels", [((480, 640), 0, 4), ((480, 640), 1, 4), ((241, 323), 1, 3), ((97, 131), 0, 2)])
def test_camera_pyramid_matches_oracle(oracle, shape, first_shift, levels):
    bgr, depth = frame_gen.camera_frame(3, *shape)
    ref = oracle.build_pyramid(bgr, depth, levels, first_shift)
    with _ctx() as ctx:
        ctx.frames_upload_cameras([bgr], [depth], n_levels=levels, first_shift=first_shift)
        for l, (g, d16) in enumerate(ref):
            grey, dep, edge, ne = ctx.frame_level(0, l)
            assert grey.shape == g.shape
            assert np.array_equal(grey, g), f"grey level {l}"
            assert np.array_equal(dep, d16.astype(np.float32)), f"depth level {l}"          # includes NaN/0 -> 1
            assert np.array_equal(edge, oracle.canny(g)), f"canny level {l}"
            assert ne == int((edge > 0).sum())


@pytest.mark.parametrize("shape,first_shift,levels,K,D", [
    ((480, 640), 1, 4, (525.0, 525.0, 319.5, 239.5), (0.12, -0.25, 0.0012, -0.0009, 0.11)),       # Xtion-like calibration
    ((480, 640), 0, 3, (517.3, 516.5, 318.6, 255.3), (0.2624, -0.9531, -0.0054, 0.0026, 1.1633)),  # TUM freiburg1 camera_info
    ((241, 323), 1, 3, (260.0, 262.0, 160.2, 119.1), (-0.3, 0.1, 0.002, 0.001, -0.02)),            # barrel, odd size, stripes of 12 rows
    ((97, 131), 0, 2, (100.0, 101.0, 65.0, 48.0), (0.0, 0.0, 0.0, 0.0, 0.0)),                      # zero distortion = identity map
])
def test_undistorted_camera_pyramid_matches_oracle(oracle, shape, first_shift, levels, K, D):
    """cv::undistort of the publisher (camTopic2PublisherPyD.cpp:88-107, :306-308) folded into the pyramid kernel: bgr and
    the 16-bit depth are remapped with OpenCV's fixed-point bilinear tables; bit-equal to the oracle's restatement"""
    bgr, depth = frame_gen.camera_frame(5, *shape)
    ref = oracle.build_pyramid(bgr, depth, levels, first_shift, undistort=(K, D))
    plain = oracle.build_pyramid(bgr, depth, levels, first_shift)
    with _ctx() as ctx:
        ctx.frames_set_undistort(shape[0], shape[1], K, D)
        ctx.frames_upload_cameras([bgr], [depth], n_levels=levels, first_shift=first_shift)
        for l, (g, d16) in enumerate(ref):
            grey, dep, edge, ne = ctx.frame_level(0, l)
            assert np.array_equal(grey, g), f"grey level {l}: {np.abs(grey.astype(int) - g.astype(int)).max()}"
            assert np.array_equal(dep, d16.astype(np.float32)), f"depth level {l}"
            assert np.array_equal(edge, oracle.canny(g)), f"canny level {l}"
        if any(D):
            assert not np.array_equal(ref[0][0], plain[0][0])          # the map really moves pixels
        else:
            assert all(np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) for a, b in zip(ref, plain))
        # another image size is refused; switching off restores the plain pyramid
        from rgbd_odometry_amd import DvoError
        with pytest.raises(DvoError):
            ctx.frames_upload_cameras([bgr[:-2]], [depth[:-2]], n_levels=levels, first_shift=first_shift)
        ctx.frames_set_undistort(0, 0, None, None)
        ctx.frames_upload_cameras([bgr], [depth], n_levels=levels, first_shift=first_shift)
        for l, (g, d16) in enumerate(plain):
            grey, dep, _, _ = ctx.frame_level(0, l)
            assert np.array_equal(grey, g) and np.array_equal(dep, d16.astype(np.float32))


def _canny_images():
    rng = np.random.default_rng(7)
    yield "noise", rng.integers(0, 256, (75, 101)).astype(np.uint8)
    yield "noise_small_amplitude", (120 + rng.integers(0, 40, (64, 64))).astype(np.uint8)
    g = np.zeros((48, 64), np.uint8); g[:, 32:] = 200
    yield "step_x", g
    yield "step_y", np.ascontiguousarray(g.T)
    yy, xx = np.mgrid[0:90, 0:70]
    yield "diag", ((xx + yy) % 23 < 11).astype(np.uint8) * 180 + 20
    yield "disc_ties", (((xx - 35) ** 2 + (yy - 45) ** 2) < 900).astype(np.uint8) * 255       # plateaus -> equal magnitudes
    yield "checker", (((xx // 4) + (yy // 4)) % 2).astype(np.uint8) * 255
    yield "flat", np.full((33, 47), 99, np.uint8)
    yield "tiny3", rng.integers(0, 256, (3, 3)).astype(np.uint8)
    yield "row1", rng.integers(0, 256, (1, 19)).astype(np.uint8)
    yield "col1", rng.integers(0, 256, (23, 1)).astype(np.uint8)
    yield "long_chain", np.kron(rng.integers(0, 2, (30, 40)).astype(np.uint8) * 200, np.ones((6, 6), np.uint8))
    yield "frame", oracle_lib.load().bgr2gray(frame_gen.camera_frame(11, 240, 320)[0])


@pytest.mark.parametrize("name,img", list(_canny_images()))
def test_canny_matches_oracle(oracle, name, img):
    with _ctx() as ctx:
        ctx.frames_upload_pyramids([[(img, None)]])
        _, _, edge, ne = ctx.frame_level(0, 0, want_depth=False)
        ref = oracle.canny(img)
        assert np.array_equal(edge, ref), f"{name}: {int((edge != ref).sum())} pixels differ"
        assert ne == int((ref > 0).sum())


@pytest.mark.parametrize("shape", [(37, 3000), (20, 5000), (12, 9000), (6, 40000), (3000, 41), (1100, 700), (9000, 12)])
def test_extreme_aspect_ratios_through_every_row_pass_variant(oracle, shape):
    """long rows exercise the 8-, 4-, 2- and 1-row LDS tiles of the exact distance transform's row pass, tall images the
    multi-chunk column pass (beyond 8192 rows: one wave per workgroup); the large edge-free region puts pixels further than
    511 pixels from every edge, which the native compact form refuses -- those images take the 16-byte texel fallback"""
    rng = np.random.default_rng(shape[1])
    img = np.kron(rng.integers(0, 2, ((shape[0] + 7) // 8, (shape[1] + 15) // 16)).astype(np.uint8) * 210 + 20,
                  np.ones((8, 16), np.uint8))[:shape[0], :shape[1]]
    img = np.ascontiguousarray(img)
    img[:, : shape[1] // 3] = 128                           # a large edge-free region: long scans
    with _ctx() as ctx:
        ctx.frames_upload_pyramids([[(img, None)]])
        ctx.frames_as_now(0, 0, 1)
        edge = ctx.frame_level(0, 0, want_depth=False)[2]
        assert np.array_equal(edge, oracle.canny(img))
        dt, gx, gy, _ = oracle.now_level_from_grey(img)
        for got, want in zip(ctx.get_now_level(0), (dt, gx, gy)):
            assert np.array_equal(got, want)


def test_canny_random_stress(oracle):
    """many random sizes and textures, batched, so that the tile-local and the cross-tile union-find meet every border case"""
    rng = np.random.default_rng(2024)
    for trial in range(12):
        rows, cols = int(rng.integers(2, 200)), int(rng.integers(2, 260))
        batch = []
        for k in range(6):
            kind = (trial + k) % 4
            if kind == 0:
                img = rng.integers(0, 256, (rows, cols))
            elif kind == 1:                                        # blocky: long chains across tiles
                b = int(rng.integers(2, 9))
                img = np.kron(rng.integers(0, 2, ((rows + b - 1) // b, (cols + b - 1) // b)) * 200 + 20, np.ones((b, b), np.int64))[:rows, :cols]
            elif kind == 2:                                        # smooth + noise near the thresholds
                yy, xx = np.mgrid[0:rows, 0:cols]
                img = 128 + 60 * np.sin(xx / 3.0) * np.cos(yy / 2.5) + rng.integers(-12, 13, (rows, cols))
            else:                                                  # sparse spikes
                img = np.full((rows, cols), 40); img[rng.random((rows, cols)) < 0.05] = 255
            batch.append(np.clip(img, 0, 255).astype(np.uint8))
        with _ctx(6) as ctx:
            ctx.frames_reserve(6)
            ctx.frames_upload_pyramids([[(im, None)] for im in batch])
            for k, im in enumerate(batch):
                edge = ctx.frame_level(k, 0, want_depth=False)[2]
                assert np.array_equal(edge, oracle.canny(im)), (trial, k, rows, cols)


@pytest.mark.parametrize("shrink", [1, 1000000])
def test_canny_all_levels_launch_stress(oracle, shrink, monkeypatch):
    """the all-levels launch (round 6: the tile kernel writes the edge map and two lists -- weak candidates, roots of strong
    components -- and the two passes after the border unions walk the lists): every level of a pyramid its own random texture, batched;
    with the lists' capacities shrunk to nothing every image overflows them and takes the dense form of both passes"""
    if shrink > 1:
        monkeypatch.setenv("DVO_CANNY_LIST_SHRINK", str(shrink))
    rng = np.random.default_rng(77 + shrink)

    def texture(kind, rows, cols):
        if kind == 0:
            img = rng.integers(0, 256, (rows, cols))
        elif kind == 1:                                            # blocky: long chains across tiles
            b = int(rng.integers(2, 9))
            img = np.kron(rng.integers(0, 2, ((rows + b - 1) // b, (cols + b - 1) // b)) * 200 + 20, np.ones((b, b), np.int64))[:rows, :cols]
        elif kind == 2:                                            # smooth + noise near the thresholds: many weak candidates
            yy, xx = np.mgrid[0:rows, 0:cols]
            img = 128 + 60 * np.sin(xx / 3.0) * np.cos(yy / 2.5) + rng.integers(-12, 13, (rows, cols))
        elif kind == 3:                                            # sparse spikes: many small strong components
            img = np.full((rows, cols), 40); img[rng.random((rows, cols)) < 0.05] = 255
        else:                                                      # low-amplitude noise: gradients between the thresholds
            img = 120 + rng.integers(0, 40, (rows, cols))
        return np.clip(img, 0, 255).astype(np.uint8)
    for trial, (rows, cols) in enumerate([(128, 96), (200, 260), (64, 32), (332, 132)]):
        sizes = [(rows, cols), (rows // 2, cols // 2), (rows // 4, cols // 4)]
        frames = [[(texture((trial + k + l) % 5, *sizes[l]), None) for l in range(3)] for k in range(5)]
        with _ctx(5) as ctx:
            ctx.frames_reserve(5)
            ctx.frames_upload_pyramids(frames)
            for k, fr in enumerate(frames):
                for l, (im, _) in enumerate(fr):
                    edge = ctx.frame_level(k, l, want_depth=False)[2]
                    ref = oracle.canny(im)
                    assert np.array_equal(edge, ref), (shrink, trial, k, l, int((edge != ref).sum()))


def test_canny_thresholds_from_params(oracle):
    img = oracle.bgr2gray(frame_gen.camera_frame(5, 120, 160)[0])
    with _ctx(canny_threshold1=40, canny_threshold2=90) as ctx:
        ctx.frames_upload_pyramids([[(img, None)]])
        edge = ctx.frame_level(0, 0, want_depth=False)[2]
        assert np.array_equal(edge, oracle.canny(img, 40.0, 90.0))
        assert not np.array_equal(edge, oracle.canny(img))


def test_layout_and_dtype_variants_agree(oracle):
    """mono8/mono16 row-major (the message) == float column-major (im_n/dim_n, after the node's 0 -> 1)"""
    from rgbd_odometry_amd.capi import DVO_LAYOUT_COL_MAJOR, DVO_LAYOUT_ROW_MAJOR
    bgr, depth = frame_gen.camera_frame(9, 120, 160)
    pyr = oracle.build_pyramid(bgr, depth, 2, 0)
    with _ctx() as a, _ctx() as b:
        a.frames_upload_pyramids([pyr], layout=DVO_LAYOUT_ROW_MAJOR)
        eig = [(g.astype(np.float32), np.where(d == 0, 1, d).astype(np.float32)) for g, d in pyr]
        b.frames_upload_pyramids([eig], layout=DVO_LAYOUT_COL_MAJOR)
        for l in range(2):
            for x, y in zip(a.frame_level(0, l)[:3], b.frame_level(0, l)[:3]):
                assert np.array_equal(x, y)


def test_now_frame_matches_oracle(oracle):
    bgr, depth = frame_gen.camera_frame(21, 240, 320)
    pyr = oracle.build_pyramid(bgr, depth, 3, 0)
    with _ctx() as ctx:
        ctx.frames_upload_pyramids([pyr])
        ctx.frames_as_now(0, 0, 1)
        for l, (g, _) in enumerate(pyr):
            dt, gx, gy, _ = oracle.now_level_from_grey(g)
            ddt, dgx, dgy = ctx.get_now_level(l)
            assert np.array_equal(ddt, dt) and np.array_equal(dgx, gx) and np.array_equal(dgy, gy), f"level {l}"


def test_ref_frame_matches_oracle(oracle):
    bgr, depth = frame_gen.camera_frame(22, 240, 320)
    pyr = oracle.build_pyramid(bgr, depth, 3, 0)
    K = tuple(np.float32(k * 0.5) for k in K640)
    with _ctx() as ctx:
        ctx.set_intrinsics(*[float(k) for k in K])
        ctx.frames_upload_pyramids([pyr])
        N = ctx.frames_as_ref(0, 0, 1)
        ctx.frames_as_now(0, 0, 1)
        for l, (g, d) in enumerate(pyr):
            xyz, uv, _ = oracle.ref_level_from_grey(l, g, d, K)
            assert N[0, l] == len(xyz)
            assert np.array_equal(ctx.get_ref_level(l), xyz), f"level {l}"


def test_frames_end_to_end_alignment(oracle):
    """frames in -> poses out, against the oracle's preprocessing + align_pyramid on the same frames"""
    iters = [10, 10, 10, 10]
    K = tuple(np.float32(k) for k in K640)
    ref_bgr, ref_d = frame_gen.camera_frame(31, 480, 640, holes=True)
    now_bgr, now_d = frame_gen.camera_frame(31, 480, 640, shift=(2, -3), holes=True)
    rp, nw = oracle.build_pyramid(ref_bgr, ref_d, 4, 0), oracle.build_pyramid(now_bgr, now_d, 4, 0)
    levels = []
    for l in range(4):
        xyz, uv, _ = oracle.ref_level_from_grey(l, rp[l][0], rp[l][1], K)
        dt, gx, gy, _ = oracle.now_level_from_grey(nw[l][0])
        levels.append(dict(xyz=xyz, uv=uv, dt=dt, gx=gx, gy=gy, rows=rp[l][0].shape[0], cols=rp[l][0].shape[1]))
    want = oracle.align_pyramid(iters, levels, K, np.eye(3), np.zeros(3))
    with _ctx() as ctx:
        ctx.set_intrinsics(*[float(k) for k in K])
        ctx.frames_upload_cameras([ref_bgr, now_bgr], [ref_d, now_d], n_levels=4, first_shift=0)
        ctx.frames_as_ref(0, 0, 1)
        ctx.frames_as_now(1, 0, 1)
        R, t = ctx.align_batch(iters, np.eye(3)[None], np.zeros((1, 3)))
        assert rot_angle(want["R"], R[0]) <= 1e-5 and np.linalg.norm(want["t"] - t[0]) <= 1e-4       # north_star tolerance
        for l in range(4):
            e, best, ratio = ctx.level_report(0, l, iters[l])
            assert np.array_equal(e, want["levels"][l]["energy"]) and best == want["levels"][l]["best_idx"]
        assert np.linalg.norm(t[0]) > 1e-4          # the shifted frame really moved the pose


def test_batched_calls_equal_single_calls(oracle):
    frames = [frame_gen.camera_frame(40 + i, 120, 160) for i in range(5)]
    K = tuple(float(k) * 0.25 for k in K640)
    with _ctx(5) as a, _ctx(1) as b:
        a.set_intrinsics(*K); b.set_intrinsics(*K)
        a.frames_reserve(8)
        a.frames_upload_cameras([f[0] for f in frames], [f[1] for f in frames], n_levels=3, first_shift=0, first_slot=2)
        Na = a.frames_as_ref(2, 0, 5)
        a.frames_as_now(2, 0, 5)
        for i, (bgr, d) in enumerate(frames):
            b.frames_upload_cameras([bgr], [d], n_levels=3, first_shift=0)
            Nb = b.frames_as_ref(0, 0, 1)
            b.frames_as_now(0, 0, 1)
            assert np.array_equal(Na[i], Nb[0])
            for l in range(3):
                for x, y in zip(a.frame_level(2 + i, l)[:3], b.frame_level(0, l)[:3]):
                    assert np.array_equal(x, y)
                for x, y in zip(a.get_now_level(l, pair=i), b.get_now_level(l)):
                    assert np.array_equal(x, y)
                ea, eb = a.eval_points(l, np.eye(3), np.zeros(3), pair=i), b.eval_points(l, np.eye(3), np.zeros(3))
                assert np.array_equal(ea["reproj"], eb["reproj"]) and np.array_equal(ea["eps"], eb["eps"])


def test_uploads_spanning_several_landing_buffers(oracle):
    """40 camera frames of 2.15 MB cross the 32 MB landing buffers three times (copy streams, events, both buffers)"""
    distinct = [frame_gen.camera_frame(60 + i, 480, 640) for i in range(4)]
    n = 40
    with _ctx(n) as ctx, _ctx(1) as one:
        ctx.frames_reserve(n)
        ctx.frames_upload_cameras([distinct[i % 4][0] for i in range(n)], [distinct[i % 4][1] for i in range(n)],
                                  n_levels=3, first_shift=0, now_first_pair=0)
        want = {}
        for i in range(4):
            one.frames_upload_cameras([distinct[i][0]], [distinct[i][1]], n_levels=3, first_shift=0)
            one.frames_as_now(0, 0, 1)
            want[i] = ([one.frame_level(0, l)[:3] for l in range(3)], [one.get_now_level(l) for l in range(3)])
        for slot in (0, 1, 13, 14, 15, 16, 28, 29, 30, 39):
            lv, nw = want[slot % 4]
            for l in range(3):
                for x, y in zip(ctx.frame_level(slot, l)[:3], lv[l]):
                    assert np.array_equal(x, y), (slot, l)
                for x, y in zip(ctx.get_now_level(l, pair=slot), nw[l]):
                    assert np.array_equal(x, y), (slot, l)


def test_prev_now_frame_becomes_reference_without_upload(oracle):
    """setPrevFrameAsRefFrame (SolveDVO.cpp:559-583): a stored frame serves as now frame, then as reference"""
    K = tuple(float(k) * 0.5 for k in K640)
    f0, f1 = frame_gen.camera_frame(50, 240, 320), frame_gen.camera_frame(50, 240, 320, shift=(1, 2))
    with _ctx() as ctx:
        ctx.set_intrinsics(*K)
        ctx.frames_upload_cameras([f0[0], f1[0]], [f0[1], f1[1]], n_levels=3, first_shift=0)
        ctx.frames_as_ref(0, 0, 1); ctx.frames_as_now(1, 0, 1)
        R1, t1 = ctx.align_batch([8, 8, 8], np.eye(3)[None], np.zeros((1, 3)))
        ctx.frames_as_ref(1, 0, 1); ctx.frames_as_now(0, 0, 1)           # roles swapped, nothing uploaded
        R2, t2 = ctx.align_batch([8, 8, 8], np.eye(3)[None], np.zeros((1, 3)))
        assert np.all(np.isfinite(t1)) and np.all(np.isfinite(t2))
        assert np.dot(t1[0], t2[0]) < 0                                  # opposite motions


def test_frame_api_errors():
    from rgbd_odometry_amd.capi import DvoError
    bgr, depth = frame_gen.camera_frame(1, 60, 80)
    with _ctx() as ctx:
        with pytest.raises(DvoError):
            ctx.frames_upload_pyramids([[(np.zeros((6, 66000), np.uint8), None)]])    # (rows+cols+1)^2 would overflow int32
        with pytest.raises(DvoError):
            ctx.frames_as_now(0, 0, 1)                                   # empty store
        ctx.frames_upload_cameras([bgr], None, n_levels=2, first_shift=0)
        with pytest.raises(DvoError):
            ctx.frames_as_ref(0, 0, 1)                                   # no intrinsics
        ctx.set_intrinsics(60.0, 60.0, 40.0, 30.0)
        with pytest.raises(DvoError):
            ctx.frames_as_ref(0, 0, 1)                                   # no depth in that slot
        with pytest.raises(DvoError):
            ctx.frames_as_now(1, 0, 1)                                   # slot never filled
        with pytest.raises(DvoError):
            ctx.frames_upload_cameras([bgr], None, n_levels=2, first_shift=0, first_slot=10 ** 6)
        flat = np.full((60, 80, 3), 77, np.uint8)
        ctx.frames_upload_cameras([flat], [np.full((60, 80), 2.0, np.float32)], n_levels=2, first_shift=0)
        with pytest.raises(DvoError):
            ctx.frames_as_ref(0, 0, 1)                                   # no edge => no reference point (:282)
        ctx.frames_as_now(0, 0, 1)                                       # an edge-free now frame is all zeros
        dt, gx, gy = ctx.get_now_level(0)
        assert not dt.any() and not gx.any() and not gy.any()


def _sparse_edge_maps():
    rng = np.random.default_rng(77)
    def blank(r, c): return np.zeros((r, c), np.uint8)
    e = blank(480, 640); e[5, 7] = 255                                   # one edge pixel: distances to 800, squared beyond 2^18
    yield "single_pixel_corner", e
    e = blank(480, 640); e[200:203, 300:340] = 255                       # one short stroke: every scan is long, < 512 everywhere
    yield "one_stroke_centre", e
    e = blank(300, 700); e[:, 0] = 255                                   # left border column only: pure horizontal distances up to 699
    yield "left_border_line", e
    e = blank(301, 397); e[150, :] = 255                                 # odd sizes (byte-wise column pass, odd last row pair), pure vertical
    yield "odd_sizes_middle_row", e
    e = blank(481, 637); e[rng.integers(0, 481, 40), rng.integers(0, 637, 40)] = 255
    yield "odd_sizes_40_points", e                                       # distances of a few dozen to ~150 pixels: pad and 8-bit limits
    e = blank(480, 640); e[rng.integers(0, 480, 6), rng.integers(0, 640, 6)] = 200
    yield "six_points_non_255", e                                        # any non-zero byte is an edge; squared distances past 65535
    e = (rng.random((480, 640)) < 0.3).astype(np.uint8) * 255
    yield "dense_random", e
    e = blank(1030, 40); e[515, 20] = 255                                 # three 512-row chunks in the column pass
    yield "tall_three_chunks", e
    e = blank(3000, 48); e[[7, 1499, 2990], [3, 40, 20]] = 255            # six 512-row chunks, edges in the first, third and last: the
    yield "tall_six_chunks", e                                            # chunk borders' distances travel over empty chunks both ways


@pytest.mark.parametrize("name,edge", list(_sparse_edge_maps()))
def test_distance_transform_of_sparse_and_odd_edge_maps(oracle, name, edge):
    """the exact distance transform behind dvo_set_now_level_from_edges (eight rows per lane in the column pass; packed 16-bit
    row scan with its exact 32-bit finish for pixels further than the tile's pad or than 255 pixels from every edge): DT, gx, gy
    bit-equal to the oracle whichever form the level ends up in"""
    rows, cols = edge.shape
    want = oracle.now_level_from_edges(_cm(edge), rows, cols)
    with _ctx() as ctx:
        ctx.set_intrinsics(500.0, 500.0, cols / 2, rows / 2)
        ctx.set_now_level_from_edges(0, _cm(edge), rows, cols)
        got = ctx.get_now_level(0)
        for g, w, what in zip(got, want, ("DT", "gx", "gy")):
            assert np.array_equal(g, w), f"{name}: {what} differs in {int((g != w).sum())} pixels"


def test_distance_transform_refuses_a_mask_without_edges():
    from rgbd_odometry_amd import DvoError
    with _ctx() as ctx:
        ctx.set_intrinsics(500.0, 500.0, 160.0, 120.0)
        with pytest.raises(DvoError):
            ctx.set_now_level_from_edges(0, np.zeros(240 * 320, np.uint8), 240, 320)


@pytest.mark.parametrize("first_shift,misalign", [(0, 0), (1, 0), (0, 1), (0, 2)])
def test_camera_frames_in_device_memory_equal_host_uploads(oracle, first_shift, misalign):
    """DVO_UPLOAD_DEVICE: BGR8 + depth images that already sit in HBM (torch tensors here) go through the same pyramid / Canny /
    distance transform as host uploads; as_now fused into the upload (now_first_pair) as well.  Round 6: aligned frames are READ WHERE
    THEY ARE through a pointer table (the full-resolution kernel at shift 0, the general one at shift 1; tables prepared once or built
    per call; a second batch through the same staging table); a frame at an odd address (misalign 1: BGR, 2: depth) takes the landing
    copy as before"""
    import torch
    frames = [frame_gen.camera_frame(300 + i, 240, 320) for i in range(3)]
    with _ctx(3) as a, _ctx(3) as b:
        for c in (a, b):
            c.set_intrinsics(262.5, 262.5, 159.75, 119.75)
            c.frames_reserve(3)
        a.frames_upload_cameras([f[0] for f in frames], [f[1] for f in frames], n_levels=3, first_shift=first_shift, now_first_pair=0)

        def dev(arr, off):
            raw = torch.zeros(arr.nbytes + 64, dtype=torch.uint8, device="cuda")
            raw[off:off + arr.nbytes] = torch.from_numpy(np.frombuffer(np.ascontiguousarray(arr).tobytes(), np.uint8).copy()).cuda()
            return raw
        tb = [dev(f[0], 1 if misalign == 1 else 0) for f in frames]
        td = [dev(np.asarray(f[1], np.float32), 4 if misalign == 2 else 0) for f in frames]
        pb = [t.data_ptr() + (1 if misalign == 1 else 0) for t in tb]
        pd = [t.data_ptr() + (4 if misalign == 2 else 0) for t in td]
        # first other frames through the same staging table (its reuse), then the ones compared -- as prepared tables
        b.frames_upload_cameras_device(list(reversed(pb)), list(reversed(pd)), 240, 320, n_levels=3, first_shift=first_shift, now_first_pair=0)
        b.frames_upload_cameras_device(b.pointer_table(pb), b.pointer_table(pd), 240, 320, n_levels=3, first_shift=first_shift,
                                       now_first_pair=0)
        for slot in range(3):
            for l in range(3):
                for x, y in zip(a.frame_level(slot, l)[:3], b.frame_level(slot, l)[:3]):
                    assert np.array_equal(x, y)
        for l in range(3):
            for p in range(3):
                for x, y in zip(a.get_now_level(l, pair=p), b.get_now_level(l, pair=p)):
                    assert np.array_equal(x, y)


def test_camera_frames_pulled_out_of_mapped_host_memory_equal_dma_uploads(oracle):
    """DVO_UPLOAD_MAPPED: pinned host buffers the GPU addresses are gathered by a kernel in the upload pipeline's own chunks;
    same frame store as the DMA path, more frames than one chunk holds"""
    import torch
    from rgbd_odometry_amd.capi import DVO_UPLOAD_MAPPED
    n = 5
    frames = [frame_gen.camera_frame(400 + i, 240, 320) for i in range(n)]
    pb = [torch.from_numpy(np.ascontiguousarray(f[0])).pin_memory() for f in frames]
    pd = [torch.from_numpy(np.ascontiguousarray(f[1], dtype=np.float32)).pin_memory() for f in frames]
    with _ctx(n) as a, _ctx(n) as b:
        for c in (a, b):
            c.set_intrinsics(262.5, 262.5, 159.75, 119.75)
            c.frames_reserve(n)
        a.frames_upload_cameras([f[0] for f in frames], [f[1] for f in frames], n_levels=3, first_shift=0, now_first_pair=0)
        b.frames_upload_cameras([t.numpy() for t in pb], [t.numpy() for t in pd], n_levels=3, first_shift=0, now_first_pair=0,
                                flags=DVO_UPLOAD_MAPPED)
        for slot in range(n):
            for l in range(3):
                for x, y in zip(a.frame_level(slot, l)[:3], b.frame_level(slot, l)[:3]):
                    assert np.array_equal(x, y)
            for l in range(3):
                for x, y in zip(a.get_now_level(l, pair=slot), b.get_now_level(l, pair=slot)):
                    assert np.array_equal(x, y)


def test_pyramid_levels_pulled_out_of_mapped_host_memory_equal_dma_uploads(oracle):
    """DVO_UPLOAD_MAPPED on the node's wire format (mono8 + mono16 levels): every level gathered by the kernel"""
    import torch
    from rgbd_odometry_amd.capi import DVO_UPLOAD_MAPPED
    n = 4
    pyrs = []
    for i in range(n):
        bgr, depth = frame_gen.camera_frame(500 + i, 240, 320)
        pyrs.append(oracle.build_pyramid(bgr, depth, 3, 0))
    keep = []
    def pinned(a):
        t = torch.from_numpy(np.ascontiguousarray(a)).pin_memory(); keep.append(t); return t.numpy()
    pinned_pyrs = [[(pinned(g), pinned(d)) for g, d in p] for p in pyrs]
    with _ctx(n) as a, _ctx(n) as b:
        for c in (a, b):
            c.set_intrinsics(262.5, 262.5, 159.75, 119.75)
            c.frames_reserve(n)
        a.frames_upload_pyramids(pyrs)
        b.frames_upload_pyramids(pinned_pyrs, flags=DVO_UPLOAD_MAPPED)
        for slot in range(n):
            for l in range(3):
                for x, y in zip(a.frame_level(slot, l)[:3], b.frame_level(slot, l)[:3]):
                    assert np.array_equal(x, y)


def test_mapped_host_allocator_of_the_c_abi(oracle):
    """dvo_host_alloc_mapped: pinned, GPU-addressable host memory for callers without the HIP runtime; frames kept there go up
    with DVO_UPLOAD_MAPPED"""
    from rgbd_odometry_amd.capi import DVO_UPLOAD_MAPPED, MappedHostArray
    bgr, depth = frame_gen.camera_frame(600, 240, 320)
    mb, md = MappedHostArray(bgr.shape, np.uint8), MappedHostArray(depth.shape, np.float32)
    mb.array[...] = bgr
    md.array[...] = depth
    with _ctx() as a, _ctx() as b:
        for c in (a, b):
            c.set_intrinsics(262.5, 262.5, 159.75, 119.75)
            c.frames_reserve(1)
        a.frames_upload_cameras([bgr], [depth], n_levels=3, first_shift=0)
        b.frames_upload_cameras([mb.array], [md.array], n_levels=3, first_shift=0, flags=DVO_UPLOAD_MAPPED)
        for l in range(3):
            for x, y in zip(a.frame_level(0, l)[:3], b.frame_level(0, l)[:3]):
                assert np.array_equal(x, y)
    mb.free(); md.free()


def test_camera_and_pyramid_uploads_interleaved_over_the_landing_buffers(oracle):
    """the two upload entry points share the landing buffers, their events and the copy streams: multi-chunk camera uploads
    (copies of chunk k+1 submitted ahead of the kernels of chunk k), a pyramid upload in between, every path (pinned mirror,
    DIRECT, MAPPED) -- every stored level equals the one-frame-at-a-time result"""
    import torch
    from rgbd_odometry_amd.capi import DVO_UPLOAD_DIRECT, DVO_UPLOAD_MAPPED
    distinct = [frame_gen.camera_frame(80 + i, 480, 640) for i in range(3)]
    n = 38
    with _ctx(1) as one:
        want = []
        for b, d in distinct:
            one.frames_upload_cameras([b], [d], n_levels=3, first_shift=0)
            want.append([one.frame_level(0, l)[:3] for l in range(3)])
        pyr = [[tuple(np.ascontiguousarray(x) for x in one.frame_level(0, l)[:2]) for l in range(3)]]     # the last frame as a pyramid
    pinned = [(torch.from_numpy(np.ascontiguousarray(b)).pin_memory(), torch.from_numpy(np.ascontiguousarray(d)).pin_memory()) for b, d in distinct]
    with _ctx(1) as ctx:
        ctx.frames_reserve(n + 2)
        for flags in (0, DVO_UPLOAD_DIRECT, DVO_UPLOAD_MAPPED, 0):
            src = pinned if flags else None
            bl = [(src[i % 3][0].numpy() if src else distinct[i % 3][0]) for i in range(n)]
            dl = [(src[i % 3][1].numpy() if src else distinct[i % 3][1]) for i in range(n)]
            ctx.frames_upload_cameras(bl, dl, n_levels=3, first_shift=0, first_slot=0, flags=flags)
            ctx.frames_upload_pyramids(pyr, first_slot=n)
            ctx.frames_upload_cameras(bl[:5], dl[:5], n_levels=3, first_shift=0, first_slot=n - 5, flags=flags)
            for slot in (0, 1, 14, 15, 16, 29, 30, n - 6, n - 5, n - 1):
                i = slot % 3 if slot < n - 5 else (slot - (n - 5)) % 3
                for l in range(3):
                    for x, y in zip(ctx.frame_level(slot, l)[:3], want[i][l]):
                        assert np.array_equal(x, y), (flags, slot, l)
            for l in range(3):
                for x, y in zip(ctx.frame_level(n, l)[:2], want[2][l][:2]):
                    assert np.array_equal(x, y), (flags, "pyramid slot", l)


def test_full_hd_frame_five_levels_through_the_level_tables(oracle):
    """1920x1080, five levels: the one-launch-per-stage kernels with the row pass's LDS tile beyond 48 KB (function attribute),
    level sizes that are not multiples of the tiles (1080 / 64, 135, 68 rows), pyramid, Canny and now levels against the oracle"""
    bgr, depth = frame_gen.camera_frame(700, 1080, 1920)
    pyr = oracle.build_pyramid(bgr, depth, 5, 0)
    with _ctx() as ctx:
        ctx.set_intrinsics(1575.0, 1575.0, 959.5, 539.5)
        ctx.frames_reserve(1)
        ctx.frames_upload_cameras([bgr], [depth], n_levels=5, first_shift=0, now_first_pair=0)
        for l, (g, d16) in enumerate(pyr):
            grey, dep, edge, _ = ctx.frame_level(0, l)
            assert np.array_equal(grey, g), f"grey level {l}"
            assert np.array_equal(dep, d16.astype(np.float32)), f"depth level {l}"
            assert np.array_equal(edge, oracle.canny(g)), f"canny level {l}"
            dt, gx, gy, _ = oracle.now_level_from_grey(g)
            for got, want, what in zip(ctx.get_now_level(l), (dt, gx, gy), ("DT", "gx", "gy")):
                assert np.array_equal(got, want), f"{what} level {l}"


def test_gpu_frame_stage_matches_reference_vectors(oracle):
    """round 6: the HIP frame stage against the REAL reference's now-frame preprocessing (OpenCV 2.4 through SolveDVO::
    computeDistTransfrmOfNow, src/SolveDVO.cpp:1740-1799), from the vectors tools/ref_dump/frames_dump.cpp produces -- skipped until
    somebody has run that driver (row f1 UNPINNED); the CPU twin and the format test live in tests/test_frames_oracle.py"""
    import os
    import test_frames_oracle as tfo
    if not os.path.exists(tfo.REFERENCE_FRAMES):
        pytest.skip("tests/golden/reference_frames_golden.npz absent: tools/ref_dump/frames_dump.cpp has not been run against the reference")
    _, export = tfo._frames_tools()
    ref = np.load(tfo.REFERENCE_FRAMES)
    for name, seed, rows, cols, levels, fs in export.FRAMES:
        bgr, depth = frame_gen.camera_frame(seed, rows, cols)
        pyr = oracle.build_pyramid(bgr, depth, levels, fs)
        got = {f"{name}_levels": np.array(levels, np.int32)}
        with _ctx() as ctx:
            ctx.frames_upload_pyramids([pyr])
            ctx.frames_as_now(0, 0, 1)
            for l, (g, _) in enumerate(pyr):
                ddt, dgx, dgy = ctx.get_now_level(l)
                edge = ctx.frame_level(0, l)[2]
                got[f"{name}_L{l}_shape"] = np.array(g.shape, np.int32)
                got[f"{name}_L{l}_edge"] = _cm(edge).astype(np.int32)
                got[f"{name}_L{l}_dt"], got[f"{name}_L{l}_gx"], got[f"{name}_L{l}_gy"] = ddt, dgx, dgy
        tfo.compare_with_reference_frames(ref, got, name)
