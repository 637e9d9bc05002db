"""Rows f1/f2, CPU side: the oracle restatement of the OpenCV-2.4 steps (oracle/dvo_oracle_frames.cpp) against its
committed golden vectors and against the properties the definitions imply.  GPU: the HIP path reproduces the vectors."""
import os

import numpy as np
import pytest

import frame_gen

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "frames_golden.npz")


def _mg():
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_frames_golden", os.path.join(os.path.dirname(GOLDEN), "make_frames_golden.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


@pytest.fixture(scope="module")
def golden():
    return np.load(GOLDEN)


def test_frame_generator_and_oracle_reproduce_golden(golden, oracle):
    fresh = _mg().build()
    assert set(fresh) == set(golden.files)
    for k in golden.files:
        assert np.array_equal(golden[k], fresh[k], equal_nan=True), k


def test_sobel_is_the_3x3_operator_with_replicated_border(oracle):
    rng = np.random.default_rng(0)
    g = rng.integers(0, 256, (19, 23)).astype(np.uint8)
    dx, dy = oracle.sobel3(g)
    p = np.pad(g.astype(np.int32), 1, mode="edge")
    kx = np.array([[-1, 0, 1], [-2, 0, 2], [-1, 0, 1]])
    ex = sum(kx[i, j] * p[i:i + 19, j:j + 23] for i in range(3) for j in range(3))
    ey = sum(kx.T[i, j] * p[i:i + 19, j:j + 23] for i in range(3) for j in range(3))
    assert np.array_equal(dx, ex) and np.array_equal(dy, ey)


def test_canny_structure(oracle):
    """edges = candidates 8-connected to a strong candidate; candidates = (mag > low^2) local maxima"""
    from scipy import ndimage
    g = oracle.bgr2gray(frame_gen.camera_frame(2, 120, 160)[0])
    edge, mag, cand = oracle.canny(g, stages=True)
    assert set(np.unique(edge)) <= {0, 255}
    assert np.all(cand[edge > 0] > 0)                                   # edges are candidates
    assert np.all(edge[cand == 2] == 255)                               # every strong candidate is an edge
    assert np.all(mag[cand > 0] > 100 * 100) and np.all(mag[cand == 2] > 150 * 150)
    lab, n = ndimage.label(cand > 0, structure=np.ones((3, 3)))
    strong = np.zeros(n + 1, bool)
    strong[np.unique(lab[cand == 2])] = True
    assert np.array_equal(edge > 0, strong[lab] & (cand > 0))           # hysteresis == connected components
    assert 0.005 < (edge > 0).mean() < 0.25


def test_canny_threshold_order_is_irrelevant_and_flat_image_has_no_edges(oracle):
    g = oracle.bgr2gray(frame_gen.camera_frame(4, 60, 80)[0])
    assert np.array_equal(oracle.canny(g, 150, 100), oracle.canny(g, 100, 150))
    assert not oracle.canny(np.full((20, 30), 200, np.uint8)).any()
    g2 = np.zeros((48, 64), np.uint8); g2[:, 32:] = 200
    e = oracle.canny(g2)
    assert np.array_equal(np.unique(np.nonzero(e)[1]), [31])            # ties: "> left, >= right" keeps the left pixel


def test_bgr2gray_resize_depth_definitions(oracle):
    rng = np.random.default_rng(1)
    bgr = rng.integers(0, 256, (31, 45, 3)).astype(np.uint8)
    b64 = bgr.astype(np.int64)
    want = (1868 * b64[..., 0] + 9617 * b64[..., 1] + 4899 * b64[..., 2] + 8192) >> 14
    assert np.array_equal(oracle.bgr2gray(bgr), want)
    assert np.array_equal(oracle.bgr2gray(np.full((2, 2, 3), 255, np.uint8)), np.full((2, 2), 255))
    half = oracle.resize_nn(bgr, 0.5)
    assert half.shape == (16, 22, 3)                                    # cvRound(15.5) = 16, cvRound(22.5) = 22
    assert np.array_equal(half, bgr[np.minimum(np.arange(16) * 2, 30)][:, np.minimum(np.arange(22) * 2, 44)])
    d = np.array([0.0, 1.2345, np.nan, 70.0, -1.0, 0.0004, np.inf, 0.0005, 0.0015, 0.0025, 65.5354], np.float32)
    assert np.array_equal(oracle.depth_m_to_mm16(d), [1, 1234, 1, 65535, 1, 1, 1, 1, 2, 2, 65535])


@pytest.mark.gpu
def test_gpu_reproduces_frames_golden(golden):
    from rgbd_odometry_amd import DvoContext
    mg = _mg()
    for name, seed, rows, cols, nl, fs in mg.CASES:
        with DvoContext(1) as ctx:
            ctx.set_intrinsics(*mg.K)
            ctx.frames_upload_cameras([golden[f"{name}_bgr"]], [golden[f"{name}_depth_m"]], n_levels=nl, first_shift=fs)
            ctx.frames_as_ref(0, 0, 1)
            ctx.frames_as_now(0, 0, 1)
            for l in range(nl):
                grey, depth, edge, ne = ctx.frame_level(0, l)
                assert np.array_equal(grey, golden[f"{name}_L{l}_grey"])
                assert np.array_equal(depth, golden[f"{name}_L{l}_depth16"].astype(np.float32))
                assert np.array_equal(edge, golden[f"{name}_L{l}_edge"])
                for got, key in zip(ctx.get_now_level(l), ("dt", "gx", "gy")):
                    assert np.array_equal(got, golden[f"{name}_L{l}_{key}"]), (name, l, key)
                assert np.array_equal(ctx.get_ref_level(l), golden[f"{name}_L{l}_xyz"])


def test_undistort_restatement(oracle):
    """cv::undistort restated (OpenCV 2.4: stripes, adjugate inverse, running sums, 5-bit fixed-point map, BilinearTab_i / _f):
    zero distortion is the identity; with distortion the result equals a float bilinear interpolation (scipy) taken at the
    map's own 1/32-pixel coordinates, to within the final rounding"""
    from scipy.ndimage import map_coordinates
    rows, cols = 96, 128
    ys, xs = np.mgrid[0:rows, 0:cols].astype(float)
    img = np.stack([128 + 100 * np.sin(xs / 9) * np.cos(ys / 7), xs + ys, 128 + 60 * np.cos(xs / 13 + ys / 5)], -1).clip(0, 255).astype(np.uint8)
    d16 = (1000 + 500 * np.sin(xs / 20) + 300 * np.cos(ys / 15)).astype(np.uint16)
    K = (120.0, 118.0, 63.5, 47.5)
    assert np.array_equal(oracle.undistort_bgr8(img, K, (0, 0, 0, 0, 0)), img)
    assert np.array_equal(oracle.undistort_u16(d16, K, (0, 0, 0, 0, 0)), d16)
    D = (0.1, -0.05, 0.001, -0.002, 0.01)
    x = (xs - K[2]) / K[0]; y = (ys - K[3]) / K[1]; r2 = x * x + y * y
    kr = 1 + ((D[4] * r2 + D[1]) * r2 + D[0]) * r2
    u = K[0] * (x * kr + D[2] * 2 * x * y + D[3] * (r2 + 2 * x * x)) + K[2]
    v = K[1] * (y * kr + D[2] * (r2 + 2 * y * y) + D[3] * 2 * x * y) + K[3]
    uq, vq = np.round(u * 32) / 32, np.round(v * 32) / 32
    inside = (uq >= 0) & (uq <= cols - 1) & (vq >= 0) & (vq <= rows - 1)
    out = oracle.undistort_bgr8(img, K, D).astype(float)
    ref = np.stack([map_coordinates(img[..., c].astype(float), [vq, uq], order=1, mode="constant", cval=0) for c in range(3)], -1)
    assert inside.mean() > 0.9 and np.abs(out - ref)[inside].max() <= 0.5 + 1e-3
    o16 = oracle.undistort_u16(d16, K, D).astype(float)
    r16 = map_coordinates(d16.astype(float), [vq, uq], order=1, mode="constant", cval=0)
    assert np.abs(o16 - r16)[inside].max() <= 0.5 + 1e-2
    # where the map leaves the source the constant border 0 comes out (strong pincushion: the corners sample far outside)
    far = oracle.undistort_bgr8(np.full_like(img, 200), K, (5.0, 0, 0, 0, 0))
    assert far[0, 0].max() == 0 and far[-1, -1].max() == 0 and far[rows // 2, cols // 2].min() == 200


# ---- pinning row f1 against the real reference (tools/ref_dump/frames_dump.cpp; round 6) ----------------------------------------
REFERENCE_FRAMES = os.path.join(os.path.dirname(GOLDEN), "reference_frames_golden.npz")


def _frames_tools():
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    mods = []
    for name in ("to_npz", "export_frames"):
        spec = importlib.util.spec_from_file_location("ref_dump_" + name, os.path.join(root, "tools", "ref_dump", name + ".py"))
        m = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(m)
        mods.append(m)
    return mods


def oracle_frame_levels(oracle, export, name):
    """what frames_dump.cpp computes for an exported frame, by the oracle: {key: array} in the converter's layout"""
    out = {}
    _, seed, rows, cols, levels, fs = [f for f in export.FRAMES if f[0] == name][0]
    gl = export.grey_levels(oracle, seed, rows, cols, levels, fs)
    out[f"{name}_levels"] = np.array(len(gl), np.int32)
    for l, g in enumerate(gl):
        dt, gx, gy, edge = oracle.now_level_from_grey(g)
        out[f"{name}_L{l}_shape"] = np.array(g.shape, np.int32)
        out[f"{name}_L{l}_edge"] = np.asarray(edge, np.int32).ravel()
        for k, a in (("dt", dt), ("gx", gx), ("gy", gy)):
            out[f"{name}_L{l}_{k}"] = np.asarray(a, np.float32).ravel()
    return out


def compare_with_reference_frames(ref, got, name):
    """edge maps equal; distance transform and gradients within one float ulp (bit-equal is reported, not required: OpenCV's
    MASK_PRECISE transform and cv::normalize are float code whose last bit this repository could only restate)"""
    n = int(ref[f"{name}_levels"])
    assert n == int(got[f"{name}_levels"])
    for l in range(n):
        assert np.array_equal(ref[f"{name}_L{l}_shape"], got[f"{name}_L{l}_shape"]), (name, l)
        assert np.array_equal(ref[f"{name}_L{l}_edge"] != 0, got[f"{name}_L{l}_edge"] != 0), (name, l, "edge map")
        for k in ("dt", "gx", "gy"):
            a, b = np.asarray(ref[f"{name}_L{l}_{k}"], np.float32), np.asarray(got[f"{name}_L{l}_{k}"], np.float32)
            ulp = np.abs(a.view(np.int32).astype(np.int64) - b.view(np.int32).astype(np.int64))
            same_sign_or_zero = (np.signbit(a) == np.signbit(b)) | ((a == 0) & (b == 0))
            assert np.all(same_sign_or_zero) and int(ulp[(a != 0) | (b != 0)].max(initial=0)) <= 1, (name, l, k, int(ulp.max()))


def test_frames_dump_text_format_round_trips(oracle, tmp_path):
    """the text format tools/ref_dump/frames_dump.cpp writes (run-length edge map, C99 hex floats) -> to_npz.parse_frames -> the arrays
    the reference tests read: written here from the ORACLE's outputs exactly as the driver's fprintf calls would"""
    to_npz, export = _frames_tools()
    name = export.FRAMES[0][0]
    want = oracle_frame_levels(oracle, export, name)
    lines = ["frame %s %d" % (name, int(want[f"{name}_levels"]))]
    for l in range(int(want[f"{name}_levels"])):
        rows, cols = (int(v) for v in want[f"{name}_L{l}_shape"])
        lines.append("level %d %d %d" % (l, rows, cols))
        e = want[f"{name}_L{l}_edge"]
        runs, i = [], 0
        while i < e.size:
            j = i
            while j < e.size and e[j] == e[i]:
                j += 1
            runs.append("%d %d" % (int(e[i]), j - i))
            i = j
        lines.append("edge " + " ".join(runs))
        for k in ("dt", "gx", "gy"):
            lines.append(k + " " + " ".join(float(x).hex() for x in want[f"{name}_L{l}_{k}"]))
    f = tmp_path / "frames.txt"
    f.write_text("\n".join(lines) + "\n")
    got = to_npz.parse_frames(str(f))
    assert set(got) == set(want)
    for k in want:
        assert np.array_equal(np.asarray(want[k]), got[k]), k
    compare_with_reference_frames(got, want, name)
    # and the exporter writes what the driver reads: column-major uint8 levels + their shapes
    export_dir = tmp_path / "inputs"
    import sys
    argv = sys.argv
    sys.argv = ["export_frames.py", str(export_dir)]
    try:
        export.main()
    finally:
        sys.argv = argv
    meta = (export_dir / name / "meta.txt").read_text().split()
    assert int(meta[0]) == int(want[f"{name}_levels"])
    g0 = np.fromfile(export_dir / name / "grey_0.u8", np.uint8)
    assert g0.size == int(meta[1]) * int(meta[2])


@pytest.mark.skipif(not os.path.exists(REFERENCE_FRAMES), reason="tests/golden/reference_frames_golden.npz absent: nobody has run tools/ref_dump/frames_dump.cpp against the real reference yet (row f1 UNPINNED)")
def test_matches_reference_vectors(oracle):
    _, export = _frames_tools()
    ref = np.load(REFERENCE_FRAMES)
    for name, *_ in export.FRAMES:
        compare_with_reference_frames(ref, oracle_frame_levels(oracle, export, name), name)
