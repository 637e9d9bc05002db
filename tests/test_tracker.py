"""Row f3: key-frame policy + GOP pose chain + pose file.  CPU: the numpy oracle (oracle/tracker_oracle.py) against
its definitions; GPU: the C++ mirror's file-replay loop (examples/track_demo.cpp on include/dvo_amd.hpp) against the
oracle chain on the same OpenCV-XML frame files."""
import os
import subprocess
import sys

import numpy as np
import pytest

import frame_gen
import frame_io

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import tracker_oracle as T  # noqa: E402

DEMO = os.path.join(ROOT, "rgbd_odometry_amd", "lib", "track_demo")


def test_quaternion_matches_scipy_up_to_sign():
    from scipy.spatial.transform import Rotation
    rng = np.random.default_rng(0)
    for _ in range(50):
        R = Rotation.from_rotvec(rng.normal(0, 1.5, 3)).as_matrix()
        q, s = T.quaternion_from_matrix(R), Rotation.from_matrix(R).as_quat()
        assert min(np.abs(q - s).max(), np.abs(q + s).max()) < 1e-12
    assert np.allclose(T.quaternion_from_matrix(np.eye(3)), [0, 0, 0, 1])
    assert np.allclose(np.abs(T.quaternion_from_matrix(np.diag([1.0, -1.0, -1.0]))), [1, 0, 0, 0])   # trace <= 0 branch


def test_gop_chain():
    from scipy.spatial.transform import Rotation
    g = T.GOP()
    R1, t1 = Rotation.from_rotvec([0.1, 0, 0]).as_matrix(), np.array([1.0, 0, 0])
    g.push_key(0, 1, np.eye(3), np.zeros(3))
    g.push_ordinary(1, R1, t1)
    assert np.allclose(g.elems[1]["R"], R1) and np.allclose(g.elems[1]["t"], t1) and not g.elems[1]["key"]
    g.update_most_recent_to_key(5)
    assert g.elems[1]["key"] and g.elems[1]["reason"] == 5
    g.push_ordinary(2, R1, t1)                               # now relative to frame 1
    assert np.allclose(g.elems[2]["R"], R1 @ R1) and np.allclose(g.elems[2]["t"], t1 + R1 @ t1)


def test_key_frame_policy_with_stub_aligner():
    """(nFrame - lastRefFrame) == 5 -> re-reference on n-1, estimate thrown away and re-run from identity"""
    calls = []

    def align(ref, now, R0, t0):
        calls.append((ref, now, bool(np.allclose(R0, np.eye(3)) and not t0.any())))
        return np.eye(3), np.array([float(now - ref), 0.0, 0.0])      # "now is (now-ref) metres from ref"

    gop, lines = T.track(13, align)
    keys = [(e["frame"], e["reason"]) for e in gop.elems if e["key"]]
    assert keys == [(0, 1), (4, 5), (8, 5)]
    assert [c[:2] for c in calls] == [(0, 1), (0, 2), (0, 3), (0, 4), (0, 5), (4, 5), (4, 6), (4, 7), (4, 8), (4, 9), (8, 9),
                                     (8, 10), (8, 11), (8, 12)]
    assert [c[2] for c in calls][:6] == [True, False, False, False, False, True]      # warm start, reset on re-reference
    assert np.allclose([e["t"][0] for e in gop.elems], np.arange(13))                 # global chain is consistent
    assert len(lines) == 12 and lines[0].split() == ["0", "0", "0", "1", "1", "0", "0"]


def test_adaptive_key_frame_exits_with_stub_aligner():
    """the three exits the reference has commented out (:2129-2152): a Laplacian scale above its threshold (reason 2), too few
    visible points (3), too few points (4); OR-ed with the every-5-frames rule (5); never a switch when n-1 is the reference already"""
    seen = {}

    def align(ref, now, R0, t0):
        info = dict(final_eps=np.full(100, 1.0, np.float32), visible_ratio=0.95, n=100)
        if now == 3 and ref == 0:
            info["final_eps"] = np.full(100, 4.0, np.float32)         # b_cap = 4 > 3
        if now == 6 and ref != 5:
            info["visible_ratio"] = 0.5
        if now == 9 and ref != 8:
            info["n"] = 10
        seen[(ref, now)] = True
        return np.eye(3), np.array([float(now - ref), 0.0, 0.0]), info

    gop, lines = T.track(12, align, adaptive=dict(laplacian_b=3.0, visible_ratio=0.8, min_points=50))
    keys = [(e["frame"], e["reason"]) for e in gop.elems if e["key"]]
    assert keys == [(0, 1), (2, 2), (5, 3), (8, 4)], keys
    assert (2, 3) in seen and (5, 6) in seen and (8, 9) in seen            # re-runs against the new reference
    assert np.allclose([e["t"][0] for e in gop.elems], np.arange(12))
    assert T.laplacian_b(np.array([1, 2, 3], np.float32)) == np.float32(2.0) and T.laplacian_b([]) == 0


def _sequence(n, rows, cols, levels, first_shift, oracle):
    frames = []
    for i in range(n):
        bgr, depth = frame_gen.camera_frame(77, rows, cols, shift=(i // 2, -i), holes=True)
        frames.append(oracle.build_pyramid(bgr, depth, levels, first_shift))
    return frames


def test_frame_xml_layout(tmp_path, oracle):
    pyr = _sequence(1, 48, 64, 2, 0, oracle)[0]
    p = tmp_path / "framemono_0000.xml"
    frame_io.write_frame_xml(str(p), pyr)
    s = p.read_text()
    assert s.startswith('<?xml version="1.0"?>\n<opencv_storage>') and '<mono_1 type_id="opencv-matrix">' in s
    assert "<dt>u</dt>" in s and "<dt>w</dt>" in s and s.rstrip().endswith("</opencv_storage>")
    body = s[s.index("<data>", s.index("<depth_0 ")) + 6:s.index("</data>", s.index("<depth_0 "))]
    assert np.array_equal(np.array(body.split(), dtype=np.int64).reshape(48, 64), pyr[0][1])


@pytest.mark.gpu
def test_cpp_file_replay_matches_oracle_chain(tmp_path, oracle):
    n, rows, cols, nl, it = 12, 240, 320, 3, 8
    K = tuple(np.float32(k) for k in (262.5, 262.5, 159.75, 119.75))
    frames = _sequence(n, rows, cols, nl, 0, oracle)
    for i, pyr in enumerate(frames):
        frame_io.write_frame_xml(str(tmp_path / ("framemono_%04d.xml" % (3 + 2 * i))), pyr)      # START 3, SKIP 2

    cache = {}

    def level_inputs(ref, now):
        if ("r", ref) not in cache:
            cache[("r", ref)] = [oracle.ref_level_from_grey(l, g, d, K) for l, (g, d) in enumerate(frames[ref])]
        if ("n", now) not in cache:
            cache[("n", now)] = [oracle.now_level_from_grey(g) for g, _ in frames[now]]
        return [dict(xyz=r[0], uv=r[1], dt=m[0], gx=m[1], gy=m[2], rows=g.shape[0], cols=g.shape[1])
                for r, m, (g, _) in zip(cache[("r", ref)], cache[("n", now)], frames[now])]

    def align(ref, now, R0, t0):
        r = oracle.align_pyramid([it] * nl, level_inputs(ref, now), K, R0, t0)
        return np.array(r["R"]), np.array(r["t"])

    gop, want = T.track(n, align)
    out = tmp_path / "estPoses.txt"
    run = subprocess.run([DEMO, str(tmp_path), "3", str(3 + 2 * (n - 1)), "2", str(nl)] + [repr(float(k)) for k in K] +
                         [str(it), str(out)], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0, run.stderr
    assert "keyframes: 0(reason 1) 4(reason 5) 8(reason 5)" in run.stdout, run.stdout
    got = out.read_text().strip().splitlines()
    assert len(got) == len(want) == n - 1
    G = np.array([[float(x) for x in ln.split()] for ln in got])
    W = np.array([[float(x) for x in ln.split()] for ln in want])
    assert np.abs(G - W).max() <= 2e-5                       # 6 significant digits in the file; poses agree to ~1e-15
    assert np.abs(W[-1, 4:]).max() > 1e-3                    # the sequence really moves


@pytest.mark.gpu
def test_cpp_file_replay_with_adaptive_key_frames(tmp_path, oracle):
    """the same replay with the reference's adaptive exits switched on (SolveDVO.cpp:2129-2152; dvo_amd::SolveDVO::adaptiveKeyFrames):
    the Laplacian scale of finalEpsilons -- a float sum in the list's order, bit-equal between the engine and the oracle -- against a
    threshold placed between the values the sequence produces, so that exits of reason 2 really happen; key frames and poses must
    match the oracle chain"""
    n, rows, cols, nl, it = 12, 240, 320, 3, 8
    K = tuple(np.float32(k) for k in (262.5, 262.5, 159.75, 119.75))
    frames = _sequence(n, rows, cols, nl, 0, oracle)
    for i, pyr in enumerate(frames):
        frame_io.write_frame_xml(str(tmp_path / ("framemono_%04d.xml" % i)), pyr)
    cache, b_seen = {}, []

    def level_inputs(ref, now):
        if ("r", ref) not in cache:
            cache[("r", ref)] = [oracle.ref_level_from_grey(l, g, d, K) for l, (g, d) in enumerate(frames[ref])]
        if ("n", now) not in cache:
            cache[("n", now)] = [oracle.now_level_from_grey(g) for g, _ in frames[now]]
        return [dict(xyz=r[0], uv=r[1], dt=m[0], gx=m[1], gy=m[2], rows=g.shape[0], cols=g.shape[1])
                for r, m, (g, _) in zip(cache[("r", ref)], cache[("n", now)], frames[now])]

    def align(ref, now, R0, t0):
        r = oracle.align_pyramid([it] * nl, level_inputs(ref, now), K, R0, t0)
        last = r["levels"][r["last_level"]]
        info = dict(final_eps=last["final_eps"], visible_ratio=last["visible_ratio"], n=len(last["final_eps"]))
        b_seen.append(float(T.laplacian_b(info["final_eps"])))
        return np.array(r["R"]), np.array(r["t"]), info

    T.track(n, align, adaptive=dict(laplacian_b=1e9, visible_ratio=0.0, min_points=0))      # first pass: the values this sequence has
    b_thresh = float(np.float32(np.sort(b_seen)[len(b_seen) * 2 // 3]))                     # a third of the frames exceed it
    gop, want = T.track(n, align, adaptive=dict(laplacian_b=b_thresh, visible_ratio=0.3, min_points=50))
    keys = " ".join("%d(reason %d)" % (e["frame"], e["reason"]) for e in gop.elems if e["key"])
    assert "reason 2" in keys, (keys, b_thresh, b_seen)
    out = tmp_path / "estPoses.txt"
    run = subprocess.run([DEMO, str(tmp_path), "0", str(n - 1), "1", str(nl)] + [repr(float(k)) for k in K] +
                         [str(it), str(out), repr(b_thresh), "0.3", "50"], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0, run.stderr
    assert "keyframes: " + keys in run.stdout, (run.stdout, keys)
    got = out.read_text().strip().splitlines()
    G = np.array([[float(x) for x in ln.split()] for ln in got])
    W = np.array([[float(x) for x in ln.split()] for ln in want])
    assert G.shape == W.shape and np.abs(G - W).max() <= 2e-5
