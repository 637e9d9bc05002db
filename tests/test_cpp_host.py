"""The C++ class mirror (include/dvo_amd.hpp: SolveDVO / PyramidalStorageStruct) through its demo binary."""
import os
import subprocess

import numpy as np
import pytest

import oracle_lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEMO = os.path.join(ROOT, "rgbd_odometry_amd", "lib", "solve_dvo_demo")


def test_cpp_mirror_compiles_and_declares_reference_surface():
    hdr = open(os.path.join(ROOT, "include", "dvo_amd.hpp")).read()
    for name in ("class SolveDVO", "class PyramidalStorageStruct", "class RGBDOdometry", "void runIterations(",
                 "void setCameraMatrix(", "void addLevel(", "void getLevel(", "void clearPyramid(", "void eventLoop(",
                 "iterationsConfig", "class GOP", "void pushAsKeyFrame(", "void pushAsOrdinaryFrame(",
                 "void updateMostRecentToKeyFrame(", "void setRcvdFrameAsRefFrame(", "void setPrevFrameAsRefFrame(",
                 "void setRcvdFrameAsNowFrame(", "void preProcessRefFrame(", "bool loadFromFile(", "static void printPose("):
        assert name in hdr, name
    assert os.path.exists(DEMO), "run __graft_entry__.build()"


@pytest.mark.gpu
def test_cpp_solve_dvo_matches_oracle(oracle):
    from rgbd_odometry_amd import SynthScene
    W, H, nl, it, seed = 320, 240, 4, 10, 2
    out = subprocess.run([DEMO, str(W), str(H), str(nl), str(it), str(seed)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    sc = SynthScene(W, H, nl, seed)
    lv = oracle_lib.scene_levels(sc, oracle)
    ref = oracle.align_pyramid([it] * nl, lv, sc.intrinsics, np.eye(3), np.zeros(3))
    lines = out.stdout.strip().splitlines()
    for ln in lines[:nl]:
        tok = ln.split()
        l, best, ratio = int(tok[1]), int(tok[3]), float(tok[5])
        energies = np.array([float(x) for x in tok[7:]], np.float32)
        rep = ref["levels"][l]
        assert best == rep["best_idx"] and np.float32(ratio) == np.float32(rep["visible_ratio"])
        assert np.array_equal(energies, rep["energy"])
    pose = np.array([float(x) for x in lines[nl].split()[1:]])
    fused = np.array([float(x) for x in lines[nl + 1].split()[1:]])
    R = pose[:9].reshape(3, 3, order="F")
    assert oracle_lib.rot_angle(ref["R"], R) <= 1e-5 and np.linalg.norm(ref["t"] - pose[9:]) <= 1e-4
    # level-by-level runIterations == the fused schedule (the matrix <-> quaternion hand-over between
    # separate calls costs ~1e-18)
    assert np.abs(pose - fused).max() < 1e-14


PHOTO_DEMO = os.path.join(ROOT, "rgbd_odometry_amd", "lib", "rgbd_odometry_demo")


@pytest.mark.gpu
@pytest.mark.parametrize("fixed", [0, 1])
def test_cpp_rgbd_odometry_matches_oracle(tmp_path, oracle, fixed):
    """dvo_amd::RGBDOdometry (eventLoop body, RGBDOdometry.cpp:138-207) on three frames: frame 0 is the reference AND the first
    now frame, the estimate is carried from frame to frame; T and the published pose equal the oracle chain"""
    import frame_gen
    K = (525.0, 525.0, 319.5, 239.5)
    rows, cols, n = 480, 640, 3
    frames = []
    for i, shift in enumerate([(0, 0), (2, -3), (3, -1)]):
        bgr, depth = frame_gen.camera_frame(21, rows, cols, shift=shift)
        d16 = np.clip(np.nan_to_num(np.round(depth * 1000.0), nan=0.0, posinf=65535, neginf=0), 1, 65535).astype(np.uint16)
        bgr.tofile(str(tmp_path / ("bgr_%04d.bin" % i)))
        d16.tofile(str(tmp_path / ("depth_%04d.bin" % i)))
        frames.append([(oracle.bgr2gray(oracle.resize_nn(bgr, 0.5 ** l)), oracle.resize_nn(d16, 0.5 ** l)) for l in range(4)])
    run = subprocess.run([PHOTO_DEMO, "photo", str(tmp_path), str(n), str(rows), str(cols)] + [repr(k) for k in K] + [str(fixed)],
                         capture_output=True, text=True, timeout=300)
    assert run.returncode == 0, run.stderr
    lines = run.stdout.strip().splitlines()
    T = np.eye(4)
    for i in range(n):
        T, rep = oracle.photo_track(frames[0], frames[i], K, T0=T, fixed=bool(fixed))
        got_T = np.array([float(x) for x in lines[2 * i + 1].split()[1:]]).reshape(4, 4)
        assert np.abs(got_T - T).max() <= 1e-9 * max(1.0, np.abs(T).max()), (i, np.abs(got_T - T).max())
        pose = np.array([float(x) for x in lines[2 * i].split()[1:]])
        np.testing.assert_allclose(pose[:3], 1000 * T[:3, 3], rtol=1e-8, atol=1e-9)            # :185-187
        from scipy.spatial.transform import Rotation
        q = Rotation.from_matrix(T[:3, :3]).as_quat()
        assert min(np.abs(pose[3:] - q).max(), np.abs(pose[3:] + q).max()) <= 1e-6
    assert lines[-1].startswith("frames 3")


@pytest.mark.gpu
def test_cpp_casual_test_function(tmp_path, oracle):
    """SolveDVO::casualTestFunction (SolveDVO.cpp:2377-2442): two XML frame files, level 0, 100 iterations from the identity"""
    import frame_gen
    import frame_io
    K = tuple(np.float32(k) for k in (262.5, 262.5, 159.75, 119.75))
    pyrs = []
    for i, shift in enumerate([(0, 0), (1, -2)]):
        bgr, depth = frame_gen.camera_frame(31, 480, 640, shift=shift)
        pyr = oracle.build_pyramid(bgr, depth, 2, 1)
        frame_io.write_frame_xml(str(tmp_path / ("f%d.xml" % i)), pyr)
        pyrs.append(pyr)
    run = subprocess.run([PHOTO_DEMO, "casual", str(tmp_path / "f0.xml"), str(tmp_path / "f1.xml"), "2"] + [repr(float(k)) for k in K] +
                         ["0", "100"], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0, run.stderr
    got = np.array([float(x) for x in run.stdout.split()], np.float32)
    g, d = pyrs[0][0]
    xyz, uv, _ = oracle.ref_level_from_grey(0, g, d, K)
    dt, gx, gy = oracle.now_level_from_grey(pyrs[1][0][0])[:3]
    ref = oracle.run_iterations(0, 100, xyz, dt, gx, gy, g.shape[0], g.shape[1], K, np.eye(3), np.zeros(3))
    assert got.shape == (100,) and np.array_equal(got, ref["energy"])
