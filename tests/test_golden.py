"""Golden vectors (tests/golden/oracle_golden.npz, made by tests/golden/make_golden.py).

CPU: the scene generator and the oracle still produce the committed numbers.
GPU: the HIP path reproduces them (float32 quantities bit-equal, pose within 1e-5 rad / 1e-4 m).
"""
import os

import numpy as np
import pytest

import oracle_lib
from oracle_lib import rot_angle

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_golden.npz")


@pytest.fixture(scope="module")
def golden():
    return np.load(GOLDEN)


def _cases():
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(os.path.dirname(GOLDEN), "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    return mg


def test_scene_generator_is_stable(golden):
    mg = _cases()
    from rgbd_odometry_amd import SynthScene
    for name, W, H, nl, it, seeds in mg.CASES:
        for seed in seeds:
            sc = SynthScene(W, H, nl, seed)
            assert mg.scene_digest(sc) == str(golden[f"{name}_s{seed}_digest"]), (name, seed)


def test_oracle_reproduces_golden(golden, oracle):
    mg = _cases()
    fresh = mg.build()
    assert set(fresh) == set(golden.files)
    for k in golden.files:
        a, b = golden[k], fresh[k]
        if a.dtype.kind in "US":
            assert str(a) == str(b), k
        elif a.dtype == np.float64:
            np.testing.assert_allclose(b, a, rtol=1e-12, atol=1e-13, err_msg=k)   # libm-level slack only
        else:
            assert np.array_equal(a, b, equal_nan=True), k


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["refdefault_s0", "refdefault_s1", "refdefault_s2", "c2_s0", "c2_s1", "c2_s2"])
def test_gpu_reproduces_golden(golden, oracle, case):
    from rgbd_odometry_amd import DvoContext, SynthScene
    from rgbd_odometry_amd.capi import DVO_FLAG_FINAL_OUTPUTS
    mg = _cases()
    spec = {f"{n}_s{s}": (W, H, nl, it, s) for n, W, H, nl, it, seeds in mg.CASES for s in seeds}[case]
    W, H, nl, it, seed = spec
    sc = SynthScene(W, H, nl, seed)
    ctx = DvoContext(1)
    try:
        ctx.set_intrinsics(*sc.intrinsics)
        for l, L in enumerate(sc.levels):
            xyz, _ = ctx.set_ref_level_from_images(l, L.ref_edge, L.ref_depth, L.rows, L.cols)   # GPU enlistRefEdgePts
            assert len(xyz) == golden[case + "_N"][l]
            ctx.set_now_level(l, L.now_dt, L.now_gx, L.now_gy, L.rows, L.cols)
        iters = [it] * nl
        R, t = ctx.align_batch(iters, np.eye(3)[None], np.zeros((1, 3)), flags=DVO_FLAG_FINAL_OUTPUTS)
        for l in range(nl):
            e, b, ratio = ctx.level_report(0, l, it)
            assert np.array_equal(e, golden[f"{case}_L{l}_energy"]), l
            assert b == int(golden[f"{case}_L{l}_best"]) and ratio == float(golden[f"{case}_L{l}_ratio"])
        assert rot_angle(golden[case + "_R"], R[0]) <= 1e-5
        assert np.linalg.norm(golden[case + "_t"] - t[0]) <= 1e-4
        feps, frep = ctx.final_outputs(0, int(golden[case + "_N"][0]))
        n = len(golden[case + "_final_eps_head"])
        assert np.array_equal(feps[:n], golden[case + "_final_eps_head"], equal_nan=True)
        assert np.array_equal(frep[:n], golden[case + "_final_reproj_head"], equal_nan=True)
        # per-point dump
        Rd, td = golden[case + "_dump_R"], golden[case + "_dump_t"]
        for l in (0, 2):
            d = ctx.eval_points(l, Rd, td)
            for k in ("reproj", "J", "eps", "w", "visible"):
                g = golden[f"{case}_dump_L{l}_{k}"]
                assert np.array_equal(d[k][:len(g)], g, equal_nan=True), (l, k)
        # first iterations of the coarsest level: accumulators of each trace step
        acc = ctx.accumulate(nl - 1, np.eye(3), np.zeros(3))
        np.testing.assert_allclose(acc[21:27], golden[case + "_trace_g"][0], rtol=1e-11,
                                   atol=1e-9 * np.abs(golden[case + "_trace_g"][0]).max())
        np.testing.assert_allclose(acc[27], golden[case + "_trace_sum_eps2"][0], rtol=1e-13)
        assert int(acc[28]) == int(golden[case + "_trace_nvis"][0])
    finally:
        ctx.close()
