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


# ---- reference-held vectors (tools/ref_dump): present only once a maintainer has run the REAL reference on the exported inputs ----------
REFERENCE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_golden.npz")
_REF_CASES = ["refdefault_s0", "refdefault_s1", "refdefault_s2", "c2_s0", "c2_s1", "c2_s2"]


def _ref_dump_tools():
    import importlib.util
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "ref_dump", "to_npz.py")
    spec = importlib.util.spec_from_file_location("ref_dump_to_npz", path)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _compare_with_reference(ref, got, case, nl):
    """what tools/ref_dump/README.md says can be demanded: pose within the north star's tolerance, best index and visible ratio equal,
    energies within 1e-5 relative (the reference's float32 norm in its compiler's order against the oracle's double sum)"""
    for l in range(nl):
        k = f"{case}_L{l}_energy"
        if k not in ref:
            continue
        e_ref, e = ref[k], got[k]
        assert len(e_ref) == len(e), (case, l)
        np.testing.assert_allclose(e, e_ref, rtol=1e-5, atol=0, err_msg=k)
        assert int(ref[f"{case}_L{l}_best"]) == int(got[f"{case}_L{l}_best"]), (case, l)
        assert float(ref[f"{case}_L{l}_ratio"]) == float(got[f"{case}_L{l}_ratio"]), (case, l)
    assert rot_angle(ref[case + "_R"], got[case + "_R"]) <= 1e-5
    assert np.linalg.norm(ref[case + "_t"] - got[case + "_t"]) <= 1e-4
    n = len(ref[case + "_final_eps_head"])
    np.testing.assert_allclose(got[case + "_final_reproj_head"][:n], ref[case + "_final_reproj_head"], atol=1e-4, rtol=0)


def test_ref_dump_text_format_round_trips(golden, tmp_path):
    """the text format tools/ref_dump/ref_dump.cpp writes (C99 hex floats) -> to_npz.parse -> the arrays test_golden reads: written here
    from the ORACLE's committed numbers exactly as the driver's fprintf calls would, so the plumbing is checked without the reference"""
    m = _ref_dump_tools()
    mg = _cases()
    lines = []
    for name, W, H, nl, it, seeds in mg.CASES:
        for seed in seeds:
            case = f"{name}_s{seed}"
            lines.append("case %s %d" % (case, nl))
            for l in range(nl - 1, -1, -1):
                e = golden[f"{case}_L{l}_energy"]
                lines.append("level %d %d %d %s" % (l, len(e), int(golden[f"{case}_L{l}_best"]), float(golden[f"{case}_L{l}_ratio"]).hex()))
                lines.append(" ".join(float(x).hex() for x in e) + " ")
                eps, rep = golden[case + "_final_eps_head"], golden[case + "_final_reproj_head"]
                lines.append("final %d" % len(eps))
                lines.append(" ".join(float(x).hex() for x in eps) + " ")
                lines.append(" ".join(float(x).hex() for x in rep.reshape(-1)) + " ")
            R, t = golden[case + "_R"], golden[case + "_t"]
            lines.append("pose " + " ".join(float(x).hex() for x in list(R.reshape(-1, order="F")) + list(t)))
    f = tmp_path / "dump.txt"
    f.write_text("\n".join(lines) + "\n")
    got = m.parse(str(f))
    for k, v in got.items():
        assert np.array_equal(np.asarray(golden[k]), v, equal_nan=True), k
    for case in _REF_CASES:
        _compare_with_reference(got, golden, case, 4)


@pytest.mark.skipif(not os.path.exists(REFERENCE), reason="tests/golden/reference_golden.npz absent: nobody has run tools/ref_dump against the real reference yet (PARITY UNPINNED)")
@pytest.mark.parametrize("case", _REF_CASES)
def test_oracle_matches_reference_vectors(golden, case):
    _compare_with_reference(np.load(REFERENCE), golden, case, 4)


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(REFERENCE), reason="tests/golden/reference_golden.npz absent (tools/ref_dump)")
@pytest.mark.parametrize("case", _REF_CASES)
def test_gpu_matches_reference_vectors(case):
    from rgbd_odometry_amd import DvoContext, SynthScene
    from rgbd_odometry_amd.capi import DVO_FLAG_FINAL_OUTPUTS
    ref = np.load(REFERENCE)
    mg = _cases()
    W, H, nl, it, seed = {f"{n}_s{s}": (W, H, nl, it, s) for n, W, H, nl, it, seeds in mg.CASES for s in seeds}[case]
    sc = SynthScene(W, H, nl, seed)
    with DvoContext(1) as ctx:
        ctx.set_intrinsics(*sc.intrinsics)
        for l, L in enumerate(sc.levels):
            ctx.set_ref_level_from_images(l, L.ref_edge, L.ref_depth, L.rows, L.cols)
            ctx.set_now_level(l, L.now_dt, L.now_gx, L.now_gy, L.rows, L.cols)
        R, t = ctx.align_batch([it] * nl, np.eye(3)[None], np.zeros((1, 3)), flags=DVO_FLAG_FINAL_OUTPUTS)
        got = {case + "_R": R[0], case + "_t": t[0]}
        for l in range(nl):
            e, b, ratio = ctx.level_report(0, l, it)
            got[f"{case}_L{l}_energy"], got[f"{case}_L{l}_best"], got[f"{case}_L{l}_ratio"] = e, b, ratio
        n = len(ref[case + "_final_eps_head"])
        feps, frep = ctx.final_outputs(0, n)
        got[case + "_final_eps_head"], got[case + "_final_reproj_head"] = feps[:n], frep[:n]
    _compare_with_reference(ref, got, case, nl)
