"""Randomised stress parity: arbitrary inputs at the C-ABI boundary, HIP path vs oracle, bit for bit."""
import numpy as np
import pytest

import oracle_lib
from oracle_lib import rot_angle

pytestmark = pytest.mark.gpu


def _same(a, b):
    return np.array_equal(np.asarray(a), np.asarray(b), equal_nan=True)


def _random_case(rng, k):
    rows = int(rng.integers(5, 90))
    cols = int(rng.integers(5, 120))
    level = int(rng.integers(0, 4))
    n = int(rng.choice([1, 3, 63, 64, 65, 257, 700, 1500]))
    s = 2.0 ** (-level)
    fx, fy = (float(x) for x in rng.uniform(0.6, 1.6, 2) * cols / s)
    cx, cy = float(rng.uniform(0.3, 0.7) * cols / s), float(rng.uniform(0.3, 0.7) * rows / s)
    K = tuple(np.float32(v) for v in (fx, fy, cx, cy))
    # image contents: normal DT-like, plus adversarial variants
    kind = k % 4
    if kind == 0:
        dt = rng.uniform(0, 255, rows * cols); gx = rng.normal(0, 3, rows * cols); gy = rng.normal(0, 3, rows * cols)
    elif kind == 1:
        dt = rng.normal(0, 1e3, rows * cols); gx = rng.normal(0, 1e4, rows * cols); gy = rng.normal(0, 1e-4, rows * cols)
    elif kind == 2:
        dt = np.abs(rng.normal(0, 1e-18, rows * cols)); gx = rng.normal(0, 1e18, rows * cols); gy = np.zeros(rows * cols)
    else:
        dt = rng.integers(0, 4, rows * cols).astype(float); gx = rng.integers(-2, 3, rows * cols).astype(float); gy = gx[::-1].copy()
    dt, gx, gy = (a.astype(np.float32) for a in (dt, gx, gy))
    # points: mostly in front of the camera and inside the view, some behind / at z = 0 / far outside
    z = rng.uniform(0.3, 6.0, n)
    u, v = rng.uniform(-0.2, 1.2, n) * cols, rng.uniform(-0.2, 1.2, n) * rows
    xyz = np.stack([(u - cx * s) / (fx * s) * z, (v - cy * s) / (fy * s) * z, z], axis=1)
    bad = rng.random(n)
    xyz[bad < 0.03, 2] *= -1.0
    xyz[(bad >= 0.03) & (bad < 0.05), 2] = 0.0
    xyz[(bad >= 0.05) & (bad < 0.06)] *= 1e20
    return rows, cols, level, K, dt, gx, gy, xyz.astype(np.float32)


@pytest.mark.parametrize("seed", range(6))
def test_random_inputs_bit_parity(oracle, seed):
    from rgbd_odometry_amd import DvoContext
    rng = np.random.default_rng(1234 + seed)
    ctx = DvoContext(1)
    try:
        for k in range(8):
            rows, cols, level, K, dt, gx, gy, xyz = _random_case(rng, k)
            ctx.set_intrinsics(*[float(x) for x in K])
            ctx.set_now_level(level, dt, gx, gy, rows, cols)
            ctx.set_ref_level(level, xyz)
            for scale in (0.0, 0.02, 0.7, 2.5):
                psi = rng.standard_normal(6) * scale
                nrm = np.linalg.norm(psi[3:])
                if nrm > 3.0:
                    psi[3:] *= 3.0 / nrm
                R, t = oracle.se3_exp(psi)
                ref = oracle.eval_points(level, xyz, dt, gx, gy, rows, cols, K, R, t)
                got = ctx.eval_points(level, R, t)
                assert np.array_equal(ref["visible"], got["visible"]), (seed, k, scale)
                for key in ("reproj", "eps", "w", "J"):
                    assert _same(ref[key], got[key]), (seed, k, scale, key)
            # a short optimisation from a random start, energies must stay bit-equal while finite
            R0, t0 = oracle.se3_exp(rng.standard_normal(6) * 0.01)
            ref = oracle.run_iterations(level, 7, xyz, dt, gx, gy, rows, cols, K, R0, t0)
            got = ctx.run_iterations(level, 7, R0, t0)
            assert _same(ref["energy"], got["energy"]), (seed, k, ref["energy"], got["energy"])
            assert ref["best_idx"] == got["best_idx"], (seed, k)
            if np.all(np.isfinite(ref["R"])) and np.all(np.isfinite(got["R"])):
                assert rot_angle(ref["R"], got["R"]) <= 1e-5 and np.linalg.norm(ref["t"] - got["t"]) <= 1e-4, (seed, k)
            assert _same(ref["final_eps"], got["final_eps"]) and _same(ref["final_reproj"], got["final_reproj"]), (seed, k)
    finally:
        ctx.close()
