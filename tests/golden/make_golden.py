#!/usr/bin/env python3
"""Generates tests/golden/oracle_golden.npz from the CPU oracle on the seeded synthetic scenes.

The reference ships no golden vectors and cannot be built here (PARITY UNPINNED, see oracle/dvo_oracle.h), so
these are the ORACLE's own outputs, committed so that (a) drift of the oracle or of the scene generator is
caught on CPU and (b) the GPU path is checked against fixed numbers as well as against a live oracle run.

    python tests/golden/make_golden.py          # rewrites oracle_golden.npz
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

CASES = [   # name, W, H, levels, iterations per level, seeds
    ("refdefault", 320, 240, 4, 50, (0, 1, 2)),     # the reference's shipped configuration (SolveDVO.cpp:30-33)
    ("c2", 640, 480, 4, 10, (0, 1, 2)),             # BASELINE.json configs[1]
]
N_DUMP = 256


def scene_digest(sc):
    h = hashlib.sha256()
    for L in sc.levels:
        for name in ("ref_edge", "ref_depth", "now_dt", "now_gx", "now_gy"):
            h.update(np.ascontiguousarray(getattr(L, name)).tobytes())
    return h.hexdigest()


def build():
    import oracle_lib
    from rgbd_odometry_amd import SynthScene
    oracle = oracle_lib.load()
    out = {}
    for name, W, H, nl, it, seeds in CASES:
        for seed in seeds:
            key = f"{name}_s{seed}"
            sc = SynthScene(W, H, nl, seed)
            lv = oracle_lib.scene_levels(sc, oracle)
            out[key + "_digest"] = np.array(scene_digest(sc))
            out[key + "_N"] = np.array([len(L["xyz"]) for L in lv], np.int32)
            r = oracle.align_pyramid([it] * nl, lv, sc.intrinsics, np.eye(3), np.zeros(3))
            out[key + "_R"], out[key + "_t"] = np.array(r["R"]), r["t"]
            for l, rep in r["levels"].items():
                out[f"{key}_L{l}_energy"] = rep["energy"]
                out[f"{key}_L{l}_best"] = np.array(rep["best_idx"], np.int32)
                out[f"{key}_L{l}_ratio"] = np.array(rep["visible_ratio"], np.float32)
            last = r["levels"][r["last_level"]]
            out[key + "_final_eps_head"] = last["final_eps"][:N_DUMP]
            out[key + "_final_reproj_head"] = last["final_reproj"][:N_DUMP]
            # per-point dump at a fixed non-trivial pose, levels 0 and 2
            P = oracle.se3_exp(np.array([0.012, -0.007, 0.009, 0.006, -0.011, 0.004]))
            out[key + "_dump_R"], out[key + "_dump_t"] = np.array(P[0]), P[1]
            for l in (0, 2):
                L = lv[l]
                d = oracle.eval_points(l, L["xyz"], L["dt"], L["gx"], L["gy"], L["rows"], L["cols"], sc.intrinsics, *P)
                for k in ("reproj", "J", "eps", "w", "visible"):
                    out[f"{key}_dump_L{l}_{k}"] = d[k][:N_DUMP]
            # first iterations of the coarsest level, step by step
            L = lv[nl - 1]
            tr = oracle.run_iterations(nl - 1, 3, L["xyz"], L["dt"], L["gx"], L["gy"], L["rows"], L["cols"],
                                       sc.intrinsics, np.eye(3), np.zeros(3), trace=True)["trace"]
            out[key + "_trace_g"] = np.array([x["g"] for x in tr])
            out[key + "_trace_psi"] = np.array([x["psi"] for x in tr])
            out[key + "_trace_sum_eps2"] = np.array([x["sum_eps2"] for x in tr])
            out[key + "_trace_nvis"] = np.array([x["n_visible"] for x in tr], np.int32)
    return out


if __name__ == "__main__":
    data = build()
    path = os.path.join(HERE, "oracle_golden.npz")
    np.savez_compressed(path, **data)
    print("wrote", path, os.path.getsize(path), "bytes,", len(data), "arrays")
