#!/usr/bin/env python3
"""Generates tests/golden/frames_golden.npz: the oracle's outputs for rows f1/f2 (pyramid, Canny, distance
transform, point extraction) on two small seeded camera frames, inputs included.

The reference ships no vectors for these steps and OpenCV 2.4 is not available here (PARITY UNPINNED, see
oracle/dvo_oracle_frames.cpp), so these pin the ORACLE against drift and give the GPU path fixed numbers.

    python tests/golden/make_frames_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

CASES = [("a", 17, 96, 128, 3, 0), ("b", 23, 121, 163, 2, 1)]     # name, seed, rows, cols, levels, first_shift
K = (131.25, 131.25, 79.875, 59.875)                               # 525/4 ... : level-0 intrinsics of a 160x120 frame


def build(inputs=None):
    import frame_gen
    import oracle_lib
    o = oracle_lib.load()
    out = {}
    for name, seed, rows, cols, nl, fs in CASES:
        if inputs is None:
            bgr, depth = frame_gen.camera_frame(seed, rows, cols)
        else:
            bgr, depth = inputs[f"{name}_bgr"], inputs[f"{name}_depth_m"]
        out[f"{name}_bgr"], out[f"{name}_depth_m"] = bgr, depth
        for l, (g, d16) in enumerate(o.build_pyramid(bgr, depth, nl, fs)):
            out[f"{name}_L{l}_grey"], out[f"{name}_L{l}_depth16"] = g, d16
            edge, mag, cand = o.canny(g, stages=True)
            out[f"{name}_L{l}_edge"] = edge
            out[f"{name}_L{l}_cand"] = cand
            out[f"{name}_L{l}_mag_sum"] = np.array(mag.astype(np.int64).sum())
            dt, gx, gy, _ = o.now_level_from_grey(g)
            out[f"{name}_L{l}_dt"], out[f"{name}_L{l}_gx"], out[f"{name}_L{l}_gy"] = dt, gx, gy
            xyz, uv, _ = o.ref_level_from_grey(l, g, d16, tuple(np.float32(k) for k in K))
            out[f"{name}_L{l}_xyz"] = xyz
    return out


if __name__ == "__main__":
    data = build()
    path = os.path.join(HERE, "frames_golden.npz")
    np.savez_compressed(path, **data)
    print("wrote", path, os.path.getsize(path), "bytes,", len(data), "arrays")
