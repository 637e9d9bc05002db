#!/usr/bin/env python3
"""Exports the grey pyramid levels of the committed synthetic camera frames (rgbd_odometry_amd/frame_gen.py, the seeds of
tests/golden/make_frames_golden.py and of the frame benchmarks) as raw column-major uint8 files for tools/ref_dump/frames_dump.cpp:
the inputs of SolveDVO::computeDistTransfrmOfNow (src/SolveDVO.cpp:1740-1799), i.e. of row f1's boundary.  Host code only.

    python tools/ref_dump/export_frames.py <out dir>
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

# (name, seed, rows, cols, levels, first_shift): small enough to commit their outputs, large enough for every branch of Canny's hysteresis
FRAMES = [("cam_s2_120x160", 2, 120, 160, 3, 0), ("cam_s5_240x320", 5, 240, 320, 4, 0), ("cam_s100_480x640", 100, 480, 640, 4, 0)]


def grey_levels(oracle, seed, rows, cols, levels, first_shift):
    from rgbd_odometry_amd import frame_gen
    bgr, depth = frame_gen.camera_frame(seed, rows, cols)
    return [g for g, _ in oracle.build_pyramid(bgr, depth, levels, first_shift)]      # row-major mono8 per level


def main():
    out = sys.argv[1]
    import oracle_lib
    oracle = oracle_lib.load()
    os.makedirs(out, exist_ok=True)
    for name, seed, rows, cols, levels, fs in FRAMES:
        d = os.path.join(out, name)
        os.makedirs(d, exist_ok=True)
        gl = grey_levels(oracle, seed, rows, cols, levels, fs)
        with open(os.path.join(d, "meta.txt"), "w") as f:
            f.write("%d\n" % len(gl))
            for g in gl:
                f.write("%d %d\n" % g.shape)
        for l, g in enumerate(gl):
            np.ascontiguousarray(g.T).astype(np.uint8).tofile(os.path.join(d, "grey_%d.u8" % l))      # (yy, xx) at yy + xx * rows
    with open(os.path.join(out, "frames.txt"), "w") as f:
        f.write("\n".join(n for n, *_ in FRAMES) + "\n")
    print("wrote %d frames to %s" % (len(FRAMES), out))


if __name__ == "__main__":
    main()
