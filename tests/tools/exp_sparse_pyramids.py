#!/usr/bin/env python3
"""The C++ file replay's exact call sequence from Python: per frame a busy host gap, dvo_frames_upload_pyramids (mono8 + mono16
levels, synchronous), dvo_frames_as_now, dvo_align_pyramid-style align (set poses, enqueue, get poses).  Same library, same
HIP runtime (DVO_NO_TORCH=1).  Under tests/: the oracle generates the pyramids.  usage: DVO_NO_TORCH=1 tests/tools/exp_sparse_pyramids.py [frames] [gap_ms]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib
from rgbd_odometry_amd import DvoContext, frame_gen
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
gap = float(sys.argv[2]) * 1e-3 if len(sys.argv) > 2 else 0.017
o = oracle_lib.load()
pyrs = []
for i in range(n):
    bgr, depth = frame_gen.camera_frame(5, 480, 640, shift=((i % 16) // 2, -(i % 16)))
    pyrs.append(o.build_pyramid(bgr, depth, 4, 0))
ctx = DvoContext(1)
ctx.set_intrinsics(525.0, 525.0, 319.5, 239.5)
ctx.frames_upload_pyramids([pyrs[0]], first_slot=0)
ctx.frames_as_ref(0, 0, 1)
w = []
for k in range(1, n):
    t_end = time.perf_counter() + gap
    while time.perf_counter() < t_end:
        pass
    ctx.frames_upload_pyramids([pyrs[k]], first_slot=1 + (k & 1))
    t0 = time.perf_counter()
    ctx.frames_as_now(1 + (k & 1), 0, 1)
    R, t = ctx.align_batch([10] * 4, np.eye(3)[None], np.zeros((1, 3)))
    w.append((time.perf_counter() - t0) * 1e3)
print("python twin of the C++ replay: frames %d  median %.3f ms  max %.3f ms  mean %.3f ms" % (len(w), float(np.median(w)), max(w), float(np.mean(w))))
ctx.close()
