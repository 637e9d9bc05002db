#!/usr/bin/env python3
"""Sparse single-stream work (one 640x480 frame, then ~17 ms of host idling, like the C++ file replay) through the C ABI from
Python, once on PyTorch's bundled HIP runtime (default) and once on /opt/rocm's (DVO_NO_TORCH=1: torch is never imported, the
library binds to the system libamdhip64).  Prints the wall time of every frame's  upload + alignment + pose.
usage: [DVO_NO_TORCH=1] exp_sparse_runtime.py [frames] [gap_ms] [spin]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from rgbd_odometry_amd import DvoContext, frame_gen
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
gap = float(sys.argv[2]) * 1e-3 if len(sys.argv) > 2 else 0.017
ctx = DvoContext(1)
ctx.set_intrinsics(525.0, 525.0, 319.5, 239.5)
ref = frame_gen.camera_frame(5, 480, 640)
nows = [frame_gen.camera_frame(5, 480, 640, shift=(1 + k % 3, -(k % 5)))[0] for k in range(4)]
ctx.frames_upload_cameras([ref[0]], [ref[1]], n_levels=4, first_shift=0, first_slot=0)
ctx.frames_as_ref(0, 0, 1)
w = []
busy = len(sys.argv) > 3 and sys.argv[3] == "busy"
for k in range(n):
    if busy:
        t_end = time.perf_counter() + gap
        x = 0
        while time.perf_counter() < t_end:
            x += 1                      # the host thread works between frames (the C++ replay parses XML) instead of sleeping
    else:
        time.sleep(gap)
    t0 = time.perf_counter()
    ctx.frames_upload_cameras([nows[k % 4]], None, n_levels=4, first_shift=0, first_slot=1, now_first_pair=0)
    R, t = ctx.align_batch([10] * 4, np.eye(3)[None], np.zeros((1, 3)))
    w.append((time.perf_counter() - t0) * 1e3)
print(("busy-gap " if busy else "sleep-gap ") + "runtime %s  frames %d gap %.0f ms: first %.2f, then median %.3f ms  max %.3f ms   (%s)" % (
    "system /opt/rocm" if os.environ.get("DVO_NO_TORCH") == "1" else "torch-bundled", n, gap * 1e3, w[0], float(np.median(w[1:])), max(w[1:]),
    " ".join("%.2f" % x for x in w[1:8])))
ctx.close()
