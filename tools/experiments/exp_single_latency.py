import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from rgbd_odometry_amd import frame_gen
from rgbd_odometry_amd import DvoContext
ctx = DvoContext(1)
ctx.set_intrinsics(525.0, 525.0, 319.5, 239.5)
f0 = frame_gen.camera_frame(5, 480, 640); f1 = frame_gen.camera_frame(5, 480, 640, shift=(1, -2))
ctx.frames_upload_cameras([f0[0], f1[0]], [f0[1], f1[1]], n_levels=4, first_shift=0)
print("N", ctx.frames_as_ref(0, 0, 1)); ctx.frames_as_now(1, 0, 1)
for it in (10, 20, 30, 40, 50):
    for rep in range(3):
        t0 = time.perf_counter()
        R, t = ctx.align_batch([it] * 4, np.eye(3)[None], np.zeros((1, 3)))
        dt = time.perf_counter() - t0
    e, best, ratio = ctx.level_report(0, 0, it)
    print("iters %d: %.3f ms  best %d ratio %.3f  t %s" % (it, dt * 1e3, best, ratio, np.round(t[0], 4)), flush=True)
for rep in range(3):
    t0 = time.perf_counter(); ctx.frames_as_now(1, 0, 1); ctx.synchronize(); print("as_now single %.3f ms" % ((time.perf_counter() - t0) * 1e3))
for rep in range(3):
    t0 = time.perf_counter(); ctx.frames_as_ref(0, 0, 1); ctx.synchronize(); print("as_ref single %.3f ms" % ((time.perf_counter() - t0) * 1e3))
for rep in range(3):
    t0 = time.perf_counter(); ctx.frames_upload_cameras([f1[0]], [f1[1]], n_levels=4, first_shift=0, first_slot=1); print("upload single %.3f ms" % ((time.perf_counter() - t0) * 1e3))
