#!/usr/bin/env python3
"""round 6: where the waves of the band kernel (dvo_edt_band.h) spend their cycles (diagnostic build:
  make -C rgbd_odometry_amd/csrc EXP=edtstamps EXPDEFS=-DDVO_EDT_STAMPS=1 ; DVO_LIB_VARIANT=_edtstamps python tools/experiments/r06_band_stamps.py [frames])"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from rgbd_odometry_amd import DvoContext, frame_gen
from rgbd_odometry_amd.capi import load_library
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
lib = load_library()
lib.dvo_debug_edt_stamps.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
frames = [frame_gen.camera_frame(100 + i % 8, 480, 640, shift=(1 + i % 2, -2))[0] for i in range(B)]
with DvoContext(B) as ctx:
    ctx.set_intrinsics(525.0, 525.0, 319.5, 239.5)
    ctx.frames_reserve(B)
    ctx.frames_upload_cameras(frames, None, n_levels=4, first_shift=0, first_slot=0)
    ctx.frames_as_now(0, 0, B); ctx.synchronize()               # warm-up
    out = (C.c_ulonglong * 8)()
    lib.dvo_debug_edt_stamps(out, 1)
    ctx.frames_as_now(0, 0, B); ctx.synchronize()
    lib.dvo_debug_edt_stamps(out, 0)
v = [int(x) for x in out]
names = ["table + g rows + barrier", "scan trips", "ranks -> result tile", "rank words (incl. barrier)", "tail"]
waves, trips, total = v[5], v[6], v[7]
px = B * sum((480 >> l) * (640 >> l) for l in range(4))
print("%d frames, %d waves, %d trips of 8 steps (%.1f steps per pixel at 128 pixels per wave-step), s_memtime ticks summed over waves: %d" %
      (B, waves, trips, trips * 8 * 128 / px, total))
for k, n in enumerate(names):
    print("   %-30s %12d  %5.1f %%" % (n, v[k], 100.0 * v[k] / max(total, 1)))
print("   ticks per trip %.2f; ticks per wave %.0f" % (v[1] / max(trips, 1), total / max(waves, 1)))
