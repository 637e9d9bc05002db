#!/usr/bin/env python3
"""Run only the fused now-frame step of tools/bench_frames.py a few times (for rocprofv3 timelines)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from rgbd_odometry_amd import frame_gen
from rgbd_odometry_amd import DvoContext
from rgbd_odometry_amd.capi import DVO_FLAG_IDENTITY_START, DVO_UPLOAD_ASYNC
B, D = 256, 8
def pin(a):
    t = torch.empty(a.shape, dtype=torch.from_numpy(a).dtype, pin_memory=True); t.numpy()[...] = a; return t.numpy()
ref = [tuple(pin(x) for x in frame_gen.camera_frame(100 + i)) for i in range(D)]
now = [pin(frame_gen.camera_frame(100 + i, shift=(1, -2))[0]) for i in range(D)]
ctx = DvoContext(B); ctx.set_intrinsics(525.0, 525.0, 319.5, 239.5); ctx.frames_reserve(2 * B)
kw = dict(n_levels=4, first_shift=0, flags=DVO_UPLOAD_ASYNC)
ctx.frames_upload_cameras([ref[i % D][0] for i in range(B)], [ref[i % D][1] for i in range(B)], first_slot=0, **kw)
ctx.frames_as_ref(0, 0, B)
nb = [now[i % D] for i in range(B)]
mode = sys.argv[1] if len(sys.argv) > 1 else "fused"
for rep in range(4):
    t0 = time.perf_counter()
    if mode == "fused":
        ctx.frames_upload_cameras(nb, None, first_slot=B, now_first_pair=0, **kw)
    else:
        ctx.frames_upload_cameras(nb, None, first_slot=B, **kw); ctx.frames_as_now(B, 0, B)
    t1 = time.perf_counter()
    ctx.enqueue([10] * 4, flags=DVO_FLAG_IDENTITY_START)
    R, t = ctx.get_poses()
    print(mode, "rep", rep, "enqueue ms %.2f total ms %.2f" % (1e3 * (t1 - t0), 1e3 * (time.perf_counter() - t0)), flush=True)
ctx.close()
