#!/usr/bin/env python3
"""GPU work per now frame, measured the same way on any tree of this repository (round 2's included): a batch of 256
640x480 BGR8 camera frames in pinned host memory -> pyramid, Canny, distance transform -> now levels -> alignment against
resident references -> poses; run under `rocprofv3 --kernel-trace`, the caller sums the kernel durations (copies excluded) and
divides by the frames processed.  usage: gpu_time_per_now_frame.py <repo root> [reps]"""
import os, sys, time
root = os.path.abspath(sys.argv[1]); reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np, torch
from rgbd_odometry_amd import DvoContext, frame_gen
from rgbd_odometry_amd import capi
B, D, H, W, L = 256, 8, 480, 640, 4
def pin(a):
    t = torch.empty(a.shape, dtype=torch.from_numpy(a).dtype, pin_memory=True); t.numpy()[...] = a; return t.numpy()
ref = [tuple(pin(x) for x in frame_gen.camera_frame(100 + i, H, W)) for i in range(D)]
now = [pin(frame_gen.camera_frame(100 + i, H, W, shift=(1 + i % 2, -2))[0]) for i in range(D)]
ctx = DvoContext(B)
ctx.set_intrinsics(525.0, 525.0, 319.5, 239.5)
ctx.frames_reserve(2 * B)
flags = capi.DVO_UPLOAD_ASYNC | getattr(capi, "DVO_UPLOAD_DIRECT", 0)
kw = dict(n_levels=L, first_shift=0, flags=flags)
ctx.frames_upload_cameras([ref[i % D][0] for i in range(B)], [ref[i % D][1] for i in range(B)], first_slot=0, **kw)
ctx.frames_as_ref(0, 0, B)
iters = [10] * L
now_b = [now[i % D] for i in range(B)]
def step():
    ctx.frames_upload_cameras(now_b, None, first_slot=B, **kw)
    ctx.frames_as_now(B, 0, B)
    ctx.enqueue(iters, flags=capi.DVO_FLAG_IDENTITY_START)
    return ctx.get_poses()
step()
print("MARK begin", flush=True)
t0 = time.perf_counter()
for _ in range(reps): step()
dt = time.perf_counter() - t0
print("frames %d wall_ms_per_256 %.3f" % (reps * B, 1e3 * dt / reps), flush=True)
ctx.close()
