#!/usr/bin/env python3
"""does the PCIe pull (gather_images_kernel on the copy streams) overlap the preprocessing kernels of the previous chunk?
Run under `rocprofv3 --kernel-trace`; tools/experiments/exp_pull_overlap.sh analyses the trace."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from rgbd_odometry_amd import DvoContext, frame_gen, capi
B, H, W, L = 256, 480, 640, 4
def pin(a):
    t = torch.empty(a.shape, dtype=torch.from_numpy(a).dtype, pin_memory=True); t.numpy()[...] = a; return t.numpy()
ref = frame_gen.camera_frame(100, H, W)
now = frame_gen.camera_frame(100, H, W, shift=(1, -2))[0]
ref_b, ref_d = [pin(ref[0]) for _ in range(B)], [pin(ref[1]) for _ in range(B)]
now_b = [pin(now) for _ in range(B)]
ctx = DvoContext(B)
ctx.set_intrinsics(525.0, 525.0, 319.5, 239.5)
ctx.frames_reserve(2 * B)
kw = dict(n_levels=L, first_shift=0, flags=capi.DVO_UPLOAD_ASYNC | capi.DVO_UPLOAD_MAPPED)
ctx.frames_upload_cameras(ref_b, ref_d, first_slot=0, **kw)
ctx.frames_as_ref(0, 0, B)
def step():
    ctx.frames_upload_cameras(now_b, None, first_slot=B, now_first_pair=0, **kw)
    ctx.enqueue([10] * L, flags=capi.DVO_FLAG_IDENTITY_START)
    return ctx.get_poses()
step(); ctx.synchronize()
t0 = time.perf_counter()
for _ in range(3): step()
print("ms per step %.3f" % ((time.perf_counter() - t0) / 3 * 1e3))
ctx.close()
