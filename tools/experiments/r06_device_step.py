"""round 6: the `camera frames in HBM -> poses` step of bench.py's frames leg, alone (for rocprofv3 timelines).  usage: r06_device_step.py [reps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from rgbd_odometry_amd import frame_gen, DvoContext
from rgbd_odometry_amd.capi import DVO_FLAG_IDENTITY_START, DVO_UPLOAD_ASYNC
B, D, H, W, NL = 256, 8, 480, 640, 4
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
ref = [frame_gen.camera_frame(100 + i, H, W) for i in range(D)]
now = [frame_gen.camera_frame(100 + i, H, W, shift=(1 + i % 2, -2))[0] for i in range(D)]
ctx = DvoContext(B)
ctx.set_intrinsics(525.0, 525.0, 319.5, 239.5)
ctx.frames_reserve(2 * B)
kw = dict(n_levels=NL, first_shift=0, flags=DVO_UPLOAD_ASYNC)
ctx.frames_upload_cameras([ref[i % D][0] for i in range(B)], [ref[i % D][1] for i in range(B)], first_slot=0, **kw)
ctx.frames_as_ref(0, 0, B)
dev_now = [torch.from_numpy(np.ascontiguousarray(now[i % D])).cuda() for i in range(B)]      # every frame in its own buffer
dev_ptrs = ctx.pointer_table([t_.data_ptr() for t_ in dev_now])
iters = [10] * NL
def step():
    ctx.frames_upload_cameras_device(dev_ptrs, None, H, W, n_levels=NL, first_shift=0, first_slot=B, flags=DVO_UPLOAD_ASYNC, now_first_pair=0)
    ctx.enqueue(iters, flags=DVO_FLAG_IDENTITY_START)
    return ctx.get_poses()
step()
t0 = time.perf_counter()
for _ in range(reps): step()
dt = (time.perf_counter() - t0) / reps
print("device_step: %.3f ms per %d frames = %.0f frames/s" % (1e3 * dt, B, B / dt), flush=True)
ctx.close()
