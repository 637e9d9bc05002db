#!/usr/bin/env python3
"""upload_cameras with DMA copies (DVO_UPLOAD_DIRECT, bgr + depth, 256 frames in their own pinned buffers): copies of chunk k+1
submitted before (DVO_COPY_AHEAD=1) or after (=0) the kernels of chunk k.  Same process order effects excluded by running each
setting in its own process, alternating (tools/experiments/exp_copy_order.sh)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from rgbd_odometry_amd import DvoContext, frame_gen, capi
B, H, W, L = 256, 480, 640, 4
def pin(a):
    t = torch.empty(a.shape, dtype=torch.from_numpy(a).dtype, pin_memory=True); t.numpy()[...] = a; return t.numpy()
ref = frame_gen.camera_frame(100, H, W)
bl, dl = [pin(ref[0]) for _ in range(B)], [pin(ref[1]) for _ in range(B)]
ctx = DvoContext(B); ctx.set_intrinsics(525.0, 525.0, 319.5, 239.5); ctx.frames_reserve(B)
for name, flags in (("DIRECT", capi.DVO_UPLOAD_DIRECT), ("MAPPED", capi.DVO_UPLOAD_MAPPED), ("mirror", 0)):
    kw = dict(n_levels=L, first_shift=0, flags=capi.DVO_UPLOAD_ASYNC | flags)
    ctx.frames_upload_cameras(bl, dl, first_slot=0, **kw); ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(4): ctx.frames_upload_cameras(bl, dl, first_slot=0, **kw)
    ctx.synchronize()
    print("%s: %.2f ms per 256 bgr+depth frames" % (name, (time.perf_counter() - t0) / 4 * 1e3))
ctx.close()
