"""create / fill / align / destroy contexts repeatedly; device memory must return to its starting level"""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
import numpy as np, torch
from rgbd_odometry_amd import DvoContext, SynthScene
from rgbd_odometry_amd import frame_gen
sc = SynthScene(320, 240, 4, 3)
frames = [frame_gen.camera_frame(i, 240, 320) for i in range(3)]
torch.cuda.init()
free0, _ = torch.cuda.mem_get_info()
for rep in range(30):
    ctx = DvoContext(16)
    ctx.set_intrinsics(*sc.intrinsics)
    for l, L in enumerate(sc.levels):
        ctx.set_ref_level_from_images(l, L.ref_edge, L.ref_depth, L.rows, L.cols)
        ctx.set_now_level_from_edges(l, (L.now_edge > 0).astype(np.uint8), L.rows, L.cols)
    ctx.replicate_pairs(1)
    ctx.align_batch([5] * 4, np.tile(np.eye(3), (16, 1, 1)), np.zeros((16, 3)), flags=1)
    ctx.align_pyramid_wide([3] * 4, np.eye(3), np.zeros(3))
    ctx.eval_points(0, np.eye(3), np.zeros(3)); ctx.accumulate(1, np.eye(3), np.zeros(3))
    # frames in (rows f1/f2): single frames (per-level streams) and a batch (landing buffers, copy streams)
    ctx.frames_upload_cameras([frames[0][0]], [frames[0][1]], n_levels=4, first_shift=0, first_slot=0)
    ctx.frames_upload_cameras([f[0] for f in frames] * 4, [f[1] for f in frames] * 4, n_levels=4, first_shift=0, first_slot=1, now_first_pair=0)
    ctx.frames_as_ref(0, 0, 1); ctx.frames_as_now(1, 0, 12)
    ctx.align_batch([3] * 4, np.tile(np.eye(3), (16, 1, 1)), np.zeros((16, 3)))
    ctx.close()
    if rep in (0, 1, 9, 19, 29):
        f = torch.cuda.mem_get_info()[0]
        if rep == 0: first = f
        print("after %2d contexts: free %.1f MB" % (rep + 1, f / 1e6), flush=True)
free1, _ = torch.cuda.mem_get_info()
print("free before %.1f MB, after %.1f MB, delta %.2f MB" % (free0 / 1e6, free1 / 1e6, (free0 - free1) / 1e6))

assert abs(first - free1) < 8e6, "device memory is not returned by dvo_destroy"
print("leak check ok (the delta to 'before' is one-time runtime / code-object overhead)")
