"""create / fill / align / destroy contexts repeatedly; device memory must return to its starting level"""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
import numpy as np, torch
from rgbd_odometry_amd import DvoContext, SynthScene
sc = SynthScene(320, 240, 4, 3)
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
    ctx.close()
    if rep in (0, 1, 9, 19, 29):
        f = torch.cuda.mem_get_info()[0]
        if rep == 0: first = f
        print("after %2d contexts: free %.1f MB" % (rep + 1, f / 1e6), flush=True)
free1, _ = torch.cuda.mem_get_info()
print("free before %.1f MB, after %.1f MB, delta %.2f MB" % (free0 / 1e6, free1 / 1e6, (free0 - free1) / 1e6))

assert abs(first - free1) < 8e6, "device memory is not returned by dvo_destroy"
print("leak check ok (the delta to 'before' is one-time runtime / code-object overhead)")
