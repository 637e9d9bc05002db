"""single-pair latency of the fused kernel vs workgroup size / points in flight / LDS budget (C2 synthetic + ref default)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from rgbd_odometry_amd import DvoContext, SynthScene
for (W, H, it) in ((640, 480, 10), (320, 240, 50), (1920, 1080, 10)):
    nl = 4 if W < 1000 else 5
    sc = SynthScene(W, H, nl, 1)
    for block in (256, 512, 1024):
        for u in (1, 2):
            for lds in (0,):
                ctx = DvoContext(1, block_threads=block, points_in_flight=u, lds_point_bytes=lds)
                ctx.set_intrinsics(*sc.intrinsics)
                for l, L in enumerate(sc.levels):
                    ctx.set_ref_level_from_images(l, L.ref_edge, L.ref_depth, L.rows, L.cols)
                    ctx.set_now_level(l, L.now_dt, L.now_gx, L.now_gy, L.rows, L.cols)
                best = 1e9
                for rep in range(6):
                    t0 = time.perf_counter()
                    R, t = ctx.align_batch([it] * nl, np.eye(3)[None], np.zeros((1, 3)))
                    best = min(best, time.perf_counter() - t0)
                print("%dx%d it %d block %4d U %d: %.3f ms" % (W, H, it, block, u, best * 1e3), flush=True)
                ctx.close()
