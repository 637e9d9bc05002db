"""Phase anatomy from the stamps build (DVO_LIB_VARIANT=_stamps)."""
import os, sys
os.environ["DVO_LIB_VARIANT"] = "_stamps"
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
import numpy as np, torch
from rgbd_odometry_amd import DvoContext, SynthScene
from rgbd_odometry_amd.capi import DVO_FLAG_IDENTITY_START
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
block = int(sys.argv[2]) if len(sys.argv) > 2 else 512
lds = int(sys.argv[3]) if len(sys.argv) > 3 else 155000
U = int(sys.argv[4]) if len(sys.argv) > 4 else 0
sc = SynthScene(640, 480, 4, 1000)
ctx = DvoContext(B, block_threads=block, lds_point_bytes=lds, points_in_flight=U)
ctx.set_intrinsics(*sc.intrinsics)
for l, L in enumerate(sc.levels):
    xyz, _ = ctx.set_ref_level_from_images(l, L.ref_edge, L.ref_depth, L.rows, L.cols, pair=0)
    for p in range(B):
        ctx.set_ref_level(l, xyz, pair=p)
        ctx.set_now_level(l, L.now_dt, L.now_gx, L.now_gy, L.rows, L.cols, pair=p)
iters = [10, 10, 10, 10]
for rep in range(3):
    ctx.enqueue(iters, flags=DVO_FLAG_IDENTITY_START); ctx.synchronize()
    st = ctx.debug_stamps(B // 2)
print("U=%d " % U, end=""); print("B=%d block=%d lds=%d   (cycles per iteration; s_memtime ticks = shader cycles)" % (B, block, lds))
for l in range(4):
    n = max(1, int(st[l, 4]))
    print("level %d: loop %7.0f  reduce %6.0f  update %6.0f  barrier %6.0f   (iters %d)" % (
        l, st[l, 0] / n, st[l, 1] / n, st[l, 2] / n, st[l, 3] / n, n))
