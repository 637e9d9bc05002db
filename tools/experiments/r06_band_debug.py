"""round 6 debug: the band stage's now levels against the scene generator's (== the oracle's) planar images"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from rgbd_odometry_amd import DvoContext, SynthScene

sc = SynthScene(640, 480, 4, 3)
with DvoContext(1) as ctx:
    ctx.set_intrinsics(*sc.intrinsics)
    for l, L in enumerate(sc.levels):
        ctx.set_ref_level_from_images(l, L.ref_edge, L.ref_depth, L.rows, L.cols)
        ctx.set_now_level_from_edges(l, (np.asarray(L.now_edge) != 0).astype(np.uint8) * 255, L.rows, L.cols)
    for l, L in enumerate(sc.levels):
        print("level", l, "info", ctx.now_compact_info(0, l), "partial", ctx.now_compact_partial(0, l))
        dt, gx, gy = ctx.get_now_level(l)
        ref = np.asarray(L.now_dt, np.float32).reshape(L.cols, L.rows)
        got = np.asarray(dt, np.float32).reshape(L.cols, L.rows)
        bad = np.argwhere(got.view(np.uint32) != ref.view(np.uint32))
        print("  dt mismatches", len(bad), "of", ref.size, "max got", got.max(), "max ref", ref.max())
        for xx, yy in bad[:12]:
            print("   xx", xx, "yy", yy, "got", got[xx, yy], "ref", ref[xx, yy])
        if len(bad):
            print("   rows hist", np.bincount(bad[:, 1] % 12, minlength=12), "cols range", bad[:, 0].min(), bad[:, 0].max())
        for nm, a, b in (("gx", gx, L.now_gx), ("gy", gy, L.now_gy)):
            a = np.asarray(a, np.float32); b = np.asarray(b, np.float32)
            print("  ", nm, "mismatches", int((a.view(np.uint32) != b.view(np.uint32)).sum()))
