#!/usr/bin/env python3
"""compact now form: what the builder says for the bench scenes (palette size per level, texel mode per level)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from rgbd_odometry_amd import DvoContext, SynthScene
W, H, NL = 640, 480, 4
ctx = DvoContext(n_pairs=4)
scs = [SynthScene(W, H, NL, 1000 + i) for i in range(4)]
ctx.set_intrinsics(*scs[0].intrinsics)
for i, sc in enumerate(scs):
    for l, L in enumerate(sc.levels):
        ctx.set_ref_level_from_images(l, L.ref_edge, L.ref_depth, L.rows, L.cols, pair=i)
        ctx.set_now_level(l, L.now_dt, L.now_gx, L.now_gy, L.rows, L.cols, pair=i)
ctx.now_prepare()
for i in range(4):
    print("pair", i, "palette sizes", [ctx.now_compact_info(i, l) for l in range(NL)])
