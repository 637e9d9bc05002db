#!/usr/bin/env python3
"""host-side cost of one bench step: the enqueue call (asynchronous), the pose read-back (synchronises), the kernel"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from rgbd_odometry_amd import DvoContext, SynthScene
from rgbd_odometry_amd.capi import DVO_FLAG_FINAL_OUTPUTS, DVO_FLAG_IDENTITY_START
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
scs = [SynthScene(640, 480, 4, 1000 + i) for i in range(8)]
ctx = DvoContext(B)
ctx.set_intrinsics(*scs[0].intrinsics)
for i, sc in enumerate(scs):
    for l, L in enumerate(sc.levels):
        ctx.set_ref_level_from_images(l, L.ref_edge, L.ref_depth, L.rows, L.cols, pair=i)
        ctx.set_now_level(l, L.now_dt, L.now_gx, L.now_gy, L.rows, L.cols, pair=i)
ctx.replicate_pairs(8); ctx.now_prepare(); ctx.synchronize()
iters = [10] * 4; flags = DVO_FLAG_IDENTITY_START | DVO_FLAG_FINAL_OUTPUTS
for _ in range(3): ctx.enqueue(iters, flags=flags); ctx.get_poses()
te = tg = 0.0; n = 30
t_all = time.perf_counter()
for _ in range(n):
    t0 = time.perf_counter(); ctx.enqueue(iters, flags=flags); t1 = time.perf_counter(); ctx.get_poses(); t2 = time.perf_counter()
    te += t1 - t0; tg += t2 - t1
t_all = time.perf_counter() - t_all
print("B=%d: enqueue call %.1f us, get_poses (incl. waiting for the kernel) %.1f us, step %.1f us" % (B, 1e6 * te / n, 1e6 * tg / n, 1e6 * t_all / n))
