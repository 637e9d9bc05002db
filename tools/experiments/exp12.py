"""Does the ORDER of the reference points change the request count / throughput?  (caller-side reordering)"""
import os, sys, json
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
import numpy as np, torch
from rgbd_odometry_amd import DvoContext, SynthScene
from rgbd_odometry_amd.capi import DVO_FLAG_IDENTITY_START, DVO_FLAG_FINAL_OUTPUTS
B, D = 1024, 16
scenes = [SynthScene(640, 480, 4, 1000 + i) for i in range(D)]
iters = [10] * 4
def order(uv, mode):
    xx, yy = uv[:, 0].astype(np.int64), uv[:, 1].astype(np.int64)
    if mode == "colmajor": return np.arange(len(uv))
    if mode == "rowmajor": return np.lexsort((xx, yy))
    if mode.startswith("block"):
        b = int(mode[5:]); return np.lexsort((yy, xx, yy // b, xx // b))      # blocks column-major, inside column-major
    if mode == "morton":
        def spread(v):
            v = v & 0xFFFF; v = (v | (v << 8)) & 0x00FF00FF; v = (v | (v << 4)) & 0x0F0F0F0F; v = (v | (v << 2)) & 0x33333333; v = (v | (v << 1)) & 0x55555555; return v
        return np.argsort(spread(xx) << 1 | spread(yy), kind="stable")
    if mode == "random": return np.random.default_rng(0).permutation(len(uv))
for mode in ["colmajor", "block4", "block8", "block16", "block32", "morton", "rowmajor", "random"]:
    ctx = DvoContext(B)
    ctx.set_intrinsics(*scenes[0].intrinsics)
    for i, sc in enumerate(scenes):
        for l, L in enumerate(sc.levels):
            xyz, uv = ctx.set_ref_level_from_images(l, L.ref_edge, L.ref_depth, L.rows, L.cols, pair=i)
            ctx.set_ref_level(l, xyz[order(uv, mode)], pair=i)
            ctx.set_now_level(l, L.now_dt, L.now_gx, L.now_gy, L.rows, L.cols, pair=i)
    ctx.replicate_pairs(D)
    stream = torch.cuda.Stream(); ctx.set_stream(stream.cuda_stream)
    fl = DVO_FLAG_IDENTITY_START | DVO_FLAG_FINAL_OUTPUTS
    for _ in range(2): ctx.enqueue(iters, flags=fl)
    ctx.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(stream)
    for _ in range(10): ctx.enqueue(iters, flags=fl)
    b.record(stream); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 10
    R, t = ctx.get_poses(0, 1)
    print("%-10s kernel %.3f ms  -> %.0f aligns/s   t0=%s" % (mode, ms, B / ms * 1e3, np.round(t[0], 6)))
    ctx.close()
