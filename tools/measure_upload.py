#!/usr/bin/env python3
"""Host-buffer (PCIe-inclusive) rate of the boundary: planar DT/gx/gy + points handed over as HOST
pointers (dvo_set_now_level / dvo_set_ref_level), packed on the device, then aligned."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from rgbd_odometry_amd import DvoContext, SynthScene
from rgbd_odometry_amd.capi import DVO_FLAG_IDENTITY_START
B = 64
sc = SynthScene(640, 480, 4, 1000)
ctx = DvoContext(B)
ctx.set_intrinsics(*sc.intrinsics)
xyzs = [ctx.set_ref_level_from_images(l, L.ref_edge, L.ref_depth, L.rows, L.cols)[0] for l, L in enumerate(sc.levels)]
def upload():
    for p in range(B):
        for l, L in enumerate(sc.levels):
            ctx.set_ref_level(l, xyzs[l], pair=p)
            ctx.set_now_level(l, L.now_dt, L.now_gx, L.now_gy, L.rows, L.cols, pair=p)
upload(); ctx.synchronize()
t0 = time.perf_counter(); upload(); ctx.synchronize(); t_up = time.perf_counter() - t0
iters = [10] * 4
ctx.enqueue(iters, flags=DVO_FLAG_IDENTITY_START); ctx.get_poses()
t0 = time.perf_counter(); ctx.enqueue(iters, flags=DVO_FLAG_IDENTITY_START); ctx.get_poses(); t_al = time.perf_counter() - t0
bytes_pair = sum(12 * L.rows * L.cols for L in sc.levels) + sum(12 * len(x) for x in xyzs)
print(json.dumps(dict(pairs=B, host_bytes_per_pair=bytes_pair, upload_pack_s=t_up, upload_GBps=B * bytes_pair / t_up / 1e9,
                      pairs_per_s_upload_only=B / t_up, align_s=t_al, pairs_per_s_including_upload=B / (t_up + t_al))))

# second variant: raw-frame hand-over (row f1): 1-byte now edge masks + reference edge/depth images, everything else on the GPU
edges = [(L.now_edge > 0).astype(np.uint8) for L in sc.levels]
def upload_raw():
    for p in range(B):
        for l, L in enumerate(sc.levels):
            ctx.set_ref_level_from_images(l, L.ref_edge, L.ref_depth, L.rows, L.cols, pair=p)
            ctx.set_now_level_from_edges(l, edges[l], L.rows, L.cols, pair=p)
upload_raw(); ctx.synchronize()
t0 = time.perf_counter(); upload_raw(); ctx.synchronize(); t_raw = time.perf_counter() - t0
raw_bytes = sum(L.rows * L.cols * (1 + 4 + 4) for L in sc.levels)
print(json.dumps(dict(variant="raw frames (edge masks + depth), preprocessing on the GPU", pairs=B, host_bytes_per_pair=raw_bytes,
                      upload_preprocess_s=t_raw, pairs_per_s=B / t_raw, ms_per_pair=1e3 * t_raw / B)))
