#!/usr/bin/env python3
"""cost of installing a now level from float images in device memory (dvo_set_now_level_device), with and without the direct
compact build (dvo_set_direct_compact): one 640x480 four-level pair per call, 200 calls, wall time per pair"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from rgbd_odometry_amd import DvoContext, SynthScene
sc = SynthScene(640, 480, 4, 3)
for mode in ("off", "on"):
    with DvoContext(1) as ctx:
        ctx.set_intrinsics(*sc.intrinsics)
        ctx.set_direct_compact(mode == "on")
        dev = [[torch.from_numpy(np.ascontiguousarray(np.asarray(a, np.float32).ravel())).cuda() for a in (L.now_dt, L.now_gx, L.now_gy)] for L in sc.levels]
        def install():
            for l, L in enumerate(sc.levels):
                ctx.set_now_level_device(l, dev[l][0].data_ptr(), dev[l][1].data_ptr(), dev[l][2].data_ptr(), L.rows, L.cols)
        install(); ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(200): install()
        ctx.synchronize()
        dt = (time.perf_counter() - t0) / 200
        print("direct build %-3s: %.1f us per four-level pair; compact info %s" % (mode, 1e6 * dt, [ctx.now_compact_info(0, l) for l in range(4)]))
