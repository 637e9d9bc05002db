#!/usr/bin/env python3
"""Launch timeline of the fused kernel from the stamps build (make STAMPS=1): start / end tick of every workgroup ->
ramp-up, duration spread, tail.  usage: exp_timeline.py [B]"""
import os, sys
os.environ["DVO_LIB_VARIANT"] = os.environ.get("DVO_TL_VARIANT", "_stamps")
if len(sys.argv) > 2: os.environ["DVO_NO_LPT"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from rgbd_odometry_amd import DvoContext, SynthScene
from rgbd_odometry_amd.capi import DVO_FLAG_IDENTITY_START, DVO_FLAG_FINAL_OUTPUTS
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
scs = [SynthScene(640, 480, 4, 1000 + i) for i in range(32)]
ctx = DvoContext(B)
ctx.set_intrinsics(*scs[0].intrinsics)
for i, sc in enumerate(scs):
    for l, L in enumerate(sc.levels):
        ctx.set_ref_level_from_images(l, L.ref_edge, L.ref_depth, L.rows, L.cols, pair=i)
        ctx.set_now_level(l, L.now_dt, L.now_gx, L.now_gy, L.rows, L.cols, pair=i)
ctx.replicate_pairs(32); ctx.now_prepare(); ctx.synchronize()
iters = [10] * 4
for rep in range(3):
    ctx.enqueue(iters, flags=DVO_FLAG_IDENTITY_START | DVO_FLAG_FINAL_OUTPUTS); ctx.synchronize()
st = np.array([ctx.debug_stamps(p).reshape(-1)[62:64] for p in range(B)], dtype=np.float64)
t0 = st[:, 0].min(); s = (st[:, 0] - t0) * 0.01; e = (st[:, 1] - t0) * 0.01          # 100 MHz device-wide counter -> microseconds
dur = e - s; total = e.max()
print("B=%d: launch %.0f us; workgroup duration mean %.0f us (min %.0f, max %.0f, sd %.0f)" % (B, total, dur.mean(), dur.min(), dur.max(), dur.std()))
print("starts within the first 2%% of the launch: %d workgroups; last start at %.2f of the launch" % ((s < 0.02 * total).sum(), s.max() / total))
edges = np.linspace(0, total, 21)
busy = [(np.minimum(e, edges[k + 1]) - np.maximum(s, edges[k])).clip(min=0).sum() / (edges[k + 1] - edges[k]) for k in range(20)]
print("workgroups in flight per 5% slice of the launch:", " ".join("%.0f" % b for b in busy))
print("ends: 50%% of the workgroups are done at %.3f of the launch, 90%% at %.3f, 99%% at %.3f" % tuple(np.percentile(e, q) / total for q in (50, 90, 99)))
print("ideal launch (sum of durations / 512 slots) = %.3f of the measured one" % (dur.sum() / min(B, 512) / total))
