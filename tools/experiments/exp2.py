"""Latency anatomy: time per iteration of one workgroup as a function of the number of points."""
import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
import numpy as np, torch
from rgbd_odometry_amd import DvoContext, SynthScene
from rgbd_odometry_amd.capi import DVO_FLAG_IDENTITY_START

sc = SynthScene(640, 480, 4, 1000)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
block = int(sys.argv[2]) if len(sys.argv) > 2 else 512
H = int(sys.argv[3]) if len(sys.argv) > 3 else 0
ctx = DvoContext(B, block_threads=block)
ctx.set_intrinsics(*sc.intrinsics)
dev = torch.device("cuda")
xyz_full = []
for l, L in enumerate(sc.levels):
    xyz, _ = ctx.set_ref_level_from_images(l, L.ref_edge, L.ref_depth, L.rows, L.cols, pair=0)
    xyz_full.append(torch.from_numpy(xyz.reshape(-1)).to(dev))
    dt, gx, gy = (torch.from_numpy(a).to(dev) for a in (L.now_dt, L.now_gx, L.now_gy))
    for p in range(B):
        ctx.set_now_level_device(l, dt.data_ptr(), gx.data_ptr(), gy.data_ptr(), L.rows, L.cols, pair=p)
stream = torch.cuda.Stream(); ctx.set_stream(stream.cuda_stream)
def run(npts, iters, reps=5):
    for l in range(4):
        n = min(npts[l], xyz_full[l].numel() // 3)
        for p in range(B):
            ctx.set_ref_level_device(l, xyz_full[l].data_ptr(), n, pair=p)
    ctx.synchronize()
    ctx.enqueue(iters, flags=DVO_FLAG_IDENTITY_START); ctx.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(stream)
    for _ in range(reps): ctx.enqueue(iters, flags=DVO_FLAG_IDENTITY_START)
    b.record(stream); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3   # us per launch
print("B=%d block=%d H=%d" % (B, block, H))
for n in (1, 64, 512, 2048, 8192, 100000):
    t = run([n] * 4, [0, 0, 0, 40])          # 40 iterations at level 3 only (tiny image: cache resident)
    print("level3 x40it  N<=%6d : %8.1f us/launch  %6.2f us/iter" % (n, t, t / 40))
for n in (512, 2048, 8192, 100000):
    t = run([n] * 4, [40, 0, 0, 0])          # 40 iterations at level 0
    print("level0 x40it  N<=%6d : %8.1f us/launch  %6.2f us/iter" % (n, t, t / 40))
t = run([10**6] * 4, [10, 10, 10, 10]); print("full C2 schedule: %.1f us/launch" % t)
