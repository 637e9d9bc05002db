#!/usr/bin/env python3
"""Rows f1+f2 measured: camera frames (BGR8 + depth in metres, host memory) -> resident now / reference levels,
batched, and the end-to-end rate frames-in -> poses-out.  Not the headline bench (bench.py); DESIGN.md quotes it.

    python tools/bench_frames.py [--batch 256] [--width 640 --height 480] [--levels 4] [--first-shift 0] [--pinned]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--levels", type=int, default=4)
    ap.add_argument("--first-shift", type=int, default=0)
    ap.add_argument("--distinct", type=int, default=8)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--pinned", action="store_true", help="frames live in pinned host memory (torch pin_memory)")
    args = ap.parse_args()
    from rgbd_odometry_amd import frame_gen
    from rgbd_odometry_amd import DvoContext
    from rgbd_odometry_amd.capi import DVO_FLAG_IDENTITY_START, DVO_UPLOAD_ASYNC, DVO_UPLOAD_DIRECT

    B, D = args.batch, min(args.distinct, args.batch)
    ref = [frame_gen.camera_frame(100 + i, args.height, args.width) for i in range(D)]
    now = [frame_gen.camera_frame(100 + i, args.height, args.width, shift=(1 + i % 2, -2)) for i in range(D)]

    def hold(a):
        if not args.pinned:
            return a
        t = torch.empty(a.shape, dtype=torch.from_numpy(a).dtype, pin_memory=True)
        t.numpy()[...] = a
        return t.numpy()
    ref = [(hold(b), hold(d)) for b, d in ref]
    now = [(hold(b), hold(d)) for b, d in now]
    ref_b, ref_d = [ref[i % D][0] for i in range(B)], [ref[i % D][1] for i in range(B)]
    now_b = [now[i % D][0] for i in range(B)]

    ctx = DvoContext(B)
    s = 2.0 ** (-args.first_shift) * args.width / 640.0
    ctx.set_intrinsics(525.0 * s, 525.0 * s, 319.5 * s, 239.5 * s * args.height / 480.0 * 640.0 / args.width)
    ctx.frames_reserve(2 * B)
    iters = [args.iters] * args.levels
    res = {}

    def timed(name, fn, units):
        fn(); ctx.synchronize()                       # warm-up (allocations)
        t0 = time.perf_counter()
        for _ in range(args.reps):
            fn()
        ctx.synchronize()
        dt = (time.perf_counter() - t0) / args.reps
        res[name] = dict(ms=1e3 * dt, per_s=units / dt)

    kw = dict(n_levels=args.levels, first_shift=args.first_shift, flags=DVO_UPLOAD_ASYNC | DVO_UPLOAD_DIRECT)    # the frames sit in pinned host memory that outlives the context
    timed("upload_ref_frames(bgr+depth: H2D, pyramid, Canny)", lambda: ctx.frames_upload_cameras(ref_b, ref_d, first_slot=0, **kw), B)
    timed("upload_now_frames(bgr only: H2D, pyramid, Canny)", lambda: ctx.frames_upload_cameras(now_b, None, first_slot=B, **kw), B)
    timed("frames_as_ref(selectedPts+enlistRefEdgePts)", lambda: ctx.frames_as_ref(0, 0, B), B)
    timed("frames_as_now(EDT+normalise+gradients+texels)", lambda: ctx.frames_as_now(B, 0, B), B)
    timed("align(%s)" % iters, lambda: ctx.enqueue(iters, flags=DVO_FLAG_IDENTITY_START), B)

    # the node's own wire format: mono8 + mono16 pyramids (RGBDFramePyd), row-major; built here by reading the device pyramid back
    pyr = []
    for i in range(D):
        ctx.frames_upload_cameras([ref[i][0]], [ref[i][1]], first_slot=0, n_levels=args.levels, first_shift=args.first_shift)
        lv = []
        for l in range(args.levels):
            g, d, _, _ = ctx.frame_level(0, l)
            lv.append((hold(np.ascontiguousarray(g)), hold(np.ascontiguousarray(d.astype(np.uint16)))))
        pyr.append(lv)
    pyr_b = [pyr[i % D] for i in range(B)]
    res["config_pyramid_bytes_per_frame"] = int(sum(g.nbytes + d.nbytes for g, d in pyr[0]))
    timed("upload_pyramids(mono8+mono16 row-major: H2D, import, Canny)", lambda: ctx.frames_upload_pyramids(pyr_b, first_slot=0, flags=DVO_UPLOAD_ASYNC | DVO_UPLOAD_DIRECT), B)
    if args.pinned:
        from rgbd_odometry_amd.capi import DVO_UPLOAD_MAPPED as _MAPPED
        timed("upload_pyramids, pinned levels pulled by the gather kernel (DVO_UPLOAD_MAPPED)", lambda: ctx.frames_upload_pyramids(pyr_b, first_slot=0, flags=DVO_UPLOAD_ASYNC | _MAPPED), B)

    def tracking_step():                              # every pair gets a fresh now frame against its resident reference
        ctx.frames_upload_cameras(now_b, None, first_slot=B, **kw)
        ctx.frames_as_now(B, 0, B)
        ctx.enqueue(iters, flags=DVO_FLAG_IDENTITY_START)
        return ctx.get_poses()
    timed("now frame in -> pose out (reference resident)", tracking_step, B)

    def tracking_step_fused():                        # as_now rides in the upload pipeline (now_first_pair)
        ctx.frames_upload_cameras(now_b, None, first_slot=B, now_first_pair=0, **kw)
        ctx.enqueue(iters, flags=DVO_FLAG_IDENTITY_START)
        return ctx.get_poses()
    timed("now frame in -> pose out, as_now inside the upload pipeline", tracking_step_fused, B)

    def tracking_step_chunked(ch=32):                 # same, in chunks: copies of chunk k+1 overlap the kernels of chunk k
        for b in range(0, B, ch):
            n = min(ch, B - b)
            ctx.frames_upload_cameras(now_b[b:b + n], None, first_slot=B + b, **kw)
            ctx.frames_as_now(B + b, b, n)
        ctx.enqueue(iters, flags=DVO_FLAG_IDENTITY_START)
        return ctx.get_poses()
    for ch in (64,):
        timed("now frame in -> pose out, chunks of %d" % ch, lambda ch=ch: tracking_step_chunked(ch), B)

    # the same camera frames as device buffers (DVO_UPLOAD_DEVICE): no PCIe
    import torch as _torch
    dev_now = [_torch.from_numpy(np.ascontiguousarray(now_b[i])).cuda() for i in range(D)]
    dev_ptrs = [dev_now[i % D].data_ptr() for i in range(B)]
    dkw = dict(n_levels=args.levels, first_shift=args.first_shift, first_slot=B, flags=DVO_UPLOAD_ASYNC)
    timed("now frames in HBM: gather, pyramid, Canny", lambda: ctx.frames_upload_cameras_device(dev_ptrs, None, args.height, args.width, **dkw), B)
    timed("now frames in HBM: gather, pyramid, Canny, EDT -> compact form", lambda: ctx.frames_upload_cameras_device(dev_ptrs, None, args.height, args.width, now_first_pair=0, **dkw), B)

    if args.pinned:      # pinned host buffers are device-accessible: the gather kernel pulls them over PCIe itself (one launch per 32 images)
        from rgbd_odometry_amd.capi import DVO_UPLOAD_MAPPED
        host_ptrs = [now_b[i].ctypes.data for i in range(B)]
        mkw = dict(dkw, flags=DVO_UPLOAD_ASYNC | DVO_UPLOAD_MAPPED)
        timed("now frames in pinned host memory, pulled by the gather kernel: pyramid, Canny", lambda: ctx.frames_upload_cameras_device(host_ptrs, None, args.height, args.width, **mkw), B)

        def tracking_step_zero_copy():
            ctx.frames_upload_cameras_device(host_ptrs, None, args.height, args.width, now_first_pair=0, **mkw)
            ctx.enqueue(iters, flags=DVO_FLAG_IDENTITY_START)
            return ctx.get_poses()
        timed("now frame in pinned host memory -> pose out, pulled by the gather kernel", tracking_step_zero_copy, B)

    def tracking_step_device():
        ctx.frames_upload_cameras_device(dev_ptrs, None, args.height, args.width, now_first_pair=0, **dkw)
        ctx.enqueue(iters, flags=DVO_FLAG_IDENTITY_START)
        return ctx.get_poses()
    timed("now frame in HBM -> pose out", tracking_step_device, B)

    def pair_step():                                  # both frames of every pair from the host
        ctx.frames_upload_cameras(ref_b, ref_d, first_slot=0, **kw)
        ctx.frames_upload_cameras(now_b, None, first_slot=B, **kw)
        ctx.frames_as_ref(0, 0, B)
        ctx.frames_as_now(B, 0, B)
        ctx.enqueue(iters, flags=DVO_FLAG_IDENTITY_START)
        return ctx.get_poses()
    timed("frame pair in -> pose out", pair_step, B)
    R, t = tracking_step()
    out = dict(config=dict(batch=B, width=args.width, height=args.height, levels=args.levels, first_shift=args.first_shift,
                           pinned=args.pinned, host_bytes_per_ref_frame=args.width * args.height * 7,
                           host_bytes_per_now_frame=args.width * args.height * 3),
               stages={k: v for k, v in res.items() if isinstance(v, dict)}, pyramid_bytes_per_frame=res.get("config_pyramid_bytes_per_frame"), mean_translation_m=float(np.linalg.norm(t, axis=1).mean()))
    print(json.dumps(out))
    ctx.close()


if __name__ == "__main__":
    main()
