#!/usr/bin/env python3
"""Single camera stream latency (the reference's use: ros::Rate(35), SolveDVO.cpp:1945, per frame :2092-2109).

One 640x480 now frame at a time through the C ABI: H2D of the BGR8 frame -> pyramid + Canny -> distance transform ->
compact now level -> alignment (4 levels x 10 iterations) against a resident reference -> pose on the host.  Per frame the
host wall time (call -> pose) and the GPU time (HIP events around the frame's work on the context stream) are recorded for
    back_to_back   frames fed as fast as the host can
    paced          frames at --hz (default 30): the GPU idles ~33 ms between frames
    paced_warm     the same with the engine's keep-warm launches between frames (dvo_set_keep_warm)
usage: single_stream.py [--frames 200] [--hz 30] [--out profiles/r03_single_stream/summary.json]
"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from rgbd_odometry_amd import DvoContext, frame_gen
from rgbd_odometry_amd.capi import DVO_FLAG_IDENTITY_START


def stats(a):
    a = np.asarray(a) * 1e3
    return dict(median_ms=float(np.median(a)), p95_ms=float(np.percentile(a, 95)), max_ms=float(a.max()), mean_ms=float(a.mean()))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=200)
    ap.add_argument("--hz", type=float, default=30.0)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--levels", type=int, default=4)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--warm-us", type=int, default=200, help="period of the keep-warm launches")
    ap.add_argument("--out", default="")
    ap.add_argument("--order", default="back_to_back,paced,paced_warm", help="which runs, in which order")
    ap.add_argument("--upload-flags", type=int, default=0, help="flags of dvo_frames_upload_cameras: 1 = ASYNC (do not wait for the frame stage "
                    "before enqueueing the alignment), 16 = MAPPED (the pinned frame is pulled by a kernel, no copy into the engine's mirror)")
    ap.add_argument("--warm-busy-us", type=int, default=0, help="> 0: keep-warm as dvo_set_keep_warm2(busy, warm_us) instead of short launches")
    args = ap.parse_args()
    iters = [args.iters] * args.levels
    ctx = DvoContext(1)
    s = args.width / 640.0
    ctx.set_intrinsics(525.0 * s, 525.0 * s, 319.5 * s, 239.5 * args.height / 480.0)
    stream = torch.cuda.Stream()
    ctx.set_stream(stream.cuda_stream)

    def pin(a):
        t = torch.empty(a.shape, dtype=torch.from_numpy(a).dtype, pin_memory=True)
        t.numpy()[...] = a
        return t.numpy()
    ref = frame_gen.camera_frame(5, args.height, args.width)
    nows = [pin(frame_gen.camera_frame(5, args.height, args.width, shift=(1 + k % 3, -(k % 5)))[0]) for k in range(8)]
    ctx.frames_upload_cameras([ref[0]], [ref[1]], n_levels=args.levels, first_shift=0, first_slot=0)
    ctx.frames_as_ref(0, 0, 1)

    def frame(k, ev):
        ev[0].record(stream)
        ctx.frames_upload_cameras([nows[k % len(nows)]], None, n_levels=args.levels, first_shift=0, first_slot=1, now_first_pair=0,
                                  flags=args.upload_flags)
        ctx.enqueue(iters, flags=DVO_FLAG_IDENTITY_START)
        ev[1].record(stream)
        return ctx.get_poses()

    def run(paced, warm):
        if warm and args.warm_busy_us > 0:
            ctx.set_keep_warm2(args.warm_busy_us, args.warm_us)
        else:
            ctx.set_keep_warm(args.warm_us if warm else 0)
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.frames)]
        if not paced:
            for k in range(5):
                frame(k, ev[0])
        wall = []
        period = 1.0 / args.hz
        t_next = time.perf_counter()
        for k in range(args.frames):
            if paced:
                t_next += period
                while True:
                    d = t_next - time.perf_counter()
                    if d <= 0:
                        break
                    time.sleep(min(d, 0.002))
            t0 = time.perf_counter()
            frame(k, ev[k])
            wall.append(time.perf_counter() - t0)
        torch.cuda.synchronize()
        gpu = [a.elapsed_time(b) * 1e-3 for a, b in ev]
        ctx.set_keep_warm(0)
        return dict(host_wall=stats(wall), gpu_events=stats(gpu), first_10_wall_ms=[round(w * 1e3, 3) for w in wall[:10]])
    out = {"config": dict(width=args.width, height=args.height, levels=args.levels, iters=args.iters, frames=args.frames, hz=args.hz,
                          warm_period_us=args.warm_us, upload_flags=args.upload_flags,
                          per_frame="H2D BGR8 -> pyramid + Canny -> distance transform -> compact now level -> alignment -> pose on the host")}
    for name in args.order.split(","):
        out[name] = run(name != "back_to_back", name == "paced_warm")
        print(name, "host wall", out[name]["host_wall"], "gpu events", out[name]["gpu_events"], flush=True)
    out["texel_modes"] = [ctx.level_texel_mode(0, l) for l in range(args.levels)]
    out["launch_shape"] = ctx.last_launch_shape()
    if args.out:
        os.makedirs(os.path.dirname(args.out), exist_ok=True)
        json.dump(out, open(args.out, "w"), indent=1)
    ctx.close()


if __name__ == "__main__":
    main()
