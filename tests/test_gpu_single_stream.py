"""Single camera stream through the C ABI (the reference's actual use: one frame every ~30 ms, SolveDVO.cpp:1945, per frame
:2092-2109): H2D of a 640x480 BGR8 frame -> pyramid + Canny -> distance transform -> compact now level -> alignment (4 levels x 10
iterations) -> pose on the host.  Measured (profiles/r03_single_stream): 0.60 ms median / 0.71 ms worst at 30 Hz over 200 frames,
12.5 ms for the very first alignment of a process (code objects, lazily allocated buffers); round 5: 0.39-0.46 ms median.  The bound
here leaves a shared test box a factor of four: steady-state frames must stay below 2 ms in the median."""
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_single_stream_frame_latency_is_bounded():
    from rgbd_odometry_amd import DvoContext, frame_gen
    from rgbd_odometry_amd.capi import DVO_FLAG_IDENTITY_START
    iters = [10, 10, 10, 10]
    ref = frame_gen.camera_frame(5, 480, 640)
    nows = [frame_gen.camera_frame(5, 480, 640, shift=(1 + k % 3, -(k % 5)))[0] for k in range(4)]
    with DvoContext(1) as ctx:
        ctx.set_intrinsics(525.0, 525.0, 319.5, 239.5)
        ctx.frames_upload_cameras([ref[0]], [ref[1]], n_levels=4, first_shift=0, first_slot=0)
        ctx.frames_as_ref(0, 0, 1)

        def frame(k):
            t0 = time.perf_counter()
            ctx.frames_upload_cameras([nows[k % 4]], None, n_levels=4, first_shift=0, first_slot=1, now_first_pair=0)
            ctx.enqueue(iters, flags=DVO_FLAG_IDENTITY_START)
            R, t = ctx.get_poses()
            return time.perf_counter() - t0, R, t
        for k in range(5):                       # the first alignment of a process loads code objects and allocates buffers
            frame(k)
        times, poses = [], []
        for k in range(40):
            if k == 20:
                time.sleep(0.2)                   # an idle gap: the next frame must not pay a clock ramp either
            dt, R, t = frame(k)
            times.append(dt)
            poses.append((R[0].copy(), t[0].copy()))
        assert [ctx.level_texel_mode(0, l) for l in range(4)] == [2, 2, 2, 2]     # the stream runs on the compact now form
        assert float(np.median(times)) < 2e-3, times        # measured 0.4-0.5 ms (round 5); a shared test box gets a factor of four
        assert max(times) < 20e-3, times
        for k in range(4, 40):                    # the same now frame gives the same pose, bit for bit, every time
            assert np.array_equal(poses[k][0], poses[k - 4][0]) and np.array_equal(poses[k][1], poses[k - 4][1])
