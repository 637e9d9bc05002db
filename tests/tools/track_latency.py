#!/usr/bin/env python3
"""Single-stream tracking latency of the C++ replay (examples/track_demo.cpp): writes a short synthetic sequence
as OpenCV-XML frame files and replays it.  Lives under tests/ because it uses the oracle (test infrastructure) as the generator of
the node's pyramid format.  usage: tests/tools/track_latency.py [width height levels iters n_frames]"""
import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import frame_io, oracle_lib
from rgbd_odometry_amd import frame_gen
W, H, nl, it, n = [int(x) for x in (sys.argv[1:6] if len(sys.argv) > 5 else (320, 240, 4, 50, 16))]
o = oracle_lib.load()
d = os.environ.get("FRAMES_DIR") or tempfile.mkdtemp()
os.makedirs(d, exist_ok=True)
for i in range(n):
    bgr, depth = frame_gen.camera_frame(5, H, W, shift=((i % 16) // 2, -(i % 16)))
    frame_io.write_frame_xml(os.path.join(d, "framemono_%04d.xml" % i), o.build_pyramid(bgr, depth, nl, 0))
s = W / 640.0
if os.environ.get("ONLY_GENERATE"):
    print(d); sys.exit(0)
out = subprocess.run([os.path.join(ROOT, "rgbd_odometry_amd", "lib", "track_demo"), d, "0", str(n - 1), "1", str(nl),
                      repr(525.0 * s), repr(525.0 * s), repr(319.5 * s), repr(239.5 * s), str(it), os.path.join(d, "poses.txt")],
                     capture_output=True, text=True, env={**os.environ, **({"TRACK_DEMO_VERBOSE": "1"} if os.environ.get("VERBOSE") else {})})
print("%dx%d levels %d iters %d frames %d" % (W, H, nl, it, n)); print(out.stdout, out.stderr[-500:])
