#!/bin/bash
OUT=$PWD/gpurun_out/r04_exit_segv; mkdir -p $OUT
REPO=$PWD; export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/min_x -o t -- python3 $REPO/tools/experiments/r04_exit_segv_min.py 2 0 x > $OUT/min_x.log 2>&1
echo "torch only, cross-stream event wait: rc=$?"
cat > /tmp/fr.py <<'PY'
import os, sys
sys.path.insert(0, os.environ["REPO"]); sys.path.insert(0, os.path.join(os.environ["REPO"], "tests"))
import numpy as np, torch
from rgbd_odometry_amd import DvoContext, frame_gen
from rgbd_odometry_amd.capi import DVO_UPLOAD_ASYNC
what = sys.argv[1]
B = 8
fr = [frame_gen.camera_frame(100 + i, 480, 640) for i in range(B)]
ctx = DvoContext(B)
ctx.set_intrinsics(525.0, 525.0, 319.5, 239.5)
ctx.frames_reserve(2 * B)
if what in ("upload", "all"):
    ctx.frames_upload_cameras([f[0] for f in fr], [f[1] for f in fr], first_slot=0, n_levels=4, first_shift=0, flags=0)
if what == "all":
    ctx.frames_as_ref(0, 0, B); ctx.frames_as_now(0, 0, B)
ctx.synchronize()
ctx.close()
print("done", what, flush=True)
PY
for w in none upload all; do
  REPO=$REPO timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/fr_$w -o t -- python3 /tmp/fr.py $w > $OUT/fr_$w.log 2>&1
  echo "engine frames path, $w: rc=$?"
  rm -rf $OUT/fr_$w
done
rm -rf $OUT/min_x
