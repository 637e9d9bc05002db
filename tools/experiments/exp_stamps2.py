"""Phase anatomy of the packed fused kernel from the stamps build (make STAMPS=1; DVO_LIB_VARIANT=_stamps).
usage: exp_stamps2.py [B] [block] [engine_variant] [alias]"""
import os, sys
os.environ["DVO_LIB_VARIANT"] = os.environ.get("DVO_STAMPS_VARIANT", "_stamps")   # e.g. _stamps_r4base for an A/B
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
import numpy as np, torch
from rgbd_odometry_amd import DvoContext, SynthScene
from rgbd_odometry_amd.capi import DVO_FLAG_IDENTITY_START, DVO_FLAG_FINAL_OUTPUTS
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
block = int(sys.argv[2]) if len(sys.argv) > 2 else 0
variant = int(sys.argv[3]) if len(sys.argv) > 3 else 0
alias = int(sys.argv[4]) if len(sys.argv) > 4 else 0
sc = SynthScene(640, 480, 4, 1000)
ctx = DvoContext(B, block_threads=block, engine_variant=variant, debug_alias_mod=alias)
ctx.set_intrinsics(*sc.intrinsics)
for l, L in enumerate(sc.levels):
    ctx.set_ref_level_from_images(l, L.ref_edge, L.ref_depth, L.rows, L.cols, pair=0)
    ctx.set_now_level(l, L.now_dt, L.now_gx, L.now_gy, L.rows, L.cols, pair=0)
ctx.replicate_pairs(1)
if os.environ.get('PREP', '1') == '1':
    ctx.now_prepare()
iters = [10, 10, 10, 10]
for rep in range(3):
    ctx.enqueue(iters, flags=DVO_FLAG_IDENTITY_START | DVO_FLAG_FINAL_OUTPUTS); ctx.synchronize()
    st = ctx.debug_stamps(B // 2)
print("B=%d block=%d variant=%d alias=%d modes=%s  (cycles per iteration; s_memtime ticks)" % (
    B, block, variant, alias, [ctx.level_texel_mode(B // 2, l) for l in range(4)]))
tot = 0
for l in range(4):
    n = max(1, int(st[l, 4]))
    print("level %d: points %7.0f  reduce %6.0f  update %6.0f  barrier %6.0f   setup %7.0f (once)  (iters %d)" % (
        l, st[l, 0] / n, st[l, 1] / n, st[l, 2] / n, st[l, 3] / n, st[l, 5] / 3, n))
    tot += (st[l, 0] + st[l, 1] + st[l, 2] + st[l, 3] + st[l, 5]) / 3
print("sum of stamped phases per alignment: %.0f ticks" % tot)
