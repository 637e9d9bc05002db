#!/usr/bin/env python3
"""round 5: which scene / level / iteration of a parity sweep differs from the oracle, and by how much.
usage (GPU box): python tools/experiments/r05_parity_diag.py W H levels iters n_scenes"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib
from rgbd_odometry_amd import DvoContext, SynthScene
from rgbd_odometry_amd.capi import DVO_FLAG_IDENTITY_START, DVO_FLAG_FINAL_OUTPUTS
W, H, nl, it, D = (int(x) for x in sys.argv[1:6])
kw = {}
for a in sys.argv[6:]:
    k, v = a.split("="); kw[k] = int(v)
oracle = oracle_lib.load() if hasattr(oracle_lib, "load") else oracle_lib.Oracle()
scenes = [SynthScene(W, H, nl, 1000 + i) for i in range(D)]
iters = [it] * nl
with DvoContext(D, **kw) as ctx:
    ctx.set_intrinsics(*scenes[0].intrinsics)
    for i, sc in enumerate(scenes):
        for l, L in enumerate(sc.levels):
            ctx.set_ref_level_from_images(l, L.ref_edge, L.ref_depth, L.rows, L.cols, pair=i)
            ctx.set_now_level_from_edges(l, (np.asarray(L.now_edge) != 0).astype(np.uint8) * 255, L.rows, L.cols, pair=i)
    ctx.enqueue(iters, flags=DVO_FLAG_IDENTITY_START | DVO_FLAG_FINAL_OUTPUTS)
    R, t = ctx.get_poses()
    print("launch shape", ctx.last_launch_shape())
    bad = 0
    for i, sc in enumerate(scenes):
        lv = oracle_lib.scene_levels(sc, oracle)
        ref = oracle.align_pyramid(iters, lv, sc.intrinsics, np.eye(3), np.zeros(3))
        for l, rep in sorted(ref["levels"].items(), reverse=True):
            e, bi, ratio = ctx.level_report(i, l, iters[l])
            if not (np.array_equal(e, rep["energy"]) and bi == rep["best_idx"] and ratio == rep["visible_ratio"]):
                d = np.nonzero(e != rep["energy"])[0]
                k = int(d[0]) if len(d) else -1
                print("scene %d level %d: %d energies differ, first at iteration %d: gpu %r oracle %r (rel %.2e); best %d / %d ratio %r / %r; N %d; modes %s exact %s" % (
                    i, l, len(d), k, float(e[k]) if k >= 0 else None, float(rep["energy"][k]) if k >= 0 else None,
                    abs(float(e[k]) - float(rep["energy"][k])) / max(abs(float(rep["energy"][k])), 1e-30) if k >= 0 else 0.0,
                    bi, rep["best_idx"], ratio, rep["visible_ratio"], len(lv[l]["xyz"]), ctx.level_texel_mode(i, l), ctx.level_exact_fallback(i, l)))
                bad += 1
                break
    print("%d of %d scenes differ" % (bad, D))
