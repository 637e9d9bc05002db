"""single-pair latency of the team kernel vs team size (and the wide path), for one frame size
usage: exp_team_single.py W H levels [iters]"""
import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from rgbd_odometry_amd import DvoContext, SynthScene
W, H, nl = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
it = int(sys.argv[4]) if len(sys.argv) > 4 else 10
sc = SynthScene(W, H, nl, 7)
iters = [it] * nl
res = {}
for team in [int(x) for x in os.environ.get("TEAMS", "1,2,4,8,16,32,64,128,256,0").split(",")]:
    with DvoContext(1, team_size=team) as ctx:
        ctx.set_intrinsics(*sc.intrinsics)
        for l, L in enumerate(sc.levels):
            ctx.set_ref_level_from_images(l, L.ref_edge, L.ref_depth, L.rows, L.cols)
            ctx.set_now_level(l, L.now_dt, L.now_gx, L.now_gy, L.rows, L.cols)
        if os.environ.get("PREP"): ctx.now_prepare()        # the compact now form at once (otherwise it appears after 16 alignments: the loops below would mix the two)
        R, t = ctx.align_batch(iters, np.eye(3)[None], np.zeros((1, 3)))
        t0 = time.perf_counter()
        for _ in range(20): R, t = ctx.align_batch(iters, np.eye(3)[None], np.zeros((1, 3)))
        res[team] = 1e3 * (time.perf_counter() - t0) / 20
        if team == 0 and "wide" not in res:
            Rw, tw = ctx.align_pyramid_wide(iters, np.eye(3), np.zeros(3))
            t0 = time.perf_counter()
            for _ in range(20): Rw, tw = ctx.align_pyramid_wide(iters, np.eye(3), np.zeros(3))
            res["wide"] = 1e3 * (time.perf_counter() - t0) / 20
            res["maxdiff_vs_wide"] = float(max(np.abs(R[0] - Rw).max(), np.abs(t[0] - tw).max()))
print("%dx%dx%d, %d it/level, N0 = %d: ms per alignment by team size (0 = auto):" % (W, H, nl, it, ctx.n_points(0) if False else 0), {k: (round(v, 3) if isinstance(v, float) else v) for k, v in res.items()})
