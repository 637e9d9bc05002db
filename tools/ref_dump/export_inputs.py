#!/usr/bin/env python3
"""Exports the inputs of the golden cases (tests/golden/make_golden.py: the seeded synthetic scenes of SURVEY.md section 8d) as raw
little-endian float32 files for tools/ref_dump/ref_dump.cpp.  Runs anywhere this repository is built (host code only: the scene
generator and the CPU oracle's enlistRefEdgePts restatement, which produces the 3 x N / 2 x N lists runIterations reads).

    python tools/ref_dump/export_inputs.py <out dir>
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))


def main():
    out = sys.argv[1]
    import make_golden
    import oracle_lib
    from rgbd_odometry_amd.synth import SynthScene
    oracle = oracle_lib.load()
    os.makedirs(out, exist_ok=True)
    names = []
    for name, W, H, nl, it, seeds in make_golden.CASES:
        for seed in seeds:
            key = f"{name}_s{seed}"
            names.append(key)
            d = os.path.join(out, key)
            os.makedirs(d, exist_ok=True)
            sc = SynthScene(W, H, nl, seed)
            lv = oracle_lib.scene_levels(sc, oracle)
            with open(os.path.join(d, "meta.txt"), "w") as f:
                f.write("%d %r %r %r %r\n" % ((nl,) + tuple(float(np.float32(k)) for k in sc.intrinsics)))
                for l, L in enumerate(lv):
                    f.write("%d %d %d %d\n" % (it, L["rows"], L["cols"], len(L["xyz"])))
            for l, L in enumerate(lv):
                # N x 3 / N x 2 row-major here == 3 x N / 2 x N column-major in Eigen; images are column-major already
                np.ascontiguousarray(L["xyz"], np.float32).tofile(os.path.join(d, f"level_{l}_xyz.f32"))
                np.ascontiguousarray(L["uv"], np.float32).tofile(os.path.join(d, f"level_{l}_uv.f32"))
                for k in ("dt", "gx", "gy"):
                    np.ascontiguousarray(L[k], np.float32).tofile(os.path.join(d, f"level_{l}_{k}.f32"))
    with open(os.path.join(out, "cases.txt"), "w") as f:
        f.write("\n".join(names) + "\n")
    print("wrote %d cases to %s" % (len(names), out))


if __name__ == "__main__":
    main()
