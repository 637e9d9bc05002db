#!/usr/bin/env python3
"""One worker of bench.py's all-cores CPU baseline: aligns ONE synthetic scene with the CPU oracle over and over for
`budget` seconds and prints "<alignments> <seconds>".  Deliberately light (numpy + ctypes, no torch, no HIP): bench.py starts
one of these per host core.  Test infrastructure -- the oracle is the thing timed beside the GPU, never a product path.

    python tests/cpu_baseline_worker.py W H levels iters seed budget_s
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))


def main():
    W, H, levels, iters, seed = (int(x) for x in sys.argv[1:6])
    budget = float(sys.argv[6])
    import oracle_lib
    from rgbd_odometry_amd.synth import SynthScene        # the scene generator only (host code); the package root imports no GPU code
    oracle = oracle_lib.load()
    sc = SynthScene(W, H, levels, seed)
    lv = oracle_lib.scene_levels(sc, oracle)
    it = [iters] * levels
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < budget:
        oracle.align_pyramid(it, lv, sc.intrinsics, np.eye(3), np.zeros(3))
        n += 1
    print(n, time.perf_counter() - t0)


if __name__ == "__main__":
    main()
