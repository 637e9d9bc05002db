"""HBM capacity of a resident batch: the 16-byte texel slab of a level is a virtual address range whose
chunks are backed by memory only when some pair's texels are asked for (dvo_capi.cpp: ensure_texels / map_texels).  The engine's
own now levels exist in the compact form only, so 40 000 resident 640x480x4 pairs fit one MI355X (with every pair's texels
reserved AND backed, as in rounds 1-3, 28 000 did not: 9.7 MB per pair).  Everything that reads texels -- dvo_get_now_level,
dvo_eval_points -- still works for any pair: its chunk is mapped and decoded from the compact form on demand."""
import os
import subprocess
import sys

import numpy as np
import pytest

import oracle_lib
from oracle_lib import rot_angle

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _u8(a):
    return (np.asarray(a) != 0).astype(np.uint8) * 255


def test_40000_resident_pairs_align_and_decode_on_demand(oracle):
    from rgbd_odometry_amd import DvoContext, SynthScene
    from rgbd_odometry_amd.capi import DVO_FLAG_IDENTITY_START, DVO_FLAG_FINAL_OUTPUTS
    import torch
    free, total = torch.cuda.mem_get_info()
    if free < 200 * (1 << 30):
        pytest.skip("needs ~150 GB of free HBM")
    B, D = 40000, 8
    scenes = [SynthScene(640, 480, 4, 1000 + i) for i in range(D)]
    iters = [10, 10, 10, 10]
    with DvoContext(B) as ctx:
        ctx.set_intrinsics(*scenes[0].intrinsics)
        for p, sc in enumerate(scenes):
            for l, L in enumerate(sc.levels):
                ctx.set_ref_level_from_images(l, L.ref_edge, L.ref_depth, L.rows, L.cols, pair=p)
                ctx.set_now_level_from_edges(l, _u8(L.now_edge), L.rows, L.cols, pair=p)
        ctx.replicate_pairs(D)
        ctx.enqueue(iters, flags=DVO_FLAG_IDENTITY_START | DVO_FLAG_FINAL_OUTPUTS)
        R, t = ctx.get_poses()
        assert ctx.level_texel_mode(B - 1, 0) == 2                      # the compact form: no texels were ever touched
        for p in list(range(D)) + list(range(B - D, B)):
            sc = scenes[p % D]
            lv = oracle_lib.scene_levels(sc, oracle)
            ref = oracle.align_pyramid(iters, lv, sc.intrinsics, np.eye(3), np.zeros(3))
            for l, rep in ref["levels"].items():
                e, b, ratio = ctx.level_report(p, l, iters[l])
                assert np.array_equal(e, rep["energy"]) and b == rep["best_idx"] and ratio == rep["visible_ratio"], (p, l)
            assert rot_angle(ref["R"], R[p]) <= 1e-5 and np.linalg.norm(ref["t"] - t[p]) <= 1e-4
        # texels on demand, far down the batch: decoded from the compact form into a freshly mapped chunk, bit-equal to the oracle's images
        p = B - 3
        lv = oracle_lib.scene_levels(scenes[p % D], oracle)
        for l in (0, 3):
            dt, gx, gy = ctx.get_now_level(l, pair=p)
            assert np.array_equal(dt, lv[l]["dt"]) and np.array_equal(gx, lv[l]["gx"]) and np.array_equal(gy, lv[l]["gy"])
        got = ctx.eval_points(0, np.eye(3), np.zeros(3), pair=p)
        L0 = lv[0]
        want = oracle.eval_points(0, L0["xyz"], L0["dt"], L0["gx"], L0["gy"], L0["rows"], L0["cols"], scenes[p % D].intrinsics, np.eye(3), np.zeros(3))
        assert np.array_equal(got["eps"], want["eps"]) and np.array_equal(got["J"], want["J"])


def test_suite_subset_on_sparse_texel_slabs():
    """the compact-now and frame tests again with DVO_TEX_SLAB=sparse (every context's texel slab sparse, whatever its size): covers
    the deferred texel pass for images the compact form cannot hold (too many distances, a far pixel, a rank step), float images,
    replication and decode on demand -- in a child process, because the policy is read once per process"""
    env = dict(os.environ, DVO_TEX_SLAB="sparse")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "tests/test_gpu_compact_now.py", "tests/test_gpu_frames.py",
                        "tests/test_gpu_parity.py", "-p", "no:cacheprovider"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1000:]


@pytest.mark.parametrize("env_kv", [("DVO_EDT_BAND_T", "5"), ("DVO_EDT_BAND_T", "2"), ("DVO_EDT_FUSED", "0")],
                         ids=["band-T5-512-threads", "band-T2-256-threads", "three-pass-stage"])
def test_frame_suites_on_every_shape_of_the_distance_transform_stage(env_kv):
    """round 6: the stage that turns edge masks into compact now levels has three forms -- the band kernel with five tile rows per
    512-thread workgroup (large batches), with two per 256 threads (small batches, the default of these tests) and the three-pass stage
    of rounds 3-5 (columns, rows, rank pack: wide or tall images, and the images on the band stage's list).  The compact-now and frame
    suites with each of them forced -- in a child process, because the choice is read once per process."""
    env = dict(os.environ, **{env_kv[0]: env_kv[1]})
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "tests/test_gpu_compact_now.py", "tests/test_gpu_frames.py",
                        "tests/test_gpu_sparse_scenes.py", "-p", "no:cacheprovider"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1000:]
