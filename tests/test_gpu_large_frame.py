"""BASELINE configs[4] at its workload on ONE GPU: a single 4096x3072 frame pair, 5-level pyramid (seed 7),
through every single-pair path -- the C-driven all-CU path (dvo_align_pyramid_wide), the Python-driven tiled loop with
the RCCL all-reduce forced at world size 1 (TiledAligner), and the C-driven tiled entry point with a raw RCCL communicator
(dvo_tiled_attach / dvo_align_pyramid_tiled) -- each against the CPU oracle: energies, best index and visible ratio
bit-equal, pose within 1e-5 rad / 1e-4 m.  (More than one GPU is not available to the tests; the multi-rank logic is
covered by the gloo world-size-2 tests in test_distributed_cpu.py and, on the GPU, by host threads as ranks over a loopback
all-reduce in test_gpu_tiled_ranks.py.)"""
import os

import numpy as np
import pytest

import oracle_lib
from oracle_lib import rot_angle

pytestmark = pytest.mark.gpu

ROT_TOL, TRANS_TOL = 1e-5, 1e-4


@pytest.fixture(scope="module")
def scene4096(oracle):
    from rgbd_odometry_amd import SynthScene
    sc = SynthScene(4096, 3072, 5, 7)
    assert [(L.rows, L.cols) for L in sc.levels] == [(3072, 4096), (1536, 2048), (768, 1024), (384, 512), (192, 256)]
    lv = oracle_lib.scene_levels(sc, oracle)
    iters = [10, 10, 10, 10, 10]
    ref = oracle.align_pyramid(iters, lv, sc.intrinsics, np.eye(3), np.zeros(3))
    return sc, lv, iters, ref


@pytest.fixture(scope="module")
def ctx4096(scene4096):
    from rgbd_odometry_amd import DvoContext
    sc, lv, iters, ref = scene4096
    ctx = DvoContext(1)
    ctx.set_intrinsics(*sc.intrinsics)
    for l, L in enumerate(sc.levels):
        xyz, _ = ctx.set_ref_level_from_images(l, L.ref_edge, L.ref_depth, L.rows, L.cols)
        assert len(xyz) == len(lv[l]["xyz"])
        ctx.set_now_level(l, L.now_dt, L.now_gx, L.now_gy, L.rows, L.cols)
    yield ctx
    ctx.close()


def _check_reports(ctx, ref, iters):
    for l, rep in ref["levels"].items():
        e, b, ratio = ctx.level_report(0, l, iters[l])
        assert np.array_equal(e, rep["energy"]), (l, e, rep["energy"])
        assert b == rep["best_idx"] and ratio == rep["visible_ratio"], l


def test_config5_wide_path(scene4096, ctx4096):
    sc, lv, iters, ref = scene4096
    assert len(lv[0]["xyz"]) > 500000                      # ~0.63 M reference points at level 0
    Rw, tw = ctx4096.align_pyramid_wide(iters, np.eye(3), np.zeros(3))
    if not os.environ.get("DVO_TILED_PACKED"):      # lists built by the engine's own kernels: every level on the packed step kernel (round 5) ...
        team = ctx4096.wide_team_levels()           # ... or, round 6, inside the fused kernel's team launches (the coarse levels of a large frame)
        assert ctx4096.wide_packed_levels() | team == (1 << len(iters)) - 1 and not (ctx4096.wide_packed_levels() & team)
        if not os.environ.get("DVO_WIDE_TEAM_MAX"):
            assert team == 0b11111, bin(team)       # 29 k / 72 k points: a team of 32 in one XCD; 159 k / 327 k: a team of 128; 629 k (one rank): a team of 256
    _check_reports(ctx4096, ref, iters)
    assert rot_angle(ref["R"], Rw) <= ROT_TOL and np.linalg.norm(ref["t"] - tw) <= TRANS_TOL


@pytest.mark.parametrize("team", [0, 64, 128, 256, 32])
def test_config5_team_over_all_xcds(scene4096, team):
    """the fused kernel with ONE pair spread over the whole GPU: a team of 8 x g1 workgroups, sums exchanged inside every XCD
    and then between the 8 XCDs (auto = a launch per tier of level sizes, 32 / 128 / 256 members); same bits as the oracle, run-to-run deterministic, final outputs
    of the best iterate included"""
    from rgbd_odometry_amd import DvoContext
    from rgbd_odometry_amd.capi import DVO_FLAG_FINAL_OUTPUTS
    sc, lv, iters, ref = scene4096
    with DvoContext(1, team_size=team) as ctx:
        ctx.set_intrinsics(*sc.intrinsics)
        for l, L in enumerate(sc.levels):
            ctx.set_ref_level_from_images(l, L.ref_edge, L.ref_depth, L.rows, L.cols)
            ctx.set_now_level(l, L.now_dt, L.now_gx, L.now_gy, L.rows, L.cols)
        R, t = ctx.align_batch(iters, np.eye(3)[None], np.zeros((1, 3)), flags=DVO_FLAG_FINAL_OUTPUTS)
        blk, g, packed = ctx.last_launch_shape()
        assert packed and blk == 512 and g == (team if team else 256), (blk, g, packed)      # auto: tiers of 32 / 128 / 256, the finest level last
        _check_reports(ctx, ref, iters)
        assert rot_angle(ref["R"], R[0]) <= ROT_TOL and np.linalg.norm(ref["t"] - t[0]) <= TRANS_TOL
        last = ref["levels"][ref["last_level"]]
        feps, frep = ctx.final_outputs(0, len(last["final_eps"]))
        assert np.array_equal(feps, last["final_eps"]) and np.array_equal(frep, last["final_reproj"], equal_nan=True)
        R2, t2 = ctx.align_batch(iters, np.eye(3)[None], np.zeros((1, 3)), flags=DVO_FLAG_FINAL_OUTPUTS)
        assert np.array_equal(R, R2) and np.array_equal(t, t2)
        if team in (0, 256):                      # the same with the now levels in the compact form
            ctx.now_prepare()
            R3, t3 = ctx.align_batch(iters, np.eye(3)[None], np.zeros((1, 3)), flags=DVO_FLAG_FINAL_OUTPUTS)
            _check_reports(ctx, ref, iters)
            assert np.abs(R3 - R).max() <= 1e-12 and np.abs(t3 - t).max() <= 1e-12


def test_config5_exact_energy_sweep_over_the_whole_chip(scene4096):
    """every energy of the 4096x3072 pyramid from the exact sweep (engine_variant 5): 256 workgroups add their three limbs through the
    two-stage exchange -- integers below 2^53, exact in any order -- and round once; same bits as the oracle's exact sum"""
    from rgbd_odometry_amd import DvoContext
    sc, lv, iters, ref = scene4096
    with DvoContext(1, engine_variant=5) as ctx:
        ctx.set_intrinsics(*sc.intrinsics)
        for l, L in enumerate(sc.levels):
            ctx.set_ref_level_from_images(l, L.ref_edge, L.ref_depth, L.rows, L.cols)
            ctx.set_now_level(l, L.now_dt, L.now_gx, L.now_gy, L.rows, L.cols)
        ctx.now_prepare()
        R, t = ctx.align_batch(iters, np.eye(3)[None], np.zeros((1, 3)))
        assert ctx.last_launch_shape()[1] == 256
        _check_reports(ctx, ref, iters)
        assert [ctx.level_energy_sweeps(0, l) for l in range(5)] == iters
        assert rot_angle(ref["R"], R[0]) <= ROT_TOL and np.linalg.norm(ref["t"] - t[0]) <= TRANS_TOL


def test_config5_tiled_loop_with_forced_collective(scene4096, ctx4096):
    """the multi-GPU loop (accumulate -> all_reduce of 32 doubles -> update) with the RCCL collective really issued"""
    import torch
    import torch.distributed as dist
    from rgbd_odometry_amd.distributed import HipTiledEngine, TiledAligner
    sc, lv, iters, ref = scene4096
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29541", rank=0, world_size=1,
                            device_id=torch.device("cuda", torch.cuda.current_device()))
    try:
        res = TiledAligner(HipTiledEngine(ctx4096), force_collective=True).align(iters, np.eye(3), np.zeros(3))
        torch.cuda.synchronize()
    finally:
        dist.destroy_process_group()
        ctx4096.use_own_stream()
    for l, rep in ref["levels"].items():
        got = res["levels"][l]
        assert np.array_equal(got["energy"], rep["energy"]) and got["best_idx"] == rep["best_idx"], l
        assert got["visible_ratio"] == rep["visible_ratio"], l
    assert rot_angle(ref["R"], res["R"]) <= ROT_TOL and np.linalg.norm(ref["t"] - res["t"]) <= TRANS_TOL


def test_config5_tiled_from_c_with_rccl(scene4096, ctx4096):
    """dvo_tiled_attach + dvo_align_pyramid_tiled with a raw ncclComm_t (world size 1): bit-identical to the wide path"""
    from rgbd_odometry_amd.capi import RcclComm
    sc, lv, iters, ref = scene4096
    Rw, tw = ctx4096.align_pyramid_wide(iters, np.eye(3), np.zeros(3))
    comm = RcclComm(RcclComm.unique_id(), 0, 1)
    try:
        ctx4096.tiled_attach(comm.comm, 0, 1, RcclComm.RCCL)
        Rt, tt = ctx4096.align_pyramid_tiled(iters, np.eye(3), np.zeros(3))
        _check_reports(ctx4096, ref, iters)
        assert np.array_equal(Rt, Rw) and np.array_equal(tt, tw)
        assert rot_angle(ref["R"], Rt) <= ROT_TOL and np.linalg.norm(ref["t"] - tt) <= TRANS_TOL
        # skipped levels and a warm start
        it2 = [3, 0, 4, 0, 2]
        ref2 = oracle_lib.load().align_pyramid(it2, lv, sc.intrinsics, Rw, tw)
        R2, t2 = ctx4096.align_pyramid_tiled(it2, Rw, tw)
        _check_reports(ctx4096, ref2, it2)
        assert rot_angle(ref2["R"], R2) <= ROT_TOL and np.linalg.norm(ref2["t"] - t2) <= TRANS_TOL
    finally:
        ctx4096.tiled_detach()
        comm.close()


def test_tiled_from_c_needs_a_communicator():
    from rgbd_odometry_amd import DvoContext, DvoError
    with DvoContext(1) as ctx:
        with pytest.raises(DvoError):
            ctx.align_pyramid_tiled([3], np.eye(3), np.zeros(3))
        with pytest.raises(DvoError):
            ctx.tiled_attach(0, 0, 1)
