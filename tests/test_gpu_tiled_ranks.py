"""The C-driven tiled mode (dvo_tiled_attach / dvo_align_pyramid_tiled: RCCL called from C between the accumulate and the update
kernels of every iteration) launched the way the driver launches it: `python -m torch.distributed.run ... bench.py --mode tiled`.

world = 1 always runs (one GPU: RCCL is really called, with a communicator of size one).  world = 2 needs two GPUs and skips
otherwise: every rank must end with bit-identical poses, equal to the oracle's (checked inside bench.py on rank 0 at world 1,
across ranks through the all_gather at world 2), and the JSON line must say so."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(world, extra, env=None):
    """`python bench.py --gpus N ...` exactly as a user (or the driver without its launcher) types it: for N > 1 bench.py itself
    starts the N ranks under torch.distributed.run as a child process and relays rank 0's line and the return code"""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "3", "--warmup", "1"] + extra
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900, env=e)


def _run(world, extra):
    r = _bench(world, ["--mode", "tiled"] + extra)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    return json.loads(line)


def test_bench_refuses_more_gpus_than_the_box_has():
    """`bench.py --gpus 8` on a box with fewer GPUs must not print a 1-GPU line and exit 0"""
    import torch
    n = torch.cuda.device_count()
    r = _bench(n + 1, ["--batch", "64", "--cpu-seconds", "0", "--no-extra-legs"])
    assert r.returncode != 0 and not [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert "device(s) visible" in r.stderr
    # a launcher that started another number of ranks than --gpus says
    r = _bench(2, ["--batch", "64", "--cpu-seconds", "0", "--no-extra-legs"], env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE" in r.stderr and not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_bench_one_gpu_line_names_its_ranks():
    r = _bench(1, ["--batch", "512", "--cpu-seconds", "0", "--no-extra-legs"])
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 1 and d["rccl_ranks"] == 1 and d["scaling"] == "weak"


def test_tiled_bench_one_rank_matches_the_oracle():
    d = _run(1, ["--width", "640", "--height", "480", "--levels", "4", "--cpu-seconds", "1"])
    assert d["n_gpus"] == 1 and d["rccl_ranks"] == 1 and d["config"]["mode"] == "tiled" and d["scaling"] == "strong"
    assert d["parity_check"]["pass"], d["parity_check"]
    assert d["parity_check"]["final_outputs_bit_equal"] and d["parity_check"]["energies_bit_equal"]
    assert d["roofline"]["kernel_ms"] > 0 and "cpu_baseline" in d


def test_tiled_bench_two_ranks_bit_identical():
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    d1 = _run(1, ["--width", "640", "--height", "480", "--levels", "4", "--cpu-seconds", "0"])
    d2 = _run(2, ["--width", "640", "--height", "480", "--levels", "4", "--cpu-seconds", "0"])
    assert d2["n_gpus"] == 2 and d2["rccl_ranks"] == 2 and d2["config"]["all_ranks_bit_identical"]
    assert d2["config"]["points_per_level"] == d1["config"]["points_per_level"]


def test_bench_shrinks_the_default_batch_to_the_free_memory():
    """a box with less free HBM than 40 000 resident pairs need: the default run must still produce its line, on a smaller batch,
    and say so"""
    r = _bench(1, ["--cpu-seconds", "0", "--no-extra-legs", "--assume-free-gb", "14"])
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["config"]["batch_reduced_from"] == 40000 and d["config"]["pairs_per_gpu"] == 256 * int((14e9 - 12e9) / (9.0 * 408000 + 0.3e6) / 256)
    assert d["value"] > 0 and d["roofline"]["frac"] > 0.2
    r = _bench(1, ["--batch", "512", "--cpu-seconds", "0", "--no-extra-legs"])
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert "batch_reduced_from" not in d["config"] and d["config"]["pairs_per_gpu"] == 512


# ---- rank > 0 of the C tiled path on ONE GPU: host threads as ranks, a loopback stand-in for ncclAllReduce ---------------------
LOOPBACK_DIR = os.path.join(ROOT, "tests", "loopback_collective")
LOOPBACK_SO = os.path.join(LOOPBACK_DIR, "libloopback_rccl.so")


def _loopback_library():
    src = os.path.join(LOOPBACK_DIR, "loopback_rccl.hip")
    if not os.path.exists(LOOPBACK_SO) or os.path.getmtime(LOOPBACK_SO) < os.path.getmtime(src):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O2", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-shared", "-o", LOOPBACK_SO, src])
    import ctypes as C
    lib = C.CDLL(LOOPBACK_SO)
    lib.loopback_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
    lib.loopback_destroy.argtypes = [C.POINTER(C.c_void_p), C.c_int]
    lib.loopback_calls.argtypes = [C.c_void_p]
    lib.loopback_calls.restype = C.c_long
    return lib


@pytest.mark.parametrize("world", [2, 3, 8])
def test_c_tiled_path_with_several_ranks_on_one_gpu(world, oracle):
    """dvo_align_pyramid_tiled has only ever met world size 1 on hardware (RCCL refuses two ranks on one device, the boxes have one).
    Here `world` host threads each drive their own context as rank r of `world`, with tests/loopback_collective standing in for
    ncclAllReduce (resolved from the library dvo_tiled_attach is given, like librccl.so.1): every rank works on ITS index range of
    every level, the sums meet in the all-reduce, every rank takes the update.  All ranks must end with identical bits; energies /
    best index / visible ratio bit-equal to the oracle's; the final outputs of the ranks' shards, concatenated, are the oracle's."""
    import ctypes as C
    import threading
    import numpy as np
    import oracle_lib
    from oracle_lib import rot_angle
    from rgbd_odometry_amd import DvoContext, SynthScene
    from rgbd_odometry_amd.capi import DVO_FLAG_FINAL_OUTPUTS
    lib = _loopback_library()
    sc = SynthScene(640, 480, 4, 9)
    lv = oracle_lib.scene_levels(sc, oracle)
    iters = [6, 0, 5, 7]
    R0, t0 = oracle.se3_exp(np.array([0.004, -0.003, 0.002, 0.001, -0.002, 0.0015]))
    ref = oracle.align_pyramid(iters, lv, sc.intrinsics, R0, t0)
    comms = (C.c_void_p * world)()
    assert lib.loopback_create(world, comms) == 0
    ctxs = []
    try:
        for r in range(world):
            ctx = DvoContext(1)
            ctx.set_intrinsics(*sc.intrinsics)
            for l, L in enumerate(sc.levels):
                ctx.set_ref_level_from_images(l, L.ref_edge, L.ref_depth, L.rows, L.cols)
                ctx.set_now_level(l, L.now_dt, L.now_gx, L.now_gy, L.rows, L.cols)
            ctx.tiled_attach(comms[r], r, world, LOOPBACK_SO)
            ctxs.append(ctx)
        out, err = [None] * world, [None] * world

        def run(r):
            try:
                out[r] = ctxs[r].align_pyramid_tiled(iters, R0, t0, flags=DVO_FLAG_FINAL_OUTPUTS)
            except Exception as e:          # a failing rank must not leave the others at the barrier for ever: the library times out
                err[r] = e
        for rep in range(2):                # the second alignment starts from scratch on the same attachment
            th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
            for x in th:
                x.start()
            for x in th:
                x.join(120)
            assert not any(x.is_alive() for x in th) and err == [None] * world, err
            for r in range(1, world):
                assert np.array_equal(out[r][0], out[0][0]) and np.array_equal(out[r][1], out[0][1]), r
            assert rot_angle(ref["R"], out[0][0]) <= 1e-5 and np.linalg.norm(ref["t"] - out[0][1]) <= 1e-4
            # one all-reduce per iteration of a level that is sharded over the ranks; small levels (round 5: one launch of one workgroup,
            # every rank runs them whole) need none
            solo = ctxs[0].wide_solo_levels()
            assert all(ctxs[r].wide_solo_levels() == solo for r in range(world))
            n_it = sum(it for l, it in enumerate(iters) if not (solo >> l) & 1)
            if not os.environ.get("DVO_TILED_SOLO_MAX") and not os.environ.get("DVO_TILED_PACKED"):
                assert solo != 0 and n_it > 0                                # this scene has levels on both sides of the threshold
            for r in range(world):
                assert lib.loopback_calls(comms[r]) == (rep + 1) * n_it, (r, lib.loopback_calls(comms[r]))
                assert not ctxs[r].tiled_graph_replayed()                    # the loopback refuses a capturing stream: direct submission
                for l, rp in ref["levels"].items():
                    e, b, ratio = ctxs[r].level_report(0, l, iters[l])
                    assert np.array_equal(e, rp["energy"]) and b == rp["best_idx"] and ratio == rp["visible_ratio"], (r, l)
        # shards: contiguous, disjoint, covering; final outputs of the last level rank by rank
        last = ref["last_level"]
        N = len(lv[last]["xyz"])
        eps_all = np.full(N, np.nan, np.float32)
        rep_all = np.full((N, 3), np.nan, np.float32)
        nxt = 0
        for r in range(world):
            first, count = ctxs[r].tiled_shard(last)
            assert first == nxt and count in (N // world, N // world + 1)
            nxt = first + count
            eps, reproj = ctxs[r].final_outputs(0, N)
            assert len(eps) == N
            eps_all[first:first + count] = eps[first:first + count]
            rep_all[first:first + count] = reproj[first:first + count]
        assert nxt == N
        want = ref["levels"][last]
        assert np.array_equal(eps_all, want["final_eps"]) and np.array_equal(rep_all, want["final_reproj"])
    finally:
        for ctx in ctxs:
            ctx.close()
        lib.loopback_destroy(comms, world)


def test_coarse_levels_as_team_launches_on_every_rank(oracle, monkeypatch):
    """round 6: the coarse levels of a large frame run as team launches of the fused kernel -- over several ranks every rank runs them
    WHOLE (identical inputs, a fixed order of additions: identical bits, no collective), only the finest level is sharded and
    all-reduced.  Two host threads as ranks 0 / 1 of 2 on one GPU (the library hands the coarse levels over in this arrangement only when
    asked to: two teams of 32 fit the chip side by side), 1920x1080x5."""
    import ctypes as C
    import threading
    import numpy as np
    import oracle_lib
    from oracle_lib import rot_angle
    from rgbd_odometry_amd import DvoContext, SynthScene
    monkeypatch.setenv("DVO_TILED_TEAM_SHARED", "1")
    if os.environ.get("DVO_WIDE_TEAM_MAX") or os.environ.get("DVO_TILED_PACKED"):
        pytest.skip("the default limits of the hand-over are what this test is about")
    world = 2
    lib = _loopback_library()
    sc = SynthScene(1920, 1080, 5, 4)
    lv = oracle_lib.scene_levels(sc, oracle)
    iters = [3, 3, 2, 3, 4]
    ref = oracle.align_pyramid(iters, lv, sc.intrinsics, np.eye(3), np.zeros(3))
    comms = (C.c_void_p * world)()
    assert lib.loopback_create(world, comms) == 0
    ctxs = []
    try:
        for r in range(world):
            ctx = DvoContext(1)
            ctx.set_intrinsics(*sc.intrinsics)
            for l, L in enumerate(sc.levels):
                ctx.set_ref_level_from_images(l, L.ref_edge, L.ref_depth, L.rows, L.cols)
                ctx.set_now_level_from_edges(l, (np.asarray(L.now_edge) != 0).astype(np.uint8) * 255, L.rows, L.cols)
            ctx.tiled_attach(comms[r], r, world, LOOPBACK_SO)
            ctxs.append(ctx)
        out, err = [None] * world, [None] * world

        def run(r):
            try:
                out[r] = ctxs[r].align_pyramid_tiled(iters, np.eye(3), np.zeros(3))
            except Exception as e:
                err[r] = e
        th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
        for x in th:
            x.start()
        for x in th:
            x.join(180)
        assert not any(x.is_alive() for x in th) and err == [None] * world, err
        assert np.array_equal(out[1][0], out[0][0]) and np.array_equal(out[1][1], out[0][1])
        assert rot_angle(ref["R"], out[0][0]) <= 1e-5 and np.linalg.norm(ref["t"] - out[0][1]) <= 1e-4
        for r in range(world):
            assert ctxs[r].wide_team_levels() == 0b11110, bin(ctxs[r].wide_team_levels())      # 67 k / 32 k / 14 k / 5.5 k points: one team launch
            assert lib.loopback_calls(comms[r]) == iters[0]                                      # only the finest level met the collective
            for l, rp in ref["levels"].items():
                e, b, ratio = ctxs[r].level_report(0, l, iters[l])
                assert np.array_equal(e, rp["energy"]) and b == rp["best_idx"] and ratio == rp["visible_ratio"], (r, l)
    finally:
        for ctx in ctxs:
            ctx.close()
        lib.loopback_destroy(comms, world)


def test_batch_bench_with_two_and_three_ranks_sharing_the_gpu():
    """The N > 1 path of the batch bench -- bench.py starting its own ranks under torch.distributed.run, per-rank scene seeds,
    barriers, MAX over ranks, the rank-0 line -- has never run on hardware (one-GPU boxes).  --ranks-share-gpu puts all ranks on
    device 0 over gloo: not a scaling measurement (the line says so), but every line of that path executes."""
    for world, extra in ((2, ["--batch", "256"]), (3, ["--total-pairs", "200"])):
        r = _bench(world, extra + ["--ranks-share-gpu", "--cpu-seconds", "0", "--no-extra-legs"])
        assert r.returncode == 0, r.stderr[-3000:]
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        assert len(lines) == 1                                     # rank 0 only
        d = json.loads(lines[0])
        # one GPU, no RCCL rank (the ranks meet over gloo): the line must say so
        assert d["n_gpus"] == 1 and d["rccl_ranks"] == 0 and d["config"]["ranks_share_one_gpu"] is True
        assert d["scaling"] == ("weak" if world == 2 else "strong")
        units = 256 * world if world == 2 else 200
        assert abs(d["value"] - units * d["steps"] / (d["ms_per_step"] * 1e-3 * d["steps"])) <= 1e-6 * d["value"]
        assert "cpu_baseline" not in d or d["cpu_baseline"] is None or world == 1
    r = _bench(2, ["--mode", "tiled", "--ranks-share-gpu"])
    assert r.returncode == 2 and "loopback" in r.stderr
