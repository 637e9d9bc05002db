"""Multi-process CPU tests (gloo, world_size 2) of the N>1 host logic: batch sharding and the tiled-mode
loop accumulate -> all_reduce(32 doubles) -> identical update.  The engine here is an oracle-backed
stand-in with the same `iter_*` protocol as the HIP engine (rgbd_odometry_amd.distributed.HipTiledEngine)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from rgbd_odometry_amd.distributed import ACC_LEN, TiledAligner, shard_range, shard_sizes  # noqa: E402


def test_shard_range_partitions_exactly():
    for n in (0, 1, 7, 256, 1000, 18349):
        for world in (1, 2, 3, 4, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0
            for (f0, c0), (f1, _) in zip(spans, spans[1:]):
                assert f0 + c0 == f1                         # contiguous, in rank order
            assert spans[-1][0] + spans[-1][1] == n
            sizes = shard_sizes(n, world)
            assert max(sizes) - min(sizes) <= 1              # balanced
    with pytest.raises(ValueError):
        shard_range(10, 2, 2)


class OracleTiledEngine:
    """CPU stand-in for HipTiledEngine: per-shard sums from the oracle, update from its state machine."""

    def __init__(self, oracle, levels, K):
        self.o, self.lv, self.K = oracle, levels, K
        self.st = None

    def n_points(self, level):
        return len(self.lv[level]["xyz"])

    def new_acc(self):
        return torch.zeros(ACC_LEN, dtype=torch.float64)

    @staticmethod
    def acc_ptr(t):
        return t

    def iter_begin(self, level, max_iters, R, t):
        self.st = self.o.state_begin(R, t)
        self.energy = np.zeros(max_iters, np.float32)
        self.broke = False

    def iter_accumulate(self, level, first, count, acc):
        L = self.lv[level]
        R, t = self.o.state_pose(self.st)
        # slots 29..31: the three limbs of the shard's exact sum of eps^2 -- they add exactly in the all-reduce (round 6)
        a = self.o.accumulate32(level, L["xyz"], first, count, L["dt"], L["gx"], L["gy"], L["rows"], L["cols"], self.K, R, t)
        acc.copy_(torch.from_numpy(a))

    def iter_update(self, level, itr, n_total, acc):
        if self.broke:
            return
        a = acc.numpy()
        sum_eps2 = self.o.e2_from_limbs(a[29:32], float(a[27]))      # the correctly rounded exact sum over ALL shards: no order, no sharding enters
        e, broke, _ = self.o.state_update(self.st, itr, n_total, a[21:27], sum_eps2, int(a[28]))
        self.energy[itr] = e
        self.broke = broke

    def iter_end(self, level):
        R, t = self.o.state_finish(self.st)
        return dict(R=R, t=t, energy=self.energy, best_idx=self.st.bestItr, visible_ratio=self.st.bestRatio)


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import oracle_lib
        from rgbd_odometry_amd import SynthScene
        oracle = oracle_lib.load()
        sc = SynthScene(160, 120, 3, 7)
        lv = oracle_lib.scene_levels(sc, oracle)
        iters = [6, 6, 6]
        al = TiledAligner(OracleTiledEngine(oracle, lv, sc.intrinsics))
        assert al.world == world and al.rank == rank
        res = al.align(iters, np.eye(3), np.zeros(3))
        # every rank must hold the identical pose (they all applied the same reduced sums)
        buf = torch.from_numpy(np.concatenate([res["R"].ravel(), res["t"]]))
        gathered = [torch.zeros_like(buf) for _ in range(world)]
        dist.all_gather(gathered, buf)
        same = all(torch.equal(gathered[0], g) for g in gathered)
        if rank == 0:
            ref = oracle.align_pyramid(iters, lv, sc.intrinsics, np.eye(3), np.zeros(3))
            ok_e = all(np.array_equal(res["levels"][l]["energy"], ref["levels"][l]["energy"]) for l in ref["levels"])
            ok_b = all(res["levels"][l]["best_idx"] == ref["levels"][l]["best_idx"] for l in ref["levels"])
            dR = float(np.abs(res["R"] - ref["R"]).max())
            dt = float(np.abs(res["t"] - ref["t"]).max())
            q.put((same, ok_e, ok_b, dR, dt))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2])
def test_tiled_mode_two_ranks_gloo(world):
    """point shards + all_reduce of the 29 sums reproduce the single-process oracle run"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    same, ok_e, ok_b, dR, dt = q.get(timeout=5)
    assert same, "ranks disagree on the final pose"
    assert ok_e and ok_b, "energies / best iterate differ from the single-process oracle"
    assert dR < 1e-9 and dt < 1e-9          # sums are associated differently (2 partials): ~1e-16 noise


def _rate_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from rgbd_odometry_amd.distributed import whole_job_throughput
        dist.barrier()
        value, elapsed = whole_job_throughput(1024, 20, 0.050 * (rank + 1))      # rank 1 is the slow one
        q.put((rank, value, elapsed))
    finally:
        dist.destroy_process_group()


def test_batch_mode_whole_job_throughput_two_ranks():
    """bench.py's N>1 accounting: total units of all ranks / MAX over ranks of the timed region"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_rate_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    got = sorted(q.get(timeout=5) for _ in range(2))
    for rank, value, elapsed in got:
        assert abs(elapsed - 0.100) < 1e-12                       # the slower rank's time, on both ranks
        assert abs(value - 2 * 1024 * 20 / 0.100) < 1e-6


def test_batch_mode_sharding_covers_all_pairs():
    """BASELINE config 4: 256 pairs over 8 GPUs -> 32 contiguous pairs each, no overlap"""
    owners = np.full(256, -1)
    for r in range(8):
        f, c = shard_range(256, r, 8)
        assert c == 32
        assert (owners[f:f + c] == -1).all()
        owners[f:f + c] = r
    assert (owners >= 0).all()


def test_bench_gpus_flag_is_checked_before_anything_runs():
    """`bench.py --gpus N` must never print an N-GPU line from fewer GPUs: without a launcher it refuses when
    fewer than N devices are visible; under a launcher WORLD_SIZE has to agree with --gpus.  Neither path touches a GPU."""
    import subprocess
    import sys
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("this box could really run two ranks: covered by tests/test_gpu_tiled_ranks.py")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 3 and "device(s) visible" in r.stderr and "{" not in r.stdout
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4"], capture_output=True, text=True,
                       env=dict(env, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"), timeout=300)
    assert r.returncode == 2 and "WORLD_SIZE" in r.stderr and "{" not in r.stdout
