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
