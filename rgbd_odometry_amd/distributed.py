"""Multi-GPU sharding of the edge-alignment hot path (one process per GPU, torch.distributed).

Two modes, as in SURVEY.md section 8(e):

* batch mode   -- independent frame pairs: `shard_range` gives every rank a contiguous block of pairs,
                  each rank runs the fused kernel on its block, no data-path collective at all.
* tiled mode   -- ONE large frame: the level's reference point list is split into contiguous index
                  ranges (= vertical strips of the reference image, because enlistRefEdgePts scans
                  column-major, SolveDVO.cpp:237-239); the now-level texels are replicated.  The only
                  coupling of the per-point work is the sum, so each iteration is
                      accumulate(own range) -> all_reduce(32 doubles, SUM) -> identical pose update on every rank.
                  `TiledAligner` drives that loop against any engine that offers the four `iter_*`
                  calls (the HIP engine: DvoContext; the CPU tests: an oracle-backed stand-in).

The collective is RCCL over xGMI when the process group backend is "nccl"; 256 bytes per iteration,
so it is latency- not bandwidth-bound, and every rank ends with bit-identical sums (ring/tree
all-reduce delivers the same reduced buffer to all ranks), hence identical pose steps.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import numpy as np

ACC_LEN = 32    # 29 accumulators padded to 32 doubles (dvo_amd.h: dvo_iter_accumulate)


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block decomposition: (first, count) of `rank`; the remainder goes to the lowest ranks."""
    if world < 1 or not (0 <= rank < world) or n_items < 0:
        raise ValueError("bad shard arguments")
    base, rem = divmod(n_items, world)
    count = base + (1 if rank < rem else 0)
    first = rank * base + min(rank, rem)
    return first, count


def shard_sizes(n_items: int, world: int) -> List[int]:
    return [shard_range(n_items, r, world)[1] for r in range(world)]


def whole_job_throughput(units_per_rank: int, steps: int, elapsed_local_s: float, device=None) -> float:
    """Batch mode accounting of bench.py: every rank processed units_per_rank * steps units in its own
    timed region; the job's rate is the total over all ranks divided by the SLOWEST rank's time
    (MAX all-reduce; RCCL when the group backend is nccl, gloo in the CPU tests)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size() if dist.is_initialized() else 1
    elapsed = elapsed_local_s
    if world > 1:
        t = torch.tensor([elapsed_local_s], dtype=torch.float64, device=device or "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return units_per_rank * steps * world / elapsed, elapsed


class TiledAligner:
    """Coarse-to-fine alignment of one frame pair whose point lists are sharded over the ranks of a
    process group (level schedule of SolveDVO::loop, SolveDVO.cpp:2097-2104).

    engine protocol (pair 0 of a DvoContext implements it; see tests/test_distributed_cpu.py for a CPU one):
        n_points(level) -> int                       total reference points of the level
        iter_begin(level, max_iters, R, t)
        iter_accumulate(level, first, count, acc)    partial sums of points [first, first+count) into `acc`
        iter_update(level, itr, n_total, acc)        the 6-DoF update from the (reduced) sums
        iter_end(level) -> dict(R, t, energy, best_idx, visible_ratio)
        new_acc() -> tensor of ACC_LEN float64 on the engine's device; acc_ptr(t) -> what iter_* take
    """

    def __init__(self, engine, group=None, force_collective: bool = False):
        self.engine = engine
        self.group = group
        self.force_collective = force_collective      # run the all-reduce even at world size 1 (testing)
        import torch.distributed as dist
        self.dist = dist
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0

    def run_level(self, level: int, iters: int, R, t):
        eng = self.engine
        n_total = eng.n_points(level)
        first, count = shard_range(n_total, self.rank, self.world)
        acc = eng.new_acc()
        eng.iter_begin(level, iters, R, t)
        for itr in range(iters):
            eng.iter_accumulate(level, first, count, eng.acc_ptr(acc))
            if self.world > 1 or self.force_collective:
                self.dist.all_reduce(acc, op=self.dist.ReduceOp.SUM, group=self.group)
            eng.iter_update(level, itr, n_total, eng.acc_ptr(acc))
        return eng.iter_end(level)

    def align(self, iters_per_level: Sequence[int], R, t):
        R, t = np.array(R, dtype=np.float64), np.array(t, dtype=np.float64)
        reports = {}
        for level in range(len(iters_per_level) - 1, -1, -1):          # :2097
            if iters_per_level[level] <= 0:                            # :2099
                continue
            rep = self.run_level(level, iters_per_level[level], R, t)
            R, t = rep["R"], rep["t"]
            reports[level] = rep
        return dict(R=R, t=t, levels=reports)


class HipTiledEngine:
    """`TiledAligner` engine on top of pair `pair` of a DvoContext; the accumulators live in a torch
    tensor so that torch.distributed (RCCL) can all-reduce them in place on the context's stream."""

    def __init__(self, ctx, pair: int = 0, device: Optional[str] = None):
        import torch
        self.torch = torch
        self.ctx, self.pair = ctx, pair
        self.device = torch.device(device or ("cuda:%d" % torch.cuda.current_device()))
        # one stream for kernels and collectives: torch's current stream
        ctx.set_stream(torch.cuda.current_stream(self.device).cuda_stream)

    def n_points(self, level):
        return self.ctx.n_points(level, self.pair)

    def new_acc(self):
        return self.torch.zeros(ACC_LEN, dtype=self.torch.float64, device=self.device)

    @staticmethod
    def acc_ptr(t):
        return t.data_ptr()

    def iter_begin(self, level, max_iters, R, t):
        self.ctx.iter_begin(level, max_iters, R, t, pair=self.pair)

    def iter_accumulate(self, level, first, count, acc_ptr):
        self.ctx.iter_accumulate(level, first, count, acc_ptr, pair=self.pair)

    def iter_update(self, level, itr, n_total, acc_ptr):
        self.ctx.iter_update(level, itr, n_total, acc_ptr, pair=self.pair)

    def iter_end(self, level):
        return self.ctx.iter_end(level, pair=self.pair)
