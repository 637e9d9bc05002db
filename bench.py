#!/usr/bin/env python3
"""Benchmark of the edge-alignment hot path (SolveDVO::runIterations + level schedule) on MI355X.

Two modes (BASELINE.json configs):

  --mode batch (default; configs[1], [2], [3])
      A "step" is one pass of the hot path over one batch of synthetic frame pairs: ONE launch of the fused alignment
      kernel over `--batch` pairs per GPU (inputs already resident in HBM, in the form the engine's own preprocessing
      stage writes them), followed by the delivery of the poses to the host.  Default workload = configs[1]:
      640x480, 4-level pyramid, 10 iterations per level.  Multi-GPU: independent pairs sharded over the ranks, no
      data-path collective (weak scaling: the per-GPU batch is fixed; --total-pairs N: strong scaling, configs[3]).

  --mode tiled (configs[4])
      A "step" is ONE alignment of one large frame (default 4096x3072, 5 levels) whose reference point lists are
      sharded over the GPUs: per iteration accumulate(own shard) -> ncclAllReduce(32 doubles) over RCCL/xGMI ->
      identical update on every rank, all enqueued from C (dvo_align_pyramid_tiled).  Strong scaling by construction.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W [--mode tiled]

torch.distributed (RCCL) is used for the barrier and the MAX over ranks of the timed region (and, in tiled mode, to hand
rank 0's ncclUniqueId to the other ranks; the per-iteration all-reduce is RCCL called from C).
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch  # imported before the HIP library on purpose: one HIP runtime per process (capi.py)


def keep_host_memory_mapped():
    """Host memory hygiene (profiles/r03_single_stream): a process that hands multi-hundred-kilobyte
    blocks back to the kernel between GPU submissions (free -> munmap; here: the pose arrays numpy allocates per step) can
    stall its own GPU queues for 10-30 ms each time on this pool.  Tell glibc to keep what it has: M_MMAP_THRESHOLD (-3) and
    M_TRIM_THRESHOLD (-1) to 1 GiB, M_TOP_PAD (-2) to 64 MiB."""
    try:
        import ctypes
        libc = ctypes.CDLL("libc.so.6")
        libc.mallopt(-3, 1 << 30)
        libc.mallopt(-1, 1 << 30)
        libc.mallopt(-2, 64 << 20)
    except Exception:
        pass

HBM_PEAK_GBPS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
HBM_COPY_GBPS = 6290.0      # measured float4 copy (same guide): what a streaming kernel reaches

#: the sources the fused alignment kernels are built from: their hash ties profiles/pmc_traffic.json to a kernel build
KERNEL_SOURCES = ["dvo_fused.hip", "dvo_point_pk.h", "dvo_palette.h", "dvo_kernels.hip", "dvo_kernel_common.h", "dvo_device_math.h", "dvo_tiled_step.h"]


def kernel_source_hash():
    import hashlib
    h = hashlib.sha256()
    for name in KERNEL_SOURCES:
        with open(os.path.join(ROOT, "rgbd_odometry_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", choices=["batch", "tiled"], default="batch")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=0, help="timed steps (default: 40 in batch mode = 2.1 s, 200 in tiled mode)")
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=40000,
                    help="frame pairs per GPU per step (weak scaling: fixed per GPU).  40000 resident 640x480x4 pairs = 140 GB of HBM (3.5 MB each: "
                         "compact now form, point lists, outputs; their 16-byte texels are address space only, dvo_capi.cpp: ensure_texels); "
                         "one step = 54 ms, so the driver's --steps 20 times 1.07 s")
    ap.add_argument("--total-pairs", type=int, default=0,
                    help="strong scaling: this many pairs IN TOTAL per step, split over the GPUs by shard_range "
                         "(BASELINE configs[3]: --total-pairs 256); overrides --batch")
    ap.add_argument("--width", type=int, default=0, help="level-0 width (default 640; tiled mode 4096)")
    ap.add_argument("--height", type=int, default=0, help="level-0 height (default 480; tiled mode 3072)")
    ap.add_argument("--levels", type=int, default=0, help="pyramid levels (default 4; tiled mode 5)")
    ap.add_argument("--iters", type=int, default=10, help="iterations per level")
    ap.add_argument("--distinct", type=int, default=32, help="distinct synthetic scenes (cycled over the batch)")
    ap.add_argument("--block", type=int, default=0, help="workgroup size of the fused kernel (0 = default)")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="CPU-oracle baseline budget per leg (0 = skip both legs)")
    ap.add_argument("--no-cpu-all-cores", action="store_true", help="skip the one-oracle-process-per-host-core leg of the CPU baseline")
    ap.add_argument("--no-final-outputs", action="store_true")
    ap.add_argument("--normal-matrix", action="store_true", help="DVO_FLAG_NORMAL_MATRIX: also accumulate H = sum w J^T J per iterate (cost measurement)")
    ap.add_argument("--inflight", type=int, default=0, help="points in flight per lane (1/2/4; 0 = default)")
    ap.add_argument("--lds-point-bytes", type=int, default=0, help="LDS bytes per workgroup for resident points (0 auto, <0 none)")
    ap.add_argument("--variant", type=int, default=0, help="engine_variant (0 auto, 1 = one-point-per-lane fused kernel, 4 = 16-byte texels)")
    ap.add_argument("--team", type=int, default=0, help="team_size: workgroups per pair for small batches (0 auto, 1 off)")
    ap.add_argument("--debug-alias", type=int, default=0, help="diagnostics: pair p reads data of pair p %% N")
    ap.add_argument("--float-now-levels", action="store_true",
                    help="install the now levels as caller-supplied float images (dvo_set_now_level: 16-byte texels; the generic "
                         "compact form only after 16 alignments or with --prepare) instead of through the engine's distance transform")
    ap.add_argument("--prepare", action="store_true", help="with --float-now-levels: dvo_now_prepare at set-up (round 2's headline)")
    ap.add_argument("--float-ref-lists", action="store_true",
                    help="install the reference points as caller-supplied 3xN float lists (dvo_set_ref_level, the literal drop-in of INTEGRATION.md) "
                         "instead of edge + depth images; round 6: a list that verifies as an enlistRefEdgePts list gets its compact twin")
    ap.add_argument("--no-extra-legs", "--no-frames-leg", dest="no_extra_legs", action="store_true",
                    help="skip the extra (never `value`) measurements: 16-byte texels, camera frames in host memory -> poses out, ...")
    ap.add_argument("--ranks-share-gpu", action="store_true",
                    help="testing only (batch mode): the N ranks of --gpus N all use device 0 and meet over gloo, so that the N > 1 code path "
                         "-- launcher, barriers, MAX over ranks, rank-0 line -- runs on a one-GPU box; the line says so (config.ranks_share_one_gpu) "
                         "and is NOT a scaling measurement")
    ap.add_argument("--assume-free-gb", type=float, default=0.0, help="testing: pretend this much HBM is free when sizing the resident batch")
    a = ap.parse_args()
    tiled = a.mode == "tiled"
    a.width = a.width or (4096 if tiled else 640)
    a.height = a.height or (3072 if tiled else 480)
    a.levels = a.levels or (5 if tiled else 4)
    a.steps = a.steps or (200 if tiled else 40)
    return a


def u8_edges(a):
    return (np.asarray(a) != 0).astype(np.uint8) * 255


def build_batch(ctx, args, rank):
    """Generate `distinct` scenes, extract their reference points on the GPU (enlistRefEdgePts), install their now levels
    and make every pair slot of the context resident in HBM.

    Now levels go in the way the engine's own preprocessing produces them: edge map -> exact distance transform ->
    the level's compact form (dvo_set_now_level_from_edges; rgbd_odometry_amd/csrc/dvo_frames.hip).  That stage writes
    the compact form natively and nothing else -- there is no separate re-encoding pass to leave out of the timing.
    --float-now-levels installs the reference's three float images instead (dvo_set_now_level)."""
    from rgbd_odometry_amd import SynthScene
    D = max(1, min(args.distinct, args.batch))
    scenes = [SynthScene(args.width, args.height, args.levels, 1000 + rank * D + i) for i in range(D)]
    ctx.set_intrinsics(*scenes[0].intrinsics)
    t0 = time.perf_counter()
    for i, sc in enumerate(scenes):
        for l, L in enumerate(sc.levels):
            xyz, _ = ctx.set_ref_level_from_images(l, L.ref_edge, L.ref_depth, L.rows, L.cols, pair=i)   # GPU enlistRefEdgePts
            if args.float_ref_lists:      # the literal drop-in: the reference's 3xN float list (dvo_set_ref_level), after the now level's size is known
                ctx.set_ref_level(l, xyz, pair=i)
            if args.float_now_levels:
                ctx.set_now_level(l, L.now_dt, L.now_gx, L.now_gy, L.rows, L.cols, pair=i)
            else:
                ctx.set_now_level_from_edges(l, u8_edges(L.now_edge), L.rows, L.cols, pair=i)
    ctx.replicate_pairs(D)          # slots D.. <- device copies of the D distinct pairs (own HBM each)
    ctx.synchronize()
    args.install_s = time.perf_counter() - t0
    args.now_prepare_ms = 0.0
    if args.float_now_levels and args.prepare:
        t0 = time.perf_counter()
        ctx.now_prepare()
        ctx.synchronize()
        args.now_prepare_ms = 1e3 * (time.perf_counter() - t0)
    return scenes


# ---- CPU baseline (the oracle = the reference path restated; test infrastructure, used here only as the thing timed
# ---- beside the GPU and as the parity checker) -------------------------------------------------------------------------
def cpu_baseline(args, scenes, iters, budget_s, max_n=2000):
    """single thread like the reference (EIGEN_DONT_PARALLELIZE, SolveDVO.h:14), timed over the span the reference times
    (SolveDVO.cpp:2092-2109): all levels of one alignment, preprocessing excluded"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    oracle = oracle_lib.load()
    lvs = [oracle_lib.scene_levels(sc, oracle) for sc in scenes[:8]]
    n, t0 = 0, time.perf_counter()
    while True:
        lv = lvs[n % len(lvs)]
        oracle.align_pyramid(iters, lv, scenes[n % len(lvs)].intrinsics, np.eye(3), np.zeros(3))
        n += 1
        el = time.perf_counter() - t0
        if el >= budget_s or n >= max_n:
            break
    return dict(value=n / el, unit="aligns/s", cores=1, kind="port",
                sample=f"{n} alignments of the same workload ({len(lvs)} distinct scenes) in {el:.1f} s, "
                       f"1 thread of {os.cpu_count()} host cores, oracle/ built -O2 -ffp-contract=off"), oracle, lvs


def host_cpu_info():
    """what the CPU legs ran on: the cores this process may use (affinity), the container's CPU quota (cgroup cpu.max), model, SMT"""
    info = {"os_cpu_count": os.cpu_count()}
    try:
        info["sched_affinity"] = len(os.sched_getaffinity(0))
    except Exception:
        info["sched_affinity"] = None
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                info["cgroup_cpu_max"] = " ".join(txt)
                info["cgroup_cpus"] = None if txt[0] == "max" else float(txt[0]) / float(txt[1])
            else:
                q = int(txt[0])
                per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read().split()[0])
                info["cgroup_cpu_max"] = "%d %d" % (q, per)
                info["cgroup_cpus"] = None if q < 0 else q / per
            break
        except Exception:
            continue
    try:
        model, phys, cores_per, siblings = None, set(), None, None
        for line in open("/proc/cpuinfo"):
            k, _, v = line.partition(":")
            k, v = k.strip(), v.strip()
            if k == "model name" and model is None:
                model = v
            elif k == "physical id":
                phys.add(v)
            elif k == "cpu cores" and cores_per is None:
                cores_per = int(v)
            elif k == "siblings" and siblings is None:
                siblings = int(v)
        info.update(model=model, sockets=len(phys) or None, cores_per_socket=cores_per,
                    threads_per_core=(siblings // cores_per) if cores_per and siblings else None)
    except Exception:
        pass
    try:
        info["loadavg_1min"] = float(open("/proc/loadavg").read().split()[0])
    except Exception:
        pass
    return info


def usable_cpus(info):
    n = info.get("sched_affinity") or info.get("os_cpu_count") or 1
    if info.get("cgroup_cpus"):
        n = max(1, min(n, int(info["cgroup_cpus"] + 0.5)))
    return n


def cpu_baseline_openmp(args, oracle, lvs, scenes, iters, budget_s, single_rate):
    """BASELINE.md section 4 (ii), the form it asks for: ONE process, OpenMP over the independent pairs of the batch configuration
    (oracle/dvo_oracle_batch.cpp: every thread runs the single-threaded reference path on its own pair), as many threads as this
    process has CPUs (affinity, capped by the container's quota); per-thread statistics say how evenly the cores delivered"""
    info = host_cpu_info()
    threads = usable_cpus(info)
    n_pairs = int(max(threads, min(20000, single_rate * threads * budget_s * 0.6)))
    r = oracle.align_batch_omp(iters, lvs, scenes[0].intrinsics, n_pairs, n_threads=threads)
    per = r["thread_pairs"] / np.maximum(r["thread_seconds"], 1e-9)
    return dict(value=n_pairs / r["seconds"], unit="aligns/s", cores=int(r["threads"]), kind="port",
                sample=f"{n_pairs} alignments ({len(lvs)} distinct scenes) in {r['seconds']:.1f} s, one process, OpenMP over pairs, "
                       f"{r['threads']} threads",
                per_thread_aligns_per_s={"min": float(per.min()), "median": float(np.median(per)), "max": float(per.max())},
                speedup_over_one_thread=(n_pairs / r["seconds"]) / single_rate, host=info)


def cpu_baseline_all_cores(args, iters, budget_s):
    """one alignment stream per host core -- one light worker process per core
    (tests/cpu_baseline_worker.py: numpy + the oracle library, no torch, no HIP), each aligning its own scene for
    `budget_s` seconds; the rate is the sum over the workers"""
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        cores = os.cpu_count() or 1
    worker = os.path.join(ROOT, "tests", "cpu_baseline_worker.py")
    procs = [subprocess.Popen([sys.executable, worker, str(args.width), str(args.height), str(args.levels), str(iters[0]),
                               str(1000 + i % 8), str(budget_s)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
             for i in range(cores)]
    n_tot, rate, ok, per = 0, 0.0, 0, []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=budget_s * 6 + 120)
            n, el = out.split()[-2:]
            n_tot += int(n); rate += int(n) / float(el); ok += 1
            per.append(int(n) / float(el))
        except Exception:
            p.kill()
    info = host_cpu_info()
    return dict(value=rate, unit="aligns/s", cores=ok, kind="port",
                sample=f"{n_tot} alignments, one oracle process per CPU of the affinity mask ({ok} of {cores} workers reported) for {budget_s:.0f} s each",
                per_worker_aligns_per_s=({"min": min(per), "median": float(np.median(per)), "max": max(per)} if per else None),
                host=info,
                note="one PROCESS per CPU of the affinity mask, whatever the container's quota (host.cgroup_cpus): where the quota is below "
                     "the mask the workers share it and each runs at quota / workers of a core -- cpu_baseline_openmp sizes its team by the quota")


# ---- extra legs (never `value`) ----------------------------------------------------------------------------------------
def frames_leg(args, iters):
    """Rows f1+f2: the same workload fed from camera frames.  (a) in (pinned) HOST memory -- BGR8 + depth uploaded over PCIe,
    pyramid / Canny / distance transform / point extraction on the GPU -- to poses on the host; (b) `resident`: the frames
    already in the frame store (uploaded, pyramid + Canny done): per step the now frames' distance transform -> compact
    form + the alignment, i.e. what one alignment costs INCLUDING the production of its now level.  Bounded: 256 pairs."""
    from rgbd_odometry_amd import frame_gen
    from rgbd_odometry_amd import DvoContext
    from rgbd_odometry_amd.capi import DVO_FLAG_IDENTITY_START, DVO_UPLOAD_ASYNC, DVO_UPLOAD_MAPPED
    B, D = 256, 8

    def pin(a):
        t = torch.empty(a.shape, dtype=torch.from_numpy(a).dtype, pin_memory=True)
        t.numpy()[...] = a
        return t.numpy()
    # D distinct scenes, but every one of the B frames in its OWN pinned buffer: the link moves B frames' worth of bytes per step
    ref = [frame_gen.camera_frame(100 + i, args.height, args.width) for i in range(D)]
    now = [frame_gen.camera_frame(100 + i, args.height, args.width, shift=(1 + i % 2, -2))[0] for i in range(D)]
    ref_b, ref_d = [pin(ref[i % D][0]) for i in range(B)], [pin(ref[i % D][1]) for i in range(B)]
    now_b = [pin(now[i % D]) for i in range(B)]
    ctx = DvoContext(B)
    s = args.width / 640.0
    ctx.set_intrinsics(525.0 * s, 525.0 * s, 319.5 * s, 239.5 * args.height / 480.0)
    ctx.frames_reserve(2 * B)
    # the frames sit in pinned host memory that outlives the context and that the GPU can address (torch pin_memory): the engine
    # pulls them over PCIe with a kernel (DVO_UPLOAD_MAPPED) instead of one DMA per image
    kw = dict(n_levels=args.levels, first_shift=0, flags=DVO_UPLOAD_ASYNC | DVO_UPLOAD_MAPPED)

    def pair_step():
        ctx.frames_upload_cameras(ref_b, ref_d, first_slot=0, **kw)
        ctx.frames_upload_cameras(now_b, None, first_slot=B, **kw)
        ctx.frames_as_ref(0, 0, B)
        ctx.frames_as_now(B, 0, B)
        ctx.enqueue(iters, flags=DVO_FLAG_IDENTITY_START)
        return ctx.get_poses()

    def now_step():                                  # the now levels are produced inside the upload pipeline (now_first_pair)
        ctx.frames_upload_cameras(now_b, None, first_slot=B, now_first_pair=0, **kw)
        ctx.enqueue(iters, flags=DVO_FLAG_IDENTITY_START)
        return ctx.get_poses()

    def resident_step():
        ctx.frames_as_now(B, 0, B)
        ctx.enqueue(iters, flags=DVO_FLAG_IDENTITY_START)
        return ctx.get_poses()

    # the same now frames as DEVICE buffers (a decoder / camera driver that lands frames in HBM): everything the GPU does per
    # now frame -- landing copy, pyramid, Canny, distance transform -> compact now level, alignment -- and no PCIe
    # every one of the B frames in its OWN device buffer (D distinct scenes): the engine reads such frames where they are (round 6), and
    # B pointers to D buffers would let the caches serve what a real ring of frames has to stream from HBM
    dev_now = [torch.from_numpy(np.ascontiguousarray(now[i % D])).cuda() for i in range(B)]
    dev_ptrs = ctx.pointer_table([t_.data_ptr() for t_ in dev_now])                 # the decoder's ring: the table is built once

    def device_step():
        ctx.frames_upload_cameras_device(dev_ptrs, None, args.height, args.width, n_levels=args.levels, first_shift=0,
                                         first_slot=B, flags=DVO_UPLOAD_ASYNC, now_first_pair=0)
        ctx.enqueue(iters, flags=DVO_FLAG_IDENTITY_START)
        return ctx.get_poses()
    out = {}
    for name, fn, reps in (("frame_pairs_per_s", pair_step, 3), ("now_frames_per_s_reference_resident", now_step, 3),
                           ("now_frames_per_s_camera_frames_in_hbm", device_step, 10),
                           ("now_frames_per_s_frames_resident", resident_step, 10)):
        fn()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        out[name] = reps * B / (time.perf_counter() - t0)
    out["now_level_texel_modes"] = [ctx.level_texel_mode(0, l) for l in range(args.levels)]
    out["note"] = ("never `value`.  frame_pairs / now_frames_per_s_reference_resident are PCIe-inclusive: %dx%d BGR8 (+ depth f32 for "
                   "reference frames) in pinned host memory, every frame in its own buffer, pulled by a kernel (DVO_UPLOAD_MAPPED) -> pyramid, Canny, distance transform -> compact now level, edge points on "
                   "the GPU -> %s iterations -> poses on the host.  now_frames_per_s_camera_frames_in_hbm: the same with the BGR8 frames "
                   "already in device memory (DVO_UPLOAD_DEVICE): all the GPU work of a now frame, no PCIe.  now_frames_per_s_frames_resident: the now frames already in the "
                   "frame store (pyramid + Canny done): distance transform -> compact now level + alignment per step.  Batches of %d; "
                   "now_level_texel_modes 2 = the alignment read the natively produced compact form" % (args.width, args.height, iters, B))
    ctx.close()
    return out


def energy_ulps(e_gpu, e_ref):
    """(compared, differing, largest distance in float32 ulps) of two energy traces.  An energy is (float)sqrt(sum of (double)eps^2)
    (SolveDVO.cpp:689, :1312; the oracle adds in list order, the GPU per lane, wave and workgroup): where the two double sums straddle a
    float rounding boundary the narrowed values differ by ONE ulp -- observed at a rate of 5e-5 per energy (profiles/r05_final/
    parity_sweep.txt); the update never reads the energy, so nothing else moves unless the best-iterate choice (:696) hangs on it."""
    a = np.ascontiguousarray(e_gpu, dtype=np.float32).ravel()
    b = np.ascontiguousarray(e_ref, dtype=np.float32).ravel()
    if a.shape != b.shape:
        return int(max(a.size, b.size)), int(max(a.size, b.size)), 1 << 30
    neq = ~((a == b) | (np.isnan(a) & np.isnan(b)))
    if not neq.any():
        return int(a.size), 0, 0
    d = np.abs(a.view(np.int32).astype(np.int64) - b.view(np.int32).astype(np.int64))
    d[~np.isfinite(a) | ~np.isfinite(b)] = 1 << 30
    return int(a.size), int(neq.sum()), int(d[neq].max())


PARITY_TOLERANCE = ("poses 1e-5 rad / 1e-4 m; best index and visible ratio equal; EVERY energy bit-equal (round 6: the energy is "
                    "(float)sqrt of the correctly rounded exact sum of eps^2 on both sides -- no order of additions enters; "
                    "energies_from_exact_sweep counts the iterations whose fast sum the kernel could not certify and settled with exact limbs)")


def sparse_scenes_leg(stream, quick=False):
    """extra leg, never `value` (round 5): the engine on scenes that do NOT flatter it.  The bench's standard
    scenes have 5-6 % edge pixels everywhere; these have 0.5 / 1 / 2 % edge pixels, all in the left half of the frame -- the right
    half is empty (320 px at 640x480, 960 px at 1920x1080, 2048 px at 4096x3072: pixels hundreds of pixels from every edge, many
    distinct distances).  Per configuration: what form every level of every distinct scene got (the compact form, or the reason it was
    refused: dvo_get_now_compact_info), what the launch read, aligns/s and the roofline fraction by the same algorithmic-byte
    definition as the headline (it counts whole images: with few points it exceeds what the launch touches, so the ratio to the HBM
    peak -- `algorithmic_rate_over_hbm_peak`, deliberately not called a roofline fraction -- can pass 1 here: a rate, not an efficiency).  Rounds 3-4 REFUSED the finest levels of such scenes (more than 8191 distinct
    distances / a pixel 512 px or more from every edge) and read them as 16-byte texels; round 5 writes a PARTIAL compact form
    (dvo_palette.h): `levels_partial`, and `exact_fallback_ran_pair0` says whether any wave met a pixel the form cannot express."""
    from rgbd_odometry_amd import DvoContext, SynthScene
    from rgbd_odometry_amd.capi import DVO_FLAG_FINAL_OUTPUTS, DVO_FLAG_IDENTITY_START
    flags = DVO_FLAG_IDENTITY_START | DVO_FLAG_FINAL_OUTPUTS
    reasons = {-1: "not a distance transform", -2: "more than 8191 distinct distances", -3: "rank step beyond +-127", -4: "foreign gradients",
               -7: "a pixel 512 px or more from every edge"}
    out = []
    cfgs = [(640, 480, 4, 1024, 8, 10), (1920, 1080, 5, 256, 4, 4), (4096, 3072, 5, 1, 1, 20)]
    if quick:
        cfgs = cfgs[:1]
    for W, H, nl, B, D, steps in cfgs:
        iters = [10] * nl
        for dens in (0.005, 0.01, 0.02):
            n_seg = max(2, int(round(dens * W * H / (70.0 * W / 320.0))))
            scenes = [SynthScene(W, H, nl, 2000 + i, n_seg=n_seg, x_frac=0.5) for i in range(D)]
            ctx = DvoContext(B)
            try:
                ctx.set_intrinsics(*scenes[0].intrinsics)
                for i, sc in enumerate(scenes):
                    for l, L in enumerate(sc.levels):
                        ctx.set_ref_level_from_images(l, L.ref_edge, L.ref_depth, L.rows, L.cols, pair=i)
                        ctx.set_now_level_from_edges(l, u8_edges(L.now_edge), L.rows, L.cols, pair=i)
                info = [[ctx.now_compact_info(i, l) for l in range(nl)] for i in range(D)]
                part = [[ctx.now_compact_partial(i, l) for l in range(nl)] for i in range(D)]
                if B > D:
                    ctx.replicate_pairs(D)
                ctx.set_stream(stream.cuda_stream)
                for _ in range(2):
                    ctx.enqueue(iters, flags=flags); ctx.get_poses()
                ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for k in range(steps):
                    ev[k][0].record(stream)
                    ctx.enqueue(iters, flags=flags)
                    ev[k][1].record(stream)
                    ctx.get_poses()
                torch.cuda.synchronize()
                el = time.perf_counter() - t0
                k_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
                nbytes = sum(ctx.algorithmic_bytes(iters, pair=p % D, flags=DVO_FLAG_FINAL_OUTPUTS) for p in range(B))
                refused = sum(1 for row in info for v in row if v <= 0)
                blk, team, packed = ctx.last_launch_shape()
                out.append({
                    "workload": "%dx%dx%d, %d pairs (%d distinct), %.1f %% edge pixels in the left half" % (W, H, nl, B, D, 100 * dens),
                    "edge_density_level0": float(np.mean([(np.asarray(sc.levels[0].now_edge) != 0).mean() for sc in scenes])),
                    "aligns_per_s": B * steps / el, "kernel_ms": k_ms,
                    # NOT an efficiency: SURVEY 8(d)'s bytes count every level's whole image, a launch over a few thousand points touches a
                    # fraction of it -- the ratio passes 1 on the sparsest scenes
                    "algorithmic_rate_over_hbm_peak": nbytes / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                    "levels_refused": refused, "levels_partial": sum(1 for row in part for v in row if v), "levels_total": D * nl,
                    "exact_fallback_ran_pair0": [bool(ctx.level_exact_fallback(0, l)) for l in range(nl)],
                    "compact_info_per_level_scene0": info[0],          # > 0: palette size; < 0: refusal reason
                    "refusal_reasons": sorted({reasons.get(v, str(v)) for row in info for v in row if v <= 0}),
                    "texel_modes_pair0": [ctx.level_texel_mode(0, l) for l in range(nl)],
                    "launch_shape": {"block_threads": blk, "team": team},
                    "reference_points_level0": int(ctx.n_points(0)),
                })
            except Exception as e:
                out.append({"workload": "%dx%dx%d %.1f %%" % (W, H, nl, 100 * dens), "error": repr(e)})
            finally:
                ctx.close()
    return out


def float_boundary_leg(args, iters, flags, stream, per_scene_bytes):
    """extra leg, never `value`: 256 pairs whose now levels arrive as the reference keeps them -- DT, gradX, gradY float images in
    host memory (SolveDVO.cpp:1788-1795 -> dvo_set_now_level)"""
    from rgbd_odometry_amd import DvoContext, SynthScene
    B, D = 256, min(8, args.distinct)
    scenes = [SynthScene(args.width, args.height, args.levels, 1000 + i) for i in range(D)]
    ctx = DvoContext(B)
    try:
        ctx.set_intrinsics(*scenes[0].intrinsics)
        ctx.set_direct_compact(False)                  # the first two measurements: the boundary as rounds 2-5 had it (16-byte texels first)
        xyz_lists = {}
        for p in range(D):
            for l, L in enumerate(scenes[p].levels):
                xyz_lists[(p, l)] = ctx.set_ref_level_from_images(l, L.ref_edge, L.ref_depth, L.rows, L.cols, pair=p)[0]
                ctx.set_now_level(l, L.now_dt, L.now_gx, L.now_gy, L.rows, L.cols, pair=p)
        ctx.replicate_pairs(D)
        ctx.synchronize()

        def install():
            for p in range(B):
                for l, L in enumerate(scenes[p % D].levels):
                    ctx.set_now_level(l, L.now_dt, L.now_gx, L.now_gy, L.rows, L.cols, pair=p)
            ctx.synchronize()
        install()
        t0 = time.perf_counter()
        install()
        install_us = 1e6 * (time.perf_counter() - t0) / B
        ctx.set_stream(stream.cuda_stream)

        def rate(n):
            for _ in range(2):
                ctx.enqueue(iters, flags=flags); ctx.get_poses()
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for a, b in ev:
                a.record(stream); ctx.enqueue(iters, flags=flags); b.record(stream); ctx.get_poses()
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
            k = float(np.mean([a.elapsed_time(b) for a, b in ev]))
            by = sum(per_scene_bytes[p % len(per_scene_bytes)] for p in range(B))
            return {"aligns_per_s": B * n / el, "kernel_ms": k, "roofline_frac": by / (k * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                    "texel_modes": [ctx.level_texel_mode(0, l) for l in range(args.levels)]}
        r16 = rate(8)                                  # 2 + 8 alignments: below DVO_COMPACT_NOW_AFTER, still on 16-byte texels
        install()                                      # fresh now levels (use count 0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ctx.now_prepare()
        ctx.synchronize()
        build_us = 1e6 * (time.perf_counter() - t0) / B
        r4 = rate(8)
        # round 6: the LITERAL drop-in of INTEGRATION.md -- the reference's 3xN float lists through dvo_set_ref_level
        # and its three float images through dvo_set_now_level, nothing else -- as a batch context gets it by default: lists that verify
        # as enlistRefEdgePts lists get their compact twin, exact distance transforms their compact form at installation
        ctx.set_direct_compact(True)
        for p in range(B):
            for l, L in enumerate(scenes[p % D].levels):
                ctx.set_ref_level(l, xyz_lists[(p % D, l)], pair=p)
        install()
        t0 = time.perf_counter()
        install()
        install_direct_us = 1e6 * (time.perf_counter() - t0) / B
        lit = rate(8)
        lit["install_us_per_pair_pcie_inclusive"] = install_direct_us
        lit["launch_shape"] = ctx.last_launch_shape() if hasattr(ctx, "last_launch_shape") else None
        return {"pairs": B, "install_us_per_pair_pcie_inclusive": install_us, "on_16_byte_texels": r16,
                "compact_build_us_per_pair": build_us, "on_compact_form": r4, "literal_drop_in_float_lists_and_float_images": lit,
                "break_even_alignments_per_now_level": (build_us * 1e-6) / max(1e-12, (1.0 / r16["aligns_per_s"] - 1.0 / r4["aligns_per_s"])),
                "note": "never `value`.  Three float images per level from pageable host memory (12 B/pixel over PCIe) dominate this boundary: "
                        "installing a pair costs a hundred alignments.  The alignment starts on the 16-byte texels the installation "
                        "packs; the batched compact build (verified bit for bit against those texels) pays for itself after "
                        "break_even_alignments_per_now_level alignments of the SAME now level at this batch size, which is when the "
                        "engine builds it on its own (DVO_COMPACT_NOW_AFTER; dvo_now_prepare builds it at once) -- the reference aligns "
                        "a now level once or twice (SolveDVO.cpp:2097-2104, :2220-2227).  Callers after throughput hand over edge "
                        "maps or camera frames instead (dvo_set_now_level_from_edges, dvo_frames_*: `frames_in`)."}
    finally:
        ctx.close()


def dist_setup(share_gpu=False):
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device; the engine has no CPU fallback")
    if share_gpu:
        local_rank = 0                      # --ranks-share-gpu (testing): every rank on device 0, RCCL would refuse that -> gloo
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if share_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    return rank, local_rank, world, dist


def traffic_record(args, key, default_knobs):
    """HBM-side traffic of this launch shape from the PMC passes (profiles/pmc_traffic.json, tools/update_pmc_traffic.py):
    only valid for the kernel build it was measured on -- the record carries the hash of the kernel sources, and a stale
    record is reported as null with the reason instead of being pasted in"""
    pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    reason = None
    try:
        rec = json.load(open(pmc)).get(key)
    except Exception as e:
        rec, reason = None, "profiles/pmc_traffic.json unreadable: %r" % (e,)
    if rec is None:
        reason = reason or "no PMC record for workload %s" % key
    elif not default_knobs:
        reason = "non-default engine knobs: the PMC record describes the default launch"
    elif rec.get("kernel_source_sha256") != kernel_source_hash():
        reason = "PMC record was measured on another kernel build (source hash %s, now %s): re-run tools/update_pmc_traffic.py on the GPU box" % (
            rec.get("kernel_source_sha256"), kernel_source_hash())
    return (rec if reason is None else None), reason


def main_batch(args):
    rank, local_rank, world, dist = dist_setup(args.ranks_share_gpu)
    from rgbd_odometry_amd import DvoContext
    from rgbd_odometry_amd.capi import DVO_FLAG_FINAL_OUTPUTS, DVO_FLAG_IDENTITY_START, DVO_FLAG_NORMAL_MATRIX
    from rgbd_odometry_amd.distributed import shard_range, whole_job_throughput
    total_pairs = args.total_pairs
    if total_pairs > 0:
        if total_pairs < world:
            raise SystemExit("--total-pairs must be at least the number of GPUs")
        args.batch = shard_range(total_pairs, rank, world)[1]      # this rank's contiguous block of pairs

    iters = [args.iters] * args.levels
    flags = DVO_FLAG_IDENTITY_START | (0 if args.no_final_outputs else DVO_FLAG_FINAL_OUTPUTS) | (DVO_FLAG_NORMAL_MATRIX if args.normal_matrix else 0)
    default_knobs = not (args.team or args.normal_matrix or args.variant or args.block or args.inflight or args.lds_point_bytes or
                         args.debug_alias or args.no_final_outputs or args.float_now_levels)
    frames_in = None
    if world == 1 and default_knobs and not total_pairs and not args.no_extra_legs:
        # extra leg, never `value`; run BEFORE the large resident batch exists (measured: host-to-device copies of a process that
        # has allocated and freed tens of GB of HBM run at a third of their rate -- nothing the hot path touches, but this leg does)
        try:
            frames_in = frames_leg(args, iters)
        except Exception as e:                       # the extra legs must never cost the headline line
            frames_in = {"error": repr(e)}
    # a box with less free HBM than the resident batch needs (another tenant, a partitioned GPU): shrink the batch to what fits
    # instead of dying in an allocation -- the line then says so (config.batch_reduced_from)
    batch_asked = args.batch
    sum_px = sum((args.width >> l) * (args.height >> l) for l in range(args.levels))
    per_pair = 9.0 * sum_px + 0.3e6                  # compact now form + palettes + point lists + outputs (3.5 MB at 640x480x4)
    free_b, _total_b = torch.cuda.mem_get_info()
    if args.assume_free_gb > 0:
        free_b = args.assume_free_gb * 1e9
    fits = int((free_b / (world if args.ranks_share_gpu else 1) - 12e9) / per_pair)
    if dist is not None and world > 1:
        # every rank must size the SAME batch (value = rank 0's batch x world): take the minimum over the ranks
        t_fits = torch.tensor([fits], dtype=torch.int64, device="cpu" if args.ranks_share_gpu else "cuda")
        dist.all_reduce(t_fits, op=dist.ReduceOp.MIN)
        fits = int(t_fits.item())
    if not total_pairs and fits < args.batch:
        args.batch = max(256, fits // 256 * 256)
    ctx = DvoContext(args.batch, block_threads=args.block, debug_alias_mod=args.debug_alias,
                     points_in_flight=args.inflight, lds_point_bytes=args.lds_point_bytes,
                     engine_variant=args.variant, team_size=args.team)
    scenes = build_batch(ctx, args, rank)
    stream = torch.cuda.Stream()
    ctx.set_stream(stream.cuda_stream)

    def step(ev=None):
        if ev is not None:
            ev[0].record(stream)
        ctx.enqueue(iters, flags=flags)                 # one launch: the whole batch, all levels
        if ev is not None:
            ev[1].record(stream)
        return ctx.get_poses()                          # deliver poses to the host (synchronises)

    def barrier():
        if dist is not None:
            dist.barrier()

    step()      # initialisation, not a measured or counted step: the engine allocates its output buffers at the first launch
    for _ in range(args.warmup):
        step()
    events = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        R, t = step(events[k])
    torch.cuda.synchronize()
    barrier()
    elapsed_local = time.perf_counter() - t0
    value, elapsed = whole_job_throughput(args.batch, args.steps, elapsed_local, device="cpu" if args.ranks_share_gpu else "cuda")   # MAX over ranks
    if total_pairs > 0:
        value = total_pairs * args.steps / elapsed            # the blocks differ by at most one pair: count the real total

    kernel_ms = float(np.mean([a.elapsed_time(b) for a, b in events]))
    D = len(scenes)
    per_scene_bytes = [ctx.algorithmic_bytes(iters, pair=p, flags=flags & DVO_FLAG_FINAL_OUTPUTS) for p in range(D)]
    per_scene_pi = [ctx.point_iterations(iters, pair=p) for p in range(D)]
    bytes_per_launch = sum(per_scene_bytes[p % D] for p in range(args.batch))     # slot p holds a copy of scene p % D
    point_iters = sum(per_scene_pi[p % D] for p in range(args.batch))
    achieved = bytes_per_launch / (kernel_ms * 1e-3) / 1e9

    blk, team, packed = ctx.last_launch_shape()
    modes = [ctx.level_texel_mode(0, l) for l in range(args.levels)]
    mode_names = {0: "16-byte texels", 1: "LDS-staged texels", 2: "compact 4-byte form"}
    kernel_label = ("align_fused2_kernel<%d,%s> (packed, two points per lane; dvo_fused.hip; %s; now levels read as %s)" %
                    (blk, "true" if team > 1 else "false",
                     "teams of %d workgroups per pair" % team if team > 1 else ("two workgroups per CU" if blk == 256 else "one workgroup per CU"),
                     "/".join(mode_names.get(m, "?") for m in modes))
                    if packed else "align_fused_kernel<%d> (one point per lane; dvo_kernels.hip)" % blk)
    if rank != 0:
        ctx.close()
        if dist is not None:
            dist.destroy_process_group()
        return
    if args.float_now_levels:
        now_desc = ("caller-supplied float images (dvo_set_now_level): 16-byte texels" +
                    ("; generic compact form built at set-up by dvo_now_prepare in %.1f ms (NOT in the timed region)" % args.now_prepare_ms
                     if args.prepare else ""))
    elif args.variant == 4:
        now_desc = "16-byte texels written by the engine's distance-transform stage (compact form switched off)"
    else:
        now_desc = ("compact 4-byte form, written natively by the engine's distance-transform stage (dvo_set_now_level_from_edges: "
                    "ranks from the integer squared distances; no re-encoding pass exists, 16-byte texels are never written)")
    out = {
        "metric": "frame-pair aligns/sec (%dx%d, %d-lvl pyr)" % (args.width, args.height, args.levels),
        "value": value, "unit": "aligns/s", "n_gpus": (1 if args.ranks_share_gpu else world), "rccl_ranks": (0 if args.ranks_share_gpu else (dist.get_world_size() if dist is not None else 1)),
        "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "strong" if total_pairs > 0 else "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {
            "workload": "%dx%d edge-alignment, %d-level pyramid, %d iters/level, batch of %d independent "
                        "frame pairs per GPU (%d distinct synthetic scenes), identity start, "
                        "sub-gradient policy of SolveDVO::runIterations" %
                        (args.width, args.height, args.levels, args.iters, args.batch, D),
            **({"ranks_share_one_gpu": True} if args.ranks_share_gpu else {}),
            "pairs_per_gpu": args.batch, **({"batch_reduced_from": batch_asked} if args.batch != batch_asked and not total_pairs else {}),
            "iters_per_level": iters,
            **({"total_pairs": total_pairs} if total_pairs > 0 else {}),
            "final_outputs": not args.no_final_outputs,
            "now_levels": now_desc,
            "timed_region_s": elapsed,
            "block_threads": args.block or ("auto: %d" % blk),
            "points_in_flight": args.inflight or 1,
            **({"debug_alias_mod": args.debug_alias} if args.debug_alias else {}),
            **({"engine_variant": args.variant} if args.variant else {}),
            **({"normal_matrix": True} if args.normal_matrix else {}),
            "point_iterations_per_launch": point_iters,
        },
        "roofline": {
            "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBPS,
            "traffic": None,
            "kernel": kernel_label, "kernel_ms": kernel_ms,
            "algorithmic_bytes_per_launch": bytes_per_launch,
            "algorithmic_bytes_per_alignment": bytes_per_launch / args.batch,
            "definition": "achieved = SURVEY 8(d) bytes (12 B/pixel of the reference's three float images per level + 12 B/point "
                          "+ 16 B/point of final outputs) x pairs per launch / HIP-event duration of the launch; `traffic` and "
                          "`hbm_side` say what the kernel really moved (the compact form is 4 B/pixel and only lines under the "
                          "contours are fetched)",
        },
    }
    key = "%dx%dx%dx%d_b%d" % (args.width, args.height, args.levels, args.iters, args.batch)
    rec, reason = traffic_record(args, key, default_knobs)
    if rec is not None:
        out["roofline"]["traffic"] = rec["hbm_bytes_per_launch"]
        out["roofline"]["traffic_source"] = rec.get("source")
        side = rec["hbm_bytes_per_launch"] / (kernel_ms * 1e-3) / 1e9
        out["roofline"]["hbm_side"] = {"GBps": side, "frac_of_peak": side / HBM_PEAK_GBPS, "frac_of_measured_copy_peak": side / HBM_COPY_GBPS,
                                       "traffic_over_algorithmic": rec["hbm_bytes_per_launch"] / bytes_per_launch,
                                       "read_bytes": rec.get("fetch_bytes"), "write_bytes": rec.get("write_bytes"),
                                       "fetch_size_as_reported_bytes": rec.get("fetch_size_as_reported_bytes"),
                                       "note": "what the launch really drew from HBM / the Infinity Cache: TCC_EA0_RDREQ x 128 B (every L2 -> fabric "
                                               "read request is a whole 128-byte line on gfx950, gathers included: profiles/r04_line_fetch; FETCH_SIZE "
                                               "tallies them at 64 B) + WRITE_SIZE, per launch / live kernel time"}
        if rec.get("l2_read_requests"):
            rate = rec["l2_read_requests"] / (kernel_ms * 1e-3) / 1e9
            out["roofline"]["request_rate"] = {
                "achieved_G_lines_per_s": rate, "calibrated_ceiling_G_lines_per_s": [44.0, 50.0],
                "frac_of_ceiling": rate / 47.0,
                "requests_per_alignment": rec["l2_read_requests"] / args.batch,
                "l2_hit_rate": (rec.get("l2_hits", 0) / rec["l2_requests"]) if rec.get("l2_requests") else None,
                "note": "TCC_EA0_RDREQ per launch (PMC profile) / live kernel time.  The ceiling (tools/exhaustive/fetch_calib.hip, "
                        "profiles/r01_fetch_size_calibration: 44-50 G requests/s for streams and sparse gathers alike) is the HBM bandwidth "
                        "itself: 128 B per request = 5.6-6.4 TB/s",
            }
    else:
        out["roofline"]["traffic_reason"] = reason
    if world == 1 and args.cpu_seconds > 0:
        base, oracle, lvs = cpu_baseline(args, scenes, iters, args.cpu_seconds)
        out["cpu_baseline"] = base
        if not args.no_cpu_all_cores:
            try:
                out["cpu_baseline_openmp"] = cpu_baseline_openmp(args, oracle, lvs, scenes, iters, args.cpu_seconds, base["value"])
            except Exception as e:
                out["cpu_baseline_openmp"] = {"error": repr(e)}
            try:
                out["cpu_baseline_all_cores"] = cpu_baseline_all_cores(args, iters, args.cpu_seconds)
            except Exception as e:
                out["cpu_baseline_all_cores"] = {"error": repr(e)}
        # parity check in the same run: EVERY distinct scene of the batch against the oracle on the same inputs
        # (pairs 0..D-1 are the distinct ones, the rest of the batch are device copies of them)
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib
        worst_r = worst_t = 0.0
        bit_equal = decisions_equal = replicas_equal = True
        n_e = n_diff = max_ulp = n_sweeps = 0
        for i, sc_i in enumerate(scenes):
            lv_i = lvs[i] if i < len(lvs) else oracle_lib.scene_levels(sc_i, oracle)
            ref = oracle.align_pyramid(iters, lv_i, sc_i.intrinsics, np.eye(3), np.zeros(3))
            worst_r = max(worst_r, oracle_lib.rot_angle(ref["R"], R[i]))
            worst_t = max(worst_t, float(np.linalg.norm(ref["t"] - t[i])))
            for l, rep in ref["levels"].items():
                e, bi, ratio = ctx.level_report(i, l, iters[l])
                bit_equal = bit_equal and bool(np.array_equal(e, rep["energy"]))
                decisions_equal = decisions_equal and bi == rep["best_idx"] and ratio == rep["visible_ratio"]
                n, k, u = energy_ulps(e, rep["energy"])
                n_e += n; n_diff += k; max_ulp = max(max_ulp, u)
                n_sweeps += ctx.level_energy_sweeps(i, l)
            # a replica far down the batch must carry the same bits as its source
            j = i + D * ((args.batch - 1 - i) // D)
            replicas_equal = replicas_equal and bool(np.array_equal(R[i], R[j])) and bool(np.array_equal(t[i], t[j]))
        out["parity_check"] = {
            "pairs_checked": D, "max_rot_err_rad": worst_r, "max_trans_err_m": worst_t,
            "energies_bit_equal": bool(bit_equal), "energies_compared": n_e, "energies_differing": n_diff, "max_energy_ulp": max_ulp,
            "energies_from_exact_sweep": n_sweeps,
            "best_index_and_ratio_equal": bool(decisions_equal), "replicas_bit_identical": bool(replicas_equal),
            "tolerance": PARITY_TOLERANCE,
            # round 6: strict at every size.  (Round 5 allowed 1e-4 of the energies of a large comparison to differ by one ulp: the
            # kernel's order of additions was not the oracle's.  The energy no longer has an order: DESIGN.md section 2.)
            "pass": bool(decisions_equal and replicas_equal and bit_equal and worst_r <= 1e-5 and worst_t <= 1e-4),
        }
    # transparency legs, never `value`; default launch only
    default_launch = default_knobs and not total_pairs
    if world == 1 and default_launch and not args.no_extra_legs:
        ctx.close()
        ctx = None
        try:
            # the same workload with the compact form switched off: the distance-transform stage then writes 16-byte texels
            b2 = min(args.batch, 1024)
            saved = args.batch
            args.batch = b2
            ctx = DvoContext(b2, engine_variant=4)
            build_batch(ctx, args, rank)
            args.batch = saved
            ctx.set_stream(stream.cuda_stream)
            n2 = 10
            for _ in range(2):
                step()
            ev2 = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n2)]
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for k in range(n2):
                step(ev2[k])
            torch.cuda.synchronize()
            el2 = time.perf_counter() - t0
            k2 = float(np.mean([a.elapsed_time(b) for a, b in ev2]))
            bytes2 = sum(per_scene_bytes[p % D] for p in range(b2))
            out["without_compact_now_form"] = {
                "value": b2 * n2 / el2, "unit": "aligns/s", "steps": n2, "pairs": b2, "kernel_ms": k2,
                "roofline_frac": bytes2 / (k2 * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                "note": "same workload with dvo_params.engine_variant = 4: the distance-transform stage writes 16-byte texels "
                        "{DT, gx, gy, w} instead of the compact form and the alignment gathers those"}
        except Exception as e:
            out["without_compact_now_form"] = {"error": repr(e)}
        if ctx is not None:
            ctx.close()
            ctx = None
        # the section-8(b) boundary as the reference would use it: the now levels handed over as three float images per level
        # (dvo_set_now_level).  What it costs to INSTALL a pair that way (PCIe-inclusive), the alignment rate on the 16-byte texels
        # such a level starts with, the cost of the batched compact build (dvo_now_prepare / automatic after DVO_COMPACT_NOW_AFTER
        # alignments of the same now level) and the rate on its result
        try:
            out["float_now_levels"] = float_boundary_leg(args, iters, flags, stream, per_scene_bytes)
        except Exception as e:
            out["float_now_levels"] = {"error": repr(e)}
        try:
            out["sparse_scenes"] = sparse_scenes_leg(stream)
        except Exception as e:
            out["sparse_scenes"] = {"error": repr(e)}
    if frames_in is not None:
        out["frames_in"] = frames_in
    print(json.dumps(out), flush=True)
    if ctx is not None:
        ctx.close()
    if dist is not None:
        dist.destroy_process_group()


# ---- tiled mode (BASELINE configs[4]) ----------------------------------------------------------------------------------
def main_tiled(args):
    rank, local_rank, world, dist = dist_setup()
    from rgbd_odometry_amd import DvoContext, SynthScene
    from rgbd_odometry_amd.capi import RcclComm, DVO_FLAG_FINAL_OUTPUTS, DVO_FLAG_NORMAL_MATRIX
    from rgbd_odometry_amd.distributed import shard_range
    iters = [args.iters] * args.levels
    # --normal-matrix: the all-reduced sums also carry H = sum w J J^T (BASELINE configs[4] names "6x6 JtJ + 6x1 Jtr"; the reference's
    # update reads J^T W eps only, SolveDVO.cpp:777, so the default leaves the 21 slots of H zero)
    flags = (0 if args.no_final_outputs else DVO_FLAG_FINAL_OUTPUTS) | (DVO_FLAG_NORMAL_MATRIX if args.normal_matrix else 0)
    sc = SynthScene(args.width, args.height, args.levels, 7)                   # C5 is seed 7
    ctx = DvoContext(1)
    ctx.set_intrinsics(*sc.intrinsics)
    for l, L in enumerate(sc.levels):                                            # every rank holds the whole pair (replicated)
        ctx.set_ref_level_from_images(l, L.ref_edge, L.ref_depth, L.rows, L.cols)
        ctx.set_now_level_from_edges(l, u8_edges(L.now_edge), L.rows, L.cols)
    n_pts = [ctx.n_points(l) for l in range(args.levels)]
    stream = torch.cuda.Stream()
    ctx.set_stream(stream.cuda_stream)
    # the communicator of the C path: rank 0's ncclUniqueId travels over the torch process group
    uid = [RcclComm.unique_id() if rank == 0 else None]
    if dist is not None:
        dist.broadcast_object_list(uid, src=0)
    comm = RcclComm(uid[0], rank, world)
    ctx.tiled_attach(comm.comm, rank, world)
    I, z = np.eye(3), np.zeros(3)

    def step():
        return ctx.align_pyramid_tiled(iters, I, z, flags=flags)                 # synchronous: poses on the host

    def barrier():
        if dist is not None:
            dist.barrier()
    for _ in range(max(1, args.warmup)):
        R, t = step()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        R, t = step()
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    value = args.steps / elapsed                                                 # all ranks work on the SAME alignment
    # every rank must hold the same bits
    same = True
    if dist is not None:
        mine = torch.tensor(np.concatenate([R.reshape(-1), t]), dtype=torch.float64, device="cuda")
        allp = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(allp, mine)
        same = all(bool(torch.equal(allp[0], a)) for a in allp)
    graph_replayed = ctx.tiled_graph_replayed()
    pk_mask, solo_mask = ctx.wide_packed_levels(), ctx.wide_solo_levels()        # of the schedule that was timed (the step measurement below enqueues its own)
    team_mask = ctx.wide_team_levels()
    team_g = ctx.last_launch_shape()[1] if team_mask else 0                       # members of the team launch that ran the finest level
    # per-level reports of the LAST timed alignment, read before anything else touches the context's outputs
    reports = {l: ctx.level_report(0, l, iters[l]) for l in range(args.levels) if iters[l] > 0}
    finals = None
    if flags:
        finals = ctx.final_outputs(0, n_pts[[l for l in range(args.levels) if iters[l] > 0][0]])
    # the dominant kernel: tiled_step_kernel over this rank's shard of the finest level.  The schedule is one replayed graph, so the
    # launch is timed through the product path itself: alignments with 10 and with 60 iterations at the finest level only -- the
    # slope is one iteration there (the launch, its boundary and the collective)
    first, count = shard_range(n_pts[0], rank, world)

    def level0_ms(n_it, reps=5):
        it = [n_it] + [0] * (args.levels - 1)
        ctx.align_pyramid_tiled(it, I, z, flags=0)
        barrier(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            ctx.align_pyramid_tiled(it, I, z, flags=0)
        torch.cuda.synchronize(); barrier()
        return 1e3 * (time.perf_counter() - t0) / reps
    # --no-extra-legs (the PMC passes of tools/update_pmc_traffic.py): whole alignments only in the kernel trace, no slope measurement
    acc_ms = max(1e-6, (level0_ms(60) - level0_ms(10)) / 50.0) if not args.no_extra_legs else 1e-6
    if dist is not None:
        tt = torch.tensor([acc_ms], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        acc_ms = float(tt.item())
    L0 = sc.levels[0]
    bytes_launch = 12 * count + 12 * L0.rows * L0.cols // max(1, iters[0])
    bytes_align = ctx.algorithmic_bytes(iters, flags=flags)
    if rank != 0:
        ctx.tiled_detach(); ctx.close(); comm.close()
        if dist is not None:
            dist.destroy_process_group()
        return
    achieved = bytes_launch / (acc_ms * 1e-3) / 1e9
    # traffic of the whole alignment from the PMC passes: only the one-launch form (one rank, every level inside the fused team launch)
    # has a single kernel per alignment to attribute it to
    all_team = world == 1 and team_mask == (1 << args.levels) - 1
    rec, traffic_reason = traffic_record(args, "tiled_%dx%dx%dx%d" % (args.width, args.height, args.levels, args.iters), all_team and not args.normal_matrix)
    out = {
        "metric": "frame-pair aligns/sec (%dx%d, %d-lvl pyr)" % (args.width, args.height, args.levels),
        "value": value, "unit": "aligns/s", "n_gpus": world, "rccl_ranks": comm.count(), "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {
            "workload": "single %dx%d frame pair, %d-level pyramid, %d iters/level, reference point lists tiled over %d GPU(s): "
                        "per iteration accumulate(own shard) -> ncclAllReduce(32 doubles, RCCL called from C) -> identical update "
                        "(dvo_align_pyramid_tiled); identity start" % (args.width, args.height, args.levels, args.iters, world),
            "mode": "tiled", "points_per_level": n_pts, "iters_per_level": iters, "final_outputs": not args.no_final_outputs,
            "us_per_iteration": 1e6 * elapsed / args.steps / sum(iters), "timed_region_s": elapsed,
            "all_ranks_bit_identical": same, "graph_replayed": graph_replayed,
            "levels_on_the_packed_step_kernel": [l for l in range(args.levels) if (pk_mask >> l) & 1],
            "levels_as_one_launch": [l for l in range(args.levels) if (solo_mask >> l) & 1],
            "levels_as_team_launches": [l for l in range(args.levels) if (team_mask >> l) & 1],   # round 6: team launches of the fused kernel, run whole by every rank (no collective)
            "points_per_level": [int(ctx.n_points(l)) for l in range(args.levels)],
            "algorithmic_bytes_per_alignment": bytes_align,
            "alignment_GBps": bytes_align * value / 1e9,
        },
        "roofline": {
            "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS, "traffic": None,
            "kernel": ("align_fused2_kernel<512,true>: ONE launch of the whole pyramid as a team of %d workgroups (one rank: nothing is sharded); "
                       "kernel_ms = one iteration of level 0 (%d points) inside that launch" % (team_g, n_pts[0])) if (team_mask & 1) else ("tiled_step_kernel (dvo_kernels.hip: update of the previous iteration + this rank's shard of level 0, %d of %d points, + the sums of the launch) and its ncclAllReduce" % (count, n_pts[0])),
            "kernel_ms": acc_ms, "algorithmic_bytes_per_launch": bytes_launch,
            # the per-launch fraction above is one level-0 step with the images amortised over its iterations; the whole
            # alignment -- SURVEY 8(d)'s bytes over the time of ALL its launches -- is latency-bound and sits far lower
            "per_alignment": {"achieved": bytes_align * value / 1e9, "frac": bytes_align * value / 1e9 / HBM_PEAK_GBPS, "unit": "GB/s",
                              "algorithmic_bytes": bytes_align},
            "definition": "12 B x points of the shard + the level's 12 B/pixel images amortised over its iterations, per accumulate "
                          "launch; kernel_ms = one iteration at level 0 measured through the product path (slope of the alignment time over "
                          "the number of level-0 iterations): the launch, its boundary and the 256-byte all-reduce",
            **({"traffic_reason": traffic_reason} if rec is None else
               {"traffic_per_alignment": rec["hbm_bytes_per_launch"], "traffic_over_algorithmic": rec["hbm_bytes_per_launch"] / bytes_align,
                "traffic_note": "HBM bytes of ONE alignment = one team launch (PMC passes: TCC_EA0_RDREQ x 128 + WRITE_SIZE; "
                                "roofline.traffic stays null: it is defined per `kernel_ms`, which here is one level-0 iteration)"}),
        },
    }
    if world == 1 and args.cpu_seconds > 0:
        base, oracle, lvs = cpu_baseline(args, [sc], iters, args.cpu_seconds, max_n=50)
        out["cpu_baseline"] = base
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib
        ref = oracle.align_pyramid(iters, lvs[0], sc.intrinsics, I, z)
        bit_equal = decisions_equal = True
        n_e = n_diff = max_ulp = 0
        for l, rep in ref["levels"].items():
            e, bi, ratio = reports[l]
            bit_equal = bit_equal and bool(np.array_equal(e, rep["energy"]))
            decisions_equal = decisions_equal and bi == rep["best_idx"] and ratio == rep["visible_ratio"]
            n, k, u = energy_ulps(e, rep["energy"])
            n_e += n; n_diff += k; max_ulp = max(max_ulp, u)
        fin = True
        if flags:
            last = ref["levels"][ref["last_level"]]
            fe, fr = finals
            fin = bool(np.array_equal(fe, last["final_eps"])) and bool(np.array_equal(fr, last["final_reproj"], equal_nan=True))
        wr, wt = oracle_lib.rot_angle(ref["R"], R), float(np.linalg.norm(ref["t"] - t))
        out["parity_check"] = {"max_rot_err_rad": wr, "max_trans_err_m": wt, "energies_bit_equal": bool(bit_equal), "energies_compared": n_e,
                               "energies_differing": n_diff, "max_energy_ulp": max_ulp, "best_index_and_ratio_equal": bool(decisions_equal),
                               "final_outputs_bit_equal": fin, "tolerance": PARITY_TOLERANCE + "; final outputs bit-equal",
                               "pass": bool(decisions_equal and bit_equal and fin and wr <= 1e-5 and wt <= 1e-4)}      # one pair, 50 energies: strict
    print(json.dumps(out), flush=True)
    ctx.tiled_detach(); ctx.close(); comm.close()
    if dist is not None:
        dist.destroy_process_group()


def launch_ranks(args):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: THIS process -- which has made no GPU call and never will
    -- starts `python -m torch.distributed.run --nproc-per-node N bench.py <same arguments>` as a child process, relays its output
    (rank 0's JSON line) and its return code.  Under a launcher (WORLD_SIZE set) nothing is started: the process IS a rank, and
    WORLD_SIZE must then agree with --gpus.  Returns only when this process should run a rank itself."""
    world_env = os.environ.get("WORLD_SIZE")
    if world_env is not None:
        if int(world_env) != args.gpus:
            sys.stderr.write("bench.py: --gpus %d but the launcher set WORLD_SIZE=%s: start as many ranks as GPUs are asked for "
                             "(python -m torch.distributed.run --nproc-per-node %d ... bench.py --gpus %d)\n" % (args.gpus, world_env, args.gpus, args.gpus))
            sys.exit(2)
        return
    if args.gpus <= 1:
        return
    n_dev = torch.cuda.device_count()           # counts devices without initialising the GPU (no HIP context in this process)
    if args.ranks_share_gpu and args.mode != "batch":
        sys.stderr.write("bench.py: --ranks-share-gpu is a batch-mode test switch (the tiled path's ranks need RCCL: tests/test_gpu_tiled_ranks.py runs them over a loopback)\n")
        sys.exit(2)
    if n_dev < (1 if args.ranks_share_gpu else args.gpus):
        sys.stderr.write("bench.py: --gpus %d asked for, %d HIP device(s) visible: refusing to print a %d-GPU line from fewer GPUs\n" % (args.gpus, n_dev, args.gpus))
        sys.exit(3)
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # dmabuf IPC: RCCL across processes needs it on this driver
    proc = subprocess.run(cmd, env=env)                      # a child process, never exec: stdout / stderr are inherited
    sys.exit(proc.returncode)


def main():
    args = parse_args()
    launch_ranks(args)
    keep_host_memory_mapped()
    if args.mode == "tiled":
        main_tiled(args)
    else:
        main_batch(args)
    if os.environ.get("DVO_DUMP_MAPS"):       # diagnostics (tools/experiments/r04_exit_segv.sh): which library owns an address of a crash at exit
        with open("/proc/self/maps") as f:
            sys.stderr.write("".join(l for l in f if " r-xp " in l))


if __name__ == "__main__":
    main()
