/*
 * dvo_ctx.h -- the engine context (struct dvo_ctx) and the host helpers shared by the two halves of the C-ABI
 * implementation: dvo_capi.cpp (lifecycle, inputs, hot path, inspection) and dvo_capi_frames.cpp (frame store,
 * rows f1 + f2).  Internal; not installed.
 */
#ifndef DVO_CTX_H_
#define DVO_CTX_H_

#include "../../include/dvo_amd.h"
#include "dvo_launch.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <vector>

namespace dvo_host {

struct Level {
    int rows = 0, cols = 0;
    float4 *tex = nullptr;
    size_t tex_stride = 0;
    /* SPARSE texel slab (round 4): the 16-byte texels of all pairs are one virtual address range (hipMemAddressReserve) of which
     * only the chunks that some pair's texels were ever asked for are backed by memory (map_texels).  The engine's own now levels
     * exist in the compact form only, so a large resident batch never maps any: 6.5 of the 9.7 MB a 640x480x4 pair used to hold.
     * Kernels keep addressing base + pair * stride.  Dense (one hipMalloc, everything backed) for small slabs. */
    bool tex_sparse = false;
    size_t tex_chunk = 0, tex_va_bytes = 0;
    std::vector<hipMemGenericAllocationHandle_t> tex_handles;   /* per chunk; tex_mapped says which are live */
    std::vector<char> tex_mapped;
    float *pts = nullptr;
    uint2 *cpts = nullptr;          /* compact twin of pts (8 B / point), same capacity; valid where compact_ok */
    unsigned *cidx = nullptr;       /* per compact point: its index in the 3 x N list (the compact twin is in block order) */
    unsigned *cpt4 = nullptr;       /* the compact twin again in 4 bytes per point (dvo_device_math.h: pt4_decode), same capacity */
    unsigned *chdr = nullptr;       /* per chunk of 64 points: linear block index of its first point; pt_cap / 64 per pair */
    int *d_pt4_ok = nullptr;        /* per pair: the 4-byte list decodes to the 8-byte one bit for bit (written by the builder) */
    std::vector<char> compact_ok;   /* per pair: the list came from the engine's enlist kernels */
    /* per pair: a stamp that changes whenever the pair's reference list is rewritten (ref_list_written).  The packed kernel stores
     * its final outputs in the order of the compact list; the getter permutes them with cidx and must not do so with the index of a
     * list written after the launch */
    std::vector<unsigned long long> list_gen;
    std::vector<int> pt4_rows;      /* per pair: image rows the 4-byte twin of the list was encoded against (pt4_decode needs the same) */
    int pt_cap = 0;
    int *dN = nullptr;
    std::vector<int> hN;            /* 0 = not set */
    std::vector<char> have_now;
    /* compact form of the now levels (dvo_palette.h), allocated at the first build */
    unsigned *p4 = nullptr;         /* n_pairs x p4_stride rank words */
    float2 *pal = nullptr;          /* n_pairs x DVO_PAL_MAX {DT value, weight} */
    int *d_pal_n = nullptr;         /* n_pairs: > 0 palette size, <= 0 no compact form (kept 0 while the form is stale) */
    size_t p4_stride = 0;
    std::vector<char> pal_built;    /* per pair: the compact form was built from the CURRENT now level (or IS how it was written) */
    std::vector<int> now_uses;      /* per pair: alignments enqueued since the now level was last written */
    /* per pair: the now level was written in its compact form only (the engine's own distance-transform stage, round 3) and its
     * 16-byte texels have not been decoded from it yet.  (A pair whose image the compact form could not hold got its texels
     * from the same launch; the decode launch skips it on the device: pal_n <= 0.) */
    std::vector<char> tex16_stale;
    /* per pair: what the HOST knows about the compact form of the current now level -- the device's pal_n says whether the builder
     * (the distance-transform stage, or the generic one) could make it, and the host only learns that where it reads pal_n back:
     * at once on a sparse texel slab (the refused images need memory mapped), otherwise on demand (refresh_p4_known: before a
     * replication and before a launch-shape decision that depends on it).  A REFUSED pair's 16-byte texels are its real form:
     * they are current (tex16_stale = 0) and travel with it (round 5, ADVICE r4: a replicated refused source left its
     * destinations without texels -- on a sparse slab without memory behind them). */
    enum : char { P4_UNKNOWN = 0, P4_OK = 1, P4_REFUSED = 2, P4_PARTIAL = 3 /* a partial compact form (dvo_palette.h): read as the compact form, 16-byte texels real too */ };
    std::vector<char> p4_known;
    /* set where the engine's own distance-transform stage wrote the pair's compact form (now_written_compact): such a form is complete or
     * PARTIAL, never refused (dvo_frames.hip) -- what dvo_enqueue's choice of launch shape asks about -- so that choice needs no read-back
     * and no wait on the stream for these pairs (one host sync per level per step of a frames -> align pipeline otherwise) */
    std::vector<char> p4_native;
    std::vector<char> p4_fresh;     /* set where sparse_map_compact_failures has just read pal_n; consumed by now_written_compact */
};

/* frame store (rows f1/f2): per level one slab per plane for all slots, slot s at base + s*npx */
struct FrameLevel {
    int rows = 0, cols = 0;
    size_t npx = 0;
    unsigned char *grey = nullptr, *edge = nullptr;
    float *depth = nullptr;
};
struct FrameStore {
    int n_slots = 0, n_levels = 0;
    FrameLevel lv[DVO_LEVELS];
    std::vector<char> valid, has_depth;
};
/* diagnostics: DVO_COMPACT_ORDER=colmajor in the environment keeps the compact point lists in the reference's column-major
 * order (A/B measurement of the block order; results are the same up to the double rounding of the sums) */
inline bool compact_block_order() {
    static const bool on = [] { const char *e = std::getenv("DVO_COMPACT_ORDER"); return !(e && std::strcmp(e, "colmajor") == 0); }();
    return on;
}
/* diagnostics / tests: DVO_COMPACT_NOW=eager builds the compact form of a now level (dvo_palette.h) at its FIRST alignment
 * (so that every test of the suite runs through it), =off never builds it; default: after DVO_COMPACT_NOW_AFTER alignments */
inline int compact_now_policy() {
    static const int pol = [] {
        const char *e = std::getenv("DVO_COMPACT_NOW");
        return (e && std::strcmp(e, "eager") == 0) ? 1 : ((e && std::strcmp(e, "off") == 0) ? 2 : 0);
    }();
    return pol;
}
}  // namespace dvo_host

struct dvo_ctx {
    dvo_params prm;
    dvo::DevParams dprm;
    int n_pairs = 0;
    int device = 0;                 /* HIP device the context was created on: every entry point makes it current (DeviceGuard) */
    int n_cu = 256;                 /* compute units of the device (auto tuning of launch shapes) */
    hipStream_t stream = nullptr;
    hipStream_t own_stream = nullptr;
    dvo::Intrinsics K{0, 0, 0, 0, 0};
    bool have_K = false;
    dvo_host::Level lv[DVO_LEVELS];
    float *staging = nullptr;       /* 3 planes (or one point list) of the largest upload so far */
    size_t staging_bytes = 0;
    double *d_poses = nullptr;
    float *d_energy = nullptr;
    size_t energy_floats = 0;
    int *d_best = nullptr;
    float *d_ratio = nullptr;
    float *d_final_eps = nullptr, *d_final_reproj = nullptr;
    int *d_final_N = nullptr;
    double *d_H = nullptr;          /* DVO_FLAG_NORMAL_MATRIX output, n_pairs x e_stride x 21 */
    size_t H_doubles = 0;
    int *d_tex_mode = nullptr;
    /* team mode of the packed kernel (G workgroups per pair for small batches): exchange slots, arrival counters, error flag */
    double *d_team_buf = nullptr;
    unsigned *d_team_cnt = nullptr;      /* n_pairs counters followed by one int error flag */
    int last_block = 0, last_team = 0, last_packed = 0;   /* shape of the last fused launch (dvo_get_last_launch_shape) */
    unsigned team_epoch = 0;             /* team mode: exchanges every launch so far may have run (Schedule.team_epoch0 of the next launch) */
    bool team_legacy = false;            /* a team launch was captured into a caller's graph once: zero the records before every launch again */
    bool team_err_dirty = false;         /* the device's error word is set (reported by the getters): the next team launch clears it */
    bool team_used = false;              /* the last enqueue ran in team mode: dvo_get_poses checks the error flag */      /* n_pairs x DVO_LEVELS, written by the packed fused kernel */
    int final_cap = 0;
    double *d_scratch = nullptr;    /* partials (1024 x 32) + acc (32) + misc doubles */
    /* dvo_align_pyramid_wide as a replayable hipGraph (the schedule is ~2 dependent launches per iteration) */
    hipGraphExec_t wide_exec = nullptr;
    unsigned long long wide_sig = 0;
    int step_solo_mask = 0;              /* ... and those that ran as one launch of one workgroup (tiled_level_solo_kernel) */
    int step_pk_mask = 0;                /* levels of the last enqueued step schedule that ran tiled_step_pk_kernel (inspection) */
    double *h_pose = nullptr;       /* pinned: in/out pose of the graph's copy nodes */
    int wide_team_mask = 0;         /* levels the last dvo_align_pyramid_wide handed to the fused team kernel (round 6) */
    int direct_compact = -1;        /* dvo_set_direct_compact: float now levels go to the compact form at installation (-1: auto, by batch size) */
    double *h_poses = nullptr;      /* pinned: dvo_get_poses / dvo_set_poses staging, 12 doubles per pair */
    unsigned long long *d_dbg = nullptr;
    char *d_states = nullptr;       /* n_pairs x pose_state_bytes(): host-driven iteration state */
    /* one-launch-per-iteration schedule (dvo_align_pyramid_wide / _tiled, round 4): two optimiser states (double-buffered), two
     * rows of 32 reduced sums (alternating), the arrival ticket of the partial rows */
    char *d_step_state = nullptr;
    double *d_step_acc = nullptr;
    unsigned *d_step_ticket = nullptr;
    hipGraphExec_t tiled_exec = nullptr;     /* the tiled schedule (incl. its ncclAllReduce calls) as a replayable graph */
    unsigned long long tiled_sig = 0;
    bool tiled_graph_used = false;           /* inspection: the last dvo_align_pyramid_tiled replayed its graph */
    float *d_iter_energy = nullptr; /* n_pairs x iter_energy_cap */
    int iter_energy_cap = 0;
    std::vector<int> iter_max;      /* per pair: max_iters of the running dvo_iter_begin (0 = none) */
    int *d_colcounts = nullptr;
    int *d_order = nullptr;         /* launch order of the pairs of a large batch (longest first) */
    size_t order_cap = 0;
    std::vector<int> h_order;
    unsigned long long points_gen = 1;   /* bumped whenever a reference list changes (its length is the launch-order key) */
    unsigned long long order_key[4] = {0, 0, 0, 0};   /* points_gen, first_pair, n_pairs, hash of the schedule the resident order was made for */
    unsigned *pal_work = nullptr;   /* scratch of the compact-now-form builder (dvo_palette.hip) */
    size_t pal_work_ints = 0;
    size_t colcounts_cap = 0;
    dvo_host::FrameStore fs;
    /* cv::undistort of the publisher: fixed-point map of the camera's full resolution (dvo_frames_set_undistort) */
    short2 *d_umap_xy = nullptr;
    unsigned short *d_umap_frac = nullptr;
    int umap_rows = 0, umap_cols = 0;
    int *work = nullptr;            /* preprocessing scratch (Canny / distance transform / point counts) */
    size_t work_bytes = 0;
    /* frame uploads: two landing buffers filled by a copy stream while the context stream preprocesses the other */
    /* camera frames that already sit in HBM are read where they are (round 6): their addresses go up as a table (pinned staging -> device) */
    void **src_tab_dev = nullptr, **src_tab_host = nullptr;
    int src_tab_cap = 0;
    hipEvent_t ev_src_tab = nullptr;
    unsigned char *up_buf[2] = {nullptr, nullptr};
    unsigned char *up_host[2] = {nullptr, nullptr};      /* pinned mirrors: small images are gathered here and go up in one copy */
    size_t up_bytes = 0, up_host_bytes = 0;              /* landing buffers / their pinned mirrors (no mirror for device sources) */
    hipStream_t copy_stream = nullptr, copy_stream2 = nullptr;      /* two SDMA queues: frames alternate between them */
    hipEvent_t ev_copied[2] = {nullptr, nullptr}, ev_copied2[2] = {nullptr, nullptr}, ev_done[2] = {nullptr, nullptr};
    bool up_used[2] = {false, false};
    int up_next = 0;
    /* small batches (a single camera stream): the pyramid levels are independent kernel chains, run side by side */
    hipStream_t lvl_stream[DVO_LEVELS] = {};
    hipEvent_t ev_fork = nullptr, ev_join[DVO_LEVELS] = {};
    struct dvo_photo_state *photo = nullptr;     /* dvo_capi_photo.cpp: the photometric engine's reference data */
    struct dvo_keep_warm_state *warm = nullptr;  /* dvo_set_keep_warm: the thread that keeps the GPU at its active clocks */
    dvo::Schedule sched{};
    bool have_sched = false;
    /* which pairs the per-pair outputs (energies, final outputs) currently describe: the output buffers are laid out by
     * the schedule of the enqueue that wrote them, so a pair aligned under an older, differently shaped schedule (or before
     * the buffers were re-allocated) has nothing valid to report */
    int sched_gen = 0;
    std::vector<int> pair_gen;      /* per pair: sched_gen of the enqueue that last aligned it (0 = never) */
    std::vector<unsigned long long> final_list_gen;   /* per pair: list_gen of the finest level's list the last enqueue aligned */
    std::string err;
};


namespace dvo_host {

/* A context lives on the device that was current when it was created.  Every C entry point makes that device current for its
 * duration and restores the caller's afterwards, so a context can be driven from a thread whose current device differs
 * (one host thread per GPU). */
struct DeviceGuard {
    int prev = -1, dev = -1;
    explicit DeviceGuard(const dvo_ctx *c) {
        if (!c) return;
        dev = c->device;
        if (hipGetDevice(&prev) != hipSuccess) { prev = -1; return; }
        if (prev != dev) (void)hipSetDevice(dev);
    }
    ~DeviceGuard() { if (prev >= 0 && prev != dev) (void)hipSetDevice(prev); }
    DeviceGuard(const DeviceGuard &) = delete;
    DeviceGuard &operator=(const DeviceGuard &) = delete;
};
#define DVO_ENTER(c)                         \
    if (!(c)) return DVO_ERR_INVALID;        \
    dvo_host::DeviceGuard dvo_device_guard_(c)

int fail(dvo_ctx *c, int code, const std::string &msg);
/* Wait for a stream WITHOUT going to sleep on an interrupt first: poll its completion (hipStreamQuery reads the queue's
 * signal) for up to two milliseconds, only then block in hipStreamSynchronize.  On this pool a process can find the wake-up
 * from a blocking wait delayed by 14-33 ms (measured on the single-camera-stream path, profiles/r03_single_stream: kernels of
 * 0.2 ms, waits of 24 ms, intermittently per process); a single stream's frame is over in well under a millisecond, so the
 * poll costs one busy host thread for that long.  DVO_WAIT=block in the environment restores the plain blocking wait. */
hipError_t stream_wait(hipStream_t s);
#define HIPCHK(c, expr)                                                                     \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess)                                                               \
            return dvo_host::fail((c), DVO_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

int ensure_staging(dvo_ctx *c, size_t bytes);
int ensure_work(dvo_ctx *c, size_t bytes);
bool pair_ok(const dvo_ctx *c, int pair);
bool level_ok(int level);
int ensure_points(dvo_ctx *c, int level, int N);                 /* room for N points per pair at `level` (keeps contents) */
int ensure_texels(dvo_ctx *c, int level, int rows, int cols);
/* the now level of `pair` at `level` was (re)written: mark it present and its compact form stale */
int now_written(dvo_ctx *c, int level, int first_pair, int count);
/* the slabs of the compact form of `level` (allocated at first use) */
int ensure_compact_slabs(dvo_ctx *c, int level);
/* the now level of these pairs was (re)written in its compact form only (launch_edges_to_now with p4 != NULL) */
int now_written_compact(dvo_ctx *c, int level, int first_pair, int count);
/* make the 16-byte texels of these pairs' now level valid (decoded from the compact form where that is all there is): for
 * every path that reads texels -- everything except the packed fused kernel on the compact form */
int ensure_tex16(dvo_ctx *c, int level, int first_pair, int count);
/* true if now levels written by the engine's own distance-transform stage should be compact (false: DVO_COMPACT_NOW=off or
 * dvo_params.engine_variant = 4 -- plain 16-byte texels) */
bool native_compact_wanted(const dvo_ctx *c);
/* build the compact form of the stale now levels among [first_pair, first_pair+count) at `level`; with only_reused, only
 * of those that have been aligned DVO_COMPACT_NOW_AFTER times (the build costs about 4.4 alignments) */
int build_compact_now(dvo_ctx *c, int level, int first_pair, int count, bool only_reused);
/* schedule / readiness / output bookkeeping of the align entry points (dvo_capi.cpp) */
dvo::LevelSlab slab_of(const dvo_ctx *c, int level);
int check_ready(dvo_ctx *c, int pair, int level);
int build_schedule(dvo_ctx *c, int n_levels, const int *iters, int flags, dvo::Schedule &sc);
int ensure_outputs(dvo_ctx *c, const dvo::Schedule &sc);
void stamp_outputs(dvo_ctx *c, const dvo::Schedule &sc, int first, int n);
/* backs the texels of pairs [first, first + count) of a level with memory (no-op for a dense slab) */
int map_texels(dvo_ctx *c, int level, int first, int count, hipStream_t stream = nullptr);
void free_texels(dvo_ctx *c, Level &L);
/* sparse slabs, after a distance-transform launch over pairs [first, first + count) that ran WITHOUT texel output: waits for it,
 * reads the palette sizes back and maps the texels of the images the compact form could not hold; *n_failed = how many */
int sparse_map_compact_failures(dvo_ctx *c, int level, int first, int count, hipStream_t stream, int *n_failed);
int refresh_p4_known(dvo_ctx *c, int level, int first, int count, bool skip_native = false);      /* reads pal_n back where the host does not know it yet (skip_native: only where a refusal is possible) */
/* enqueues the level schedule of one pair as ONE launch per iteration on c->stream (dvo_kernels.hip: tiled_step_kernel): this
 * rank's contiguous share of every level's points (rank / world: dvo_tiled_shard's decomposition), `all_reduce` (may be empty:
 * one GPU) called on the 32 sums between two launches.  Pose in / out through d_pose (12 doubles on the device). */
hipError_t enqueue_step_schedule(dvo_ctx *c, const dvo::Schedule &sc, int pair, int flags, double *d_pose, int rank, int world,
                                 const std::function<hipError_t(double *)> &all_reduce);
int ensure_step_buffers(dvo_ctx *c);
hipError_t team_err_fetch(dvo_ctx *c, int *pinned_slot);      /* the error word of the last team launch -> a pinned slot, on the stream (no wait) */
int team_err_result(dvo_ctx *c, const int *pinned_slot);       /* after the caller's wait: DVO_ERR_HIP if it was set */
int team_err_check(dvo_ctx *c);       /* after a wait: DVO_ERR_HIP if a member of the last team launch gave up waiting for its team */
/* the wide / tiled schedule's coarse levels as one team launch of the fused kernel (dvo_capi.cpp) */
int wide_coarse_levels_as_team(dvo_ctx *c, int pair, int n_levels, const int *iters, int flags, dvo::Schedule &sc, const double *h_pose_in,
                               double *d_pose, unsigned &coarse_mask, bool &coarse_team, bool allow_finest);
int check_step_lost(dvo_ctx *c);      /* after the wait of a wide / tiled alignment: DVO_ERR_HIP if a step launch lost a workgroup's rows */
unsigned long long step_schedule_signature(dvo_ctx *c, const dvo::Schedule &sc, int pair, int n_levels, int flags, int rank, int world);
/* the reference lists of pairs [first, first + n) of a level were (re)written: bumps points_gen and the pairs' list stamps;
 * rows > 0: the rows of the image their 4-byte twins were encoded against (0: no valid 4-byte twin) */
void ref_list_written(dvo_ctx *c, int level, int first, int n, int rows);
void tiled_forget(dvo_ctx *c);
void photo_forget(dvo_ctx *c);           /* dvo_capi_photo.cpp */          /* dvo_capi_tiled.cpp: drop the RCCL attachment of a context */

}  // namespace dvo_host
#endif
