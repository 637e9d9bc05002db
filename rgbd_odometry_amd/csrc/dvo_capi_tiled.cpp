/*
 * dvo_capi_tiled.cpp -- the tiled mode of SURVEY.md 8(e) behind the C ABI: ONE large frame whose reference point lists are
 * sharded over the GPUs of a node, the per-iteration all-reduce of the 32 accumulator doubles done by RCCL over xGMI, the
 * whole level schedule of SolveDVO::loop (reference src/SolveDVO.cpp:2097-2104) enqueued from C on the context stream:
 *
 *     per iteration:  ONE kernel -- the previous iteration's 6-DoF update (:724-920) at its head, executed identically by every
 *                     workgroup of every rank; this rank's point range at the new pose; the 32 sums written by the workgroup that
 *                     arrives last (dvo_kernels.hip: tiled_step_kernel) -- then ncclAllReduce(32, ncclDouble, ncclSum);
 *                     the whole schedule captured into one graph (rounds 1-3: accumulate, reduce, all-reduce, update per iteration)
 *
 * No Python in the loop (rgbd_odometry_amd/distributed.py::TiledAligner is the torch.distributed twin used by the CPU/gloo
 * tests).  RCCL is NOT a link-time dependency of libdvo_amd.so: the communicator is created by the caller (a C++ ROS node
 * links RCCL itself), and ncclAllReduce is resolved when the communicator is attached -- from the library the caller names,
 * else from what is already loaded in the process, else from librccl.so.1.
 *
 * Every rank holds the full now pyramid (replicated) and the full point lists; rank r works on the contiguous index range
 * shard(N_level, r, world) (= a vertical strip of the reference image, :237-239).  All ranks receive the same reduced
 * bits from the all-reduce, hence take identical pose steps; best-iterate bookkeeping is replicated, not communicated.
 */
#include "dvo_ctx.h"

#include <dlfcn.h>
#include <mutex>

using namespace dvo;
using namespace dvo_host;

namespace {

enum { kNcclSum = 0, kNcclDouble = 8 };     /* rccl.h: ncclRedOp_t ncclSum = 0, ncclDataType_t ncclDouble = ncclFloat64 = 8 */
typedef int (*nccl_allreduce_fn)(const void *, void *, size_t, int, int, void *, hipStream_t);
typedef const char *(*nccl_errstr_fn)(int);

struct Tiled {
    void *comm = nullptr;
    int rank = 0, world = 1;
    nccl_allreduce_fn all_reduce = nullptr;
    nccl_errstr_fn errstr = nullptr;
    void *lib = nullptr;            /* dlopen handle we own (or nullptr) */
    bool no_graph = false;          /* the runtime refused to capture the schedule with this communicator: direct submission */
};

/* one record per context, kept outside struct dvo_ctx (only this file knows RCCL) */
struct Entry { dvo_ctx *ctx; Tiled t; };
/* One process may drive several contexts from several host threads (one per GPU): the registry is
 * the only state they share, so every access takes its lock.  Entries are heap records: the pointer find() returns stays
 * valid while other threads attach or detach THEIR contexts (a context itself is still one-thread-at-a-time). */
std::mutex &registry_mutex() { static std::mutex m; return m; }
std::vector<Entry *> &registry() { static std::vector<Entry *> r; return r; }
Tiled *find(dvo_ctx *c) {
    std::lock_guard<std::mutex> lock(registry_mutex());
    for (Entry *e : registry()) if (e->ctx == c) return &e->t;
    return nullptr;
}

/* communicators attached to contexts on this context's device: more than one = several ranks of one process share a GPU (the thread-rank
 * arrangement of tests/test_gpu_tiled_ranks.py; a real node has one rank per GPU).  Their team launches would compete for co-residency --
 * three half-resident teams of 128 can hold each other's compute units -- so the coarse levels then keep the step launches. */
int ranks_on_device_of(dvo_ctx *c) {
    std::lock_guard<std::mutex> lock(registry_mutex());
    int n = 0;
    for (Entry *e : registry()) n += (e->ctx->device == c->device) ? 1 : 0;
    return n;
}

void shard(int n, int rank, int world, int &first, int &count) {      /* same decomposition as distributed.py::shard_range */
    const int base = n / world, rem = n % world;
    count = base + (rank < rem ? 1 : 0);
    first = rank * base + (rank < rem ? rank : rem);
}

}  // namespace

namespace dvo_host {
void tiled_forget(dvo_ctx *c) {           /* called by dvo_destroy */
    Entry *gone = nullptr;
    {
        std::lock_guard<std::mutex> lock(registry_mutex());
        auto &r = registry();
        for (size_t i = 0; i < r.size(); i++)
            if (r[i]->ctx == c) { gone = r[i]; r.erase(r.begin() + i); break; }
    }
    if (!gone) return;
    if (gone->t.lib) dlclose(gone->t.lib);
    delete gone;
}
}  // namespace dvo_host

extern "C" {

int dvo_tiled_attach(dvo_ctx *c, void *nccl_comm, int rank, int world, const char *rccl_library) {
    DVO_ENTER(c);
    if (!nccl_comm || world < 1 || rank < 0 || rank >= world) return fail(c, DVO_ERR_INVALID, "bad communicator / rank / world size");
    tiled_forget(c);
    Tiled t;
    t.comm = nccl_comm; t.rank = rank; t.world = world;
    void *sym = nullptr;
    if (rccl_library && *rccl_library) {
        t.lib = dlopen(rccl_library, RTLD_NOW | RTLD_GLOBAL);
        if (!t.lib) return fail(c, DVO_ERR_INVALID, std::string("cannot load ") + rccl_library + ": " + dlerror());
        sym = dlsym(t.lib, "ncclAllReduce");
    } else {
        sym = dlsym(RTLD_DEFAULT, "ncclAllReduce");             /* the RCCL the caller's process already uses */
        if (!sym) {
            t.lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
            if (t.lib) sym = dlsym(t.lib, "ncclAllReduce");
        }
    }
    if (!sym) {
        if (t.lib) dlclose(t.lib);
        return fail(c, DVO_ERR_INVALID, "ncclAllReduce not found: link/load RCCL (librccl.so.1) or pass its path");
    }
    t.all_reduce = (nccl_allreduce_fn)sym;
    t.errstr = (nccl_errstr_fn)(t.lib ? dlsym(t.lib, "ncclGetErrorString") : dlsym(RTLD_DEFAULT, "ncclGetErrorString"));
    {
        std::lock_guard<std::mutex> lock(registry_mutex());
        registry().push_back(new Entry{c, t});
    }
    return DVO_OK;
}

int dvo_tiled_detach(dvo_ctx *c) {
    DVO_ENTER(c);
    if (c->stream) (void)stream_wait(c->stream);
    if (c->tiled_exec) { (void)hipGraphExecDestroy(c->tiled_exec); c->tiled_exec = nullptr; }      /* it refers to the communicator */
    tiled_forget(c);
    return DVO_OK;
}

int dvo_tiled_shard(dvo_ctx *c, int pair, int level, int *first, int *count) {
    DVO_ENTER(c);
    if (!pair_ok(c, pair) || !level_ok(level) || !first || !count) return fail(c, DVO_ERR_INVALID, "bad arguments");
    Tiled *T = find(c);
    if (!T) return fail(c, DVO_ERR_STATE, "no communicator attached (dvo_tiled_attach)");
    const int N = c->lv[level].hN.empty() ? 0 : c->lv[level].hN[pair];
    shard(N, T->rank, T->world, *first, *count);
    return DVO_OK;
}

int dvo_tiled_graph_replayed(dvo_ctx *c, int *graph_replayed) {
    DVO_ENTER(c);
    if (!graph_replayed) return fail(c, DVO_ERR_INVALID, "graph_replayed is NULL");
    *graph_replayed = c->tiled_graph_used ? 1 : 0;
    return DVO_OK;
}

int dvo_wide_packed_levels(dvo_ctx *c, int *levels_mask, int *solo_mask) {
    DVO_ENTER(c);
    if (!levels_mask) return fail(c, DVO_ERR_INVALID, "levels_mask is NULL");
    *levels_mask = c->step_pk_mask;
    if (solo_mask) *solo_mask = c->step_solo_mask;
    return DVO_OK;
}

int dvo_wide_team_levels(dvo_ctx *c, int *levels_mask) {
    DVO_ENTER(c);
    if (!levels_mask) return fail(c, DVO_ERR_INVALID, "levels_mask is NULL");
    *levels_mask = c->wide_team_mask;
    return DVO_OK;
}

int dvo_align_pyramid_tiled(dvo_ctx *c, int pair, int n_levels, const int *iters, int flags, double *R, double *t) {
    DVO_ENTER(c);
    if (!pair_ok(c, pair) || !R || !t) return fail(c, DVO_ERR_INVALID, "bad arguments");
    if (flags & ~(DVO_FLAG_FINAL_OUTPUTS | DVO_FLAG_NORMAL_MATRIX)) return fail(c, DVO_ERR_INVALID, "dvo_align_pyramid_tiled takes DVO_FLAG_FINAL_OUTPUTS and DVO_FLAG_NORMAL_MATRIX only");
    Tiled *T = find(c);
    if (!T) return fail(c, DVO_ERR_STATE, "no communicator attached (dvo_tiled_attach)");
    Schedule sc;
    int rc = dvo_host::build_schedule(c, n_levels, iters, flags, sc);
    if (rc) return rc;
    for (int l = 0; l < n_levels; l++)
        if (sc.iters[l] > 0 && ((rc = dvo_host::check_ready(c, pair, l)) || (rc = dvo_host::ensure_tex16(c, l, pair, 1)))) return rc;
    if ((rc = dvo_host::ensure_outputs(c, sc))) return rc;
    if ((rc = dvo_host::ensure_step_buffers(c))) return rc;
    if (!c->h_pose) HIPCHK(c, hipHostMalloc((void **)&c->h_pose, sizeof(double) * 14, hipHostMallocDefault));      /* [12]: the step launches' error word, [13]: the team launches' */
    double *h = c->h_pose;
    std::memcpy(h, R, sizeof(double) * 9);
    std::memcpy(h + 9, t, sizeof(double) * 3);
    double *d_pose = c->d_poses + (size_t)12 * pair;
    /* round 6: the coarse levels of a large frame as ONE team launch of the fused kernel, run whole by every rank (no collective: identical
     * inputs, a fixed order of additions, identical bits); `sc` keeps the fine levels for the step launches below */
    unsigned coarse_mask = 0;
    bool coarse_team = false;
    /* DVO_TILED_TEAM_SHARED=1 (tests): hand the coarse levels over although several ranks share the device */
    if ((ranks_on_device_of(c) <= 1 || std::getenv("DVO_TILED_TEAM_SHARED") != nullptr) &&
        (rc = dvo_host::wide_coarse_levels_as_team(c, pair, n_levels, iters, flags, sc, h, d_pose, coarse_mask, coarse_team, T->world == 1))) return rc;
    const bool all_team = coarse_mask != 0 && sc.last_level < 0;      /* one rank, every level inside team launches: the sequence below is two copies */
    /* Per iteration ONE kernel and ONE collective (round 4; rounds 1-3: accumulate, reduce, all-reduce, update): the update of an
     * iteration is applied at the head of the next iteration's launch by every workgroup of every rank from the same all-reduced
     * bits (dvo_kernels.hip: tiled_step_kernel), the 32 sums of a launch are written by its last workgroup.  The whole schedule,
     * ncclAllReduce calls included, is captured once into a graph and replayed (RCCL collectives are capturable); a capture the
     * runtime refuses falls back to direct submission (DVO_TILED_NO_GRAPH=1 forces that for A/B measurements). */
    int nccl_rc = 0;
    auto all_reduce = [&](double *buf) -> hipError_t {
        const int nrc = T->all_reduce(buf, buf, DVO_NACC_PAD, kNcclDouble, kNcclSum, T->comm, c->stream);
        if (nrc != 0 && nccl_rc == 0) nccl_rc = nrc;
        return nrc == 0 ? hipSuccess : hipErrorUnknown;
    };
    auto submit = [&]() -> hipError_t {
        hipError_t e = coarse_mask ? hipSuccess : hipMemcpyAsync(d_pose, h, sizeof(double) * 12, hipMemcpyHostToDevice, c->stream);      /* else: went in front of the team launch */
        if (e == hipSuccess) e = dvo_host::enqueue_step_schedule(c, sc, pair, flags, d_pose, T->rank, T->world, all_reduce);
        if (e == hipSuccess) e = hipMemcpyAsync(h, d_pose, sizeof(double) * 12, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(h + 12, c->d_step_ticket + 2, sizeof(unsigned), hipMemcpyDeviceToHost, c->stream);
        return e;
    };
    auto nccl_fail = [&]() { return fail(c, DVO_ERR_HIP, std::string("ncclAllReduce: ") + (T->errstr ? T->errstr(nccl_rc) : "error " + std::to_string(nccl_rc))); };
    static const bool env_no_graph = std::getenv("DVO_TILED_NO_GRAPH") != nullptr;
    /* graph replay of a schedule that contains REAL multi-rank collectives has never run (the boxes have one GPU; the
     * loopback stand-in of the tests refuses capture), and the fallback below only triggers when the capture fails, not when a
     * replay misbehaves.  So for world > 1 the graph is opt-in (DVO_TILED_GRAPH_MULTIRANK=1) until an 8-GPU node has validated it;
     * one rank -- where RCCL's in-place all-reduce launches nothing -- keeps the graph. */
    static const bool env_graph_multirank = std::getenv("DVO_TILED_GRAPH_MULTIRANK") != nullptr;
    const unsigned long long sig = dvo_host::step_schedule_signature(c, sc, pair, n_levels, flags, T->rank, T->world) ^ (unsigned long long)(size_t)T->comm ^ ((unsigned long long)coarse_mask << 48);
    bool direct = env_no_graph || c->stream == nullptr || T->no_graph || (T->world > 1 && !env_graph_multirank);
    if (!direct && (!c->tiled_exec || sig != c->tiled_sig)) {
        if (c->tiled_exec) { (void)hipGraphExecDestroy(c->tiled_exec); c->tiled_exec = nullptr; }
        hipError_t e = hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal);
        if (e == hipSuccess) {
            e = submit();
            hipGraph_t graph = nullptr;
            const hipError_t e2 = hipStreamEndCapture(c->stream, &graph);
            if (e == hipSuccess) e = e2;
            if (e == hipSuccess) e = hipGraphInstantiate(&c->tiled_exec, graph, nullptr, nullptr, 0);
            if (graph) (void)hipGraphDestroy(graph);
        }
        if (e != hipSuccess) {          /* this RCCL / runtime pair does not capture the collective: submit directly from now on */
            (void)hipGetLastError();
            if (c->tiled_exec) { (void)hipGraphExecDestroy(c->tiled_exec); c->tiled_exec = nullptr; }
            T->no_graph = true;
            direct = true;
            nccl_rc = 0;
        } else {
            c->tiled_sig = sig;
        }
    }
    if (direct) {
        const hipError_t e = submit();
        if (nccl_rc != 0) return nccl_fail();
        HIPCHK(c, e);
    } else {
        HIPCHK(c, hipGraphLaunch(c->tiled_exec, c->stream));
    }
    c->team_used = coarse_team;
    HIPCHK(c, dvo_host::team_err_fetch(c, reinterpret_cast<int *>(h + 13)));
    HIPCHK(c, stream_wait(c->stream));
    { const int lrc = dvo_host::check_step_lost(c); if (lrc) return lrc; }
    if ((rc = dvo_host::team_err_result(c, reinterpret_cast<const int *>(h + 13)))) return rc;
    std::memcpy(R, h, sizeof(double) * 9);
    std::memcpy(t, h + 9, sizeof(double) * 3);
    if (coarse_mask && (rc = dvo_host::build_schedule(c, n_levels, iters, flags, sc))) return rc;      /* the outputs follow the WHOLE schedule */
    if (all_team) sc.final_blk = 1;                        /* final outputs written by the fused kernel sit in the compact list's order */
    dvo_host::stamp_outputs(c, sc, pair, 1);
    c->sched = sc;
    c->have_sched = true;
    c->team_used = coarse_team;
    c->tiled_graph_used = !direct;
    c->wide_team_mask = (int)coarse_mask;
    return DVO_OK;
}

}  // extern "C"
