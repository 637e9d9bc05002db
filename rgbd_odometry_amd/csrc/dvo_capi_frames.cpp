/*
 * dvo_capi_frames.cpp -- the frame-store half of the C ABI (include/dvo_amd.h, "frames in"): SURVEY.md section 8f rows
 * f1 + f2.  Host side only: slabs, uploads (double-buffered landing buffers, two copy streams, pinned mirror for small
 * images), per-level execution lanes, kernel launches of dvo_frames.hip.  No CPU compute path.
 */
#include "dvo_ctx.h"
#include <chrono>
#include <cstdio>

using namespace dvo;
using namespace dvo_host;

extern "C" {

/* ---- frame store: rows f1 + f2 ------------------------------------------------------------------------ */
namespace {

constexpr size_t kWorkBudget = (size_t)1 << 30;     /* scratch per chunk of a batched preprocessing call */

int round_half_even_pos(double v) {                 /* cvRound for the non-negative sizes used here */
    const double f = std::floor(v);
    const double d = v - f;
    int i = (int)f;
    if (d > 0.5 || (d == 0.5 && (i & 1))) i++;
    return i;
}

void canny_thresholds(const dvo_ctx *c, int *low, int *high) {
    double t1 = c->prm.canny_threshold1, t2 = c->prm.canny_threshold2;
    if (t1 == 0 && t2 == 0) { t1 = 150; t2 = 100; }               /* SolveDVO.cpp:1704, :1764 */
    double lo = std::min(t1, t2), hi = std::max(t1, t2);          /* the detector swaps them */
    lo = std::min(32767.0, lo); hi = std::min(32767.0, hi);
    if (lo > 0) lo *= lo;                                         /* L2gradient: squared magnitudes */
    if (hi > 0) hi *= hi;
    *low = (int)std::floor(lo); *high = (int)std::floor(hi);
}

int frames_default_slots(const dvo_ctx *c) { return std::min(2 * c->n_pairs + 2, 64); }

void frames_free(dvo_ctx *c) {
    for (int l = 0; l < DVO_LEVELS; l++) {
        FrameLevel &F = c->fs.lv[l];
        void *fp[] = {F.grey, F.edge, F.depth};
        for (void *p : fp) if (p) (void)hipFree(p);
        F = FrameLevel();
    }
    c->fs.n_levels = 0;
    std::fill(c->fs.valid.begin(), c->fs.valid.end(), 0);
}

/* make the store hold `n_levels` levels of the given geometry (drops the stored frames if it changes) */
int frames_geometry(dvo_ctx *c, int n_levels, const int *rows, const int *cols) {
    FrameStore &S = c->fs;
    for (int l = 0; l < n_levels; l++)      /* squared distances are kept in 32-bit integers: (rows+cols+1)^2 < 2^31 */
        if ((long long)rows[l] + cols[l] + 1 > 46340)
            return fail(c, DVO_ERR_INVALID, "image too large for the exact distance transform (rows + cols must stay below 46339)");
    if (S.n_slots == 0) {
        S.n_slots = frames_default_slots(c);
        S.valid.assign(S.n_slots, 0); S.has_depth.assign(S.n_slots, 0);
    }
    bool same = S.n_levels == n_levels;
    for (int l = 0; same && l < n_levels; l++) same = S.lv[l].rows == rows[l] && S.lv[l].cols == cols[l];
    if (same) return DVO_OK;
    HIPCHK(c, stream_wait(c->stream));
    frames_free(c);
    for (int l = 0; l < n_levels; l++) {
        FrameLevel &F = S.lv[l];
        F.rows = rows[l]; F.cols = cols[l]; F.npx = (size_t)rows[l] * cols[l];
        HIPCHK(c, hipMalloc((void **)&F.grey, F.npx * S.n_slots));
        HIPCHK(c, hipMalloc((void **)&F.edge, F.npx * S.n_slots));
        HIPCHK(c, hipMalloc((void **)&F.depth, sizeof(float) * F.npx * S.n_slots));
    }
    S.n_levels = n_levels;
    return DVO_OK;
}

bool slots_ok(const dvo_ctx *c, int first, int count) {
    return first >= 0 && count >= 1 && first + count <= c->fs.n_slots;
}

int chunk_for(size_t bytes_per_image, int count) {
    size_t k = kWorkBudget / std::max<size_t>(bytes_per_image, 1);
    if (k < 1) k = 1;
    return (int)std::min<size_t>(k, (size_t)count);
}

/* Per-level execution lanes.  Large batches: every level on the context stream, scratch shared and chunked.
 * Small batches (one camera stream): each level's kernel chain on its own stream with its own scratch, forked from
 * and joined back into the context stream -- the chains are independent, so a frame costs the longest chain instead
 * of their sum. */
struct LevelLanes {
    bool parallel = false;
    hipStream_t s[DVO_LEVELS];
    int *work[DVO_LEVELS];          /* nullptr: use c->work with chunking */
};
constexpr size_t kParallelPixels = (size_t)4 << 20;

int lanes_begin(dvo_ctx *c, int n_levels, int count, bool with_now, LevelLanes &ln) {
    int rc;
    if (with_now)
        for (int l = 0; l < n_levels; l++)
            if ((rc = ensure_texels(c, l, c->fs.lv[l].rows, c->fs.lv[l].cols))) return rc;
    /* OFF unless DVO_FRAME_LANES=1 (round 3).  The lanes save ~0.13 ms of a single 640x480 frame when the hardware queues
     * behind the four extra streams are live, but the fork / join events make every frame wait on cross-queue dependencies,
     * and a process that is not the first on the GPU can find those waits taking 14-33 ms EACH FRAME (profiles/
     * r03_single_stream: the C++ file replay as second process of a box, 24 ms per frame with lanes, 0.64 ms without; neither
     * clocks nor host time -- the alignment itself stayed at 0.4 ms).  One stream, levels back to back, is the robust default. */
    static const int lanes_env = [] { const char *e = std::getenv("DVO_FRAME_LANES"); return e ? std::atoi(e) : 0; }();
    ln.parallel = n_levels > 1 && (size_t)count * c->fs.lv[0].npx <= kParallelPixels && lanes_env == 1;
    for (int l = 0; l < n_levels; l++) { ln.s[l] = c->stream; ln.work[l] = nullptr; }
    if (!ln.parallel) return DVO_OK;
    size_t off[DVO_LEVELS + 1];
    off[0] = 0;
    for (int l = 0; l < n_levels; l++) {
        const FrameLevel &F = c->fs.lv[l];
        size_t need = canny_work_ints(F.rows, F.cols, count);
        if (with_now) need = std::max(need, edt_work_ints(F.rows, F.cols, count));
        off[l + 1] = off[l] + (need + 31) / 32 * 32;
    }
    if ((rc = ensure_work(c, sizeof(int) * off[n_levels]))) return rc;
    if (!c->ev_fork) HIPCHK(c, hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
    for (int l = 0; l < n_levels; l++) {
        if (!c->lvl_stream[l]) {
            HIPCHK(c, hipStreamCreateWithFlags(&c->lvl_stream[l], hipStreamNonBlocking));
            HIPCHK(c, hipEventCreateWithFlags(&c->ev_join[l], hipEventDisableTiming));
        }
        ln.s[l] = c->lvl_stream[l];
        ln.work[l] = c->work + off[l];
    }
    return DVO_OK;
}
int lanes_fork(dvo_ctx *c, int n_levels, const LevelLanes &ln) {
    if (!ln.parallel) return DVO_OK;
    HIPCHK(c, hipEventRecord(c->ev_fork, c->stream));
    for (int l = 0; l < n_levels; l++) HIPCHK(c, hipStreamWaitEvent(ln.s[l], c->ev_fork, 0));
    return DVO_OK;
}
int lanes_join(dvo_ctx *c, int n_levels, const LevelLanes &ln) {
    if (!ln.parallel) return DVO_OK;
    for (int l = 0; l < n_levels; l++) {
        HIPCHK(c, hipEventRecord(c->ev_join[l], ln.s[l]));
        HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_join[l], 0));
    }
    return DVO_OK;
}

int run_canny(dvo_ctx *c, int level, int first_slot, int count, hipStream_t stream, int *work) {
    FrameLevel &F = c->fs.lv[level];
    int low, high;
    canny_thresholds(c, &low, &high);
    if (work) {
        const size_t off = (size_t)first_slot * F.npx;
        HIPCHK(c, launch_canny(F.grey + off, F.npx, ImgBatch{F.rows, F.cols, count}, low, high, work, F.edge + off, F.npx, stream));
        return DVO_OK;
    }
    const int chunk = chunk_for(sizeof(int) * canny_work_ints(F.rows, F.cols, 1), count);
    int rc = ensure_work(c, sizeof(int) * canny_work_ints(F.rows, F.cols, chunk));
    if (rc) return rc;
    for (int b = 0; b < count; b += chunk) {
        const int nc = std::min(chunk, count - b);
        const size_t off = (size_t)(first_slot + b) * F.npx;
        HIPCHK(c, launch_canny(F.grey + off, F.npx, ImgBatch{F.rows, F.cols, nc}, low, high, c->work,
                               F.edge + off, F.npx, stream));
    }
    return DVO_OK;
}

/* Canny of all pyramid levels of `count` stored frames: four launches for all levels when the images allow it, else per level */
int run_canny_all(dvo_ctx *c, int n_levels, int first_slot, int count, hipStream_t stream) {
    int rows[DVO_LEVELS], cols[DVO_LEVELS];
    unsigned char *edge[DVO_LEVELS]; size_t estride[DVO_LEVELS];
    for (int l = 0; l < n_levels; l++) {
        const FrameLevel &F = c->fs.lv[l];
        rows[l] = F.rows; cols[l] = F.cols; estride[l] = F.npx; edge[l] = F.edge + (size_t)first_slot * F.npx;
    }
    static const bool per_level = getenv("DVO_CANNY_PER_LEVEL") != nullptr;
    if (per_level || !canny_levels_ok(n_levels, rows, cols, edge, estride)) {
        for (int l = 0; l < n_levels; l++) { const int rc = run_canny(c, l, first_slot, count, stream, nullptr); if (rc) return rc; }
        return DVO_OK;
    }
    int low, high;
    canny_thresholds(c, &low, &high);
    const int chunk = chunk_for(sizeof(int) * canny_levels_work_ints(n_levels, rows, cols, 1), count);
    int rc = ensure_work(c, sizeof(int) * canny_levels_work_ints(n_levels, rows, cols, chunk));
    if (rc) return rc;
    for (int b = 0; b < count; b += chunk) {
        const int nc = std::min(chunk, count - b);
        const unsigned char *grey[DVO_LEVELS]; size_t gstride[DVO_LEVELS]; unsigned char *e[DVO_LEVELS];
        for (int l = 0; l < n_levels; l++) {
            const FrameLevel &F = c->fs.lv[l];
            grey[l] = F.grey + (size_t)(first_slot + b) * F.npx; gstride[l] = F.npx; e[l] = F.edge + (size_t)(first_slot + b) * F.npx;
        }
        HIPCHK(c, launch_canny_levels(n_levels, rows, cols, grey, gstride, e, estride, nc, low, high, c->work, stream));
    }
    return DVO_OK;
}

constexpr size_t kUploadHalf = (size_t)32 << 20;   /* landing buffer per pipeline stage */
constexpr size_t kSmallImage = (size_t)256 << 10;  /* images up to this size are gathered on the host before they go up */

static size_t mapped_half() {                     /* ... when a kernel pulls them out of mapped host memory (DVO_MAPPED_CHUNK_MB) */
    static const size_t v = [] { const char *e = getenv("DVO_MAPPED_CHUNK_MB"); const long m = e ? atol(e) : 0; return (size_t)(m > 0 ? m : 64) << 20; }();
    return v;
}
#define kMappedHalf mapped_half()
constexpr size_t kDeviceHalf = (size_t)512 << 20;  /* landing buffer per stage when the sources are device buffers: whole batches */

/* landing buffers of at least `bytes` each (+ their pinned mirrors when the sources are host buffers) + copy streams + events */
int ensure_upload(dvo_ctx *c, size_t bytes, bool with_host = true) {
    if (!c->copy_stream) {
        /* highest priority: the kernels that pull mapped host memory (DVO_UPLOAD_MAPPED) must not queue behind the
         * preprocessing of the previous chunk -- the PCIe link is the longer pole */
        int prio_lo = 0, prio_hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
        HIPCHK(c, hipStreamCreateWithPriority(&c->copy_stream, hipStreamNonBlocking, prio_hi));
        HIPCHK(c, hipStreamCreateWithPriority(&c->copy_stream2, hipStreamNonBlocking, prio_hi));
        const unsigned evf = hipEventDisableTiming | hipEventDisableSystemFence;      /* producers and consumers are kernels / DMAs of this device */
        for (int b = 0; b < 2; b++) {
            HIPCHK(c, hipEventCreateWithFlags(&c->ev_copied[b], evf));
            HIPCHK(c, hipEventCreateWithFlags(&c->ev_copied2[b], evf));
            HIPCHK(c, hipEventCreateWithFlags(&c->ev_done[b], evf));
        }
    }
    const bool grow_dev = bytes > c->up_bytes, grow_host = with_host && bytes > c->up_host_bytes;
    if (!grow_dev && !grow_host) return DVO_OK;
    HIPCHK(c, hipStreamSynchronize(c->copy_stream));
    HIPCHK(c, hipStreamSynchronize(c->copy_stream2));
    HIPCHK(c, stream_wait(c->stream));
    for (int b = 0; b < 2; b++) {
        if (grow_dev) {
            if (c->up_buf[b]) HIPCHK(c, hipFree(c->up_buf[b]));
            c->up_buf[b] = nullptr;
            HIPCHK(c, hipMalloc((void **)&c->up_buf[b], bytes));
        }
        if (grow_host) {
            if (c->up_host[b]) HIPCHK(c, hipHostFree(c->up_host[b]));
            c->up_host[b] = nullptr;
            HIPCHK(c, hipHostMalloc((void **)&c->up_host[b], bytes, hipHostMallocDefault));
        }
        c->up_used[b] = false;
    }
    if (grow_dev) c->up_bytes = bytes;
    if (grow_host) c->up_host_bytes = bytes;
    return DVO_OK;
}
/* stage A of a chunk: returns the landing buffer; copies must go to c->copy_stream */
int upload_begin(dvo_ctx *c, unsigned char **buf, int *slot) {
    const int b = c->up_next;
    if (c->up_used[b]) {                                /* its previous consumer finished */
        HIPCHK(c, hipStreamWaitEvent(c->copy_stream, c->ev_done[b], 0));
        HIPCHK(c, hipStreamWaitEvent(c->copy_stream2, c->ev_done[b], 0));
    }
    *buf = c->up_buf[b]; *slot = b;
    return DVO_OK;
}
/* stage B: everything enqueued on c->stream after this sees the copies */
int upload_copied(dvo_ctx *c, int b) {
    HIPCHK(c, hipEventRecord(c->ev_copied[b], c->copy_stream));
    HIPCHK(c, hipEventRecord(c->ev_copied2[b], c->copy_stream2));
    HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_copied[b], 0));
    HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_copied2[b], 0));
    return DVO_OK;
}
/* stage C: the kernels reading the landing buffer are enqueued */
int upload_consumed(dvo_ctx *c, int b) {
    HIPCHK(c, hipEventRecord(c->ev_done[b], c->stream));
    c->up_used[b] = true;
    c->up_next = b ^ 1;
    return DVO_OK;
}

size_t pix_bytes(int dtype) { return dtype == DVO_PIX_U8 ? 1 : (dtype == DVO_PIX_U16 ? 2 : 4); }

}  // namespace

static int frames_as_now_level(dvo_ctx *c, int level, int first_slot, int first_pair, int count, hipStream_t stream, int *work);
static int frames_as_now_all(dvo_ctx *c, int n_levels, int first_slot, int first_pair, int count, hipStream_t stream);

int dvo_frames_reserve(dvo_ctx *c, int n_slots) {
    DVO_ENTER(c);
    if (n_slots < 1) return fail(c, DVO_ERR_INVALID, "n_slots must be >= 1");
    HIPCHK(c, stream_wait(c->stream));
    frames_free(c);
    c->fs.n_slots = n_slots;
    c->fs.valid.assign(n_slots, 0);
    c->fs.has_depth.assign(n_slots, 0);
    return DVO_OK;
}

int dvo_frames_num_levels(const dvo_ctx *c) { return c ? c->fs.n_levels : 0; }

int dvo_frames_upload_pyramids(dvo_ctx *c, int first_slot, int count, int n_levels,
                               const dvo_image *grey, const dvo_image *depth, int now_first_pair, int flags) {
    DVO_ENTER(c);
    if (!grey || count < 1 || n_levels < 1 || n_levels > DVO_LEVELS) return fail(c, DVO_ERR_INVALID, "bad frame arguments");
    int rows[DVO_LEVELS], cols[DVO_LEVELS];
    for (int l = 0; l < n_levels; l++) { rows[l] = grey[l].rows; cols[l] = grey[l].cols; }
    for (int f = 0; f < count; f++)
        for (int l = 0; l < n_levels; l++) {
            const dvo_image &g = grey[(size_t)f * n_levels + l];
            if (!g.data || g.rows < 1 || g.cols < 1 || g.rows != rows[l] || g.cols != cols[l] ||
                (g.dtype != DVO_PIX_U8 && g.dtype != DVO_PIX_F32) || g.dtype != grey[l].dtype || g.layout != grey[l].layout)
                return fail(c, DVO_ERR_INVALID, "grey images of one level must share size, dtype (U8/F32) and layout");
            if (depth) {
                const dvo_image &d = depth[(size_t)f * n_levels + l];
                if (!d.data || d.rows != rows[l] || d.cols != cols[l] || (d.dtype != DVO_PIX_U16 && d.dtype != DVO_PIX_F32) ||
                    d.dtype != depth[l].dtype || d.layout != depth[l].layout)
                    return fail(c, DVO_ERR_INVALID, "depth images must match the grey size and share dtype (U16/F32) and layout");
            }
        }
    int rc = frames_geometry(c, n_levels, rows, cols);
    if (rc) return rc;
    if (!slots_ok(c, first_slot, count)) return fail(c, DVO_ERR_INVALID, "frame slot range out of bounds (dvo_frames_reserve)");
    if (now_first_pair >= 0 && (!pair_ok(c, now_first_pair) || now_first_pair + count > c->n_pairs))
        return fail(c, DVO_ERR_INVALID, "now_first_pair range out of bounds");
    /* chunks of frames flow through copy (copy stream) -> import + Canny (context stream), double-buffered */
    size_t g_img[DVO_LEVELS], d_img[DVO_LEVELS], g_off[DVO_LEVELS], d_off[DVO_LEVELS], frame_bytes = 0;
    for (int l = 0; l < n_levels; l++) {
        const size_t npx = c->fs.lv[l].npx;
        g_img[l] = (npx * pix_bytes(grey[l].dtype) + 15) / 16 * 16;           /* 16-byte aligned images */
        d_img[l] = depth ? (npx * pix_bytes(depth[l].dtype) + 15) / 16 * 16 : 0;
        frame_bytes += g_img[l] + d_img[l];
    }
    const bool mapped = (flags & DVO_UPLOAD_MAPPED) != 0;   /* pinned host memory the GPU addresses: pulled by a kernel (dvo_amd.h) */
    const int chunk = (int)std::min<size_t>(std::max<size_t>((mapped ? kMappedHalf : kUploadHalf) / frame_bytes, 1), (size_t)count);
    if ((rc = ensure_upload(c, frame_bytes * chunk, !mapped))) return rc;
    std::vector<const void *> srcs;
    {   size_t o = 0;                                   /* landing layout: per level, `chunk` grey images then `chunk` depth images */
        for (int l = 0; l < n_levels; l++) { g_off[l] = o; o += g_img[l] * chunk; d_off[l] = o; o += d_img[l] * chunk; }
    }
    for (int b = 0; b < count; b += chunk) {
        const int nc = std::min(chunk, count - b);
        unsigned char *buf; int ub;
        if ((rc = upload_begin(c, &buf, &ub))) return rc;
        /* A sub-megabyte hipMemcpyAsync costs ~10 us of host time however small it is, and a pyramid is mostly small
         * images (the reference's 320x240 ... 40x30 levels: eight per frame): those are gathered into the pinned mirror
         * of the landing buffer with memcpy and go up in one copy per run of small levels; big images go up directly. */
        unsigned char *hbuf = c->up_host[ub];
        if (c->up_used[ub] && !mapped) {      /* the mirror's previous copies have left: both queues (the cameras path sends its depth half on the second) */
            HIPCHK(c, hipEventSynchronize(c->ev_copied[ub]));
            HIPCHK(c, hipEventSynchronize(c->ev_copied2[ub]));
        }
        bool small[DVO_LEVELS];
        for (int l = 0; l < n_levels && mapped; l++) {      /* one gather launch per level and 32 images, grey and depth on the two copy streams */
            const size_t npx = c->fs.lv[l].npx, gb = pix_bytes(grey[l].dtype), db = depth ? pix_bytes(depth[l].dtype) : 0;
            small[l] = false;
            const int wgs = std::max(1, 32 / std::min(nc, 32));
            srcs.resize(nc);
            for (int i = 0; i < nc; i++) srcs[i] = grey[(size_t)(b + i) * n_levels + l].data;
            HIPCHK(c, launch_gather_images(srcs.data(), nc, buf + g_off[l], npx * gb, g_img[l], c->copy_stream, wgs));
            if (depth) {
                for (int i = 0; i < nc; i++) srcs[i] = depth[(size_t)(b + i) * n_levels + l].data;
                HIPCHK(c, launch_gather_images(srcs.data(), nc, buf + d_off[l], npx * db, d_img[l], c->copy_stream2, wgs));
            }
        }
        for (int l = 0; l < n_levels && !mapped; l++) {
            const size_t npx = c->fs.lv[l].npx, gb = pix_bytes(grey[l].dtype), db = depth ? pix_bytes(depth[l].dtype) : 0;
            small[l] = !(flags & DVO_UPLOAD_DIRECT) || npx * std::max(gb, db) <= kSmallImage;     /* default: through the pinned mirror */
            for (int i = 0; i < nc; i++) {
                const void *gsrc = grey[(size_t)(b + i) * n_levels + l].data;
                const void *dsrc = depth ? depth[(size_t)(b + i) * n_levels + l].data : nullptr;
                if (small[l]) {
                    std::memcpy(hbuf + g_off[l] + g_img[l] * i, gsrc, npx * gb);
                    if (depth) std::memcpy(hbuf + d_off[l] + d_img[l] * i, dsrc, npx * db);
                } else {
                    hipStream_t cs = (i & 1) ? c->copy_stream2 : c->copy_stream;
                    HIPCHK(c, hipMemcpyAsync(buf + g_off[l] + g_img[l] * i, gsrc, npx * gb, hipMemcpyHostToDevice, cs));
                    if (depth) HIPCHK(c, hipMemcpyAsync(buf + d_off[l] + d_img[l] * i, dsrc, npx * db, hipMemcpyHostToDevice, cs));
                }
            }
        }
        for (int l = 0; l < n_levels;) {                                  /* one copy per maximal run of small levels */
            if (!small[l]) { l++; continue; }
            int e = l;
            while (e + 1 < n_levels && small[e + 1]) e++;
            const size_t lo = g_off[l], hi = (e + 1 < n_levels) ? g_off[e + 1] : frame_bytes * chunk;
            HIPCHK(c, hipMemcpyAsync(buf + lo, hbuf + lo, hi - lo, hipMemcpyHostToDevice, c->copy_stream));
            l = e + 1;
        }
        LevelLanes ln;
        if ((rc = lanes_begin(c, n_levels, nc, now_first_pair >= 0, ln))) return rc;
        if ((rc = upload_copied(c, ub))) return rc;
        if ((rc = lanes_fork(c, n_levels, ln))) return rc;
        for (int pass = 0; pass < 2; pass++) {              /* sequential lanes: all imports, then release the landing buffer, then the rest */
            for (int l = 0; l < n_levels; l++) {
                FrameLevel &F = c->fs.lv[l];
                const size_t off = (size_t)(first_slot + b) * F.npx;
                const ImgBatch ib{F.rows, F.cols, nc};
                if (pass == 0 || ln.parallel) {
                    HIPCHK(c, launch_import_grey(buf + g_off[l], grey[l].dtype, grey[l].layout == DVO_LAYOUT_ROW_MAJOR,
                                                 g_img[l] / pix_bytes(grey[l].dtype), F.grey + off, F.npx, ib, ln.s[l]));
                    if (depth)
                        HIPCHK(c, launch_import_depth(buf + d_off[l], depth[l].dtype, depth[l].layout == DVO_LAYOUT_ROW_MAJOR,
                                                      d_img[l] / pix_bytes(depth[l].dtype), F.depth + off, F.npx, ib, ln.s[l]));
                }
                if (pass == 1 && !ln.parallel && l == 0) {         /* one launch per stage for all levels */
                    if ((rc = run_canny_all(c, n_levels, first_slot + b, nc, c->stream))) return rc;
                    if (now_first_pair >= 0 && (rc = frames_as_now_all(c, n_levels, first_slot + b, now_first_pair + b, nc, c->stream))) return rc;
                }
                if (ln.parallel) {
                    if ((rc = run_canny(c, l, first_slot + b, nc, ln.s[l], ln.work[l]))) return rc;
                    if (now_first_pair >= 0 &&
                        (rc = frames_as_now_level(c, l, first_slot + b, now_first_pair + b, nc, ln.s[l], ln.work[l]))) return rc;
                }
            }
            if (ln.parallel) break;
            if (pass == 0 && (rc = upload_consumed(c, ub))) return rc;
        }
        if (ln.parallel) {
            if ((rc = lanes_join(c, n_levels, ln))) return rc;
            if ((rc = upload_consumed(c, ub))) return rc;
        }
    }
    for (int f = 0; f < count; f++) { c->fs.valid[first_slot + f] = 1; c->fs.has_depth[first_slot + f] = depth ? 1 : 0; }
    if (!(flags & DVO_UPLOAD_ASYNC)) HIPCHK(c, stream_wait(c->stream));
    return DVO_OK;
}

/* cv::undistort's map for this camera (OpenCV 2.4 undistort.cpp: undistort() + initUndistortRectifyMap(), CV_16SC2 maps).
 * Built on the host in double precision exactly as OpenCV's CPU code does -- per stripe of min(max(1, 4096/cols), rows) rows
 * the new camera matrix is the camera matrix with cy moved by the stripe's first row, inverted by the 3x3 adjugate formula
 * cv::invert uses; the normalised coordinates advance by running sums along a row -- and uploaded once. */
int dvo_frames_set_undistort(dvo_ctx *c, int rows, int cols, const double *K4, const double *D5) {
    DVO_ENTER(c);
    HIPCHK(c, stream_wait(c->stream));
    if (c->d_umap_xy) { (void)hipFree(c->d_umap_xy); (void)hipFree(c->d_umap_frac); c->d_umap_xy = nullptr; c->d_umap_frac = nullptr; }
    c->umap_rows = c->umap_cols = 0;
    if (!K4 && !D5) return DVO_OK;                         /* switched off: frames are taken as already undistorted */
    if (!K4 || !D5 || rows < 1 || cols < 1 || !(K4[0] != 0.0) || !(K4[1] != 0.0)) return fail(c, DVO_ERR_INVALID, "bad calibration");
    const size_t npx = (size_t)rows * cols;
    std::vector<short> xy(2 * npx);
    std::vector<unsigned short> fr(npx);
    const double fx = K4[0], fy = K4[1], cx = K4[2], cy = K4[3];
    const double k1 = D5[0], k2 = D5[1], p1 = D5[2], p2 = D5[3], k3 = D5[4];
    int stripe0 = std::min(std::max(1, (1 << 12) / std::max(cols, 1)), rows);
    auto cvround = [](double v) { return (int)std::lrint(v); };          /* round half to even, like cvRound */
    for (int y0 = 0; y0 < rows; y0 += stripe0) {
        const int n = std::min(stripe0, rows - y0);
        /* inverse of [[fx 0 cx][0 fy cy'][0 0 1]], cy' = cy - y0, by cofactors: every entry is (minor) * (1/det) */
        const double m[3][3] = {{fx, 0, cx}, {0, fy, cy - y0}, {0, 0, 1}};
        const double det = m[0][0] * (m[1][1] * m[2][2] - m[1][2] * m[2][1]) - m[0][1] * (m[1][0] * m[2][2] - m[1][2] * m[2][0]) +
                           m[0][2] * (m[1][0] * m[2][1] - m[1][1] * m[2][0]);
        const double d = 1. / det;
        double ir[9];
        ir[0] = (m[1][1] * m[2][2] - m[1][2] * m[2][1]) * d; ir[1] = (m[0][2] * m[2][1] - m[0][1] * m[2][2]) * d; ir[2] = (m[0][1] * m[1][2] - m[0][2] * m[1][1]) * d;
        ir[3] = (m[1][2] * m[2][0] - m[1][0] * m[2][2]) * d; ir[4] = (m[0][0] * m[2][2] - m[0][2] * m[2][0]) * d; ir[5] = (m[0][2] * m[1][0] - m[0][0] * m[1][2]) * d;
        ir[6] = (m[1][0] * m[2][1] - m[1][1] * m[2][0]) * d; ir[7] = (m[0][1] * m[2][0] - m[0][0] * m[2][1]) * d; ir[8] = (m[0][0] * m[1][1] - m[0][1] * m[1][0]) * d;
        for (int i = 0; i < n; i++) {
            double X = i * ir[1] + ir[2], Y = i * ir[4] + ir[5], W = i * ir[7] + ir[8];
            short *pxy = xy.data() + 2 * ((size_t)(y0 + i) * cols);
            unsigned short *pf = fr.data() + (size_t)(y0 + i) * cols;
            for (int j = 0; j < cols; j++, X += ir[0], Y += ir[3], W += ir[6]) {
                const double w = 1. / W, x = X * w, y = Y * w;
                const double x2 = x * x, y2 = y * y, r2 = x2 + y2, two_xy = 2 * x * y;
                const double kr = (1 + ((k3 * r2 + k2) * r2 + k1) * r2) / (1 + ((0 * r2 + 0) * r2 + 0) * r2);
                const double u = fx * (x * kr + p1 * two_xy + p2 * (r2 + 2 * x2)) + cx;
                const double v = fy * (y * kr + p1 * (r2 + 2 * y2) + p2 * two_xy) + cy;
                const int iu = cvround(u * 32), iv = cvround(v * 32);
                pxy[2 * j] = (short)(iu >> 5); pxy[2 * j + 1] = (short)(iv >> 5);
                pf[j] = (unsigned short)((iv & 31) * 32 + (iu & 31));
            }
        }
    }
    HIPCHK(c, hipMalloc((void **)&c->d_umap_xy, sizeof(short) * 2 * npx));
    HIPCHK(c, hipMalloc((void **)&c->d_umap_frac, sizeof(unsigned short) * npx));
    HIPCHK(c, hipMemcpy(c->d_umap_xy, xy.data(), sizeof(short) * 2 * npx, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->d_umap_frac, fr.data(), sizeof(unsigned short) * npx, hipMemcpyHostToDevice));
    c->umap_rows = rows; c->umap_cols = cols;
    return DVO_OK;
}

int dvo_frames_upload_cameras(dvo_ctx *c, int first_slot, int count, const unsigned char *const *bgr8,
                              const float *const *depth_m, int rows, int cols, int n_levels, int first_shift,
                              int now_first_pair, int flags) {
    DVO_ENTER(c);
    if (!bgr8 || count < 1 || rows < 1 || cols < 1 || n_levels < 1 || n_levels > DVO_LEVELS || first_shift < 0 ||
        first_shift + n_levels > 16)
        return fail(c, DVO_ERR_INVALID, "bad camera frame arguments");
    for (int f = 0; f < count; f++)
        if (!bgr8[f] || (depth_m && !depth_m[f])) return fail(c, DVO_ERR_INVALID, "NULL camera image");
    int lr[DVO_LEVELS], lc[DVO_LEVELS];
    for (int l = 0; l < n_levels; l++) {          /* cv::resize(Size(), s, s): dsize = cvRound(size * s) */
        const double sc = std::ldexp(1.0, -(first_shift + l));
        lr[l] = round_half_even_pos(rows * sc); lc[l] = round_half_even_pos(cols * sc);
        if (lr[l] < 1 || lc[l] < 1) return fail(c, DVO_ERR_INVALID, "pyramid level would be empty");
    }
    int rc = frames_geometry(c, n_levels, lr, lc);
    if (rc) return rc;
    if (!slots_ok(c, first_slot, count)) return fail(c, DVO_ERR_INVALID, "frame slot range out of bounds (dvo_frames_reserve)");
    if (now_first_pair >= 0 && (!pair_ok(c, now_first_pair) || now_first_pair + count > c->n_pairs))
        return fail(c, DVO_ERR_INVALID, "now_first_pair range out of bounds");
    if (c->d_umap_xy && (c->umap_rows != rows || c->umap_cols != cols))
        return fail(c, DVO_ERR_INVALID, "the undistortion map was built for another image size (dvo_frames_set_undistort)");
    const size_t npx = (size_t)rows * cols;
    const size_t b_img = (npx * 3 + 15) / 16 * 16, d_img = depth_m ? npx * 4 : 0;
    const bool dev_src = (flags & DVO_UPLOAD_DEVICE) != 0;     /* no PCIe to overlap with: whole batches per stage */
    const bool pulled = dev_src || (flags & DVO_UPLOAD_MAPPED);  /* device-addressable sources: gathered by a kernel, no pinned mirror */
    /* Frames already in HBM are read where they are (round 6): their addresses go up as a table and the level kernels index it -- no
     * landing copy (per 256 VGA frames 236 MB each way, 71 us, and a stream hand-over: 1.60 -> 1.5 ms per `frames in HBM -> poses` step).
     * Needs the alignment of the landing buffer (4 bytes BGR, 16 depth); DVO_DEVICE_DIRECT=off keeps the copy (A/B). */
    static const bool direct_off = [] { const char *e = getenv("DVO_DEVICE_DIRECT"); return e && !std::strcmp(e, "off"); }();
    bool direct = dev_src && !direct_off;
    for (int f = 0; f < count && direct; f++)
        direct = (reinterpret_cast<size_t>(bgr8[f]) & 3) == 0 && (!depth_m || (reinterpret_cast<size_t>(depth_m[f]) & 15) == 0);
    const size_t half = dev_src ? kDeviceHalf : ((flags & DVO_UPLOAD_MAPPED) ? kMappedHalf : kUploadHalf);
    int chunk = direct ? count : (int)std::min<size_t>(std::max<size_t>(half / (b_img + d_img), 1), (size_t)count);
    if (!direct && pulled && chunk > 32) chunk -= chunk % 32;   /* whole gather launches of 32 images: a short tail launch runs far below the link rate */
    if (!direct && (rc = ensure_upload(c, (b_img + d_img) * chunk, !pulled))) return rc;
    SrcTab tab0 = {nullptr, nullptr};
    if (direct) {
        const int need = 2 * count;
        if (need > c->src_tab_cap) {
            HIPCHK(c, stream_wait(c->stream));
            if (c->src_tab_dev) HIPCHK(c, hipFree(c->src_tab_dev));
            if (c->src_tab_host) HIPCHK(c, hipHostFree(c->src_tab_host));
            c->src_tab_dev = nullptr; c->src_tab_host = nullptr; c->src_tab_cap = 0;
            HIPCHK(c, hipMalloc((void **)&c->src_tab_dev, sizeof(void *) * (size_t)need));
            HIPCHK(c, hipHostMalloc((void **)&c->src_tab_host, sizeof(void *) * (size_t)need, hipHostMallocDefault));
            c->src_tab_cap = need;
        }
        if (!c->ev_src_tab) HIPCHK(c, hipEventCreateWithFlags(&c->ev_src_tab, hipEventDisableTiming));
        else HIPCHK(c, hipEventSynchronize(c->ev_src_tab));      /* the staging table's previous copy has gone up; the DEVICE table's readers are ahead of this call's copy on the stream */
        for (int f = 0; f < count; f++) {
            c->src_tab_host[f] = const_cast<unsigned char *>(bgr8[f]);
            c->src_tab_host[count + f] = depth_m ? const_cast<float *>(depth_m[f]) : nullptr;
        }
        HIPCHK(c, hipMemcpyAsync(c->src_tab_dev, c->src_tab_host, sizeof(void *) * (size_t)need, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipEventRecord(c->ev_src_tab, c->stream));
        tab0.bgr = c->src_tab_dev;
        tab0.depth = depth_m ? c->src_tab_dev + count : nullptr;
    }
    /* chunk k+1 is copied (copy streams) while chunk k is preprocessed (context stream), over two landing buffers.  The copies
     * of chunk k+1 are SUBMITTED before the kernels of chunk k: measured on this pool (tools/experiments/exp_pull_overlap.sh),
     * work of two streams that becomes ready at the same moment starts in submission order, and a pull submitted after ~45
     * preprocessing launches waited for nearly all of them -- the link idled 0.6 ms of every 1.6 */
    struct Chunk { int b, nc, ub; unsigned char *sb; float *sd; };
    int next_ub = c->up_next;
    auto issue_copy = [&](int b, Chunk &k) -> int {
        if (direct) { k.b = b; k.nc = std::min(chunk, count - b); k.ub = -1; k.sb = nullptr; k.sd = nullptr; return DVO_OK; }
        k.b = b; k.nc = std::min(chunk, count - b); k.ub = next_ub; next_ub ^= 1;
        k.sb = c->up_buf[k.ub]; k.sd = (float *)(k.sb + b_img * chunk);
        if (c->up_used[k.ub]) {                             /* the buffer's previous consumer (two chunks back) has read it */
            HIPCHK(c, hipStreamWaitEvent(c->copy_stream, c->ev_done[k.ub], 0));
            HIPCHK(c, hipStreamWaitEvent(c->copy_stream2, c->ev_done[k.ub], 0));
        }
        if (pulled) {                                        /* in HBM already, or in pinned host memory the GPU addresses: gathered */
            /* mapped host memory: ~32 workgroups per launch of up to 32 images keep the link busy (128 KB in flight) without taking
             * the wave slots the previous chunk's preprocessing needs; a single camera frame gets all 32 */
            static const int pull_wgs = [] { const char *e = getenv("DVO_PULL_WGS"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 32; }();
            const int wgs = dev_src ? 64 : std::max(1, pull_wgs / std::min(k.nc, 32));
            HIPCHK(c, launch_gather_images(reinterpret_cast<const void *const *>(bgr8 + b), k.nc, k.sb, npx * 3, b_img, c->copy_stream, wgs));
            if (depth_m) HIPCHK(c, launch_gather_images(reinterpret_cast<const void *const *>(depth_m + b), k.nc, k.sd, npx * 4, npx * 4, c->copy_stream2, wgs));
        } else if (flags & DVO_UPLOAD_DIRECT) {
            for (int i = 0; i < k.nc; i++) {
                hipStream_t cs = (i & 1) ? c->copy_stream2 : c->copy_stream;
                HIPCHK(c, hipMemcpyAsync(k.sb + b_img * i, bgr8[b + i], npx * 3, hipMemcpyHostToDevice, cs));
                if (depth_m) HIPCHK(c, hipMemcpyAsync(k.sd + npx * i, depth_m[b + i], npx * 4, hipMemcpyHostToDevice, cs));
            }
        } else {
            /* through the engine's pinned mirror of the landing buffer: one memcpy per image on the host, then two DMAs per
             * chunk; the caller's (pageable) memory is never registered with the driver (include/dvo_amd.h, DVO_UPLOAD_DIRECT) */
            unsigned char *hb = c->up_host[k.ub];
            float *hd = (float *)(hb + b_img * chunk);
            if (c->up_used[k.ub]) { HIPCHK(c, hipEventSynchronize(c->ev_copied[k.ub])); HIPCHK(c, hipEventSynchronize(c->ev_copied2[k.ub])); }
            for (int i = 0; i < k.nc; i++) {
                std::memcpy(hb + b_img * i, bgr8[b + i], npx * 3);
                if (depth_m) std::memcpy(hd + npx * i, depth_m[b + i], npx * 4);
            }
            HIPCHK(c, hipMemcpyAsync(k.sb, hb, b_img * (size_t)k.nc, hipMemcpyHostToDevice, c->copy_stream));
            if (depth_m) HIPCHK(c, hipMemcpyAsync(k.sd, hd, npx * 4 * (size_t)k.nc, hipMemcpyHostToDevice, c->copy_stream2));
        }
        HIPCHK(c, hipEventRecord(c->ev_copied[k.ub], c->copy_stream));
        HIPCHK(c, hipEventRecord(c->ev_copied2[k.ub], c->copy_stream2));
        c->up_used[k.ub] = true;                            /* from here on the buffer has a pending ev_copied and, soon, ev_done */
        c->up_next = k.ub ^ 1;                              /* committed per chunk: an error further down leaves events and turn consistent */
        return DVO_OK;
    };
    auto issue_compute = [&](const Chunk &k) -> int {
        int rc2;
        LevelLanes ln;
        if ((rc2 = lanes_begin(c, n_levels, k.nc, now_first_pair >= 0, ln))) return rc2;
        const SrcTab tab = direct ? SrcTab{tab0.bgr + k.b, tab0.depth ? tab0.depth + k.b : nullptr} : SrcTab{nullptr, nullptr};
        const float *const dsrc = depth_m ? (direct ? reinterpret_cast<const float *>(16) /* "has depth"; the table holds the addresses */ : k.sd) : nullptr;
        if (k.ub >= 0) {
            HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_copied[k.ub], 0));
            HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_copied2[k.ub], 0));
        }
        if ((rc2 = lanes_fork(c, n_levels, ln))) return rc2;
        for (int pass = 0; pass < 2; pass++) {
            for (int l = 0; l < n_levels; l++) {
                FrameLevel &F = c->fs.lv[l];
                const size_t off = (size_t)(first_slot + k.b) * F.npx;
                if (ln.parallel || (pass == 0 && (l == 0 || n_levels == 2)))
                    HIPCHK(c, launch_camera_level(k.sb, b_img, dsrc, npx, rows, cols, first_shift + l,
                                                  c->d_umap_xy, c->d_umap_frac, (flags & DVO_UPLOAD_DEPTH_RAW) ? 1 : 0,
                                                  F.grey + off, F.depth + off, F.npx, ImgBatch{F.rows, F.cols, k.nc}, ln.s[l], tab));
                else if (pass == 0 && l == 1) {                  /* levels 1 .. n-1 in one launch */
                    int sh[DVO_LEVELS], lr2[DVO_LEVELS], lc2[DVO_LEVELS]; unsigned char *gl[DVO_LEVELS]; float *dl[DVO_LEVELS]; size_t st[DVO_LEVELS];
                    for (int m = 1; m < n_levels; m++) {
                        FrameLevel &G = c->fs.lv[m];
                        sh[m - 1] = first_shift + m; lr2[m - 1] = G.rows; lc2[m - 1] = G.cols; st[m - 1] = G.npx;
                        gl[m - 1] = G.grey + (size_t)(first_slot + k.b) * G.npx; dl[m - 1] = G.depth + (size_t)(first_slot + k.b) * G.npx;
                    }
                    static const bool decimate_off = [] { const char *e = getenv("DVO_PYRAMID_DECIMATE"); return e && !std::strcmp(e, "off"); }();
                    FrameLevel &F0 = c->fs.lv[0];
                    if (!decimate_off && camera_levels_decimate_ok(n_levels, lr, lc))      /* from level 0, written just before on this stream */
                        HIPCHK(c, launch_camera_decimate_levels(F0.grey + (size_t)(first_slot + k.b) * F0.npx, depth_m ? F0.depth + (size_t)(first_slot + k.b) * F0.npx : nullptr,
                                                                F0.npx, F0.rows, F0.cols, n_levels - 1, lr2, lc2, gl, dl, st, k.nc, c->stream));
                    else
                        HIPCHK(c, launch_camera_levels(k.sb, b_img, dsrc, npx, rows, cols, n_levels - 1, sh, lr2, lc2, c->d_umap_xy,
                                                       c->d_umap_frac, (flags & DVO_UPLOAD_DEPTH_RAW) ? 1 : 0, gl, dl, st, k.nc, c->stream, tab));
                }
                if (pass == 1 && !ln.parallel && l == 0) {         /* one launch per stage for all levels */
                    if ((rc2 = run_canny_all(c, n_levels, first_slot + k.b, k.nc, c->stream))) return rc2;
                    if (now_first_pair >= 0 && (rc2 = frames_as_now_all(c, n_levels, first_slot + k.b, now_first_pair + k.b, k.nc, c->stream))) return rc2;
                }
                if (ln.parallel) {
                    if ((rc2 = run_canny(c, l, first_slot + k.b, k.nc, ln.s[l], ln.work[l]))) return rc2;
                    if (now_first_pair >= 0 &&
                        (rc2 = frames_as_now_level(c, l, first_slot + k.b, now_first_pair + k.b, k.nc, ln.s[l], ln.work[l]))) return rc2;
                }
            }
            if (ln.parallel) break;
            if (pass == 0 && k.ub >= 0) HIPCHK(c, hipEventRecord(c->ev_done[k.ub], c->stream));     /* the landing buffer is free again */
        }
        if (ln.parallel) {
            if ((rc2 = lanes_join(c, n_levels, ln))) return rc2;
            if (k.ub >= 0) HIPCHK(c, hipEventRecord(c->ev_done[k.ub], c->stream));
        }
        return DVO_OK;
    };
    {
        Chunk cur, nxt;
        if ((rc = issue_copy(0, cur))) return rc;
        for (int b = 0; b < count; b += chunk) {
            const bool more = b + chunk < count;
            /* pull kernels (few workgroups, latency-bound on the link) go in ahead of the chunk's preprocessing and run beside
             * it; DMA / blit copies submitted ahead would hold the preprocessing back instead (measured: 14.6 -> 22 ms per 256
             * frames with depth), so they keep their place behind it */
            static const int ahead_env = [] { const char *e = getenv("DVO_COPY_AHEAD"); return e ? atoi(e) : -1; }();
            const bool ahead = ahead_env >= 0 ? ahead_env != 0 : pulled;
            if (more && ahead && (rc = issue_copy(b + chunk, nxt))) return rc;
            if ((rc = issue_compute(cur))) return rc;
            if (more && !ahead && (rc = issue_copy(b + chunk, nxt))) return rc;
            cur = nxt;
        }
        if (!direct) c->up_next = next_ub;
    }
    for (int f = 0; f < count; f++) { c->fs.valid[first_slot + f] = 1; c->fs.has_depth[first_slot + f] = depth_m ? 1 : 0; }
    if (!(flags & DVO_UPLOAD_ASYNC)) HIPCHK(c, stream_wait(c->stream));
    return DVO_OK;
}

static int frames_check_use(dvo_ctx *c, int first_slot, int first_pair, int count, bool need_depth) {
    if (c->fs.n_levels < 1) return fail(c, DVO_ERR_STATE, "frame store is empty (dvo_frames_upload_*)");
    if (!slots_ok(c, first_slot, count)) return fail(c, DVO_ERR_INVALID, "frame slot range out of bounds");
    if (!pair_ok(c, first_pair) || first_pair + count > c->n_pairs) return fail(c, DVO_ERR_INVALID, "pair range out of bounds");
    for (int f = first_slot; f < first_slot + count; f++) {
        if (!c->fs.valid[f]) return fail(c, DVO_ERR_STATE, "frame slot " + std::to_string(f) + " holds no frame");
        if (need_depth && !c->fs.has_depth[f]) return fail(c, DVO_ERR_STATE, "frame slot " + std::to_string(f) + " has no depth");
    }
    return DVO_OK;
}

static int frames_as_now_level(dvo_ctx *c, int l, int first_slot, int first_pair, int count, hipStream_t stream, int *work) {
    int rc;
    FrameLevel &F = c->fs.lv[l];
    if ((rc = ensure_texels(c, l, F.rows, F.cols))) return rc;
    /* the now level is written in its compact form (dvo_palette.h) straight from the integer squared distances; 16-byte texels
     * only for images that form cannot hold, or when the caller switched the compact form off */
    const bool compact = native_compact_wanted(c);
    if (compact && (rc = ensure_compact_slabs(c, l))) return rc;
    Level &L = c->lv[l];
    unsigned *p4 = compact ? L.p4 : nullptr;
    /* sparse texel slab (dvo_ctx.h): the stage runs without texel output; the images the compact form cannot hold get their
     * texels mapped and written by a second run of the last pass -- per chunk, while the chunk's scratch still holds them */
    const bool defer = compact && L.tex_sparse;
    if (L.tex_sparse && !compact && (rc = map_texels(c, l, first_pair, count, stream))) return rc;
    auto run = [&](const unsigned char *edge, int nc, int *wk, int pair0) -> int {
        HIPCHK(c, launch_edges_to_now(edge, F.npx, ImgBatch{F.rows, F.cols, nc}, wk, defer ? nullptr : L.tex + (size_t)pair0 * L.tex_stride,
                                      L.tex_stride, p4, L.p4_stride, L.pal, L.d_pal_n, pair0, stream));
        if (defer) {
            int n_failed = 0, rc2;
            if ((rc2 = sparse_map_compact_failures(c, l, pair0, nc, stream, &n_failed))) return rc2;
            if (n_failed)
                HIPCHK(c, launch_edges_to_now(edge, F.npx, ImgBatch{F.rows, F.cols, nc}, wk, L.tex + (size_t)pair0 * L.tex_stride, L.tex_stride,
                                              p4, L.p4_stride, L.pal, L.d_pal_n, pair0, stream, true));
        }
        return DVO_OK;
    };
    if (work) {
        if ((rc = run(F.edge + (size_t)first_slot * F.npx, count, work, first_pair))) return rc;
    } else {
        const int chunk = chunk_for(sizeof(int) * edt_work_ints(F.rows, F.cols, 1), count);
        if ((rc = ensure_work(c, sizeof(int) * edt_work_ints(F.rows, F.cols, chunk)))) return rc;
        for (int b = 0; b < count; b += chunk) {
            const int nc = std::min(chunk, count - b);
            if ((rc = run(F.edge + (size_t)(first_slot + b) * F.npx, nc, c->work, first_pair + b))) return rc;
        }
    }
    return compact ? now_written_compact(c, l, first_pair, count) : now_written(c, l, first_pair, count);
}

/* every level of `count` stored frames as now levels of pairs first_pair..: one launch per stage for all levels when the
 * geometry allows it (dvo_frames.hip, launch_edges_to_now_levels), else level by level */
static int frames_as_now_all(dvo_ctx *c, int n_levels, int first_slot, int first_pair, int count, hipStream_t stream) {
    int rc;
    int rows[DVO_LEVELS], cols[DVO_LEVELS];
    for (int l = 0; l < n_levels; l++) { rows[l] = c->fs.lv[l].rows; cols[l] = c->fs.lv[l].cols; }
    static const bool per_level = getenv("DVO_EDT_PER_LEVEL") != nullptr;
    if (per_level || !edt_levels_ok(n_levels, rows, cols)) {
        for (int l = 0; l < n_levels; l++)
            if ((rc = frames_as_now_level(c, l, first_slot, first_pair, count, stream, nullptr))) return rc;
        return DVO_OK;
    }
    const bool compact = native_compact_wanted(c);
    for (int l = 0; l < n_levels; l++) {
        if ((rc = ensure_texels(c, l, rows[l], cols[l]))) return rc;
        if (compact && (rc = ensure_compact_slabs(c, l))) return rc;
    }
    /* Tried and dropped (round 4): the batch as two halves on two streams with their own scratch, so that the issue-bound row scan
     * of one half runs beside the memory-bound rank pack of the other -- 1.02-1.05 ms per 256 frames against 0.86 ms on one stream
     * (half-size launches fill the GPU less well, and work of two streams starts in submission order on this pool). */
    const int n_lanes = 1;
    const int per_lane = count;
    const int chunk = chunk_for(sizeof(int) * edt_levels_work_ints(n_levels, rows, cols, 1), count);
    const size_t lane_ints = edt_levels_work_ints(n_levels, rows, cols, chunk);
    if ((rc = ensure_work(c, sizeof(int) * lane_ints))) return rc;
    for (int lane = 0; lane < n_lanes; lane++) {
    hipStream_t ls = stream;
    int *lwork = c->work + lane * lane_ints;
    const int lane_first = lane * per_lane, lane_end = std::min(count, lane_first + per_lane);
    for (int b = lane_first; b < lane_end; b += chunk) {
        const int nc = std::min(chunk, lane_end - b);
        const unsigned char *edge[DVO_LEVELS]; size_t estride[DVO_LEVELS], tstride[DVO_LEVELS], pstride[DVO_LEVELS];
        float4 *tex[DVO_LEVELS]; unsigned *p4[DVO_LEVELS]; float2 *pal[DVO_LEVELS]; int *pal_n[DVO_LEVELS];
        bool defer = false;
        for (int l = 0; l < n_levels; l++) {
            const FrameLevel &F = c->fs.lv[l];
            Level &L = c->lv[l];
            edge[l] = F.edge + (size_t)(first_slot + b) * F.npx; estride[l] = F.npx;
            if (L.tex_sparse && !compact && (rc = map_texels(c, l, first_pair + b, nc, ls))) return rc;
            const bool dl = compact && L.tex_sparse;                 /* sparse texel slab: no texel output in the first run */
            defer = defer || dl;
            tex[l] = dl ? nullptr : L.tex + (size_t)(first_pair + b) * L.tex_stride; tstride[l] = L.tex_stride;
            p4[l] = compact ? L.p4 : nullptr; pstride[l] = L.p4_stride; pal[l] = L.pal; pal_n[l] = L.d_pal_n;
        }
        HIPCHK(c, launch_edges_to_now_levels(n_levels, rows, cols, edge, estride, nc, lwork, tex, tstride, p4, pstride, pal, pal_n,
                                             first_pair + b, ls));
        if (defer) {
            int failed_any = 0;
            for (int l = 0; l < n_levels; l++) {
                Level &L = c->lv[l];
                if (!L.tex_sparse) continue;
                int n_failed = 0;
                if ((rc = sparse_map_compact_failures(c, l, first_pair + b, nc, ls, &n_failed))) return rc;
                failed_any += n_failed;
                tex[l] = L.tex + (size_t)(first_pair + b) * L.tex_stride;      /* only the pairs just mapped are written (pal_n < 0) */
            }
            if (failed_any)
                HIPCHK(c, launch_edges_to_now_levels(n_levels, rows, cols, edge, estride, nc, lwork, tex, tstride, p4, pstride, pal, pal_n,
                                                     first_pair + b, ls, true));
        }
    }
    }
    for (int l = 0; l < n_levels; l++)
        if ((rc = compact ? now_written_compact(c, l, first_pair, count) : now_written(c, l, first_pair, count))) return rc;
    return DVO_OK;
}

int dvo_frames_as_now(dvo_ctx *c, int first_slot, int first_pair, int count) {
    DVO_ENTER(c);
    int rc = frames_check_use(c, first_slot, first_pair, count, false);
    if (rc) return rc;
    const int nl = c->fs.n_levels;
    LevelLanes ln;
    if ((rc = lanes_begin(c, nl, count, true, ln))) return rc;
    if (!ln.parallel) return frames_as_now_all(c, nl, first_slot, first_pair, count, c->stream);
    if ((rc = lanes_fork(c, nl, ln))) return rc;
    for (int l = 0; l < nl; l++)
        if ((rc = frames_as_now_level(c, l, first_slot, first_pair, count, ln.s[l], ln.work[l]))) return rc;
    return lanes_join(c, nl, ln);
}

int dvo_frames_as_ref(dvo_ctx *c, int first_slot, int first_pair, int count, int *N_out) {
    DVO_ENTER(c);
    if (!c->have_K) return fail(c, DVO_ERR_STATE, "intrinsics not set (dvo_set_intrinsics)");
    int rc = frames_check_use(c, first_slot, first_pair, count, true);
    if (rc) return rc;
    const int nl = c->fs.n_levels;
    size_t cc_off[DVO_LEVELS + 1];                      /* per level: count x (cols+2) column counters ... */
    size_t bc_off[DVO_LEVELS + 1];                      /* ... and count x enlist_block_ints() block-order counters */
    cc_off[0] = 0;
    for (int l = 0; l < nl; l++) cc_off[l + 1] = cc_off[l] + (size_t)count * (c->fs.lv[l].cols + 2);
    bc_off[0] = cc_off[nl];
    for (int l = 0; l < nl; l++) bc_off[l + 1] = bc_off[l] + (size_t)count * enlist_block_ints(c->fs.lv[l].rows, c->fs.lv[l].cols);
    if ((rc = ensure_work(c, sizeof(int) * bc_off[nl]))) return rc;
    std::vector<int> hN((size_t)count * nl);
    for (int l = 0; l < nl; l++) {
        FrameLevel &F = c->fs.lv[l];
        const size_t off = (size_t)first_slot * F.npx;
        int *cc = c->work + cc_off[l];
        HIPCHK(c, launch_enlist_count(F.edge + off, 1, F.npx, F.depth + off, F.npx, ImgBatch{F.rows, F.cols, count}, cc,
                                      compact_block_order() ? c->work + bc_off[l] : nullptr, c->stream));
        HIPCHK(c, hipMemcpy2DAsync(hN.data() + (size_t)l * count, sizeof(int), cc + F.cols, sizeof(int) * (F.cols + 2),
                                   sizeof(int), count, hipMemcpyDeviceToHost, c->stream));
    }
    HIPCHK(c, stream_wait(c->stream));
    int bad_level = -1, bad_frame = -1;
    for (int l = 0; l < nl; l++) {
        FrameLevel &F = c->fs.lv[l];
        int maxN = 0;
        for (int i = 0; i < count; i++) {
            const int N = hN[(size_t)l * count + i];
            maxN = std::max(maxN, N);
            if (N < 1 && bad_level < 0) { bad_level = l; bad_frame = i; }
            if (N_out) N_out[(size_t)i * nl + l] = N;
        }
        if ((rc = ensure_points(c, l, std::max(maxN, 1)))) return rc;
        Level &L = c->lv[l];
        const size_t off = (size_t)first_slot * F.npx;
        HIPCHK(c, launch_enlist_write(F.edge + off, 1, F.npx, F.depth + off, F.npx, ImgBatch{F.rows, F.cols, count}, l, c->K,
                                      c->work + cc_off[l], compact_block_order() ? c->work + bc_off[l] : nullptr, L.pts + (size_t)first_pair * L.pt_cap * 3, (size_t)L.pt_cap * 3,
                                      L.cpts + (size_t)first_pair * L.pt_cap, L.cidx + (size_t)first_pair * L.pt_cap, nullptr, L.pt_cap,
                                      L.dN + first_pair, c->stream));
        HIPCHK(c, launch_points4_build(L.cpts, L.dN, L.pt_cap, F.rows, L.cpt4, L.chdr, L.d_pt4_ok, first_pair, count, c->stream));
        for (int i = 0; i < count; i++) { L.hN[first_pair + i] = hN[(size_t)l * count + i]; L.compact_ok[first_pair + i] = 1; }
        ref_list_written(c, l, first_pair, count, F.rows);
    }
    if (bad_level >= 0)
        return fail(c, DVO_ERR_INVALID, "no reference point selected in frame " + std::to_string(first_slot + bad_frame) +
                                            " level " + std::to_string(bad_level) +
                                            " (reference asserts nSelectedPts > 0, SolveDVO.cpp:282)");
    return DVO_OK;
}

int dvo_frame_get_level(dvo_ctx *c, int slot, int level, int *rows, int *cols, unsigned char *grey,
                        float *depth_mm, unsigned char *edge, int *n_edges) {
    DVO_ENTER(c);
    if (level < 0 || level >= c->fs.n_levels) return fail(c, DVO_ERR_INVALID, "frame level out of range");
    if (!slots_ok(c, slot, 1) || !c->fs.valid[slot]) return fail(c, DVO_ERR_STATE, "frame slot holds no frame");
    FrameLevel &F = c->fs.lv[level];
    if (rows) *rows = F.rows;
    if (cols) *cols = F.cols;
    const size_t off = (size_t)slot * F.npx;
    if (grey) HIPCHK(c, hipMemcpyAsync(grey, F.grey + off, F.npx, hipMemcpyDeviceToHost, c->stream));
    if (edge) HIPCHK(c, hipMemcpyAsync(edge, F.edge + off, F.npx, hipMemcpyDeviceToHost, c->stream));
    if (depth_mm) {
        if (!c->fs.has_depth[slot]) return fail(c, DVO_ERR_STATE, "frame slot has no depth");
        HIPCHK(c, hipMemcpyAsync(depth_mm, F.depth + off, sizeof(float) * F.npx, hipMemcpyDeviceToHost, c->stream));
    }
    if (n_edges) {
        int rc = ensure_work(c, sizeof(int));
        if (rc) return rc;
        HIPCHK(c, launch_count_edges(F.edge + off, F.npx, c->work, c->stream));
        HIPCHK(c, hipMemcpyAsync(n_edges, c->work, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    }
    /* round 6: a question about the level's size alone touches no device memory and must not wait for the stream (the Python binding asks
     * after every upload: four host syncs per step of a frames -> poses pipeline, 45 us of idle GPU in front of every alignment) */
    if (grey || edge || depth_mm || n_edges) HIPCHK(c, stream_wait(c->stream));
    return DVO_OK;
}

}  // extern "C"
