/* dvo_tiled_step.h -- the launch of one iteration of the tiled / wide schedule, but for its per-point phase (see the comment above
 * tiled_step_kernel in dvo_kernels.hip: head = pending update, body = this workgroup's share of the points, tail = row + ticket +
 * last arriver).  Two kernels share it: tiled_step_kernel (dvo_kernels.hip: one point per lane, the reference's 3 x N list, H on
 * request) and tiled_step_pk_kernel (dvo_fused.hip, round 5: the packed two-points-per-lane loop over the compact list). */
#ifndef DVO_TILED_STEP_H
#define DVO_TILED_STEP_H
#include "dvo_kernel_common.h"

namespace dvo {

DVO_DEV void store_sc1_f64(double *p, double v) { asm volatile("global_store_dwordx2 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory"); }
/* a relaxed agent-scope atomic load IS global_load_dwordx2 sc1 (MI355X_MICROARCH.md, the HIP construct table), with the wait
 * counters left to the compiler: the sixteen loads of a lane below are all in flight before the first is consumed (an inline-asm
 * load would have to wait for itself: 16 dependent memory latencies in the last workgroup of every launch) */
DVO_DEV double load_sc1_f64(const double *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
/* Tagged rows (round 5, the launches without H): a workgroup's eight sums travel as eight 16-byte records {value lo, tag, value hi, tag},
 * one store each -- a reader that sees the tag in both halves has the value (the team exchange's record, dvo_fused.hip) -- and the
 * storing workgroup is DONE: no wait for the stores to drain, no barrier, no ticket.  Workgroup 0, which keeps the state and has no
 * points, polls the rows of all others and adds them in a fixed order.  The tag is the launch's sequence number + 1, kept in device
 * memory (`ticket[1]`, advanced by workgroup 0 at the end of every launch, read by every workgroup at its head): a replayed graph
 * passes the same arguments again, so the tag cannot be one.  Rounds 4-5a: sc1 stores, s_waitcnt, barrier, an atomic ticket, a second
 * barrier, and the LAST workgroup loading all rows -- three dependent memory round trips at the end of every launch. */
typedef unsigned step_v4u __attribute__((ext_vector_type(4)));
DVO_DEV void step_store_rec(step_v4u *p, double v, unsigned tag) {
    const unsigned long long bits = (unsigned long long)__double_as_longlong(v);
    step_v4u rec;
    rec.x = (unsigned)bits; rec.y = tag; rec.z = (unsigned)(bits >> 32); rec.w = tag;
    asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(rec) : "memory");
}
/* one poll round: the loads of a lane's four records and their wait in ONE statement (the compiler never sees a register in flight) */
DVO_DEV void step_poll4(step_v4u (&r)[4], const step_v4u *p0, const step_v4u *p1, const step_v4u *p2, const step_v4u *p3) {
    asm volatile("global_load_dwordx4 %0, %4, off sc1\n\tglobal_load_dwordx4 %1, %5, off sc1\n\t"
                 "global_load_dwordx4 %2, %6, off sc1\n\tglobal_load_dwordx4 %3, %7, off sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]) : "v"(p0), "v"(p1), "v"(p2), "v"(p3) : "memory");
}
DVO_DEV double step_rec_value(const step_v4u &r) { return __longlong_as_double((long long)(((unsigned long long)r.z << 32) | r.x)); }
#ifndef DVO_STEP_THREADS
#define DVO_STEP_THREADS 512
#endif
static_assert(sizeof(PoseState) % 8 == 0, "the state is copied 8 bytes per lane");
static_assert(DVO_STEP_THREADS == 512, "the last arriver of a step launch adds the rows in 16 chains of 32 lanes");

/* LDS of a step launch (declared by the kernel, handed to the body) */
struct TiledStepLds {
    double tot[DVO_NACC_PAD];
    double part[16][DVO_NACC_PAD + 1];
    PoseState s;
    PoseCur nxt;                                                            /* the iterate the pending update produces */
    double g_s[8];                                                          /* sums 21..28 of the previous launch */
    int s_last, s_stop0;
    unsigned seq;                                                           /* the launch's sequence number (ticket[1]) */
};

/* points(pc, run, b0, b1, tot): EVERY thread of the workgroup calls it; it leaves the workgroup's sums of points [b0, b1) at pose pc in
 * tot[0 .. DVO_NACC) (zeros when !run), behind a workgroup barrier */
template <bool WITH_H, typename Points>
DVO_DEV void tiled_step_body(TiledStepLds &m, const PoseState *st_in, PoseState *st_out, const double *__restrict__ acc_in, int itr, int apply_prev,
                             int n_total, int first, int n, double *partials, unsigned *ticket, double *acc_out, float *energy, double *H_prev,
                             Points points) {
    PoseState &s = m.s;
    PoseCur &nxt = m.nxt;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    /* head (round 5): everything the update needs is requested at once -- the state (8 bytes per lane), the eight sums, the stop
     * flag as the launch found it -- so the head pays ONE memory latency before its barrier (rounds 1-4: state, barrier, sums).
     * Then the packed kernel's split (dvo_fused.hip, serial part): wave 0 takes direction (one component per lane) and step into
     * `nxt`, wave 1 the energy and best-iterate bookkeeping of the same iterate; log(pose) for the regulariser was taken by
     * workgroup 0 of the previous launch (or by iter_begin_kernel / tiled_finish_kernel) while the others worked. */
    {
        const unsigned long long *src = reinterpret_cast<const unsigned long long *>(st_in);
        unsigned long long *dst = reinterpret_cast<unsigned long long *>(&s);
        for (int i = tid; i < (int)(sizeof(PoseState) / 8); i += DVO_STEP_THREADS) dst[i] = src[i];
        if (wave == 7) {
            if (lane < 8) m.g_s[lane] = apply_prev ? acc_in[21 + lane] : 0.0;
            if (lane == 8) m.s_stop0 = st_in->stop;
            if (lane == 9) m.seq = ticket[1];
        }
    }
    __syncthreads();
    const bool upd = apply_prev && !m.s_stop0;                              /* after :877 nothing runs */
    if (upd) {
        if (wave == 0) {
            double psi[6];
            pose_direction_lanes(s, s.u, pose_neg_step(s.u, itr - 1), m.g_s[lane < 6 ? lane : 5], lane, psi);
            if (lane == 0) pose_apply(s, s.p[0], nxt, s.u, psi);
        } else if (wave == 1 && lane == 0) {
            const float e = pose_bookkeep(s, s.p[0], itr - 1, n_total, m.g_s[6], (int)m.g_s[7]);
            if (blockIdx.x == 0) energy[itr - 1] = e;                       /* :690 */
        } else if (WITH_H && wave == 2) {
            /* DVO_FLAG_NORMAL_MATRIX: H = sum w J J^T of the previous iterate (reduced over all ranks), kept per iterate like the batch kernels do */
            if (H_prev && blockIdx.x == 0 && lane < 21) H_prev[lane] = acc_in[lane];
        }
    }
    __syncthreads();
    const bool moved = upd && !s.stop;                                      /* the points run at nxt, else at the state's iterate */
    const PoseCur &pc = moved ? nxt : s.p[0];
    /* workgroup 0 keeps the state and, with more than one workgroup, has no share of the points (below) */
    const int nshare = (gridDim.x > 1) ? (int)gridDim.x - 1 : 1;
    const int share = (gridDim.x > 1) ? (int)blockIdx.x - 1 : 0;
    const int per = (n + nshare - 1) / nshare;
    const int b0 = first + share * per;
    int b1 = b0 + per;
    if (b1 > first + n) b1 = first + n;
    points(pc, !s.stop && share >= 0 && b0 < b1 /* wave-uniform (LDS) */, b0, b1, m.tot);
    /* workgroup 0 keeps the state: the new iterate, log(pose) for the next update's regulariser, st_out (double-buffered: no
     * workgroup of this launch reads what it writes).  With more than one workgroup it has no share of the points, so none of this
     * is on the launch's critical path. */
    if (blockIdx.x == 0) {
        if (moved && tid < (int)(sizeof(PoseCur) / 8))
            reinterpret_cast<unsigned long long *>(&s.p[0])[tid] = reinterpret_cast<const unsigned long long *>(&nxt)[tid];
        __syncthreads();
        if (tid == 0 && !s.stop) pose_regulariser_precompute(s, s.p[0], s.u);
        __syncthreads();
        if (apply_prev) {
            const unsigned long long *src = reinterpret_cast<const unsigned long long *>(&s);
            unsigned long long *dst = reinterpret_cast<unsigned long long *>(st_out);
            for (int i = tid; i < (int)(sizeof(PoseState) / 8); i += DVO_STEP_THREADS) dst[i] = src[i];
        }
    }
    double *tot = m.tot;
    double (*part)[DVO_NACC_PAD + 1] = m.part;
    if constexpr (!WITH_H) {
        /* tail without H: tagged rows to workgroup 0 (see step_store_rec) */
        const unsigned tag = m.seq + 1u;
        step_v4u *recs = reinterpret_cast<step_v4u *>(partials);             /* [workgroup][8] records */
        if (gridDim.x == 1) {                                               /* the only workgroup: its sums are the launch's */
            if (tid < DVO_NACC_PAD) acc_out[tid] = (tid >= 21 && tid < 29) ? tot[tid] : 0.0;
            if (tid == 0) ticket[1] = tag;
            return;
        }
        if (blockIdx.x != 0) {
            if (tid < 8) step_store_rec(recs + (size_t)blockIdx.x * 8 + tid, tot[21 + tid], tag);
            return;
        }
        /* workgroup 0: records 8 .. 8 * gridDim.x - 1; thread t takes t, t + 512, t + 1024, t + 1536 (+ 2048 ...): all of one sum k = t & 7 */
        const int n_rec = ((int)gridDim.x - 1) * 8;
        const step_v4u *first_rec = recs + 8;
        double sum = 0.0;
        bool lost = false;
        for (int r0 = tid; r0 < n_rec; r0 += 4 * DVO_STEP_THREADS) {
            step_v4u r[4];
            const step_v4u *pq[4];
            bool valid[4];
#pragma unroll
            for (int q = 0; q < 4; q++) { const int i = r0 + q * DVO_STEP_THREADS; valid[q] = i < n_rec; pq[q] = first_rec + (valid[q] ? i : r0); }
            int spins = 0;
            for (;;) {
                step_poll4(r, pq[0], pq[1], pq[2], pq[3]);
                bool ok = true;
#pragma unroll
                for (int q = 0; q < 4; q++) ok = ok && (!valid[q] || (r[q].y == tag && r[q].w == tag));
                if (ok) break;
                if (++spins > (1 << 22)) { lost = true; break; }            /* seconds: a workgroup of this launch never delivered -- NaN sums, not a hang */
                __builtin_amdgcn_s_sleep(1);
            }
#pragma unroll
            for (int q = 0; q < 4; q++) sum += valid[q] ? step_rec_value(r[q]) : 0.0;      /* fixed order: workgroups ascending */
        }
        if (lost) sum = __longlong_as_double(0x7ff8000000000000ll);
        double *flat = &part[0][0];                                         /* 64 chains x 8 sums */
        static_assert(sizeof(m.part) >= sizeof(double) * DVO_STEP_THREADS, "one partial sum per thread");
        flat[tid] = sum;
        __syncthreads();
        if (tid < DVO_NACC_PAD) {
            double t = 0.0;
            if (tid >= 21 && tid < 29) {
                const int k = tid - 21;
                for (int ch = 0; ch < DVO_STEP_THREADS / 8; ch++) t += flat[ch * 8 + k];      /* fixed order */
            }
            acc_out[tid] = t;
        }
        if (tid == 0) ticket[1] = tag;                                      /* the next launch's sequence number (visible at the kernel boundary) */
        return;
    }
    /* tail: this workgroup's row, then the ticket.  Without H only the eight sums 21..28 exist (round 5: the row is those eight
     * doubles, a quarter of the bytes the last arriver has to collect) */
    constexpr int ROW = WITH_H ? DVO_NACC_PAD : 8;
    if (WITH_H) { if (tid < DVO_NACC_PAD) store_sc1_f64(partials + (size_t)blockIdx.x * ROW + tid, (tid < DVO_NACC) ? tot[tid] : 0.0); }
    else if (tid < 8) store_sc1_f64(partials + (size_t)blockIdx.x * ROW + tid, tot[21 + tid]);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                        /* every storing wave: its stores have left */
    __syncthreads();
    if (tid == 0) {
        const unsigned t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        m.s_last = (t == gridDim.x - 1) ? 1 : 0;
    }
    __syncthreads();
    if (!m.s_last) return;
    {   /* the last arriver: all rows in the fixed two-level order of reduce_partials_kernel (16 interleaved chains, then in order) */
        const int nb = (int)gridDim.x;
        if constexpr (WITH_H) {
        const int k = tid & 31, ch = tid >> 5;                              /* 512 threads: 16 chains x 32 values */
        double sum = 0.0;
        for (int b0 = ch; b0 < nb && ch < 16; b0 += 16 * 16) {                          /* up to 16 rows of this chain at a time, all loads issued first */
            double v[16];
#pragma unroll
            for (int q = 0; q < 16; q++) {
                const int b = b0 + 16 * q;
                v[q] = (b < nb) ? load_sc1_f64(partials + (size_t)b * ROW + k) : 0.0;
            }
#pragma unroll
            for (int q = 0; q < 16; q++) sum += v[q];                       /* fixed order: rows ch, ch + 16, ch + 32, ... */
        }
        if (ch < 16) part[ch][k] = sum;
        __syncthreads();
        if (tid < DVO_NACC_PAD) {
            double t = 0.0;
#pragma unroll
            for (int j = 0; j < 16; j++) t += part[j][tid];
            acc_out[tid] = t;
        }
        } else {
        /* eight values per row: the same 16 chains (rows ch, ch + 16, ...), eight lanes each; 128 of the 512 threads load */
        const int k = tid & 7, ch = tid >> 3;
        double sum = 0.0;
        for (int b0 = ch; b0 < nb && ch < 16; b0 += 16 * 16) {
            double v[16];
#pragma unroll
            for (int q = 0; q < 16; q++) {
                const int b = b0 + 16 * q;
                v[q] = (b < nb) ? load_sc1_f64(partials + (size_t)b * ROW + k) : 0.0;
            }
#pragma unroll
            for (int q = 0; q < 16; q++) sum += v[q];
        }
        if (ch < 16) part[ch][k] = sum;
        __syncthreads();
        if (tid < DVO_NACC_PAD) {
            double t = 0.0;
            if (tid >= 21 && tid < 29) {
#pragma unroll
                for (int j = 0; j < 16; j++) t += part[j][tid - 21];
            }
            acc_out[tid] = t;
        }
        }
        if (tid == 0) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      /* for the next launch (visible at the kernel boundary) */
    }
}

}  // namespace dvo
#endif
