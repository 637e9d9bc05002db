/* dvo_tiled_step.h -- the launch of one iteration of the tiled / wide schedule, but for its per-point phase (see the comment above
 * tiled_step_kernel in dvo_kernels.hip: head = pending update, body = this workgroup's share of the points, tail = tagged rows to
 * workgroup 0).  Two kernels share it: tiled_step_kernel (dvo_kernels.hip: one point per lane, the reference's 3 x N list, H on
 * request) and tiled_step_pk_kernel (dvo_fused.hip, round 5: the packed two-points-per-lane loop over the compact list). */
#ifndef DVO_TILED_STEP_H
#define DVO_TILED_STEP_H
#include "dvo_kernel_common.h"

namespace dvo {

/* Tagged rows (round 5): a workgroup's sums (eight, or all 32 slots with H) travel as 16-byte records {value lo, tag, value hi, tag},
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
static_assert(DVO_STEP_THREADS == 512, "wave 7 fetches the sums for the head; workgroup 0 adds the rows as 512 / ROW chains");

/* LDS of a step launch (declared by the kernel, handed to the body) */
struct TiledStepLds {
    double tot[DVO_NACC_PAD];
    double part[16][DVO_NACC_PAD + 1];
    PoseState s;
    PoseCur nxt;                                                            /* the iterate the pending update produces */
    double g_s[12];                                                         /* sums 21..31 of the previous launch (29..31: the limbs of the exact sum of eps^2) */
    int s_stop0;
    unsigned seq;                                                           /* the launch's sequence number (ticket[1]) */
};

/* points(pc, run, b0, b1, tot): EVERY thread of the workgroup calls it; it leaves the workgroup's sums of points [b0, b1) at pose pc in
 * tot[0 .. DVO_NACC_PAD) (zeros when !run), behind a workgroup barrier */
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
            if (lane < 11) m.g_s[lane] = apply_prev ? acc_in[21 + lane] : 0.0;
            if (lane == 11) m.s_stop0 = st_in->stop;
            if (lane == 12) m.seq = ticket[1];
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
            /* the energy without an order (dvo_device_math.h): the limbs were added exactly over workgroups and ranks */
            const float e = pose_bookkeep(s, s.p[0], itr - 1, n_total, e2_from_limbs(m.g_s[8], m.g_s[9], m.g_s[10], m.g_s[6]), (int)m.g_s[7]);
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
    {
        /* tail: tagged rows to workgroup 0 (see step_store_rec).  A row is the sums 21..31 (sixteen records: g, sum eps^2, the visible
         * count, the three limbs of the exact sum of eps^2 -- round 6; eight until then), or with H all DVO_NACC_PAD slots */
        constexpr int ROW = WITH_H ? DVO_NACC_PAD : 16;
        constexpr int OFF = WITH_H ? 0 : 21;                                /* slot of the row's first value */
        const unsigned tag = m.seq + 1u;
        step_v4u *recs = reinterpret_cast<step_v4u *>(partials);             /* [workgroup][ROW] records */
        if (gridDim.x == 1) {                                               /* the only workgroup: its sums are the launch's */
            if (tid < DVO_NACC_PAD) acc_out[tid] = (tid >= OFF) ? tot[tid] : 0.0;
            if (tid == 0) ticket[1] = tag;
            return;
        }
        if (blockIdx.x != 0) {
            if (tid < ROW) step_store_rec(recs + (size_t)blockIdx.x * ROW + tid, (OFF + tid < DVO_NACC_PAD) ? tot[OFF + tid] : 0.0, tag);
            return;
        }
        /* workgroup 0: records ROW .. ROW * gridDim.x - 1; thread t takes t, t + 512, t + 1024, ...: all of one sum k = t & (ROW - 1) */
        static_assert(DVO_STEP_THREADS % ROW == 0, "a thread's records all belong to one sum");
        const int n_rec = ((int)gridDim.x - 1) * ROW;
        const step_v4u *first_rec = recs + ROW;
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
        if (lost) {                                                         /* NaN sums AND an error word the host checks: DVO_ERR_HIP, not DVO_OK with NaN poses */
            sum = __longlong_as_double(0x7ff8000000000000ll);
            ticket[2] = 1u;
        }
        double *flat = &part[0][0];                                         /* (512 / ROW) chains x ROW sums */
        static_assert(sizeof(m.part) >= sizeof(double) * DVO_STEP_THREADS, "one partial sum per thread");
        flat[tid] = sum;
        __syncthreads();
        if (tid < DVO_NACC_PAD) {
            double t = 0.0;
            if (tid >= OFF) {
                const int k = tid - OFF;
                for (int ch = 0; ch < DVO_STEP_THREADS / ROW; ch++) t += flat[ch * ROW + k];      /* fixed order */
            }
            acc_out[tid] = t;
        }
        if (tid == 0) ticket[1] = tag;                                      /* the next launch's sequence number (visible at the kernel boundary) */
    }
}

}  // namespace dvo
#endif
