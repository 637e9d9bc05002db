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
