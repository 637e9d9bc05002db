/*
 * dvo_kernel_common.h -- device helpers shared by the kernel files (dvo_kernels.hip, dvo_fused.hip):
 * level constants, the per-lane accumulators of one iteration and the fixed-shape wave / workgroup reductions.
 * Internal; compiled with -ffp-contract=off like everything that touches the per-point float32 math.
 */
#ifndef DVO_KERNEL_COMMON_H_
#define DVO_KERNEL_COMMON_H_

#include "dvo_launch.h"

namespace dvo {

/* ------------------------------------------------------------------------- */
/* small helpers                                                              */
/* ------------------------------------------------------------------------- */
DVO_DEV float uniform_f(float x) {          /* wave-uniform value -> SGPR */
    return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(x)));
}
DVO_DEV float pow2_neg(int level) {          /* (float)pow(2,-level), exact (:231,:334) */
    return __int_as_float((127 - level) << 23);
}
DVO_DEV void level_consts(IterConst &c, const Intrinsics &K, int level, int rows, int cols) {
    c.interp = K.interp;
    c.cols = cols;
    const float s = pow2_neg(level);
    c.m00 = s * K.fx; c.m02 = s * K.cx;       /* (scaleMatrix*K), :344 */
    c.m11 = s * K.fy; c.m12 = s * K.cy;
    c.m00_z1 = exact_div_z1(c.m00); c.m11_z1 = exact_div_z1(c.m11);
    c.ncols_f = (float)cols; c.nrows_f = (float)rows;
    c.rows = rows;
    c.tiles_per_col = texel_tiles_per_col(rows);
    c.pfx = (float)(1. / (double)(s * K.fx));  /* :232 double division, narrowed */
    c.pfy = (float)(1. / (double)(s * K.fy));  /* :233 */
    c.pcx = s * K.cx;                          /* :234 */
    c.pcy = s * K.cy;                          /* :235 */
}

/* per-lane partial sums of one iteration */
struct Acc {
    double g[6];        /* J^T W eps          (:777)   exact products, double fma */
    double e2;          /* sum eps^2          (:1312) */
    double H[21];       /* sum w J J^T upper triangle, row-major (the 21 normal-equation accumulators; pattern SolvePnP.cpp:168-182).
                           Exact float x float products summed in double, like g.  Only the kernels instantiated WITH_H carry
                           them: the reference's sub-gradient policy never forms H (SolveDVO.cpp:777) */
    int nvis;           /* visible points, counted per WAVE with ballots (uniform) */
};
DVO_DEV void acc_zero(Acc &a) {
#pragma unroll
    for (int k = 0; k < 6; k++) a.g[k] = 0.0;
    a.e2 = 0.0;
#pragma unroll
    for (int k = 0; k < 21; k++) a.H[k] = 0.0;
    a.nvis = 0;
}
/* visible point -> accumulators.  jw = (float)(J_k*w) (:716) widened, times eps
 * widened (:719-720): both factors are floats, so the double product is exact and
 * fma(a,b,c) == c + a*b bit for bit. */
template <bool WITH_H>
DVO_DEV void acc_add(Acc &a, const float *J, float eps, float w) {
    const double e = (double)eps;
    float jw[6];
#pragma unroll
    for (int k = 0; k < 6; k++) {
        jw[k] = J[k] * w;
        a.g[k] = fma((double)jw[k], e, a.g[k]);
    }
    a.e2 = fma(e, e, a.e2);
    if (WITH_H) {
        /* upper triangle, row-major: index(i,j) = i*6 - i*(i-1)/2 + (j-i); written with
         * compile-time indices so H stays in registers */
#pragma unroll
        for (int i = 0; i < 6; i++)
#pragma unroll
            for (int j = 0; j < 6; j++)
                if (j >= i) a.H[i * 6 - (i * (i - 1)) / 2 + (j - i)] = fma((double)jw[i], (double)J[j], a.H[i * 6 - (i * (i - 1)) / 2 + (j - i)]);
    }
}

/* Butterfly reduce-scatter inside a wave: NV values per lane; at step s lanes
 * that differ in bit (5-s) exchange half of their values and add the other half,
 * so the work halves every step (NV-1 exchanges instead of 6*NV).  The remaining
 * lane bits are folded with plain xor steps.  On return every lane L holds, in
 * v[0], the wave total of value index L >> (6 - log2 NV).  Fixed shape:
 * deterministic. */
template <typename T, int NV, int HALF, int BIT>
struct ReduceScatterStep {
    static DVO_DEV void run(T (&v)[NV], int lane) {
        const bool up = (lane & BIT) != 0;
#pragma unroll
        for (int j = 0; j < HALF; j++) {
            const T keep = up ? v[j + HALF] : v[j];
            const T send = up ? v[j] : v[j + HALF];
            v[j] = keep + __shfl_xor(send, BIT, 64);
        }
        ReduceScatterStep<T, NV, HALF / 2, BIT / 2>::run(v, lane);
    }
};
template <typename T, int NV, int BIT>
struct ReduceScatterStep<T, NV, 0, BIT> {          /* one value per lane left: fold the remaining lane bits */
    static DVO_DEV void run(T (&v)[NV], int) {
#pragma unroll
        for (int b = BIT; b >= 1; b >>= 1) v[0] += __shfl_xor(v[0], b, 64);
    }
};
template <typename T, int NV>
DVO_DEV void wave_reduce_scatter(T (&v)[NV]) {
    ReduceScatterStep<T, NV, NV / 2, 32>::run(v, threadIdx.x & 63);   /* all indices are compile-time */
}

/* Fixed-shape reduction of the 29 accumulators over a workgroup.
 * Result in tot[0..28] (valid after the trailing barrier). */
template <int BLOCK, bool WITH_H>
DVO_DEV void block_reduce(const Acc &a, double (*red)[DVO_NACC_PAD], double *tot) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (WITH_H) {
        double h[32];
#pragma unroll
        for (int k = 0; k < 32; k++) h[k] = (k < 21) ? a.H[k] : 0.0;
        wave_reduce_scatter<double, 32>(h);
        const int idx = lane >> 1;
        if ((lane & 1) == 0 && idx < 21) red[wave][idx] = h[0];
    } else if (lane < 21) {
        red[wave][lane] = 0.0;
    }
    {
        double d[8];
#pragma unroll
        for (int k = 0; k < 6; k++) d[k] = a.g[k];
        d[6] = a.e2;
        d[7] = 0.0;
        wave_reduce_scatter<double, 8>(d);
        const int idx = lane >> 3;
        if ((lane & 7) == 0 && idx < 7) red[wave][21 + idx] = d[0];
    }
    if (lane == 0) red[wave][28] = (double)a.nvis;
    __syncthreads();
    if (threadIdx.x < DVO_NACC) {
        double s = 0.0;
#pragma unroll
        for (int w = 0; w < BLOCK / 64; w++) s += red[w][threadIdx.x];
        tot[threadIdx.x] = s;
    }
    __syncthreads();
}

}  // namespace dvo
#endif
