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
    c.nby = (unsigned)((rows + 15) >> 4);
    c.inv_nby = 1.0f / (float)c.nby;
    c.half_inv_nby = 0.5f * c.inv_nby;
}

/* per-lane partial sums of one iteration */
struct Acc {
    double g[6];        /* J^T W eps          (:777)   exact products, double fma */
    double e2;          /* sum eps^2          (:1312) */
    double H[21];       /* sum w J J^T upper triangle, row-major (the 21 normal-equation accumulators; pattern SolvePnP.cpp:168-182).
                           Exact float x float products summed in double, like g.  Only the kernels instantiated WITH_H carry
                           them: the reference's sub-gradient policy never forms H (SolveDVO.cpp:777) */
    int nvis;           /* visible points, counted per WAVE with ballots (uniform) */
    E2Limbs l;          /* the exact sum of eps^2 (round 6, dvo_device_math.h: the energy without an order): these kernels' sums travel
                           between launches, workgroups and ranks, so the limbs ride along always -- slots 29..31 of the 32 */
};
/* the 32 slots of a reduced accumulator row -> sum eps^2: the correctly rounded exact sum (slot 27 -- the sum as added -- only if a
 * residual was outside the limbs' range) */
DVO_DEV double acc_sum_eps2(const double *acc) { return e2_from_limbs(acc[29], acc[30], acc[31], acc[27]); }
DVO_DEV void acc_zero(Acc &a) {
#pragma unroll
    for (int k = 0; k < 6; k++) a.g[k] = 0.0;
    a.e2 = 0.0;
#pragma unroll
    for (int k = 0; k < 21; k++) a.H[k] = 0.0;
    a.nvis = 0;
    e2_limbs_zero(a.l);
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
    e2_limbs_add(a.l, eps);
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

/* The same reduce-scatter for exactly 8 doubles with the NEAR lane exchanges done by DPP (data-parallel-primitive operand
 * modifiers: a register move with a lane permutation, a few cycles) instead of ds_bpermute (a round trip through the LDS
 * crossbar, ~100 cycles each, twenty of them in a row in wave_reduce_scatter<double, 8>): bits 1, 2, 4 scatter the 8 values
 * (4 + 2 + 1 doubles exchanged), bit 8 folds by DPP too, bits 16 and 32 by lane-permute swaps (xor16_sum, xor32_sum).
 * On return lane L holds, in v[0], the wave total of value index  4*(L&1) + 2*((L>>1)&1) + ((L>>2)&1).  Fixed shape:
 * deterministic. */
#define DVO_DPP_QUAD_XOR1 0xB1      /* quad_perm [1,0,3,2] */
#define DVO_DPP_QUAD_XOR2 0x4E      /* quad_perm [2,3,0,1] */
#define DVO_DPP_ROW_SHL(n) (0x100 + (n))   /* lane i <- lane i+n of its row of 16 */
#define DVO_DPP_ROW_SHR(n) (0x110 + (n))   /* lane i <- lane i-n */
template <int CTRL>
DVO_DEV double dpp_quad(double x) {
    const long long b = __double_as_longlong(x);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)b, CTRL, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xF, 0xF, false);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
/* partner lane L ^ N for N = 4 or 8: lanes whose bit N is clear read lane L+N (row_shl), the others lane L-N (row_shr);
 * the bank mask (banks = groups of 4 lanes of a row) picks which lanes each of the two moves writes */
template <int N>
DVO_DEV double dpp_xor_row(double x) {
    static_assert(N == 4 || N == 8, "row-local partner");
    constexpr int LOW = (N == 4) ? 0x5 : 0x3, HIGH = (N == 4) ? 0xA : 0xC;
    const long long b = __double_as_longlong(x);
    int lo = __builtin_amdgcn_update_dpp(0, (int)b, DVO_DPP_ROW_SHL(N), 0xF, LOW, false);
    lo = __builtin_amdgcn_update_dpp(lo, (int)b, DVO_DPP_ROW_SHR(N), 0xF, HIGH, false);
    int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), DVO_DPP_ROW_SHL(N), 0xF, LOW, false);
    hi = __builtin_amdgcn_update_dpp(hi, (int)(b >> 32), DVO_DPP_ROW_SHR(N), 0xF, HIGH, false);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
/* x[L] + x[L ^ 16] and x[L] + x[L ^ 32] in every lane, through gfx950's v_permlane16_swap / v_permlane32_swap (register moves
 * between the rows of 16 lanes / the halves of the wave) instead of ds_bpermute (round 5: the two crossbar round trips per
 * double were ~200 cycles of every wave's reduction).  swap(a, a) leaves the even rows' (low half's) values in both rows of a
 * pair in the first result and the odd rows' (high half's) in the second: their sum is the pairwise sum, the same bits in both
 * partners. */
DVO_DEV double xor16_sum(double x) {
    const long long b = __double_as_longlong(x);
    const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)b, (unsigned)b, false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)(b >> 32), (unsigned)(b >> 32), false, false);
    return __longlong_as_double((long long)(((unsigned long long)hi[0] << 32) | lo[0])) +
           __longlong_as_double((long long)(((unsigned long long)hi[1] << 32) | lo[1]));
}
DVO_DEV double xor32_sum(double x) {
    const long long b = __double_as_longlong(x);
    const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)b, (unsigned)b, false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)(b >> 32), (unsigned)(b >> 32), false, false);
    return __longlong_as_double((long long)(((unsigned long long)hi[0] << 32) | lo[0])) +
           __longlong_as_double((long long)(((unsigned long long)hi[1] << 32) | lo[1]));
}
DVO_DEV void wave_reduce_scatter8_dpp(double (&v)[8]) {
    const int lane = threadIdx.x & 63;
    {   /* bit 1: 8 -> 4 values */
        const bool up = (lane & 1) != 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const double keep = up ? v[j + 4] : v[j], send = up ? v[j] : v[j + 4];
            v[j] = keep + dpp_quad<DVO_DPP_QUAD_XOR1>(send);
        }
    }
    {   /* bit 2: 4 -> 2 */
        const bool up = (lane & 2) != 0;
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const double keep = up ? v[j + 2] : v[j], send = up ? v[j] : v[j + 2];
            v[j] = keep + dpp_quad<DVO_DPP_QUAD_XOR2>(send);
        }
    }
    {   /* bit 4: 2 -> 1 */
        const bool up = (lane & 4) != 0;
        const double keep = up ? v[1] : v[0], send = up ? v[0] : v[1];
        v[0] = keep + dpp_xor_row<4>(send);
    }
    v[0] += dpp_xor_row<8>(v[0]);
    v[0] = xor16_sum(v[0]);
    v[0] = xor32_sum(v[0]);
}
/* value index held by lane L after wave_reduce_scatter8_dpp */
DVO_DEV int reduce_scatter8_dpp_index(int lane) { return 4 * (lane & 1) + 2 * ((lane >> 1) & 1) + ((lane >> 2) & 1); }

/* Fixed-shape reduction of the 29 accumulators and the three limbs of the exact sum of eps^2 over a workgroup.
 * Result in tot[0..31] (valid after the trailing barrier). */
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
    {   /* round 5: the DPP / lane-permute reduce-scatter of the packed kernel (twenty ds_bpermute round trips shorter) */
        double d[8];
#pragma unroll
        for (int k = 0; k < 6; k++) d[k] = a.g[k];
        d[6] = a.e2;
        d[7] = 0.0;
        wave_reduce_scatter8_dpp(d);
        const int idx = reduce_scatter8_dpp_index(lane);
        if (lane < 8 && idx < 7) red[wave][21 + idx] = d[0];
    }
    {   /* the three limbs: integers below 2^53, every addition from here on is exact */
        double d[8];
        d[0] = (double)a.l.l0; d[1] = (double)a.l.l1; d[2] = (double)a.l.l2;
#pragma unroll
        for (int k = 3; k < 8; k++) d[k] = 0.0;
        wave_reduce_scatter8_dpp(d);
        const int idx = reduce_scatter8_dpp_index(lane);
        if (lane < 8 && idx < 3) red[wave][29 + idx] = d[0];
    }
    if (lane == 0) red[wave][28] = (double)a.nvis;
    __syncthreads();
    if (threadIdx.x < DVO_NACC_PAD) {
        double s = 0.0;
#pragma unroll
        for (int w = 0; w < BLOCK / 64; w++) s += red[w][threadIdx.x];
        tot[threadIdx.x] = s;
    }
    __syncthreads();
}

}  // namespace dvo
#endif
