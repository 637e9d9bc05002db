/*
 * dvo_kernels.hip -- hand-written HIP kernels (gfx950 / CDNA4, wave64) for the
 * SolveDVO edge-alignment hot path (reference src/SolveDVO.cpp:306-462, :619-1017).
 *
 * Kernels
 *   align_fused_kernel   the whole coarse-to-fine schedule of one frame pair in
 *                        ONE workgroup: per iteration every lane warps its share
 *                        of the reference edge points, gathers one 16-byte texel
 *                        {DT,gx,gy} of the now level, forms the 1x6 Jacobian row
 *                        and accumulates 29 sums in registers; a wave-shuffle +
 *                        LDS tree reduces them in a fixed order; lane 0 then
 *                        performs the reference's double-precision sub-gradient /
 *                        heavy-ball / trust-region update and publishes the next
 *                        float pose through LDS.  grid = number of frame pairs.
 *   pack_texels_kernel   planar DT/gx/gy (reference layout) -> float4 texels.
 *   eval_points_kernel, accumulate_kernel (+reduce_partials_kernel)
 *                        single-evaluation forms used for inspection, large
 *                        frames and the multi-GPU tiled mode.
 *   enlist_* kernels     selectedPts + enlistRefEdgePts (:1230-1264, :224-264).
 *
 * No MFMA: the path is per-point arithmetic plus a tree reduction, not a dense
 * contraction.  Compile with -ffp-contract=off (see Makefile): bit-parity of the
 * float32 per-point math with the CPU oracle depends on it.
 */
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
    const float s = pow2_neg(level);
    c.m00 = s * K.fx; c.m02 = s * K.cx;       /* (scaleMatrix*K), :344 */
    c.m11 = s * K.fy; c.m12 = s * K.cy;
    c.ncols_f = (float)cols; c.nrows_f = (float)rows;
    c.rows = rows;
}

/* per-lane partial sums of one iteration */
struct Acc {
    double g[6];        /* J^T W eps          (:777)   exact products, double fma */
    double e2;          /* sum eps^2          (:1312) */
    float H[21];        /* sum w J J^T upper triangle; per-lane float, tree in double */
    int nvis;
};
DVO_DEV void acc_zero(Acc &a) {
#pragma unroll
    for (int k = 0; k < 6; k++) a.g[k] = 0.0;
    a.e2 = 0.0;
#pragma unroll
    for (int k = 0; k < 21; k++) a.H[k] = 0.0f;
    a.nvis = 0;
}
/* visible point -> accumulators.  jw = (float)(J_k*w) (:716) widened, times eps
 * widened (:719-720): both factors are floats, so the double product is exact and
 * fma(a,b,c) == c + a*b bit for bit. */
DVO_DEV void acc_add(Acc &a, const float *J, float eps, float w) {
    const double e = (double)eps;
    float jw[6];
#pragma unroll
    for (int k = 0; k < 6; k++) {
        jw[k] = J[k] * w;
        a.g[k] = fma((double)jw[k], e, a.g[k]);
    }
    a.e2 = fma(e, e, a.e2);
    int h = 0;
#pragma unroll
    for (int i = 0; i < 6; i++)
#pragma unroll
        for (int j = i; j < 6; j++) { a.H[h] = fmaf(jw[i], J[j], a.H[h]); h++; }
    a.nvis += 1;
}

DVO_DEV double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

/* Fixed-shape reduction of the 29 accumulators over a workgroup: shuffle tree
 * inside each wave, one LDS row per wave, lanes 0..28 of the workgroup add the
 * rows in wave order.  Deterministic: same inputs -> same bits.
 * Result in tot[0..28] (valid after the trailing barrier). */
template <int BLOCK>
DVO_DEV void block_reduce(const Acc &a, double (*red)[DVO_NACC_PAD], double *tot) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double v;
#pragma unroll
    for (int k = 0; k < 21; k++) {
        v = wave_sum((double)a.H[k]);
        if (lane == 0) red[wave][k] = v;
    }
#pragma unroll
    for (int k = 0; k < 6; k++) {
        v = wave_sum(a.g[k]);
        if (lane == 0) red[wave][21 + k] = v;
    }
    v = wave_sum(a.e2);
    if (lane == 0) red[wave][27] = v;
    v = wave_sum((double)a.nvis);
    if (lane == 0) red[wave][28] = v;
    __syncthreads();
    if (threadIdx.x < DVO_NACC) {
        double s = 0.0;
#pragma unroll
        for (int w = 0; w < BLOCK / 64; w++) s += red[w][threadIdx.x];
        tot[threadIdx.x] = s;
    }
    __syncthreads();
}

/* ------------------------------------------------------------------------- */
/* texel packing: planar reference layout -> {DT,gx,gy,0}                      */
/* ------------------------------------------------------------------------- */
__global__ void __launch_bounds__(256)
pack_texels_kernel(const float *__restrict__ dt, const float *__restrict__ gx,
                   const float *__restrict__ gy, float4 *__restrict__ out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) out[i] = make_float4(dt[i], gx[i], gy[i], 0.0f);
}

hipError_t launch_pack_texels(const float *dt, const float *gx, const float *gy, float4 *out,
                              size_t n, hipStream_t s) {
    if (n == 0) return hipSuccess;
    size_t blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(pack_texels_kernel, dim3((unsigned)blocks), dim3(256), 0, s, dt, gx, gy, out, n);
    return hipGetLastError();
}

/* ------------------------------------------------------------------------- */
/* fused coarse-to-fine alignment: one workgroup per frame pair                */
/* ------------------------------------------------------------------------- */
template <int BLOCK>
__global__ void __launch_bounds__(BLOCK)
align_fused_kernel(LevelSet lv, Schedule sc, Intrinsics K, DevParams prm, Outputs out, int first_pair) {
    const int pair = first_pair + blockIdx.x;
    const int tid = threadIdx.x;
    __shared__ PoseState st;
    __shared__ double red[BLOCK / 64][DVO_NACC_PAD];
    __shared__ double tot[DVO_NACC_PAD];

    if (tid == 0) {
        const double *p = out.poses + (size_t)pair * 12;
        const bool ident = (sc.flags & 2) != 0;                  /* DVO_FLAG_IDENTITY_START (:2210-2211) */
#pragma unroll
        for (int k = 0; k < 9; k++) st.R[k] = ident ? ((k % 4 == 0) ? 1.0 : 0.0) : p[k];
#pragma unroll
        for (int k = 0; k < 3; k++) st.t[k] = ident ? 0.0 : p[9 + k];
    }
    __syncthreads();

    for (int l = sc.n_levels - 1; l >= 0; --l) {                 /* SolveDVO.cpp:2097 */
        const int iters = sc.iters[l];
        if (iters <= 0) continue;                                 /* :2099 */
        const LevelSlab &L = lv.l[l];
        const int N = L.N[pair];
        const float4 *__restrict__ tex = L.tex + (size_t)pair * L.tex_stride;
        const float *__restrict__ pts = L.pts + (size_t)pair * L.pt_cap * 3;
        float *energy = out.energy + (size_t)pair * sc.e_stride + sc.e_off[l];

        IterConst c;
        level_consts(c, K, l, L.rows, L.cols);

        for (int i = tid; i < iters; i += BLOCK) energy[i] = 0.0f;          /* :634 */
        if (tid == 0) pose_state_begin(st);                                  /* :642-657 */
        __syncthreads();

        for (int itr = 0; itr < iters; ++itr) {                              /* :658 */
#pragma unroll
            for (int k = 0; k < 9; k++) c.r[k] = uniform_f(st.Rf[k]);        /* :673 */
#pragma unroll
            for (int k = 0; k < 3; k++) c.t[k] = uniform_f(st.tf[k]);        /* :674 */

            Acc a;
            acc_zero(a);
            for (int i = tid; i < N; i += BLOCK) {                           /* :369, :433 */
                const float X = pts[3 * i], Y = pts[3 * i + 1], Z = pts[3 * i + 2];
                float xn, yn, zn, u, v;
                if (project_point(c, X, Y, Z, xn, yn, zn, u, v)) {
                    const float4 tx = tex[texel_index((int)v, (int)u, c.rows)];
                    float J[6];
                    jacobian_row(c, xn, yn, zn, tx.y, tx.z, J);
                    acc_add(a, J, tx.x, weight_of(tx.x));
                }
            }
            block_reduce<BLOCK>(a, red, tot);
            if (tid == 0) {
                const float e = pose_update(st, prm, itr, N, &tot[21], tot[27], (int)tot[28]);
                energy[itr] = e;                                             /* :690 */
            }
            __syncthreads();
            if (st.stop) break;                                              /* :877 */
        }

        /* finalEpsilons / finalReprojections = those of the best iterate (:703-704,
         * :1002-1003); recomputed once from the same float pose -> same bits. */
        if ((sc.flags & 1) && l == sc.last_level) {
            if (st.bestItr >= 0) {
#pragma unroll
                for (int k = 0; k < 9; k++) c.r[k] = uniform_f((float)st.bestR[k]);
#pragma unroll
                for (int k = 0; k < 3; k++) c.t[k] = uniform_f((float)st.bestT[k]);
                float *fe = out.final_eps + (size_t)pair * out.final_cap;
                float *fr = out.final_reproj + (size_t)pair * out.final_cap * 3;
                for (int i = tid; i < N; i += BLOCK) {
                    const float X = pts[3 * i], Y = pts[3 * i + 1], Z = pts[3 * i + 2];
                    float xn, yn, zn, u, v;
                    const bool vis = project_point(c, X, Y, Z, xn, yn, zn, u, v);
                    float e = 0.0f;
                    if (vis) e = tex[texel_index((int)v, (int)u, c.rows)].x;
                    fe[i] = e;
                    fr[3 * i] = u; fr[3 * i + 1] = v; fr[3 * i + 2] = zn;
                }
            }
            if (tid == 0) out.final_N[pair] = (st.bestItr >= 0) ? N : 0;
        }
        __syncthreads();
        if (tid == 0) {                                                      /* :997-1005 */
#pragma unroll
            for (int k = 0; k < 9; k++) st.R[k] = st.bestR[k];
            if (prm.enable_rotationize) rotationize(st.R);
#pragma unroll
            for (int k = 0; k < 3; k++) st.t[k] = st.bestT[k];
            out.best_idx[pair * DVO_LEVELS + l] = st.bestItr;
            out.ratio[pair * DVO_LEVELS + l] = st.bestRatio;
        }
        __syncthreads();
    }

    if (tid == 0) {
        double *p = out.poses + (size_t)pair * 12;
#pragma unroll
        for (int k = 0; k < 9; k++) p[k] = st.R[k];
#pragma unroll
        for (int k = 0; k < 3; k++) p[9 + k] = st.t[k];
    }
}

hipError_t launch_align_fused(int block_threads, const LevelSet &lv, const Schedule &sc,
                              const Intrinsics &K, const DevParams &prm, const Outputs &out,
                              int first_pair, int n_pairs, hipStream_t s) {
    if (n_pairs <= 0) return hipSuccess;
    switch (block_threads) {
    case 256:
        hipLaunchKernelGGL(align_fused_kernel<256>, dim3(n_pairs), dim3(256), 0, s, lv, sc, K, prm, out, first_pair);
        break;
    case 1024:
        hipLaunchKernelGGL(align_fused_kernel<1024>, dim3(n_pairs), dim3(1024), 0, s, lv, sc, K, prm, out, first_pair);
        break;
    default:
        hipLaunchKernelGGL(align_fused_kernel<512>, dim3(n_pairs), dim3(512), 0, s, lv, sc, K, prm, out, first_pair);
        break;
    }
    return hipGetLastError();
}

/* ------------------------------------------------------------------------- */
/* single evaluation kernels                                                   */
/* ------------------------------------------------------------------------- */
struct FloatPose { float r[9]; float t[3]; };

__global__ void __launch_bounds__(256)
eval_points_kernel(LevelSlab L, int pair, int level, Intrinsics K, FloatPose P,
                   float *reproj, float *Jout, float *eps, float *w, int *vis) {
    const int N = L.N[pair];
    const float4 *__restrict__ tex = L.tex + (size_t)pair * L.tex_stride;
    const float *__restrict__ pts = L.pts + (size_t)pair * L.pt_cap * 3;
    IterConst c;
    level_consts(c, K, level, L.rows, L.cols);
#pragma unroll
    for (int k = 0; k < 9; k++) c.r[k] = P.r[k];
#pragma unroll
    for (int k = 0; k < 3; k++) c.t[k] = P.t[k];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x) {
        const PointEval o = eval_point(c, tex, pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]);
        if (reproj) { reproj[3 * i] = o.u; reproj[3 * i + 1] = o.v; reproj[3 * i + 2] = o.zn; }
        if (Jout) {
#pragma unroll
            for (int k = 0; k < 6; k++) Jout[6 * i + k] = o.J[k];
        }
        if (eps) eps[i] = o.eps;
        if (w) w[i] = o.w;
        if (vis) vis[i] = o.vis ? 1 : 0;
    }
}

hipError_t launch_eval_points(const LevelSlab &L, int pair, int level, const Intrinsics &K,
                              const float *Rf, const float *tf,
                              float *reproj, float *J, float *eps, float *w, int *vis, hipStream_t s) {
    FloatPose P;
    for (int k = 0; k < 9; k++) P.r[k] = Rf[k];
    for (int k = 0; k < 3; k++) P.t[k] = tf[k];
    hipLaunchKernelGGL(eval_points_kernel, dim3(1024), dim3(256), 0, s, L, pair, level, K, P,
                       reproj, J, eps, w, vis);
    return hipGetLastError();
}

/* points [first, first+n) of one pair/level -> one row of 32 partial sums per block */
__global__ void __launch_bounds__(256)
accumulate_kernel(LevelSlab L, int pair, int level, Intrinsics K, FloatPose P,
                  int first, int n, double *__restrict__ partials) {
    __shared__ double red[256 / 64][DVO_NACC_PAD];
    __shared__ double tot[DVO_NACC_PAD];
    const float4 *__restrict__ tex = L.tex + (size_t)pair * L.tex_stride;
    const float *__restrict__ pts = L.pts + (size_t)pair * L.pt_cap * 3;
    IterConst c;
    level_consts(c, K, level, L.rows, L.cols);
#pragma unroll
    for (int k = 0; k < 9; k++) c.r[k] = P.r[k];
#pragma unroll
    for (int k = 0; k < 3; k++) c.t[k] = P.t[k];
    Acc a;
    acc_zero(a);
    const int end = first + n;
    for (int i = first + blockIdx.x * 256 + threadIdx.x; i < end; i += gridDim.x * 256) {
        const float X = pts[3 * i], Y = pts[3 * i + 1], Z = pts[3 * i + 2];
        float xn, yn, zn, u, v;
        if (project_point(c, X, Y, Z, xn, yn, zn, u, v)) {
            const float4 tx = tex[texel_index((int)v, (int)u, c.rows)];
            float J[6];
            jacobian_row(c, xn, yn, zn, tx.y, tx.z, J);
            acc_add(a, J, tx.x, weight_of(tx.x));
        }
    }
    block_reduce<256>(a, red, tot);
    if (threadIdx.x < DVO_NACC_PAD)
        partials[(size_t)blockIdx.x * DVO_NACC_PAD + threadIdx.x] = (threadIdx.x < DVO_NACC) ? tot[threadIdx.x] : 0.0;
}

/* acc[k] = sum over blocks in block order (fixed order => reproducible) */
__global__ void __launch_bounds__(64)
reduce_partials_kernel(const double *__restrict__ partials, int nblocks, double *__restrict__ acc) {
    const int k = threadIdx.x;
    if (k < DVO_NACC_PAD) {
        double s = 0.0;
        for (int b = 0; b < nblocks; b++) s += partials[(size_t)b * DVO_NACC_PAD + k];
        acc[k] = s;
    }
}

int accumulate_blocks_for(int n_points) {
    int b = (n_points + 255) / 256;
    if (b < 1) b = 1;
    if (b > 1024) b = 1024;
    return b;
}

hipError_t launch_accumulate(const LevelSlab &L, int pair, int level, const Intrinsics &K,
                             const float *Rf, const float *tf, int first_point, int n_points,
                             double *partials, int nblocks, double *acc, hipStream_t s) {
    FloatPose P;
    for (int k = 0; k < 9; k++) P.r[k] = Rf[k];
    for (int k = 0; k < 3; k++) P.t[k] = tf[k];
    hipLaunchKernelGGL(accumulate_kernel, dim3(nblocks), dim3(256), 0, s, L, pair, level, K, P,
                       first_point, n_points, partials);
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(1), dim3(64), 0, s, partials, nblocks, acc);
    return hipGetLastError();
}

/* ------------------------------------------------------------------------- */
/* SE(3) helpers on one lane (property tests of the device math)               */
/* ------------------------------------------------------------------------- */
__global__ void se3_exp_kernel(const double *psi, double *Rt) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        double p[6], R[9], t[3];
        for (int k = 0; k < 6; k++) p[k] = psi[k];
        se3_exp(p, R, t);
        for (int k = 0; k < 9; k++) Rt[k] = R[k];
        for (int k = 0; k < 3; k++) Rt[9 + k] = t[k];
    }
}
__global__ void se3_log_kernel(const double *Rt, double *psi) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        double p[6], R[9], t[3];
        for (int k = 0; k < 9; k++) R[k] = Rt[k];
        for (int k = 0; k < 3; k++) t[k] = Rt[9 + k];
        se3_log(R, t, p);
        for (int k = 0; k < 6; k++) psi[k] = p[k];
    }
}
__global__ void rotationize_kernel(double *R9) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        double R[9];
        for (int k = 0; k < 9; k++) R[k] = R9[k];
        rotationize(R);
        for (int k = 0; k < 9; k++) R9[k] = R[k];
    }
}
hipError_t launch_se3_exp(const double *psi, double *Rt12, hipStream_t s) {
    hipLaunchKernelGGL(se3_exp_kernel, dim3(1), dim3(64), 0, s, psi, Rt12);
    return hipGetLastError();
}
hipError_t launch_se3_log(const double *Rt12, double *psi, hipStream_t s) {
    hipLaunchKernelGGL(se3_log_kernel, dim3(1), dim3(64), 0, s, Rt12, psi);
    return hipGetLastError();
}
hipError_t launch_rotationize(double *R9, hipStream_t s) {
    hipLaunchKernelGGL(rotationize_kernel, dim3(1), dim3(64), 0, s, R9);
    return hipGetLastError();
}

/* ------------------------------------------------------------------------- */
/* selectedPts + enlistRefEdgePts  (SolveDVO.cpp:1230-1264, :224-264)          */
/* Column-major scan order (xx outer, yy inner): one wave per image column.     */
/* ------------------------------------------------------------------------- */
DVO_DEV bool ref_selected(int e, float d) { return (e > 0) && (d > 100.0f); }   /* :1251 */

__global__ void __launch_bounds__(64)
enlist_count_kernel(const int32_t *__restrict__ edge, const float *__restrict__ depth,
                    int rows, int *__restrict__ col_counts) {
    const int xx = blockIdx.x, lane = threadIdx.x;
    const size_t base = (size_t)xx * rows;
    int cnt = 0;
    for (int y0 = 0; y0 < rows; y0 += 64) {
        const int yy = y0 + lane;
        const bool sel = (yy < rows) && ref_selected(edge[base + yy], depth[base + yy]);
        cnt += __popcll(__ballot(sel));
    }
    if (lane == 0) col_counts[xx] = cnt;
}

/* exclusive scan of col_counts[0..cols) in place; col_counts[cols] = total */
__global__ void __launch_bounds__(1024)
enlist_scan_kernel(int *__restrict__ col_counts, int cols, int *__restrict__ N_out) {
    __shared__ int part[1024];
    const int tid = threadIdx.x;
    const int per = (cols + 1023) / 1024;
    const int b = tid * per;
    int s = 0;
    for (int k = 0; k < per; k++) if (b + k < cols) s += col_counts[b + k];
    part[tid] = s;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {          /* Hillis-Steele inclusive scan */
        int v = (tid >= off) ? part[tid - off] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    int run = (tid == 0) ? 0 : part[tid - 1];
    for (int k = 0; k < per; k++) {
        if (b + k < cols) { const int cval = col_counts[b + k]; col_counts[b + k] = run; run += cval; }
    }
    if (tid == 1023) { col_counts[cols] = part[1023]; *N_out = part[1023]; }
}

__global__ void __launch_bounds__(64)
enlist_write_kernel(const int32_t *__restrict__ edge, const float *__restrict__ depth,
                    int rows, int level, Intrinsics K, const int *__restrict__ col_offsets,
                    float *__restrict__ xyz, float *__restrict__ uv, int capacity) {
    const int xx = blockIdx.x, lane = threadIdx.x;
    const size_t base = (size_t)xx * rows;
    const float scaleFac = pow2_neg(level);                             /* :231 */
    const float tmpfx = (float)(1. / (double)(scaleFac * K.fx));        /* :232 double division */
    const float tmpfy = (float)(1. / (double)(scaleFac * K.fy));        /* :233 */
    const float tmpcx = scaleFac * K.cx;                                /* :234 */
    const float tmpcy = scaleFac * K.cy;                                /* :235 */
    int run = col_offsets[xx];
    for (int y0 = 0; y0 < rows; y0 += 64) {
        const int yy = y0 + lane;
        float d = 0.0f;
        bool sel = false;
        if (yy < rows) { d = depth[base + yy]; sel = ref_selected(edge[base + yy], d); }
        const unsigned long long m = __ballot(sel);
        if (sel) {
            const int nC = run + __popcll(m & ((1ull << lane) - 1ull));
            if (nC < capacity) {
                const float Z = d / 1000.0f;                            /* :248 */
                const float X = Z * ((float)xx - tmpcx) * tmpfx;        /* :249 */
                const float Y = Z * ((float)yy - tmpcy) * tmpfy;        /* :250 */
                xyz[3 * nC] = X; xyz[3 * nC + 1] = Y; xyz[3 * nC + 2] = Z;   /* :254-256 */
                if (uv) { uv[2 * nC] = (float)xx; uv[2 * nC + 1] = (float)yy; }   /* :244-245 */
            }
        }
        run += __popcll(m);
    }
}

hipError_t launch_enlist_ref_points(const int32_t *edge, const float *depth_mm, int rows, int cols,
                                    int level, const Intrinsics &K, int *col_counts,
                                    float *xyz, float *uv, int capacity, int *N_out, hipStream_t s) {
    hipLaunchKernelGGL(enlist_count_kernel, dim3(cols), dim3(64), 0, s, edge, depth_mm, rows, col_counts);
    hipLaunchKernelGGL(enlist_scan_kernel, dim3(1), dim3(1024), 0, s, col_counts, cols, N_out);
    hipLaunchKernelGGL(enlist_write_kernel, dim3(cols), dim3(64), 0, s, edge, depth_mm, rows, level, K,
                       col_counts, xyz, uv, capacity);
    return hipGetLastError();
}

}  // namespace dvo
