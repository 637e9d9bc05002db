/*
 * dvo_kernels.hip -- hand-written HIP kernels (gfx950 / CDNA4, wave64) for the
 * SolveDVO edge-alignment hot path (reference src/SolveDVO.cpp:306-462, :619-1017).
 *
 * Kernels
 *   align_fused_kernel   the whole coarse-to-fine schedule of one frame pair in ONE workgroup
 *                        (throughput path, grid = number of frame pairs).  Per level the reference
 *                        points are staged once into LDS; per iteration every lane warps its share of
 *                        them, gathers one 16-byte texel {DT,gx,gy,w} of the now level (software
 *                        pipelined, ping-pong registers), forms the 1x6 Jacobian row and accumulates
 *                        g = J^T W eps and sum eps^2 in double; a wave butterfly + LDS tree reduces them
 *                        in a fixed order; lane 0 performs the reference's double-precision sub-gradient /
 *                        heavy-ball / trust-region update and publishes the next float pose through LDS.
 *   accumulate_state_kernel + reduce_partials_kernel / iter_reduce_update_kernel + iter_*_kernel
 *                        one iteration spread over all CUs with the optimiser state in HBM: large single
 *                        frames (dvo_align_pyramid_wide) and the multi-GPU tiled mode (dvo_iter_*), where
 *                        the 32 sums (21 H + 6 g + sum eps^2 + visible count) are all-reduced in between.
 *   eval_points_kernel, accumulate_kernel      single evaluations for inspection / parity tests.
 *   pack_texels_kernel, replicate_level_kernel, unpack_texels_kernel     resident-data management.
 *   (per-frame preprocessing -- Canny, distance transform, point extraction -- lives in dvo_frames.hip)
 *
 * No MFMA: the path is per-point arithmetic plus a tree reduction, not a dense contraction.
 * Compile with -ffp-contract=off (see Makefile): bit-parity of the float32 per-point math with the
 * CPU oracle depends on it.
 */
#include "dvo_kernel_common.h"
#include "dvo_palette.h"
#include "dvo_tiled_step.h"

namespace dvo {

/* The per-point phase of one iteration over points [first, end) with a lane
 * stride of `stride` (:369-407 + :433-451).  U points per lane are in flight at
 * once: their coordinates are loaded, projected, and the U texel gathers are
 * issued back to back (unconditionally, index 0 when not visible) before any
 * Jacobian arithmetic, so a lane waits for one memory round trip per U points
 * instead of one per point.  The trip count is wave-uniform, so the visible
 * count can be taken from ballots. */
/* Where a lane finds reference point i: the first `n_lds` points of the level
 * live in LDS as three planes (x | y | z, each `cap` floats; conflict-free
 * ds_read_b32), the rest are read from HBM (3 x N column-major, 12 B / point). */
struct PointSrc {
    const float *__restrict__ g;    /* global list, 3 x N column-major */
    const uint2 *__restrict__ gc;   /* global compact list (or nullptr) */
    const float *l;                 /* LDS planes  */
    int n_lds, cap;
};
/* Point sources.  XYZ: the reference's 3 x N float list (12 B / point).  COMPACT: what enlistRefEdgePts was given --
 * pixel (xx, yy) and Z = depth/1000 -- in 8 bytes {xx | yy << 16, Z}; X and Y are rebuilt with the very operations of
 * :249-250 (X = Z*(xx - tmpcx)*tmpfx), so the bits are those of the 12-byte list.  A third less LDS and stream
 * traffic per point; only lists built by the engine's own enlist kernels have it. */
enum { SRC_GLOBAL_XYZ = 0, SRC_LDS_XYZ = 1, SRC_GLOBAL_COMPACT = 2, SRC_LDS_COMPACT = 3 };
/* branch-free on purpose: a (divergent) LDS-or-HBM choice per point makes the compiler drain all
 * outstanding gathers before it; whole rounds are served from one source instead */
template <int SRC>
DVO_DEV void load_point(const IterConst &c, const PointSrc &p, int i, float &X, float &Y, float &Z) {
    if (SRC == SRC_LDS_XYZ) {
        X = p.l[i]; Y = p.l[p.cap + i]; Z = p.l[2 * p.cap + i];
    } else if (SRC == SRC_GLOBAL_XYZ) {
        X = p.g[3 * i]; Y = p.g[3 * i + 1]; Z = p.g[3 * i + 2];
    } else if (SRC == SRC_LDS_COMPACT) {
        expand_compact(c, __float_as_uint(p.l[i]), p.l[p.cap + i], X, Y, Z);
    } else {
        const uint2 v = p.gc[i];
        expand_compact(c, v.x, __uint_as_float(v.y), X, Y, Z);
    }
}

/* One "round" = U points per lane.  The per-point phase is software-pipelined
 * over rounds: while the Jacobian rows of round r are computed, the U texel
 * gathers of round r+1 are already in flight (loads return in order on CDNA, so
 * waiting for round r's texels leaves the younger requests outstanding).  Two
 * named buffer sets (ping-pong, loop unrolled by two rounds) avoid register
 * copies, and the compute stage is branch-free (a lane whose point is not
 * visible works on a harmless dummy and contributes exact zeros), so the steady
 * state is straight-line code. */
template <int U> struct RoundBuf {
    float xn[U], yn[U], zn[U];
    float4 t[U];
    bool vis[U];
};

/* stage B: load U points, project, issue the gathers (texel 0 when not visible) */
template <int U, int SRC>
DVO_DEV void round_issue(const IterConst &c, const float4 *__restrict__ tex, const PointSrc &pts,
                         int base, int end, int lane_off, int stride, RoundBuf<U> &b) {
    int idx[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
        const int i = base + lane_off + u * stride;
        const bool valid = i < end;
        const int ii = valid ? i : (end - 1);
        float X, Y, Z, uu, vv;
        load_point<SRC>(c, pts, ii, X, Y, Z);
        const bool vis = project_point(c, X, Y, Z, b.xn[u], b.yn[u], b.zn[u], uu, vv) && valid;
        b.vis[u] = vis;
        idx[u] = vis ? texel_index((int)vv, (int)uu, c.tiles_per_col) : 0;
        if (!vis) { b.xn[u] = 0.0f; b.yn[u] = 0.0f; b.zn[u] = 1.0f; }    /* finite dummy */
    }
#pragma unroll
    for (int u = 0; u < U; u++) b.t[u] = tex[idx[u]];
}
/* stage C: Jacobian rows + accumulation; invisible lanes add exact zeros */
/* INTERP: 0 = nearest lookup only (the reference's build), 1 = interpolate() always, 2 = decide at run time
 * from c.interp (the non-critical single-evaluation kernels) */
template <int U, bool WITH_H, int INTERP>
DVO_DEV void round_compute(const IterConst &c, const float4 *__restrict__ tex, const RoundBuf<U> &b, Acc &a) {
#pragma unroll
    for (int u = 0; u < U; u++) {
        a.nvis += __popcll(__ballot(b.vis[u]));
        float J[6];
        jacobian_row(c, b.xn[u], b.yn[u], b.zn[u], b.t[u].y, b.t[u].z, J);
        float eps = b.vis[u] ? b.t[u].x : 0.0f;
        float w = b.vis[u] ? b.t[u].w : 0.0f;
        if (INTERP == 1 || (INTERP == 2 && c.interp)) {     /* off in the reference's build (SolveDVO.h:97) */
            if (b.vis[u]) {     /* u, v recomputed exactly as project_point does (:344) */
                const float uu = c.m00 * b.xn[u] + c.m02 * b.zn[u], vv = c.m11 * b.yn[u] + c.m12 * b.zn[u];
                eps = interpolate_dt(c, tex, vv, uu);
                w = weight_of(eps);
            }
        }
        acc_add<WITH_H>(a, J, eps, w);
    }
}

/* The per-point phase of one iteration over points [first, end) with a lane
 * stride of `stride` (:369-407 + :433-451).  The trip count is wave-uniform, so
 * the visible count can be taken from ballots. */
template <int U, bool WITH_H, int SRC, int INTERP>
DVO_DEV void accumulate_points(const IterConst &c, const float4 *__restrict__ tex,
                               const PointSrc &pts, int first, int end, int lane_off,
                               int stride, Acc &a) {
    if (first >= end) return;
    const int step = stride * U;
    /* rounds in which THIS wave still has a point (its lowest lane offset decides; wave-uniform): waves beyond the tail
     * of the last, partial round skip it instead of processing dummies */
    const int wave_off = __builtin_amdgcn_readfirstlane(lane_off - (int)(threadIdx.x & 63));
    const int n_rounds = (end - first - wave_off + step - 1) / step;
    if (n_rounds <= 0) return;
    RoundBuf<U> A, B;
    round_issue<U, SRC>(c, tex, pts, first, end, lane_off, stride, A);
    int r = 0;
    for (; r + 2 < n_rounds; r += 2) {                       /* steady state: A = round r */
        round_issue<U, SRC>(c, tex, pts, first + (r + 1) * step, end, lane_off, stride, B);
        round_compute<U, WITH_H, INTERP>(c, tex, A, a);
        round_issue<U, SRC>(c, tex, pts, first + (r + 2) * step, end, lane_off, stride, A);
        round_compute<U, WITH_H, INTERP>(c, tex, B, a);
    }
    if (r + 1 < n_rounds) {                                  /* two rounds left */
        round_issue<U, SRC>(c, tex, pts, first + (r + 1) * step, end, lane_off, stride, B);
        round_compute<U, WITH_H, INTERP>(c, tex, A, a);
        round_compute<U, WITH_H, INTERP>(c, tex, B, a);
    } else {
        round_compute<U, WITH_H, INTERP>(c, tex, A, a);
    }
}

/* ------------------------------------------------------------------------- */
/* texel packing: planar reference layout -> {DT,gx,gy,0}                      */
/* ------------------------------------------------------------------------- */
__global__ void __launch_bounds__(256)
pack_texels_kernel(const float *__restrict__ dt, const float *__restrict__ gx,
                   const float *__restrict__ gy, float4 *__restrict__ out, int rows, int cols) {
    const size_t n = (size_t)rows * cols;
    const int tpc = texel_tiles_per_col(rows);
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        const int xx = (int)(i / rows), yy = (int)(i - (size_t)xx * rows);
        out[texel_index(yy, xx, tpc)] = make_float4(dt[i], gx[i], gy[i], weight_of(dt[i]));
    }
}

hipError_t launch_pack_texels(const float *dt, const float *gx, const float *gy, float4 *out,
                              int rows, int cols, hipStream_t s) {
    const size_t n = (size_t)rows * cols;
    if (n == 0) return hipSuccess;
    size_t blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(pack_texels_kernel, dim3((unsigned)blocks), dim3(256), 0, s, dt, gx, gy, out, rows, cols);
    return hipGetLastError();
}

/* slot p in [dst_first, dst_first+dst_count) <- copy of slot (p - dst_first) % n_src : one launch per level
 * instead of a pack + copies per pair (bench / throughput set-up, warm replicas) */
__global__ void __launch_bounds__(256)
replicate_level_kernel(float4 *tex, const unsigned char *__restrict__ src_has_tex, size_t tex_stride, float *pts, uint2 *cpts, unsigned *cidx,
                       unsigned *cpt4, unsigned *chdr, int *pt4_ok, int pt_cap, int *N, int n_src, int dst_first, int dst_count) {
    const int p = dst_first + blockIdx.y;
    const int src = blockIdx.y % n_src;
    if (p == src) return;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    /* tex NULL: no source holds 16-byte texels (compact form only); otherwise per source (on a sparse slab only the destinations
     * of sources WITH texels have memory behind theirs) */
    if (tex && (!src_has_tex || src_has_tex[src])) {
        const float4 *st = tex + (size_t)src * tex_stride;
        float4 *dt = tex + (size_t)p * tex_stride;
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < tex_stride; i += stride) dt[i] = st[i];
    }
    const int n = N[src];
    const float *sp = pts + (size_t)src * pt_cap * 3;
    float *dp = pts + (size_t)p * pt_cap * 3;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < (size_t)3 * n; i += stride) dp[i] = sp[i];
    if (cpts) {
        const uint2 *sc_ = cpts + (size_t)src * pt_cap;
        uint2 *dc = cpts + (size_t)p * pt_cap;
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < (size_t)n; i += stride) dc[i] = sc_[i];
    }
    if (cidx) {
        const unsigned *si = cidx + (size_t)src * pt_cap;
        unsigned *di = cidx + (size_t)p * pt_cap;
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < (size_t)n; i += stride) di[i] = si[i];
    }
    if (cpt4) {
        const unsigned *s4 = cpt4 + (size_t)src * pt_cap, *sh = chdr + (size_t)src * (pt_cap / 64);
        unsigned *d4 = cpt4 + (size_t)p * pt_cap, *dh = chdr + (size_t)p * (pt_cap / 64);
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < (size_t)n; i += stride) d4[i] = s4[i];
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < (size_t)(n + 63) / 64; i += stride) dh[i] = sh[i];
        if (blockIdx.x == 0 && threadIdx.x == 0) pt4_ok[p] = pt4_ok[src];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) N[p] = n;
}
hipError_t launch_replicate_level(float4 *tex, const unsigned char *src_has_tex, size_t tex_stride, float *pts, uint2 *cpts, unsigned *cidx, unsigned *cpt4,
                                  unsigned *chdr, int *pt4_ok, int pt_cap, int *N, int n_src, int dst_first, int dst_count, hipStream_t s) {
    if (dst_count <= 0) return hipSuccess;
    hipLaunchKernelGGL(replicate_level_kernel, dim3(64, dst_count), dim3(256), 0, s, tex, src_has_tex, tex_stride, pts, cpts, cidx, cpt4, chdr, pt4_ok,
                       pt_cap, N, n_src, dst_first, dst_count);
    return hipGetLastError();
}

/* The compact list of one pair again in 4 bytes per point + one header per 64-point chunk (dvo_device_math.h: pt4_decode).
 * One workgroup per pair.  LOSSLESS BY VERIFICATION: every point is encoded, decoded with the function the fused kernel uses
 * and compared bit for bit with {xx | yy << 16, Z}; pt4_ok[pair] = 1 only if all of them survive. */
__global__ void __launch_bounds__(256)
points4_build_kernel(const uint2 *__restrict__ cpts, const int *__restrict__ N, int pt_cap, int rows, unsigned *__restrict__ cpt4,
                     unsigned *__restrict__ chdr, int *__restrict__ pt4_ok, int first_pair) {
    const int pair = first_pair + blockIdx.x;
    const int n = N[pair];
    cpts += (size_t)pair * pt_cap; cpt4 += (size_t)pair * pt_cap; chdr += (size_t)pair * (pt_cap / 64);
    const unsigned nby = (unsigned)((rows + 15) >> 4);
    const float inv_nby = 1.0f / (float)nby, half_inv = 0.5f * inv_nby;
    bool bad = false;
    for (int i = threadIdx.x; i < n; i += 256) {
        const uint2 p = cpts[i], p0 = cpts[i & ~63];          /* the chunk's first point (a 64-lane wave reads it once) */
        const unsigned xx = p.x & 0xffffu, yy = p.x >> 16;
        const unsigned L = (xx >> 4) * nby + (yy >> 4), L0 = ((p0.x & 0xffffu) >> 4) * nby + ((p0.x >> 16) >> 4);
        const float Z = __uint_as_float(p.y);
        const float dmm = rintf(Z * 1000.0f);
        const bool enc = (L >= L0) && (L - L0 <= 255u) && (L < (1u << 20)) && (dmm >= 0.0f) && (dmm <= 65535.0f);
        const unsigned w = enc ? ((xx & 15u) | ((yy & 15u) << 4) | ((unsigned)dmm << 8) | ((L - L0) << 24)) : 0u;
        float xf, yf, zf;
        pt4_decode(nby, inv_nby, half_inv, w, L0, xf, yf, zf);
        if (!enc || xf != (float)xx || yf != (float)yy || __float_as_uint(zf) != p.y) bad = true;
        cpt4[i] = w;
        if ((i & 63) == 0) chdr[i >> 6] = L0;
    }
    const int any_bad = __syncthreads_or(bad ? 1 : 0);
    if (threadIdx.x == 0) pt4_ok[pair] = (n > 0 && !any_bad) ? 1 : 0;
}
hipError_t launch_points4_build(const uint2 *cpts, const int *N, int pt_cap, int rows, unsigned *cpt4, unsigned *chdr, int *pt4_ok,
                                int first_pair, int count, hipStream_t s) {
    if (count <= 0) return hipSuccess;
    hipLaunchKernelGGL(points4_build_kernel, dim3(count), dim3(256), 0, s, cpts, N, pt_cap, rows, cpt4, chdr, pt4_ok, first_pair);
    return hipGetLastError();
}

/* The packed kernel writes finalEpsilons / finalReprojections in the order of its compact point list (16 x 16 blocks: the
 * order it gathers in); callers get them in the reference's order (:703-704): out[cidx[i]] = in[i], at the boundary, on demand. */
__global__ void __launch_bounds__(256)
final_permute_kernel(const unsigned *__restrict__ cidx, const float *__restrict__ fe_blk, const float *__restrict__ fr_blk, int n,
                     float *__restrict__ fe, float *__restrict__ fr) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const unsigned j = cidx[i];
        if (j >= (unsigned)n) continue;
        fe[j] = fe_blk[i];
        fr[3 * (size_t)j] = fr_blk[3 * (size_t)i]; fr[3 * (size_t)j + 1] = fr_blk[3 * (size_t)i + 1]; fr[3 * (size_t)j + 2] = fr_blk[3 * (size_t)i + 2];
    }
}
hipError_t launch_final_permute(const unsigned *cidx, const float *fe_blk, const float *fr_blk, int n, float *fe, float *fr, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    int blocks = (n + 255) / 256; if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(final_permute_kernel, dim3(blocks), dim3(256), 0, s, cidx, fe_blk, fr_blk, n, fe, fr);
    return hipGetLastError();
}

__global__ void __launch_bounds__(256)
replicate_compact_kernel(unsigned *p4, size_t p4_stride, float2 *pal, int *pal_n, int n_src, int dst_first) {
    const int p = dst_first + blockIdx.y;
    const int src = blockIdx.y % n_src;
    if (p == src) return;
    const int pn = pal_n[src];
    if (blockIdx.x == 0 && threadIdx.x == 0) pal_n[p] = pn;
    const int n = pal_count(pn);                   /* a partial form (DVO_PAL_PARTIAL) travels as it is */
    if (n <= 0) return;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const uint4 *s4 = reinterpret_cast<const uint4 *>(p4 + (size_t)src * p4_stride);       /* p4_stride % 32 == 0 */
    uint4 *d4 = reinterpret_cast<uint4 *>(p4 + (size_t)p * p4_stride);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < p4_stride / 4; i += stride) d4[i] = s4[i];
    const float2 *sp = pal + (size_t)src * DVO_PAL_MAX;
    float2 *dp = pal + (size_t)p * DVO_PAL_MAX;
    const size_t n_copy = (size_t)n + (pal_partial(pn) ? 2 : 1);       /* + the zero sentinel (+ the NaN entry) */
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_copy; i += stride) dp[i] = sp[i];
}
hipError_t launch_replicate_compact(unsigned *p4, size_t p4_stride, float2 *pal, int *pal_n, int n_src, int dst_first,
                                    int dst_count, hipStream_t s) {
    if (dst_count <= 0) return hipSuccess;
    hipLaunchKernelGGL(replicate_compact_kernel, dim3(32, dst_count), dim3(256), 0, s, p4, p4_stride, pal, pal_n, n_src, dst_first);
    return hipGetLastError();
}

/* ------------------------------------------------------------------------- */
/* fused coarse-to-fine alignment: one workgroup per frame pair                */
/* ------------------------------------------------------------------------- */
/* DVO_STAMPS: diagnostic build only (make STAMPS=1 -> libdvo_amd_stamps.so).  Lane 0 of wave 0
 * accumulates s_memtime differences of the four phases of every iteration into out.dbg
 * (a buffer nothing else reads).  Never enabled in the product library. */
#ifdef DVO_STAMPS
DVO_DEV unsigned long long stamp_now() {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
#define DVO_STAMP(var) const unsigned long long var = stamp_now()
#define DVO_STAMP_ADD(slot, a, b) do { if (tid == 0 && out.dbg) out.dbg[(size_t)pair * 64 + l * 8 + (slot)] += (b) - (a); } while (0)
#else
#define DVO_STAMP(var) do {} while (0)
#define DVO_STAMP_ADD(slot, a, b) do {} while (0)
#endif
#ifndef DVO_WAVES_PER_EU
#define DVO_WAVES_PER_EU 1       /* register budget of the fused kernel: 512 / waves VGPRs */
#endif
template <int BLOCK, int U, int INTERP, bool CP, bool WITH_H = false>
__global__ void __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(DVO_WAVES_PER_EU, 8)))
align_fused_kernel(LevelSet lv, Schedule sc, Intrinsics K, DevParams prm, Outputs out, int first_pair) {
    const int pair = first_pair + blockIdx.x;
    const int tid = threadIdx.x;
    __shared__ PoseState st;
    __shared__ double red[BLOCK / 64][DVO_NACC_PAD];
    __shared__ double tot[DVO_NACC_PAD];
    extern __shared__ float lds_points[];     /* planes of sc.lds_points words: this level's points (x|y|z, or xy|z when compact) */

    if (tid == 0) {
        const double *p = out.poses + (size_t)pair * 12;
        const bool ident = (sc.flags & 2) != 0;                  /* DVO_FLAG_IDENTITY_START (:2210-2211) */
        double R0[9], t0[3];
#pragma unroll
        for (int k = 0; k < 9; k++) R0[k] = ident ? ((k % 4 == 0) ? 1.0 : 0.0) : p[k];
#pragma unroll
        for (int k = 0; k < 3; k++) t0[k] = ident ? 0.0 : p[9 + k];
        pose_state_load(st, R0, t0);
        upd_const_build(st.u, prm);
    }
    __syncthreads();

    for (int l = sc.n_levels - 1; l >= 0; --l) {                 /* SolveDVO.cpp:2097 */
        const int iters = sc.iters[l];
        if (iters <= 0) continue;                                 /* :2099 */
        const LevelSlab &L = lv.l[l];
        const int dpair = (sc.alias_mod > 0) ? (pair % sc.alias_mod) : pair;
        const int N = L.N[dpair];
        const float4 *__restrict__ tex = L.tex + (size_t)dpair * L.tex_stride;
        const float *__restrict__ pts = L.pts + (size_t)dpair * L.pt_cap * 3;
        float *energy = out.energy + (size_t)pair * sc.e_stride + sc.e_off[l];

        IterConst c;
        level_consts(c, K, l, L.rows, L.cols);

        for (int i = tid; i < iters; i += BLOCK) energy[i] = 0.0f;          /* :634 */
        if (tid == 0) { pose_state_begin(st); pose_regulariser_precompute(st, st.p[0], st.u); }   /* :642-657 */
        /* The reference re-reads (in fact deep-copies, :670) the 3xN point list every
         * iteration; here the level's points are staged into LDS once and stay there
         * for all its iterations, so HBM sees them once per level. */
        PointSrc psrc;
        psrc.g = pts; psrc.gc = CP ? (L.cpts + (size_t)dpair * L.pt_cap) : nullptr;
        psrc.l = lds_points; psrc.cap = sc.lds_points;
        psrc.n_lds = (N < sc.lds_points) ? N : (sc.lds_points / (BLOCK * U)) * (BLOCK * U);   /* whole rounds only */
        if (CP) {
            /* two points per lane and load: 16 B x 64 lanes = whole 128-byte lines per request (the request rate, not
             * the byte rate, is what this kernel is short of; the list base is 16-byte aligned: pt_cap % 256 == 0) */
            const uint4 *g4 = reinterpret_cast<const uint4 *>(psrc.gc);
            const int n2 = psrc.n_lds >> 1;
            for (int i = tid; i < n2; i += BLOCK) {
                const uint4 v = g4[i];
                lds_points[2 * i] = __uint_as_float(v.x);
                lds_points[psrc.cap + 2 * i] = __uint_as_float(v.y);
                lds_points[2 * i + 1] = __uint_as_float(v.z);
                lds_points[psrc.cap + 2 * i + 1] = __uint_as_float(v.w);
            }
            if ((psrc.n_lds & 1) && tid == 0) {
                const uint2 v = psrc.gc[psrc.n_lds - 1];
                lds_points[psrc.n_lds - 1] = __uint_as_float(v.x);
                lds_points[psrc.cap + psrc.n_lds - 1] = __uint_as_float(v.y);
            }
        } else {
            for (int i = tid; i < psrc.n_lds; i += BLOCK) {
                lds_points[i] = pts[3 * i];
                lds_points[psrc.cap + i] = pts[3 * i + 1];
                lds_points[2 * psrc.cap + i] = pts[3 * i + 2];
            }
        }
        __syncthreads();

        for (int itr = 0; itr < iters; ++itr) {                              /* :658 */
#pragma unroll
            for (int k = 0; k < 9; k++) c.r[k] = uniform_f(st.p[0].Rf[k]);        /* :673 */
#pragma unroll
            for (int k = 0; k < 3; k++) c.t[k] = uniform_f(st.p[0].tf[k]);        /* :674 */

            DVO_STAMP(t0);
#ifdef DVO_YOUNG_WAVE_PRIO
            /* the second-dispatched half of the workgroup loses VALU arbitration to the older half
             * (MI355X_MICROARCH.md, two waves per SIMD) and would finish the loop late */
            if (__builtin_amdgcn_readfirstlane(tid >> 6) >= BLOCK / 128) __builtin_amdgcn_s_setprio(DVO_YOUNG_WAVE_PRIO);
#endif
            Acc a;
            acc_zero(a);
            /* waves take the lanes of a round in reverse order: the tail of the last, partial round goes to the high
             * waves first, so wave 0 -- whose lane 0 still has the regulariser of the new pose to finish (below) -- is the
             * one that most often has a round less */
            const int lane_off = BLOCK - 64 - (tid & ~63) + (tid & 63);
            accumulate_points<U, WITH_H, CP ? SRC_LDS_COMPACT : SRC_LDS_XYZ, INTERP>(c, tex, psrc, 0, psrc.n_lds, lane_off, BLOCK, a);    /* :369, :433 */
            accumulate_points<U, WITH_H, CP ? SRC_GLOBAL_COMPACT : SRC_GLOBAL_XYZ, INTERP>(c, tex, psrc, psrc.n_lds, N, lane_off, BLOCK, a);  /* beyond the LDS budget */
#ifdef DVO_YOUNG_WAVE_PRIO
            __builtin_amdgcn_s_setprio(0);
#endif
            DVO_STAMP(t1);
            block_reduce<BLOCK, WITH_H>(a, red, tot);
            if (WITH_H && tid < 21)     /* DVO_FLAG_NORMAL_MATRIX: H = sum w J J^T of this iterate (the 21 of the "21+6" accumulators) */
                out.H[((size_t)pair * sc.e_stride + sc.e_off[l] + itr) * 21 + tid] = tot[tid];
            DVO_STAMP(t2);
            if (tid == 0) {
                const float e = pose_update_t<true>(st, st.u, itr, N, &tot[21], acc_sum_eps2(tot), (int)tot[28]);
                energy[itr] = e;                                             /* :690 */
            }
            DVO_STAMP(t3);
            __syncthreads();
            DVO_STAMP(t4);
            DVO_STAMP_ADD(0, t0, t1); DVO_STAMP_ADD(1, t1, t2); DVO_STAMP_ADD(2, t2, t3); DVO_STAMP_ADD(3, t3, t4);
            DVO_STAMP_ADD(4, t0, t0 + 1);
            if (st.stop) break;                                              /* :877 */
            /* log(new pose) for the next iteration's regulariser: only the pose is needed, so lane 0 takes it now,
             * while the other waves are already in their point phase */
            if (tid == 0 && itr + 1 < iters) pose_regulariser_precompute(st, st.p[0], st.u);
        }

        /* finalEpsilons / finalReprojections = those of the best iterate (:703-704,
         * :1002-1003); recomputed once from the same float pose -> same bits. */
        if ((sc.flags & 1) && l == sc.last_level) {
            if (st.bestItr >= 0) {
#pragma unroll
                for (int k = 0; k < 9; k++) c.r[k] = uniform_f(st.bRf[k]);
#pragma unroll
                for (int k = 0; k < 3; k++) c.t[k] = uniform_f(st.btf[k]);
                float *fe = out.final_eps + (size_t)pair * out.final_cap;
                float *fr = out.final_reproj + (size_t)pair * out.final_cap * 3;
                for (int i = tid; i < N; i += BLOCK) {
                    float X, Y, Z, xn, yn, zn, u, v;
                    /* indexed like the reference's list: with compact points (block order, dvo_frames.hip) read the 3 x N list */
                    if (!CP && i < psrc.n_lds) load_point<SRC_LDS_XYZ>(c, psrc, i, X, Y, Z);
                    else load_point<SRC_GLOBAL_XYZ>(c, psrc, i, X, Y, Z);
                    const bool vis = project_point(c, X, Y, Z, xn, yn, zn, u, v);
                    float e = 0.0f;
                    if (vis) e = (INTERP == 1) ? interpolate_dt(c, tex, v, u) : tex[texel_index((int)v, (int)u, c.tiles_per_col)].x;
                    fe[i] = e;
                    fr[3 * i] = u; fr[3 * i + 1] = v; fr[3 * i + 2] = zn;
                }
            }
            if (tid == 0) out.final_N[pair] = (st.bestItr >= 0) ? N : 0;
        }
        __syncthreads();
        if (tid == 0) {                                                      /* :997-1005 */
            pose_state_finish(st);
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
        for (int k = 0; k < 3; k++) p[9 + k] = st.p[0].t[k];
    }
}

template <int BLOCK, int U, bool CP>
static hipError_t launch_fused_bu(const LevelSet &lv, const Schedule &sc, const Intrinsics &K,
                                  const DevParams &prm, const Outputs &out, int first_pair, int n_pairs, hipStream_t s) {
    const size_t dyn = (size_t)sc.lds_points * (CP ? 2 : 3) * sizeof(float);
    auto kern = align_fused_kernel<BLOCK, U, 0, CP>;
    const hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(n_pairs), dim3(BLOCK), dyn, s, lv, sc, K, prm, out, first_pair);
    return hipGetLastError();
}
template <int BLOCK>
static hipError_t launch_fused_b(int u, const LevelSet &lv, const Schedule &sc, const Intrinsics &K,
                                 const DevParams &prm, const Outputs &out, int first_pair, int n_pairs, hipStream_t s) {
    if (u >= 4) return launch_fused_bu<BLOCK, 4, false>(lv, sc, K, prm, out, first_pair, n_pairs, s);
    if (u == 2) return launch_fused_bu<BLOCK, 2, false>(lv, sc, K, prm, out, first_pair, n_pairs, s);
    if (sc.compact) return launch_fused_bu<BLOCK, 1, true>(lv, sc, K, prm, out, first_pair, n_pairs, s);
    return launch_fused_bu<BLOCK, 1, false>(lv, sc, K, prm, out, first_pair, n_pairs, s);
}

/* true if this launch configuration reads the compact point lists (the host sizes sc.lds_points accordingly) */
bool fused_uses_compact(int points_in_flight, int interp) { return points_in_flight <= 1 && !interp; }

hipError_t launch_align_fused(int block_threads, int points_in_flight, const LevelSet &lv,
                              const Schedule &sc, const Intrinsics &K, const DevParams &prm,
                              const Outputs &out, int first_pair, int n_pairs, hipStream_t s) {
    if (n_pairs <= 0) return hipSuccess;
    if ((sc.flags & 4) && !K.interp) {     /* DVO_FLAG_NORMAL_MATRIX: the 21 H sums ride along (one configuration, 512 threads) */
        const size_t dyn = (size_t)sc.lds_points * (sc.compact ? 2 : 3) * sizeof(float);
        hipError_t e;
        if (sc.compact) {
            auto kern = align_fused_kernel<512, 1, 0, true, true>;
            if ((e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn)) != hipSuccess) return e;
            hipLaunchKernelGGL(kern, dim3(n_pairs), dim3(512), dyn, s, lv, sc, K, prm, out, first_pair);
        } else {
            auto kern = align_fused_kernel<512, 1, 0, false, true>;
            if ((e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn)) != hipSuccess) return e;
            hipLaunchKernelGGL(kern, dim3(n_pairs), dim3(512), dyn, s, lv, sc, K, prm, out, first_pair);
        }
        return hipGetLastError();
    }
    if (K.interp) {     /* optional interpolate() lookup: one configuration only, it is not the tuned path */
        const size_t dyn = (size_t)sc.lds_points * 3 * sizeof(float);
        auto kern = align_fused_kernel<512, 1, 1, false>;
        const hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, dim3(n_pairs), dim3(512), dyn, s, lv, sc, K, prm, out, first_pair);
        return hipGetLastError();
    }
    switch (block_threads) {
    case 256: return launch_fused_b<256>(points_in_flight, lv, sc, K, prm, out, first_pair, n_pairs, s);
    case 1024: return launch_fused_b<1024>(points_in_flight, lv, sc, K, prm, out, first_pair, n_pairs, s);
    default: return launch_fused_b<512>(points_in_flight, lv, sc, K, prm, out, first_pair, n_pairs, s);
    }
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
    /* contiguous chunk per block (each block = a contiguous index range of the point list) */
    const int per = (n + gridDim.x - 1) / gridDim.x;
    const int b0 = first + blockIdx.x * per;
    int b1 = b0 + per;
    if (b1 > first + n) b1 = first + n;
    PointSrc psrc;
    psrc.g = pts; psrc.gc = nullptr; psrc.l = nullptr; psrc.n_lds = 0; psrc.cap = 0;
    if (b0 < b1) accumulate_points<4, true, SRC_GLOBAL_XYZ, 2>(c, tex, psrc, b0, b1, threadIdx.x, 256, a);
    block_reduce<256, true>(a, red, tot);
    if (threadIdx.x < DVO_NACC_PAD)
        partials[(size_t)blockIdx.x * DVO_NACC_PAD + threadIdx.x] = tot[threadIdx.x];
}

/* acc[k] = sum over blocks of partials[b][k], in a fixed two-level order (32 interleaved chains per value,
 * then the chains in order) => reproducible, and 32x shorter than one serial chain */
__global__ void __launch_bounds__(1024)
reduce_partials_kernel(const double *__restrict__ partials, int nblocks, double *__restrict__ acc) {
    __shared__ double part[32][DVO_NACC_PAD + 1];
    const int k = threadIdx.x & 31, c = threadIdx.x >> 5;
    double s = 0.0;
    for (int b = c; b < nblocks; b += 32) s += partials[(size_t)b * DVO_NACC_PAD + k];
    part[c][k] = s;
    __syncthreads();
    if (threadIdx.x < DVO_NACC_PAD) {
        double t = 0.0;
#pragma unroll
        for (int j = 0; j < 32; j++) t += part[j][threadIdx.x];
        acc[threadIdx.x] = t;
    }
}

int accumulate_blocks_for(int n_points) {
    int b = (n_points + 255) / 256;
    if (b < 1) b = 1;
    if (b > 512) b = 512;             /* two 256-thread workgroups per CU */
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
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(1), dim3(1024), 0, s, partials, nblocks, acc);
    return hipGetLastError();
}

/* ------------------------------------------------------------------------- */
/* host-driven iteration (large single frames, multi-GPU tiled mode):          */
/* the optimiser state lives in HBM, one PoseState per pair                    */
/* ------------------------------------------------------------------------- */
__global__ void __launch_bounds__(64)
iter_begin_kernel(PoseState *st, DevParams prm, const double *Rt12, float *energy, int max_iters) {
    if (threadIdx.x == 1) upd_const_build(st->u, prm);
    if (threadIdx.x == 0) {
        double R[9], t[3];
        for (int k = 0; k < 9; k++) R[k] = Rt12[k];
        for (int k = 0; k < 3; k++) t[k] = Rt12[9 + k];
        pose_state_load(*st, R, t);
        pose_state_begin(*st);
    }
    for (int i = threadIdx.x; i < max_iters; i += 64) energy[i] = 0.0f;      /* :634 */
    __syncthreads();
    if (threadIdx.x == 0) pose_regulariser_precompute(*st, st->p[0], st->u); /* log(pose) for the first update (tiled_step_kernel) */
}

/* same as accumulate_kernel, but the float pose is read from the device-resident state */
__global__ void __launch_bounds__(256)
accumulate_state_kernel(LevelSlab L, int pair, int level, Intrinsics K, const PoseState *st,
                        int first, int n, double *__restrict__ partials) {
    __shared__ double red[256 / 64][DVO_NACC_PAD];
    __shared__ double tot[DVO_NACC_PAD];
    const float4 *__restrict__ tex = L.tex + (size_t)pair * L.tex_stride;
    const float *__restrict__ pts = L.pts + (size_t)pair * L.pt_cap * 3;
    IterConst c;
    level_consts(c, K, level, L.rows, L.cols);
#pragma unroll
    for (int k = 0; k < 9; k++) c.r[k] = uniform_f(st->p[0].Rf[k]);
#pragma unroll
    for (int k = 0; k < 3; k++) c.t[k] = uniform_f(st->p[0].tf[k]);
    Acc a;
    acc_zero(a);
    const int per = (n + gridDim.x - 1) / gridDim.x;
    const int b0 = first + blockIdx.x * per;
    int b1 = b0 + per;
    if (b1 > first + n) b1 = first + n;
    PointSrc psrc;
    psrc.g = pts; psrc.gc = nullptr; psrc.l = nullptr; psrc.n_lds = 0; psrc.cap = 0;
    if (b0 < b1) accumulate_points<2, true, SRC_GLOBAL_XYZ, 2>(c, tex, psrc, b0, b1, threadIdx.x, 256, a);
    block_reduce<256, true>(a, red, tot);
    if (threadIdx.x < DVO_NACC_PAD)
        partials[(size_t)blockIdx.x * DVO_NACC_PAD + threadIdx.x] = tot[threadIdx.x];
}

__global__ void __launch_bounds__(64)
iter_update_kernel(PoseState *st, DevParams prm, int itr, int n_total, const double *__restrict__ acc,
                   float *energy) {
    if (threadIdx.x == 0 && !st->stop) {                                    /* after :877 nothing runs */
        double g[6];
        for (int k = 0; k < 6; k++) g[k] = acc[21 + k];
        energy[itr] = pose_update(*st, st->u, itr, n_total, g, acc_sum_eps2(acc), (int)acc[28]);
    }
}

/* single-GPU form: partial sums -> totals -> update in one launch (no collective in between) */
__global__ void __launch_bounds__(1024)
iter_reduce_update_kernel(PoseState *st, DevParams prm, int itr, int n_total,
                          const double *__restrict__ partials, int nblocks, float *energy) {
    __shared__ double part[32][DVO_NACC_PAD + 1];
    __shared__ double acc[DVO_NACC_PAD];
    const int k = threadIdx.x & 31, c = threadIdx.x >> 5;
    double s = 0.0;
    for (int b = c; b < nblocks; b += 32) s += partials[(size_t)b * DVO_NACC_PAD + k];
    part[c][k] = s;
    __syncthreads();
    if (threadIdx.x < DVO_NACC_PAD) {
        double t = 0.0;
#pragma unroll
        for (int j = 0; j < 32; j++) t += part[j][threadIdx.x];
        acc[threadIdx.x] = t;
    }
    __syncthreads();
    if (threadIdx.x == 0 && !st->stop) {
        double g[6];
        for (int q = 0; q < 6; q++) g[q] = acc[21 + q];
        energy[itr] = pose_update(*st, st->u, itr, n_total, g, acc_sum_eps2(acc), (int)acc[28]);
    }
}

/* ---- one launch per iteration of the tiled / wide schedule (round 4) ---------------------------------------------------
 * Rounds 1-3 ran an iteration as accumulate -> reduce partials -> [all-reduce] -> update: three dependent launches, each ~1.5 us
 * of boundary on top of its work (MI355X_MICROARCH.md, price list: boundary).  tiled_step_kernel is the whole iteration but the
 * collective:
 *   head   EVERY workgroup applies the pending update of the previous iteration -- the reduced (all-reduced) sums `acc_in` and
 *          the state st_in are the same bits for every workgroup (and, in a tiled run, for every rank), the update is a pure
 *          function of them, so every workgroup derives the same float pose; workgroup 0 also writes the new state to st_out
 *          (double-buffered: no workgroup of this launch reads what it writes) and the energy of the iterate (:690);
 *   body   the workgroup's share of this rank's point range at that pose: 29 sums (:714-720, :777);
 *   tail   the 32 doubles of every workgroup go to `partials` with write-through (sc1) stores, a ticket counts the arrivals, and
 *          the workgroup whose ticket comes last adds all rows in a fixed order (that of reduce_partials_kernel) and writes
 *          `acc_out`: the buffer the all-reduce works on, or directly the next launch's acc_in on one GPU.
 * Hand-off form: MI355X_MICROARCH.md, Workgroup dispatch ..., hand-offs measured with sc1 loads, first row -- one lane of each
 * storing workgroup adds to one unsharded counter after that workgroup's stores have drained (every storing wave's vmcnt(0),
 * then the workgroup barrier), the last arriver is told by the value its add returned, its other waves load after a barrier it
 * then joins; hipMalloc memory, one workgroup per CU (the host launches at most that many), all stores and loads of the handed-
 * off rows 8-byte sc1.  */
#ifndef DVO_STEP_U
#define DVO_STEP_U 1          /* points in flight per lane (measured 4096x3072x5: 1: 12.8 us per iteration, 2: 13.4, 4: 14.7 -- the launch is
                                 dominated by its fixed part, the smallest loop wins) */
#endif
template <bool WITH_H>
__global__ void __launch_bounds__(DVO_STEP_THREADS)
tiled_step_kernel(LevelSlab L, int pair, int level, Intrinsics K, DevParams prm, const PoseState *st_in, PoseState *st_out,
                  const double *__restrict__ acc_in, int itr, int apply_prev, int n_total, int first, int n,
                  double *partials, unsigned *ticket, double *acc_out, float *energy, double *H_prev) {
    __shared__ double red[DVO_STEP_THREADS / 64][DVO_NACC_PAD];
    __shared__ TiledStepLds m;
    tiled_step_body<WITH_H>(m, st_in, st_out, acc_in, itr, apply_prev, n_total, first, n, partials, ticket, acc_out, energy, H_prev,
        [&](const PoseCur &pc, bool run, int b0, int b1, double *tot) {
            Acc a;
            acc_zero(a);
            if (run) {
                const float4 *__restrict__ tex = L.tex + (size_t)pair * L.tex_stride;
                const float *__restrict__ pts = L.pts + (size_t)pair * L.pt_cap * 3;
                IterConst c;
                level_consts(c, K, level, L.rows, L.cols);
#pragma unroll
                for (int k = 0; k < 9; k++) c.r[k] = uniform_f(pc.Rf[k]);
#pragma unroll
                for (int k = 0; k < 3; k++) c.t[k] = uniform_f(pc.tf[k]);
                PointSrc psrc;
                psrc.g = pts; psrc.gc = nullptr; psrc.l = nullptr; psrc.n_lds = 0; psrc.cap = 0;
                /* (round 5, measured: requesting every lane's first six points before the head -- they do not depend on the pose -- and
                 * issuing their gathers together made every launch 3 us SLOWER, 4096 x 3072 included: the head then waits behind eighteen
                 * cold loads, and waves without points work through dummy rounds.  The per-point phase of this kernel is issue-bound:
                 * tiled_step_pk_kernel, dvo_fused.hip, is the answer to that.) */
                accumulate_points<DVO_STEP_U, WITH_H, SRC_GLOBAL_XYZ, 2>(c, tex, psrc, b0, b1, (int)threadIdx.x, DVO_STEP_THREADS, a);
            }
            block_reduce<DVO_STEP_THREADS, WITH_H>(a, red, tot);
        });
}
/* after the last iteration of a level: the pending update, then what iter_end_kernel does */
__global__ void __launch_bounds__(64)
tiled_finish_kernel(const PoseState *st_in, PoseState *st_out, DevParams prm, const double *__restrict__ acc_in, int itr_last,
                    int n_total, float *energy, double *Rt12, int *best_idx, float *ratio, double *H_last, float *next_energy, int next_iters) {
    __shared__ PoseState s;
    {
        const unsigned long long *src = reinterpret_cast<const unsigned long long *>(st_in);
        unsigned long long *dst = reinterpret_cast<unsigned long long *>(&s);
        for (int i = threadIdx.x; i < (int)(sizeof(PoseState) / 8); i += 64) dst[i] = src[i];
    }
    __syncthreads();
    if (H_last && threadIdx.x < 21 && !s.stop) H_last[threadIdx.x] = acc_in[threadIdx.x];
    __syncthreads();
    if (threadIdx.x == 0) {
        if (!s.stop) {
            double g[6];
            for (int k = 0; k < 6; k++) g[k] = acc_in[21 + k];
            energy[itr_last] = pose_update_t<true>(s, s.u, itr_last, n_total, g, acc_sum_eps2(acc_in), (int)acc_in[28]);
        }
        pose_state_finish(s);                                               /* :997-1001 */
        for (int k = 0; k < 9; k++) Rt12[k] = s.R[k];
        for (int k = 0; k < 3; k++) Rt12[9 + k] = s.p[0].t[k];
        *best_idx = s.bestItr;
        *ratio = s.bestRatio;
        /* round 5: the next level's iter_begin_kernel rides here (one launch less per level).  The state carries the pose as the
         * fused kernels do between levels -- the quaternion of the best iterate and its matrix (pose_state_finish) -- instead of
         * re-deriving the quaternion from the matrix in Rt12 (:642-657 start from cR / cT either way) */
        if (next_iters > 0) { pose_state_begin(s); pose_regulariser_precompute(s, s.p[0], s.u); }
    }
    for (int i = threadIdx.x; i < next_iters; i += 64) next_energy[i] = 0.0f;      /* :634 */
    __syncthreads();
    {
        const unsigned long long *src = reinterpret_cast<const unsigned long long *>(&s);
        unsigned long long *dst = reinterpret_cast<unsigned long long *>(st_out);
        for (int i = threadIdx.x; i < (int)(sizeof(PoseState) / 8); i += 64) dst[i] = src[i];
    }
}
int tiled_step_blocks(int n_points, int n_cu) {
    int b = (n_points + DVO_STEP_THREADS - 1) / DVO_STEP_THREADS;
    if (b < 1) b = 1;
    if (b > 1) b += 1;                  /* workgroup 0 keeps the state and takes no points once there are several (tiled_step_kernel) */
    if (b > n_cu) b = n_cu;             /* one workgroup per CU: the hand-off form of the tail is measured for that */
    /* the partials buffer (dvo_ctx.h: d_scratch) holds 1024 x DVO_NACC_PAD doubles before the sums: a workgroup's tagged rows are
     * 16-byte records, 32 of them with DVO_FLAG_NORMAL_MATRIX -- 512 workgroups' worth (the old bound of 1024 was that of
     * the 8-byte rows of round 4) */
    static_assert(1024 * DVO_NACC_PAD * sizeof(double) >= 512 * 32 * 16, "tagged rows of 512 workgroups fit the partials buffer");
    if (b > 512) b = 512;
    return b;
}
hipError_t launch_tiled_step(const LevelSlab &L, int pair, int level, const Intrinsics &K, const DevParams &prm, const void *st_in,
                             void *st_out, const double *acc_in, int itr, int apply_prev, int n_total, int first_point, int n_points,
                             double *partials, unsigned *ticket, double *acc_out, float *energy, int nblocks, double *H_prev, hipStream_t s) {
    if (H_prev)
        hipLaunchKernelGGL(tiled_step_kernel<true>, dim3(nblocks), dim3(DVO_STEP_THREADS), 0, s, L, pair, level, K, prm, (const PoseState *)st_in,
                           (PoseState *)st_out, acc_in, itr, apply_prev, n_total, first_point, n_points, partials, ticket, acc_out, energy, H_prev);
    else
        hipLaunchKernelGGL(tiled_step_kernel<false>, dim3(nblocks), dim3(DVO_STEP_THREADS), 0, s, L, pair, level, K, prm, (const PoseState *)st_in,
                           (PoseState *)st_out, acc_in, itr, apply_prev, n_total, first_point, n_points, partials, ticket, acc_out, energy, H_prev);
    return hipGetLastError();
}
hipError_t launch_tiled_finish(const void *st_in, void *st_out, const DevParams &prm, const double *acc_in, int itr_last, int n_total,
                               float *energy, double *Rt12, int *best_idx, float *ratio, double *H_last, float *next_energy, int next_iters,
                               hipStream_t s) {
    hipLaunchKernelGGL(tiled_finish_kernel, dim3(1), dim3(64), 0, s, (const PoseState *)st_in, (PoseState *)st_out, prm, acc_in, itr_last,
                       n_total, energy, Rt12, best_idx, ratio, H_last, next_energy, next_iters);
    return hipGetLastError();
}

__global__ void __launch_bounds__(64)
iter_end_kernel(PoseState *st, double *Rt12, int *best_idx, float *ratio) {
    if (threadIdx.x == 0) {
        pose_state_finish(*st);                                             /* :997-1001 */
        for (int k = 0; k < 9; k++) Rt12[k] = st->R[k];
        for (int k = 0; k < 3; k++) Rt12[9 + k] = st->p[0].t[k];
        *best_idx = st->bestItr;
        *ratio = st->bestRatio;
    }
}

/* finalEpsilons / finalReprojections (:703-704, :1002-1003) of points [first, first+n) from the best iterate's float pose kept in
 * the state -- the host-driven / tiled twin of the fused kernels' final pass (every rank of a tiled run produces its own shard,
 * at the points' indices in the whole list).  Same float pose, same per-point code -> same bits as the fused kernels. */
__global__ void __launch_bounds__(256)
final_outputs_state_kernel(LevelSlab L, int pair, int level, Intrinsics K, const PoseState *st, int first, int n,
                           float *__restrict__ fe, float *__restrict__ fr, int *__restrict__ final_N) {
    const int Nall = L.N[pair];
    const bool have = st->bestItr >= 0;
    if (blockIdx.x == 0 && threadIdx.x == 0) *final_N = have ? Nall : 0;
    if (!have) return;
    const float4 *__restrict__ tex = L.tex + (size_t)pair * L.tex_stride;
    const float *__restrict__ pts = L.pts + (size_t)pair * L.pt_cap * 3;
    IterConst c;
    level_consts(c, K, level, L.rows, L.cols);
#pragma unroll
    for (int k = 0; k < 9; k++) c.r[k] = uniform_f(st->bRf[k]);
#pragma unroll
    for (int k = 0; k < 3; k++) c.t[k] = uniform_f(st->btf[k]);
    for (int i = first + blockIdx.x * blockDim.x + threadIdx.x; i < first + n; i += gridDim.x * blockDim.x) {
        float xn, yn, zn, u, v;
        const bool vis = project_point(c, pts[3 * i], pts[3 * i + 1], pts[3 * i + 2], xn, yn, zn, u, v);
        float e = 0.0f;
        if (vis) e = c.interp ? interpolate_dt(c, tex, v, u) : tex[texel_index((int)v, (int)u, c.tiles_per_col)].x;
        fe[i] = e;
        fr[3 * i] = u; fr[3 * i + 1] = v; fr[3 * i + 2] = zn;
    }
}
hipError_t launch_final_outputs_state(const LevelSlab &L, int pair, int level, const Intrinsics &K, const void *state,
                                      int first_point, int n_points, float *final_eps, float *final_reproj, int *final_N,
                                      hipStream_t s) {
    const int nb = accumulate_blocks_for(n_points);
    hipLaunchKernelGGL(final_outputs_state_kernel, dim3(nb), dim3(256), 0, s, L, pair, level, K, (const PoseState *)state,
                       first_point, n_points, final_eps, final_reproj, final_N);
    return hipGetLastError();
}

hipError_t launch_iter_begin(void *state, const DevParams &prm, const double *Rt12, float *energy, int max_iters, hipStream_t s) {
    hipLaunchKernelGGL(iter_begin_kernel, dim3(1), dim3(64), 0, s, (PoseState *)state, prm, Rt12, energy, max_iters);
    return hipGetLastError();
}
hipError_t launch_iter_accumulate(const LevelSlab &L, int pair, int level, const Intrinsics &K, const void *state,
                                  int first_point, int n_points, double *partials, int nblocks, double *acc,
                                  hipStream_t s) {
    hipLaunchKernelGGL(accumulate_state_kernel, dim3(nblocks), dim3(256), 0, s, L, pair, level, K,
                       (const PoseState *)state, first_point, n_points, partials);
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(1), dim3(1024), 0, s, partials, nblocks, acc);
    return hipGetLastError();
}
hipError_t launch_iter_update(void *state, const DevParams &prm, int itr, int n_total, const double *acc,
                              float *energy, hipStream_t s) {
    hipLaunchKernelGGL(iter_update_kernel, dim3(1), dim3(64), 0, s, (PoseState *)state, prm, itr, n_total, acc, energy);
    return hipGetLastError();
}
hipError_t launch_iter_step_fused(const LevelSlab &L, int pair, int level, const Intrinsics &K, void *state,
                                  const DevParams &prm, int itr, int n_points, double *partials, int nblocks,
                                  float *energy, hipStream_t s) {
    hipLaunchKernelGGL(accumulate_state_kernel, dim3(nblocks), dim3(256), 0, s, L, pair, level, K,
                       (const PoseState *)state, 0, n_points, partials);
    hipLaunchKernelGGL(iter_reduce_update_kernel, dim3(1), dim3(1024), 0, s, (PoseState *)state, prm, itr, n_points,
                       partials, nblocks, energy);
    return hipGetLastError();
}
hipError_t launch_iter_end(void *state, double *Rt12, int *best_idx, float *ratio, hipStream_t s) {
    hipLaunchKernelGGL(iter_end_kernel, dim3(1), dim3(64), 0, s, (PoseState *)state, Rt12, best_idx, ratio);
    return hipGetLastError();
}
size_t pose_state_bytes() { return sizeof(PoseState); }

/* texels -> the three planar images (inspection / tests) */
__global__ void __launch_bounds__(256)
unpack_texels_kernel(const float4 *__restrict__ tex, int rows, int cols, float *__restrict__ dt,
                     float *__restrict__ gx, float *__restrict__ gy) {
    const size_t n = (size_t)rows * cols;
    const int tpc = texel_tiles_per_col(rows);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int xx = (int)(i / rows), yy = (int)(i - (size_t)xx * rows);
        const float4 t = tex[texel_index(yy, xx, tpc)];
        dt[i] = t.x; gx[i] = t.y; gy[i] = t.z;
    }
}

hipError_t launch_unpack_texels(const float4 *tex, int rows, int cols, float *dt, float *gx, float *gy, hipStream_t s) {
    const size_t n = (size_t)rows * cols;
    size_t blocks = (n + 255) / 256; if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(unpack_texels_kernel, dim3((unsigned)blocks), dim3(256), 0, s, tex, rows, cols, dt, gx, gy);
    return hipGetLastError();
}

/* one wave asleep for `us` microseconds of real time (the constant 100 MHz counter: the duration does not stretch when the
 * shader clock is low): the activity that keeps the GPU's power management from parking the clocks between the frames of a
 * single camera stream (dvo_set_keep_warm).  Bounded: a launch always ends by itself. */
__global__ void __launch_bounds__(64) keep_warm_kernel(unsigned ticks) {
    unsigned long long t0, t;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    do {
        __builtin_amdgcn_s_sleep(64);
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    } while (t - t0 < (unsigned long long)ticks);
}
hipError_t launch_keep_warm(int us, hipStream_t s) {
    hipLaunchKernelGGL(keep_warm_kernel, dim3(1), dim3(64), 0, s, (unsigned)(us < 1 ? 1 : (us > 20000 ? 20000 : us)) * 100u);
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

}  // namespace dvo
