/*
 * dvo_palette.hip -- builder of the compact ("P4") form of a now level (layout and rationale: dvo_palette.h).
 *
 * Generic builder: works from the 16-byte texels {DT, gx, gy, w} of a level, whoever produced them (the engine's own
 * distance-transform kernels or a caller's float images, reference SolveDVO.cpp:1768-1795, :1063-1098, :1047-1053), and
 * VERIFIES per pixel that the compact form decodes to exactly those four floats.  One 1024-thread workgroup per image:
 *   1. distinct DT bit patterns -> LDS hash set (< DVO_PAL_MAX, else "no compact form")
 *   2. compaction + bitonic sort  -> the palette P[0..n) (non-negative floats order like their bit patterns)
 *   3. per pixel: rank of its DT (binary search) written to its interior slot and to the apron slots that stand for it
 *      (row above / below of the neighbouring tiles, reflect-101 rows at the image border, cv::filter2D's default border)
 *   4. per pixel: ranks of the four neighbours, the two horizontal rank steps packed into the dword, and the check
 *      P[c] == DT, W[c] == w, 0.5*(P[r]-P[l]) == gx, 0.5*(P[d]-P[u]) == gy  bit for bit
 * Not a hot path: runs once per now level that is aligned more than once (or on request, dvo_now_prepare).
 *
 * Compile with -ffp-contract=off (the gradient formula must stay a subtraction followed by a multiplication).
 */
#include "dvo_launch.h"
#include "dvo_palette.h"

namespace dvo {

#define PAL_HASH 8192u
#define PAL_EMPTY 0xffffffffu

/* reasons for "no compact form" (pal_n = -reason) */
enum { PAL_BAD_VALUE = 1, PAL_TOO_MANY = 2, PAL_STEP = 3, PAL_GRADIENT = 4, PAL_WEIGHT = 5, PAL_SHAPE = 6 };

DVO_DEV int pal_rank(const unsigned *sorted, int n, unsigned key) {
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (sorted[mid] < key) lo = mid + 1; else hi = mid;
    }
    return lo;
}

__global__ void __launch_bounds__(1024)
palette_build_kernel(const float4 *__restrict__ tex, size_t tex_stride, int rows, int cols, unsigned *p4, size_t p4_stride,
                     float2 *__restrict__ pal, int *__restrict__ pal_n, int first_pair) {
    const int pair = first_pair + blockIdx.x;
    const int tid = threadIdx.x;
    tex += (size_t)pair * tex_stride;
    p4 += (size_t)pair * p4_stride;
    pal += (size_t)pair * DVO_PAL_MAX;
    __shared__ unsigned keys[PAL_HASH];
    __shared__ unsigned sorted[DVO_PAL_MAX];
    __shared__ unsigned wts[DVO_PAL_MAX];
    __shared__ int cnt, bad;
    const int tpc16 = texel_tiles_per_col(rows);
    const int tpc = p4_tiles_per_col(rows);
    const int npx = rows * cols;

    for (unsigned i = tid; i < PAL_HASH; i += 1024) keys[i] = PAL_EMPTY;
    if (tid == 0) { cnt = 0; bad = (rows < 2 || cols < 2) ? PAL_SHAPE : 0; }
    for (size_t i = tid; i < p4_stride; i += 1024) p4[i] = 0u;          /* slots outside the image: rank 0 */
    __syncthreads();

    /* 1. the set of distinct DT values */
    if (!bad) {
        for (int p = tid; p < npx; p += 1024) {
            const int xx = p / rows, yy = p - xx * rows;
            const unsigned key = __float_as_uint(tex[texel_index(yy, xx, tpc16)].x);
            if (key >= 0x7f800000u) { bad = PAL_BAD_VALUE; break; }        /* negative, inf or nan: not a distance */
            unsigned h = (key * 2654435761u) >> 19;
            for (;;) {
                const unsigned cur = *(volatile unsigned *)&keys[h];
                if (cur == key) break;
                if (cur == PAL_EMPTY) {
                    const unsigned old = atomicCAS(&keys[h], PAL_EMPTY, key);
                    if (old == PAL_EMPTY) { if (atomicAdd(&cnt, 1) >= DVO_PAL_MAX - 1) bad = PAL_TOO_MANY; break; }     /* one entry is the sentinel */
                    if (old == key) break;
                }
                if (*(volatile int *)&bad) break;
                h = (h + 1u) & (PAL_HASH - 1u);
            }
            if (*(volatile int *)&bad) break;
        }
    }
    __syncthreads();
    if (bad) { if (tid == 0) pal_n[pair] = -bad; return; }
    const int n = cnt;
    int m = 2;
    while (m < n) m <<= 1;

    /* 2. compaction (any order) + bitonic sort */
    if (tid == 0) cnt = 0;
    __syncthreads();
    for (unsigned i = tid; i < PAL_HASH; i += 1024) {
        const unsigned k = keys[i];
        if (k != PAL_EMPTY) sorted[atomicAdd(&cnt, 1)] = k;
    }
    for (int i = n + tid; i < m; i += 1024) sorted[i] = PAL_EMPTY;
    __syncthreads();
    for (int k = 2; k <= m; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < m; i += 1024) {
                const int q = i ^ j;
                if (q > i) {
                    const unsigned a = sorted[i], b = sorted[q];
                    const bool up = (i & k) == 0;
                    if ((a > b) == up) { sorted[i] = b; sorted[q] = a; }
                }
            }
            __syncthreads();
        }
    }

    /* the sentinel: line 0 points at entry n = {0, 0} */
    if (tid < 32) p4[tid] = (unsigned)n << 3;
    /* 3. ranks into the interior slot and into every apron slot that stands for this pixel */
    for (int p = tid; p < npx; p += 1024) {
        const int xx = p / rows, yy = p - xx * rows;
        const float4 t = tex[texel_index(yy, xx, tpc16)];
        const int c = pal_rank(sorted, n, __float_as_uint(t.x));
        wts[c] = __float_as_uint(t.w);                                /* the same for every pixel of this rank -- checked in 4 */
        const unsigned v = (unsigned)c << 3;
        const int ty = yy / DVO_P4_ROWS, ry = yy - ty * DVO_P4_ROWS;
        p4[p4_slot(ty, ry + 1, xx, tpc)] = v;
        if (ry == 0 && ty > 0) p4[p4_slot(ty - 1, 7, xx, tpc)] = v;                         /* row below the tile above */
        if (ry == DVO_P4_ROWS - 1 && yy + 1 < rows) p4[p4_slot(ty + 1, 0, xx, tpc)] = v;    /* row above the tile below */
        if (yy == 1) p4[p4_slot(0, 0, xx, tpc)] = v;                                         /* reflect-101: row -1 = row 1 */
        if (yy == rows - 2) {                                                                /* row `rows` = row rows-2 */
            const int tl = (rows - 1) / DVO_P4_ROWS;
            p4[p4_slot(tl, (rows - 1) - tl * DVO_P4_ROWS + 2, xx, tpc)] = v;
        }
    }
    __syncthreads();

    /* 4. neighbour ranks, horizontal rank steps, verification against the 16-byte texel */
    for (int p = tid; p < npx; p += 1024) {
        const int xx = p / rows, yy = p - xx * rows;
        const float4 t = tex[texel_index(yy, xx, tpc16)];
        const int ty = yy / DVO_P4_ROWS, ry = yy - ty * DVO_P4_ROWS;
        const int xr = (xx + 1 < cols) ? xx + 1 : cols - 2, xl = (xx > 0) ? xx - 1 : 1;      /* reflect-101 */
        const size_t own = p4_slot(ty, ry + 1, xx, tpc);
        const int c = (int)((p4[own] >> 3) & 0x1fffu);
        const int ru = (int)((p4[own - 1] >> 3) & 0x1fffu), rd = (int)((p4[own + 1] >> 3) & 0x1fffu);
        const int rr = (int)((p4[p4_slot(ty, ry + 1, xr, tpc)] >> 3) & 0x1fffu);
        const int rl = (int)((p4[p4_slot(ty, ry + 1, xl, tpc)] >> 3) & 0x1fffu);
        const int dr = rr - c, dl = rl - c;
        if (dr < -127 || dr > 127 || dl < -127 || dl > 127) { bad = PAL_STEP; break; }
        const float gx = 0.5f * (__uint_as_float(sorted[rr]) - __uint_as_float(sorted[rl]));
        const float gy = 0.5f * (__uint_as_float(sorted[rd]) - __uint_as_float(sorted[ru]));
        if (__float_as_uint(gx) != __float_as_uint(t.y) || __float_as_uint(gy) != __float_as_uint(t.z)) { bad = PAL_GRADIENT; break; }
        if (wts[c] != __float_as_uint(t.w)) { bad = PAL_WEIGHT; break; }
        p4[own] = ((unsigned)c << 3) | (((unsigned)dr & 0xffu) << 16) | (((unsigned)dl & 0xffu) << 24);
    }
    __syncthreads();
    for (int k = tid; k < n; k += 1024) pal[k] = make_float2(__uint_as_float(sorted[k]), __uint_as_float(wts[k]));
    if (tid == 0) pal[n] = make_float2(0.0f, 0.0f);
    if (tid == 0) pal_n[pair] = bad ? -bad : n;
}

hipError_t launch_palette_build(const float4 *tex, size_t tex_stride, int rows, int cols, unsigned *p4, size_t p4_stride,
                                float2 *pal, int *pal_n, int first_pair, int count, hipStream_t s) {
    if (count <= 0) return hipSuccess;
    hipLaunchKernelGGL(palette_build_kernel, dim3(count), dim3(1024), 0, s, tex, tex_stride, rows, cols, p4, p4_stride, pal, pal_n,
                       first_pair);
    return hipGetLastError();
}

}  // namespace dvo
