/*
 * dvo_palette.hip -- builder of the compact ("P4") form of a now level (layout and rationale: dvo_palette.h).
 *
 * Generic builder: works from the 16-byte texels {DT, gx, gy, w} of a level, whoever produced them (the engine's own
 * distance-transform kernels or a caller's float images, reference SolveDVO.cpp:1768-1795, :1063-1098, :1047-1053), and
 * VERIFIES per pixel that the compact form decodes to exactly those four floats.  Three launches over a batch of images:
 *   1. collect  (several workgroups per image, a range of pixel columns each): distinct DT bit patterns -> LDS hash set ->
 *               merged into the image's hash set in HBM (< DVO_PAL_BUILD_MAX values, else "no compact form")
 *   2. sort     (one workgroup per image): compaction + bitonic sort -> the palette {P, W = getWeightOf(P)} (non-negative
 *               floats order like their bit patterns), the sentinel entry and the sentinel line
 *   3. encode   (several workgroups per image, a range of tile columns each): palette -> LDS hash map value -> rank; one
 *               thread per stored slot: the pixel it stands for (reflect-101 rows at the image border, cv::filter2D's
 *               default border), its rank; for interior slots the ranks of the four neighbours, the horizontal rank steps
 *               packed into the dword, and the check  P[c] == DT, W[c] == w, 0.5*(P[r]-P[l]) == gx, 0.5*(P[d]-P[u]) == gy
 *               bit for bit
 * Not a hot path: runs once per now level that keeps being aligned (or on request, dvo_now_prepare).
 *
 * Compile with -ffp-contract=off (the gradient formula must stay a subtraction followed by a multiplication).
 */
#include "dvo_launch.h"
#include "dvo_palette.h"

namespace dvo {

#define PAL_HASH 8192u
#define PAL_EMPTY 0xffffffffu
#define PAL_WORK_INTS (PAL_HASH + 2)      /* per image: hash set | count | reason */

size_t palette_work_ints(int count) { return (size_t)count * PAL_WORK_INTS; }

DVO_DEV unsigned pal_hash(unsigned key) { return (key * 2654435761u) >> 19; }

/* insert into an open-addressing set; returns true if the key was new.  `stop` (may be NULL) aborts a full table. */
DVO_DEV bool pal_set_insert(unsigned *keys, unsigned key, const int *stop) {
    unsigned h = pal_hash(key);
    for (;;) {
        const unsigned cur = *(volatile unsigned *)&keys[h];
        if (cur == key) return false;
        if (cur == PAL_EMPTY) {
            const unsigned old = atomicCAS(&keys[h], PAL_EMPTY, key);
            if (old == PAL_EMPTY) return true;
            if (old == key) return false;
        }
        if (stop && *(volatile const int *)stop) return false;
        h = (h + 1u) & (PAL_HASH - 1u);
    }
}

__global__ void __launch_bounds__(256)
palette_init_kernel(unsigned *__restrict__ work, int *__restrict__ pal_n, int first_pair, int rows, int cols) {
    unsigned *w = work + (size_t)blockIdx.x * PAL_WORK_INTS;
    for (unsigned i = threadIdx.x; i < PAL_HASH; i += 256) w[i] = PAL_EMPTY;
    if (threadIdx.x == 0) { w[PAL_HASH] = 0u; w[PAL_HASH + 1] = (rows < 2 || cols < 2) ? PAL_SHAPE : 0u; pal_n[first_pair + blockIdx.x] = 0; }
}

/* 1. distinct DT values of pixel columns [x0, x1) of image blockIdx.y */
__global__ void __launch_bounds__(256)
palette_collect_kernel(const float4 *__restrict__ tex, size_t tex_stride, int rows, int cols, unsigned *__restrict__ work,
                       int first_pair, int cols_per_chunk) {
    const int pair = first_pair + blockIdx.y;
    tex += (size_t)pair * tex_stride;
    unsigned *gkeys = work + (size_t)blockIdx.y * PAL_WORK_INTS;
    int *gcnt = reinterpret_cast<int *>(gkeys + PAL_HASH), *gbad = gcnt + 1;
    __shared__ unsigned keys[PAL_HASH];
    __shared__ int lbad, lcnt;
    const int tid = threadIdx.x;
    for (unsigned i = tid; i < PAL_HASH; i += 256) keys[i] = PAL_EMPTY;
    if (tid == 0) { lbad = *gbad; lcnt = 0; }
    __syncthreads();
    if (lbad) return;
    const int tpc16 = texel_tiles_per_col(rows);
    const int x0 = blockIdx.x * cols_per_chunk, x1 = min(cols, x0 + cols_per_chunk);
    for (int xx = x0; xx < x1; xx++) {
        for (int y0 = tid; y0 < rows; y0 += 4 * 256) {          /* four independent loads in flight */
            unsigned k[4];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int yy = y0 + q * 256;
                k[q] = (yy < rows) ? __float_as_uint(tex[texel_index(yy, xx, tpc16)].x) : 0u;
            }
#pragma unroll
            for (int q = 0; q < 4; q++) {
                if (y0 + q * 256 >= rows) continue;
                if (k[q] >= 0x7f800000u) { lbad = PAL_BAD_VALUE; continue; }     /* negative, inf or nan: not a distance */
                /* the local set must never fill up (an insert into a full table would not terminate): more than
                 * DVO_PAL_BUILD_MAX distinct values in this chunk alone already means "no compact form" -- stop inserting */
                if (*(volatile int *)&lbad) continue;
                if (pal_set_insert(keys, k[q], &lbad) && atomicAdd(&lcnt, 1) >= DVO_PAL_BUILD_MAX - 1) lbad = PAL_TOO_MANY;
            }
        }
        if (*(volatile int *)&lbad) break;
    }
    __syncthreads();
    if (lbad) { if (tid == 0) atomicMax(gbad, lbad); return; }
    for (unsigned i = tid; i < PAL_HASH; i += 256) {
        const unsigned k = keys[i];
        if (k == PAL_EMPTY) continue;
        if (pal_set_insert(gkeys, k, gbad) && atomicAdd(gcnt, 1) >= DVO_PAL_BUILD_MAX - 1) atomicMax(gbad, (int)PAL_TOO_MANY);   /* one entry is the sentinel */
    }
}

/* 2. the sorted palette of image blockIdx.x */
__global__ void __launch_bounds__(1024)
palette_sort_kernel(const unsigned *__restrict__ work, unsigned *__restrict__ p4, size_t p4_stride, float2 *__restrict__ pal,
                    int *__restrict__ pal_n, int first_pair) {
    const int pair = first_pair + blockIdx.x;
    const unsigned *gkeys = work + (size_t)blockIdx.x * PAL_WORK_INTS;
    const int n = (int)gkeys[PAL_HASH], bad = (int)gkeys[PAL_HASH + 1];
    const int tid = threadIdx.x;
    if (bad || n < 1 || n > DVO_PAL_BUILD_MAX - 1) { if (tid == 0) pal_n[pair] = -(bad ? bad : (int)PAL_TOO_MANY); return; }
    p4 += (size_t)pair * p4_stride;
    pal += (size_t)pair * DVO_PAL_MAX;
    __shared__ unsigned sorted[DVO_PAL_BUILD_MAX];
    __shared__ int cnt;
    int m = 2;
    while (m < n) m <<= 1;
    if (tid == 0) cnt = 0;
    __syncthreads();
    for (unsigned i = tid; i < PAL_HASH; i += 1024) {
        const unsigned k = gkeys[i];
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
    for (int k = tid; k < n; k += 1024) {
        const float P = __uint_as_float(sorted[k]);
        pal[k] = make_float2(P, weight_of(P));                   /* getWeightOf, SolveDVO.cpp:1047-1053 */
    }
    if (tid == 0) { pal[n] = make_float2(0.0f, 0.0f); pal_n[pair] = n; }       /* the sentinel entry */
    if (tid < 32) p4[tid] = (unsigned)n << 3;                                   /* the sentinel line */
}

/* 3. rank words of tile columns [c0, c1) of image blockIdx.y.
 * value -> rank: distance values live in [0, 255] (cv::normalize), so a table of 2048 buckets of width 1/8 gives the rank
 * range of a value in one look-up and a binary search over the few palette entries of that bucket finishes it (values
 * beyond 256 -- caller-supplied images -- share the last bucket: still correct, just a longer search). */
#define PAL_BUCKETS 2048
DVO_DEV unsigned pal_bucket(unsigned key) {
    const float f = __uint_as_float(key) * 8.0f;
    return (f >= (float)(PAL_BUCKETS - 1)) ? (unsigned)(PAL_BUCKETS - 1) : (unsigned)f;
}

#define PAL_SEG 1020      /* image rows per pass of a tile column (a multiple of DVO_P4_ROWS); LDS holds their ranks + 2 halo rows */
__global__ void __launch_bounds__(256)
palette_encode_kernel(const float4 *__restrict__ tex, size_t tex_stride, int rows, int cols, unsigned *__restrict__ p4,
                      size_t p4_stride, const float2 *__restrict__ pal, int *__restrict__ pal_n, int first_pair, int tcols_per_chunk) {
    const int pair = first_pair + blockIdx.y;
    const int n = pal_n[pair];
    if (n <= 0) return;
    tex += (size_t)pair * tex_stride;
    p4 += (size_t)pair * p4_stride;
    pal += (size_t)pair * DVO_PAL_MAX;
    __shared__ unsigned sorted[DVO_PAL_BUILD_MAX];                 /* P as bits */
    __shared__ unsigned short first[PAL_BUCKETS + 2];        /* first[b] = number of palette values below bucket b */
    __shared__ unsigned short rk[6][PAL_SEG + 2];            /* ranks of pixel columns 4tc-1 .. 4tc+4, rows y0-1 .. y0+PAL_SEG */
    __shared__ int bad;
    const int tid = threadIdx.x;
    for (int k = tid; k < n; k += 256) {
        const float2 e = pal[k];
        sorted[k] = __float_as_uint(e.x);
    }
    if (tid == 0) bad = 0;
    __syncthreads();
    for (int b = tid; b <= PAL_BUCKETS; b += 256) {
        const unsigned edge = __float_as_uint((float)b * (1.0f / 8.0f));       /* exact */
        int lo = 0, hi = n;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (sorted[mid] < edge) lo = mid + 1; else hi = mid; }
        first[b] = (unsigned short)((b == PAL_BUCKETS) ? n : lo);
    }
    __syncthreads();
    auto rank_of = [&](unsigned key) -> int {
        const unsigned b = pal_bucket(key);
        int lo = first[b], hi = first[b + 1];
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (sorted[mid] < key) lo = mid + 1; else hi = mid; }
        return (lo < n && sorted[lo] == key) ? lo : -1;       /* -1 cannot happen: every pixel value was collected */
    };
    const int tpc16 = texel_tiles_per_col(rows), tpc = p4_tiles_per_col(rows);
    const int n_tcols = (cols + 3) >> 2;
    const int c0 = blockIdx.x * tcols_per_chunk, c1 = min(n_tcols, c0 + tcols_per_chunk);
    int lbad = 0;
    for (int tc = c0; tc < c1; tc++) {
        unsigned *__restrict__ col = p4 + 32u + (size_t)tc * tpc * 32u;
        for (int y0 = 0; y0 < rows; y0 += PAL_SEG) {             /* rows [y0, y1) = whole tile rows */
            const int y1 = min(rows, y0 + PAL_SEG), nr = y1 - y0 + 2;
            /* ranks of the 6 x nr pixels around this strip: every pixel's rank is looked up once (reflect-101 outside the image) */
            for (int i0 = tid; i0 < 6 * nr; i0 += 4 * 256) {            /* four independent loads in flight per lane */
                unsigned key[4];
                int cxs[4], rs[4];
                bool in[4];
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int i = i0 + q * 256;
                    const int cx = i / nr, r = i - cx * nr;
                    int x = tc * 4 - 1 + cx, y = y0 - 1 + r;
                    x = (x < 0) ? 1 : ((x >= cols) ? ((x == cols) ? cols - 2 : -1) : x);
                    y = (y < 0) ? 1 : ((y >= rows) ? ((y == rows) ? rows - 2 : -1) : y);
                    cxs[q] = cx; rs[q] = r;
                    in[q] = i < 6 * nr && x >= 0 && y >= 0;
                    key[q] = in[q] ? __float_as_uint(tex[texel_index(y, x, tpc16)].x) : 0u;
                }
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    if (i0 + q * 256 >= 6 * nr) continue;
                    int c = 0;
                    if (in[q]) {
                        c = rank_of(key[q]);
                        if (c < 0) { lbad = PAL_BAD_VALUE; c = 0; }
                    }
                    rk[cxs[q]][rs[q]] = (unsigned short)c;
                }
            }
            __syncthreads();
            /* the stored slots of tile rows y0/6 .. : interior slots get the horizontal rank steps and the bit-exact check */
            const int ty0 = y0 / DVO_P4_ROWS, ty1 = (y1 + DVO_P4_ROWS - 1) / DVO_P4_ROWS;
            for (int s0 = ty0 * 32 + tid; s0 < ty1 * 32; s0 += 4 * 256) {      /* four independent texel loads in flight per lane */
                float4 t[4];
                int ys[4], xls[4];
                bool use[4], inter[4];
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int s = s0 + q * 256;
                    const int ty = s >> 5, w = s & 31, srow = w & 7;
                    xls[q] = w >> 3;
                    const int xx = tc * 4 + xls[q];
                    ys[q] = ty * DVO_P4_ROWS + srow - 1;                  /* image row this slot stands for (-1 / rows: reflected) */
                    use[q] = s < ty1 * 32 && xx < cols && ys[q] <= rows && ys[q] <= y1;
                    inter[q] = use[q] && srow >= 1 && srow <= DVO_P4_ROWS && ys[q] < rows;
                    if (inter[q]) t[q] = tex[texel_index(ys[q], xx, tpc16)];
                }
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int s = s0 + q * 256;
                    if (s >= ty1 * 32) continue;
                    unsigned word = 0u;
                    if (use[q]) {
                        const int r = ys[q] - (y0 - 1), xl = xls[q];
                        const int c = rk[xl + 1][r];
                        word = (unsigned)c << 3;
                        if (inter[q]) {
                            const int rr = rk[xl + 2][r], rl = rk[xl][r], ru = rk[xl + 1][r - 1], rd = rk[xl + 1][r + 1];
                            const int dr = rr - c, dl = rl - c;
                            const float gx = 0.5f * (__uint_as_float(sorted[rr]) - __uint_as_float(sorted[rl]));
                            const float gy = 0.5f * (__uint_as_float(sorted[rd]) - __uint_as_float(sorted[ru]));
                            if (dr < -127 || dr > 127 || dl < -127 || dl > 127) lbad = PAL_STEP;
                            else if (sorted[c] != __float_as_uint(t[q].x)) lbad = PAL_BAD_VALUE;
                            else if (__float_as_uint(gx) != __float_as_uint(t[q].y) || __float_as_uint(gy) != __float_as_uint(t[q].z)) lbad = PAL_GRADIENT;
                            else if (__float_as_uint(pal[c].y) != __float_as_uint(t[q].w)) lbad = PAL_WEIGHT;       /* W[c] = getWeightOf(P[c]), palette_sort_kernel */
                            word |= (((unsigned)dr & 0xffu) << 16) | (((unsigned)dl & 0xffu) << 24);
                        }
                    }
                    col[s] = word;
                }
            }
            __syncthreads();
        }
    }
    if (lbad) bad = lbad;
    __syncthreads();
    if (tid == 0 && bad) atomicMin(&pal_n[pair], -bad);
}

hipError_t launch_palette_build(const float4 *tex, size_t tex_stride, int rows, int cols, unsigned *p4, size_t p4_stride,
                                float2 *pal, int *pal_n, int first_pair, int count, unsigned *work, hipStream_t s) {
    if (count <= 0) return hipSuccess;
    /* enough workgroups per image to fill the GPU for small batches, a few for large ones */
    int chunks = 2048 / count;
    chunks = chunks < 4 ? 4 : (chunks > 64 ? 64 : chunks);
    const int n_tcols = (cols + 3) >> 2;
    const int cpc = (cols + chunks - 1) / chunks, tpcn = (n_tcols + chunks - 1) / chunks;
    hipLaunchKernelGGL(palette_init_kernel, dim3(count), dim3(256), 0, s, work, pal_n, first_pair, rows, cols);
    hipLaunchKernelGGL(palette_collect_kernel, dim3((cols + cpc - 1) / cpc, count), dim3(256), 0, s, tex, tex_stride, rows, cols, work,
                       first_pair, cpc);
    hipLaunchKernelGGL(palette_sort_kernel, dim3(count), dim3(1024), 0, s, work, p4, p4_stride, pal, pal_n, first_pair);
    hipLaunchKernelGGL(palette_encode_kernel, dim3((n_tcols + tpcn - 1) / tpcn, count), dim3(256), 0, s, tex, tex_stride, rows, cols, p4,
                       p4_stride, pal, pal_n, first_pair, tpcn);
    return hipGetLastError();
}

}  // namespace dvo
