/*
 * dvo_frames.hip -- per-frame preprocessing on gfx950, the rows either side of the hot path
 * (SURVEY.md section 8f, rows f1 and f2).  Integer / byte work bound by HBM and launch latency;
 * every kernel is batched: blockIdx.y = image of a batch of `count` same-geometry images, image b
 * of a buffer lives at base + b*stride.  Internal layout is the reference's Eigen layout:
 * COLUMN-major, pixel (yy,xx) at yy + xx*rows.
 *
 *   import_*              wire / Eigen images -> resident grey (u8) and depth (f32 mm)
 *                         (imageArrivedCallBack, src/SolveDVO.cpp:508-519)
 *   camera_level_kernel   full-resolution BGR8 + depth(m) -> one pyramid level
 *                         (camTopic2PublisherPyD.cpp:73-77, :344-347: *1000 -> u16 -> 0->1,
 *                         INTER_NEAREST decimation, BGR2GRAY)
 *   canny_*               cv::Canny(img, 150, 100, 3, true) (src/SolveDVO.cpp:1704, :1764): 3x3 Sobel,
 *                         squared-L2 magnitude, sector non-maximum suppression, hysteresis as a
 *                         union-find over the candidate pixels (order-independent, so identical to
 *                         the sequential stack walk of the CPU implementation)
 *   edt_* / dt_*          distanceTransform(L2, PRECISE) -> normalize(0,255,MINMAX) -> [-.5 0 .5]
 *                         gradients -> texels (src/SolveDVO.cpp:1768-1795, :1063-1098); normalise, gradients
 *                         and the texel store are one kernel
 *   enlist_*              selectedPts + enlistRefEdgePts (src/SolveDVO.cpp:1230-1264, :224-264)
 */
#include "dvo_launch.h"

namespace dvo {

namespace {

DVO_DEV float pow2_neg_f(int level) { return __int_as_float((127 - level) << 23); }
/* one value per 256-thread block: wave shuffle, then the four wave results through LDS; valid in thread 0 */
template <bool MAX>
DVO_DEV int block_reduce_256(int v) {
    __shared__ int part[4];
    for (int off = 32; off > 0; off >>= 1) { const int o = __shfl_down(v, off, 64); v = MAX ? (o > v ? o : v) : v + o; }
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < 4; k++) v = MAX ? (part[k] > v ? part[k] : v) : v + part[k];
    }
    return v;
}
inline unsigned grid_x(size_t n, unsigned cap = 2048) {
    size_t b = (n + 255) / 256;
    if (b < 1) b = 1;
    return (unsigned)(b > cap ? cap : b);
}

}  // namespace

/* ------------------------------------------------------------------------- */
/* import: host-format images -> resident column-major grey / depth              */
/* ------------------------------------------------------------------------- */
template <typename T>
__global__ void __launch_bounds__(256)
import_grey_kernel(const T *__restrict__ src, size_t src_stride, int row_major,
                   unsigned char *__restrict__ grey, size_t stride, int rows, int cols) {
    const size_t n = (size_t)rows * cols;
    src += (size_t)blockIdx.y * src_stride;
    grey += (size_t)blockIdx.y * stride;
    for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (size_t)gridDim.x * blockDim.x) {
        const int xx = (int)(p / rows), yy = (int)(p - (size_t)xx * rows);
        const T v = src[row_major ? (size_t)yy * cols + xx : p];
        unsigned char o;
        if constexpr (sizeof(T) == 1) o = (unsigned char)v;
        else {                                              /* Mat::convertTo(CV_8U): saturate_cast<uchar>(cvRound(v)) */
            const float f = (float)v;
            const float r = rintf(f);                       /* round half to even */
            o = (!(f > -2147483648.5f && f < 2147483648.0f)) ? 0 : (unsigned char)(r < 0.0f ? 0.0f : (r > 255.0f ? 255.0f : r));
        }
        grey[p] = o;
    }
}

template <typename T>
__global__ void __launch_bounds__(256)
import_depth_kernel(const T *__restrict__ src, size_t src_stride, int row_major,
                    float *__restrict__ depth, size_t stride, int rows, int cols) {
    const size_t n = (size_t)rows * cols;
    src += (size_t)blockIdx.y * src_stride;
    depth += (size_t)blockIdx.y * stride;
    for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (size_t)gridDim.x * blockDim.x) {
        const int xx = (int)(p / rows), yy = (int)(p - (size_t)xx * rows);
        const T v = src[row_major ? (size_t)yy * cols + xx : p];
        float o;
        if constexpr (sizeof(T) == 2) o = (float)(v == 0 ? (T)1 : v);   /* dframe.setTo(1, dframe==0)  :514 */
        else o = (float)v;
        depth[p] = o;
    }
}

hipError_t launch_import_grey(const void *src, int dtype, int row_major, size_t src_stride,
                              unsigned char *grey, size_t stride, ImgBatch g, hipStream_t s) {
    const size_t n = (size_t)g.rows * g.cols;
    const dim3 grid(grid_x(n), g.count);
    if (dtype == 0)
        hipLaunchKernelGGL(import_grey_kernel<unsigned char>, grid, dim3(256), 0, s, (const unsigned char *)src, src_stride,
                           row_major, grey, stride, g.rows, g.cols);
    else
        hipLaunchKernelGGL(import_grey_kernel<float>, grid, dim3(256), 0, s, (const float *)src, src_stride,
                           row_major, grey, stride, g.rows, g.cols);
    return hipGetLastError();
}
hipError_t launch_import_depth(const void *src, int dtype, int row_major, size_t src_stride,
                               float *depth, size_t stride, ImgBatch g, hipStream_t s) {
    const size_t n = (size_t)g.rows * g.cols;
    const dim3 grid(grid_x(n), g.count);
    if (dtype == 1)
        hipLaunchKernelGGL(import_depth_kernel<unsigned short>, grid, dim3(256), 0, s, (const unsigned short *)src,
                           src_stride, row_major, depth, stride, g.rows, g.cols);
    else
        hipLaunchKernelGGL(import_depth_kernel<float>, grid, dim3(256), 0, s, (const float *)src, src_stride,
                           row_major, depth, stride, g.rows, g.cols);
    return hipGetLastError();
}

/* ------------------------------------------------------------------------- */
/* row f2: camera frame -> one pyramid level                                    */
/* ------------------------------------------------------------------------- */
DVO_DEV float depth_m_to_mm(float d_m) {
    const float mm = d_m * 1000.0f;                                  /* depth = 1000.0 * depth (32F)   :75 */
    if (!(mm > -2147483648.5f && mm < 2147483648.0f)) return 1.0f;   /* cvRound -> INT_MIN -> saturates to 0 -> 1 */
    float r = rintf(mm);                                             /* convertTo(CV_16U)              :76 */
    r = r < 0.0f ? 0.0f : (r > 65535.0f ? 65535.0f : r);
    return r == 0.0f ? 1.0f : r;                                     /* setTo(1, depth16==0)           :77 */
}

/* One 64 (yy) x 16 (xx) output tile per workgroup: the row-major source is read along rows (16 neighbouring lanes =
 * 16 neighbouring source pixels), the column-major result is written along columns (64 neighbouring lanes = 64
 * consecutive bytes / floats) -- the transpose goes through LDS instead of through uncoalesced global accesses. */
/* cv::undistort of the publisher (camTopic2PublisherPyD.cpp:88-107, :306-308) folded into the level kernel: only the
 * pixels a level keeps are remapped.  The fixed-point map (integer source pixel + 5-bit fractions, what
 * initUndistortRectifyMap writes for CV_16SC2) is built once per calibration on the host (dvo_frames_set_undistort);
 * here the INTER_LINEAR remap with BORDER_CONSTANT 0: 8-bit channels with the 15-bit integer weights of OpenCV's
 * BilinearTab_i ((32-fx)(32-fy)*32 ..., weight 1.0 stored as 32767 with the missing 1 on tap (1,1)) and (sum + 2^14) >> 15;
 * 16-bit depth with float weights, v0*w0 + v1*w1 + v2*w2 + v3*w3 left to right, cvRound. */
struct UndistortMaps { const short2 *xy; const unsigned short *frac; int depth_raw; /* depth already in sensor units: taken as is */ };
DVO_DEV void undistort_taps(int sx, int sy, int src_rows, int src_cols, size_t (&at)[4], bool (&in)[4]) {
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++) {
            const int yy = sy + a, xx = sx + b;
            in[a * 2 + b] = (yy >= 0) && (yy < src_rows) && (xx >= 0) && (xx < src_cols);
            at[a * 2 + b] = in[a * 2 + b] ? (size_t)yy * src_cols + xx : 0;
        }
}
constexpr int CAM_TY = 64, CAM_TX = 16;
__global__ void __launch_bounds__(256)
camera_level_kernel(const unsigned char *__restrict__ bgr, size_t bgr_stride,
                    const float *__restrict__ depth_m, size_t depth_stride,
                    int src_rows, int src_cols, int shift, int tiles_y, UndistortMaps um,
                    unsigned char *__restrict__ grey, float *__restrict__ depth, size_t stride, int rows, int cols) {
    __shared__ unsigned char sg[CAM_TX][CAM_TY + 4];
    __shared__ float sd[CAM_TX][CAM_TY + 1];
    bgr += (size_t)blockIdx.y * bgr_stride;
    grey += (size_t)blockIdx.y * stride;
    if (depth_m) { depth_m += (size_t)blockIdx.y * depth_stride; depth += (size_t)blockIdx.y * stride; }
    const int y0 = (blockIdx.x % tiles_y) * CAM_TY, x0 = (blockIdx.x / tiles_y) * CAM_TX;
#pragma unroll
    for (int k = 0; k < CAM_TY * CAM_TX / 256; k++) {
        const int p = threadIdx.x + k * 256;
        const int lx = p % CAM_TX, ly = p / CAM_TX;
        const int yy = y0 + ly, xx = x0 + lx;
        if (yy < rows && xx < cols) {
            int sy = yy << shift, sx = xx << shift;                  /* resizeNN: min(floor(x/scale), size-1) */
            sy = sy > src_rows - 1 ? src_rows - 1 : sy;
            sx = sx > src_cols - 1 ? src_cols - 1 : sx;
            const size_t sp = (size_t)sy * src_cols + sx;
            int b, gg, r;
            float dmm = 0.0f;
            if (um.xy) {                                             /* pixel (sy, sx) of the UNDISTORTED image */
                const short2 m = um.xy[sp];
                const int fi = um.frac[sp], fy = fi >> 5, fx = fi & 31;
                size_t at[4]; bool in[4];
                undistort_taps(m.x, m.y, src_rows, src_cols, at, in);
                int w[4] = {(32 - fy) * (32 - fx) * 32, (32 - fy) * fx * 32, fy * (32 - fx) * 32, fy * fx * 32};
                if (fi == 0) { w[0] = 32767; w[3] = 1; }
                int acc[3] = {0, 0, 0};
#pragma unroll
                for (int k = 0; k < 4; k++)
                    if (in[k]) { acc[0] += bgr[3 * at[k]] * w[k]; acc[1] += bgr[3 * at[k] + 1] * w[k]; acc[2] += bgr[3 * at[k] + 2] * w[k]; }
                b = (acc[0] + (1 << 14)) >> 15; gg = (acc[1] + (1 << 14)) >> 15; r = (acc[2] + (1 << 14)) >> 15;
                if (depth_m) {
                    const float ty[2] = {1.0f - fy * (1.0f / 32), fy * (1.0f / 32)}, tx[2] = {1.0f - fx * (1.0f / 32), fx * (1.0f / 32)};
                    float a4 = 0.0f;
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const float v = in[k] ? (um.depth_raw ? depth_m[at[k]] : depth_m_to_mm(depth_m[at[k]])) : 0.0f;   /* depth16 is converted BEFORE it is undistorted */
                        const float pw = v * (ty[k >> 1] * tx[k & 1]);
                        a4 = (k == 0) ? pw : a4 + pw;
                    }
                    float rr = rintf(a4);
                    dmm = rr < 0.0f ? 0.0f : (rr > 65535.0f ? 65535.0f : rr);
                }
            } else {
                b = bgr[3 * sp]; gg = bgr[3 * sp + 1]; r = bgr[3 * sp + 2];
                if (depth_m) dmm = um.depth_raw ? depth_m[sp] : depth_m_to_mm(depth_m[sp]);
            }
            sg[lx][ly] = (unsigned char)((1868 * b + 9617 * gg + 4899 * r + (1 << 13)) >> 14);   /* BGR2GRAY 8u */
            if (depth_m) sd[lx][ly] = dmm;
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < CAM_TY * CAM_TX / 256; k++) {
        const int p = threadIdx.x + k * 256;
        const int ly = p % CAM_TY, lx = p / CAM_TY;
        const int yy = y0 + ly, xx = x0 + lx;
        if (yy < rows && xx < cols) {
            const size_t o = (size_t)xx * rows + yy;
            grey[o] = sg[lx][ly];
            if (depth_m) depth[o] = sd[lx][ly];
        }
    }
}

hipError_t launch_camera_level(const unsigned char *bgr, size_t bgr_stride, const float *depth_m, size_t depth_stride,
                               int src_rows, int src_cols, int shift, const short2 *umap_xy, const unsigned short *umap_frac,
                               int depth_raw, unsigned char *grey, float *depth_mm, size_t stride, ImgBatch g, hipStream_t s) {
    const int tiles_y = (g.rows + CAM_TY - 1) / CAM_TY, tiles_x = (g.cols + CAM_TX - 1) / CAM_TX;
    UndistortMaps um{umap_xy, umap_frac, depth_raw};
    hipLaunchKernelGGL(camera_level_kernel, dim3(tiles_y * tiles_x, g.count), dim3(256), 0, s, bgr, bgr_stride, depth_m,
                       depth_stride, src_rows, src_cols, shift, tiles_y, um, grey, depth_mm, stride, g.rows, g.cols);
    return hipGetLastError();
}

/* ------------------------------------------------------------------------- */
/* row f1: Canny                                                                */
/* ------------------------------------------------------------------------- */
/* work layout per batch: label[count*n] (int) | cand[count*n] (u8) | flag[count*n] (u8) */
size_t canny_work_ints(int rows, int cols, int count) {
    const size_t n = (size_t)rows * cols * count;
    return n + 2 * ((n + 3) / 4);
}

/* union-find on int labels (global or LDS): a root points to itself, links only ever go to smaller indices */
DVO_DEV int uf_load(const int *L, int a) { return __hip_atomic_load(L + a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
DVO_DEV int uf_find(const int *L, int a) {
    int p;
    while ((p = uf_load(L, a)) != a) a = p;
    return a;
}
DVO_DEV void uf_union(int *L, int a, int b) {
    for (;;) {
        a = uf_find(L, a);
        b = uf_find(L, b);
        if (a == b) return;
        if (a < b) { const int t = a; a = b; b = t; }            /* a is the larger root: hang it below b */
        const int old = atomicMin(L + a, b);
        if (old == a) return;
        a = old;                                                  /* a had been re-parented meanwhile: go on from there */
    }
}

/* Canny front end, one 64 x 32 pixel tile per workgroup, everything between the grey load and the candidate
 * map in LDS: 3x3 Sobel (BORDER_REPLICATE) -> squared magnitude -> sector non-maximum suppression ->
 * union-find of the tile's candidates.  Out: cand (0 suppressed, 1 candidate, 2 candidate above `high`), and for
 * candidates label = global index of the tile-local root, flag = 0.  Pairs of candidates in different tiles
 * are joined by canny_border_kernel. */
constexpr int CT_Y = 64, CT_X = 32;
__global__ void __launch_bounds__(256)
canny_tile_kernel(const unsigned char *__restrict__ grey, size_t stride, int rows, int cols, int tiles_y, int low, int high,
                  unsigned char *__restrict__ cand, unsigned char *__restrict__ flag, int *__restrict__ label) {
    constexpr int GH = CT_Y + 4, GW = CT_X + 4, MH = CT_Y + 2, MW = CT_X + 2, NT = CT_Y * CT_X;
    __shared__ unsigned char sg[GW * GH];         /* grey, halo 2, [x][y] */
    __shared__ int smag[MW * MH];                 /* squared magnitude, halo 1 (0 outside the image) */
    __shared__ short2 sdxy[MW * MH];
    __shared__ int slab[NT];
    __shared__ unsigned char scand[NT];
    const size_t n = (size_t)rows * cols;
    grey += (size_t)blockIdx.y * stride;
    cand += (size_t)blockIdx.y * n; flag += (size_t)blockIdx.y * n; label += (size_t)blockIdx.y * n;
    const int ty = blockIdx.x % tiles_y, tx = blockIdx.x / tiles_y;
    const int y0 = ty * CT_Y, x0 = tx * CT_X;
    const int tid = threadIdx.x;

    for (int idx = tid; idx < GW * GH; idx += 256) {
        const int lx = idx / GH, ly = idx - lx * GH;
        int gy = y0 + ly - 2, gx = x0 + lx - 2;
        gy = gy < 0 ? 0 : (gy > rows - 1 ? rows - 1 : gy);                      /* BORDER_REPLICATE */
        gx = gx < 0 ? 0 : (gx > cols - 1 ? cols - 1 : gx);
        sg[idx] = grey[(size_t)gx * rows + gy];
    }
    __syncthreads();
    for (int idx = tid; idx < MW * MH; idx += 256) {
        const int lx = idx / MH, ly = idx - lx * MH;
        const int py = y0 + ly - 1, px = x0 + lx - 1;
        int m = 0, dx = 0, dy = 0;
        if (py >= 0 && py < rows && px >= 0 && px < cols) {
            const unsigned char *c = sg + (lx + 1) * GH + (ly + 1);             /* the pixel itself */
            const int a = c[-GH - 1], b = c[-1], cc = c[GH - 1];                /* row above: x-1, x, x+1 */
            const int d = c[-GH], f = c[GH];
            const int g = c[-GH + 1], h = c[1], i = c[GH + 1];
            dx = (cc - a) + 2 * (f - d) + (i - g);
            dy = (g - a) + 2 * (h - b) + (i - cc);
            m = dx * dx + dy * dy;
        }
        smag[idx] = m;
        sdxy[idx] = make_short2((short)dx, (short)dy);
    }
    __syncthreads();
    constexpr int SHIFT = 15;
    constexpr int TG22 = 13573;                                   /* round(tan(22.5 deg) * 2^15) */
    for (int idx = tid; idx < NT; idx += 256) {
        const int lx = idx / CT_Y, ly = idx - lx * CT_Y;
        const int mi = (lx + 1) * MH + (ly + 1);
        const int m = smag[mi];
        bool keep = false;
        if (m > low && y0 + ly < rows && x0 + lx < cols) {
            const short2 d = sdxy[mi];
            const int xs = d.x, ys = d.y;
            const int ax = xs < 0 ? -xs : xs, ay = (ys < 0 ? -ys : ys) << SHIFT;
            const int tg22x = ax * TG22;
            int o1, o2;                                           /* the two neighbours of the sector (offsets in smag) */
            bool ge2;                                             /* second comparison is >= (x and y sectors) */
            if (ay < tg22x) { o1 = -MH; o2 = MH; ge2 = true; }
            else if (ay > tg22x + (ax << (SHIFT + 1))) { o1 = -1; o2 = 1; ge2 = true; }
            else { const int sgn = ((xs ^ ys) < 0) ? -1 : 1; o1 = -1 - sgn * MH; o2 = 1 + sgn * MH; ge2 = false; }
            const int m1 = smag[mi + o1], m2 = smag[mi + o2];
            keep = (m > m1) && (ge2 ? (m >= m2) : (m > m2));
        }
        scand[idx] = keep ? (m > high ? 2 : 1) : 0;
        slab[idx] = keep ? idx : -1;
    }
    __syncthreads();
    for (int idx = tid; idx < NT; idx += 256) {                   /* join with the candidate neighbours of smaller index */
        if (!scand[idx]) continue;
        const int lx = idx / CT_Y, ly = idx - lx * CT_Y;
        if (ly > 0 && scand[idx - 1]) uf_union(slab, idx, idx - 1);
        if (lx > 0) {
            const int q = idx - CT_Y;
            if (scand[q]) uf_union(slab, idx, q);
            if (ly > 0 && scand[q - 1]) uf_union(slab, idx, q - 1);
            if (ly < CT_Y - 1 && scand[q + 1]) uf_union(slab, idx, q + 1);
        }
    }
    __syncthreads();
    for (int idx = tid; idx < NT; idx += 256) {
        const int lx = idx / CT_Y, ly = idx - lx * CT_Y;
        const int py = y0 + ly, px = x0 + lx;
        if (py >= rows || px >= cols) continue;
        const size_t p = (size_t)px * rows + py;
        const unsigned char c = scand[idx];
        cand[p] = c;
        if (c) {
            const int r = uf_find(slab, idx);
            const int rx = r / CT_Y, ry = r - rx * CT_Y;
            label[p] = (x0 + rx) * rows + (y0 + ry);
            flag[p] = 0;
        }
    }
}

/* candidate pairs that straddle a tile boundary: rows r = 64, 128, ... looking up, columns c = 32, 64, ... looking left */
__global__ void __launch_bounds__(256)
canny_border_kernel(const unsigned char *__restrict__ cand, int rows, int cols, int *__restrict__ label) {
    const size_t n = (size_t)rows * cols;
    cand += (size_t)blockIdx.y * n; label += (size_t)blockIdx.y * n;
    const int nA = ((rows - 1) / CT_Y) * cols, nB = ((cols - 1) / CT_X) * rows;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nA + nB; i += gridDim.x * blockDim.x) {
        if (i < nB) {                                             /* lanes along yy: coalesced */
            const int xx = (i / rows + 1) * CT_X, yy = i % rows;
            const size_t p = (size_t)xx * rows + yy, q = p - rows;
            if (!cand[p]) continue;
            if (cand[q]) uf_union(label, (int)p, (int)q);
            if (yy > 0 && cand[q - 1]) uf_union(label, (int)p, (int)q - 1);
            if (yy < rows - 1 && cand[q + 1]) uf_union(label, (int)p, (int)q + 1);
        } else {
            const int j = i - nB;
            const int yy = (j / cols + 1) * CT_Y, xx = j % cols;
            const size_t p = (size_t)xx * rows + yy;
            if (!cand[p]) continue;
            if (cand[p - 1]) uf_union(label, (int)p, (int)p - 1);
            if (xx > 0 && cand[p - 1 - rows]) uf_union(label, (int)p, (int)(p - 1 - rows));
            if (xx < cols - 1 && cand[p - 1 + rows]) uf_union(label, (int)p, (int)(p - 1 + rows));
        }
    }
}

/* label <- root; roots of components holding a strong candidate are flagged */
__global__ void __launch_bounds__(256)
canny_flag_kernel(const unsigned char *__restrict__ cand, size_t n, int *__restrict__ label, unsigned char *__restrict__ flag) {
    cand += (size_t)blockIdx.y * n; label += (size_t)blockIdx.y * n; flag += (size_t)blockIdx.y * n;
    for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (size_t)gridDim.x * blockDim.x) {
        const unsigned char c = cand[p];
        if (!c) continue;
        const int r = uf_find(label, (int)p);
        label[p] = r;                       /* racing writers only ever store ancestors: find() stays correct */
        if (c == 2) flag[r] = 1;
    }
}

__global__ void __launch_bounds__(256)
canny_final_kernel(const unsigned char *__restrict__ cand, const int *__restrict__ label, const unsigned char *__restrict__ flag,
                   size_t n, unsigned char *__restrict__ edge, size_t edge_stride) {
    cand += (size_t)blockIdx.y * n; label += (size_t)blockIdx.y * n; flag += (size_t)blockIdx.y * n;
    edge += (size_t)blockIdx.y * edge_stride;
    for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (size_t)gridDim.x * blockDim.x) {
        bool e = false;
        if (cand[p]) e = flag[uf_find(label, (int)p)] != 0;
        edge[p] = e ? 255 : 0;
    }
}

/* number of edge pixels of one image (inspection only: kept out of the per-frame pipeline, thousands of
 * blocks adding into a handful of adjacent counters serialise in one L2 channel) */
__global__ void __launch_bounds__(256)
count_edges_kernel(const unsigned char *__restrict__ edge, size_t n, int *__restrict__ out) {
    int cnt = 0;
    for (size_t p = threadIdx.x; p < n; p += blockDim.x) cnt += edge[p] ? 1 : 0;
    cnt = block_reduce_256<false>(cnt);
    if (threadIdx.x == 0) *out = cnt;
}
hipError_t launch_count_edges(const unsigned char *edge, size_t n, int *out, hipStream_t s) {
    hipLaunchKernelGGL(count_edges_kernel, dim3(1), dim3(256), 0, s, edge, n, out);
    return hipGetLastError();
}

hipError_t launch_canny(const unsigned char *grey, size_t stride, ImgBatch g, int low, int high, int *work,
                        unsigned char *edge, size_t edge_stride, hipStream_t s) {
    const size_t n = (size_t)g.rows * g.cols, nb = n * g.count;
    int *label = work;
    unsigned char *cand = reinterpret_cast<unsigned char *>(work + nb);
    unsigned char *flag = cand + ((nb + 3) / 4) * 4;
    const int tiles_y = (g.rows + CT_Y - 1) / CT_Y, tiles_x = (g.cols + CT_X - 1) / CT_X;
    const dim3 grid(grid_x(n), g.count), blk(256);
    hipLaunchKernelGGL(canny_tile_kernel, dim3(tiles_y * tiles_x, g.count), blk, 0, s, grey, stride, g.rows, g.cols, tiles_y,
                       low, high, cand, flag, label);
    const int n_border = ((g.rows - 1) / CT_Y) * g.cols + ((g.cols - 1) / CT_X) * g.rows;
    if (n_border > 0)
        hipLaunchKernelGGL(canny_border_kernel, dim3(grid_x((size_t)n_border), g.count), blk, 0, s, cand, g.rows, g.cols, label);
    hipLaunchKernelGGL(canny_flag_kernel, grid, blk, 0, s, cand, n, label, flag);
    hipLaunchKernelGGL(canny_final_kernel, grid, blk, 0, s, cand, label, flag, n, edge, edge_stride);
    return hipGetLastError();
}

/* ------------------------------------------------------------------------- */
/* now-frame preprocessing after Canny: computeDistTransfrmOfNow (SolveDVO.cpp:1768-1795) +
 * imageGradient (:1063-1098).  edge mask -> exact squared EDT in integers (two separable passes)
 * -> sqrt -> min-max normalise to [0,255] (:1774) -> central differences with a
 * reflect-101 border (:1077-1090) -> tiled texels {DT,gx,gy,w}.                              */
/* ------------------------------------------------------------------------- */
#define DVO_EDT_INF(rows, cols) ((rows) + (cols) + 1)

/* phase 1: per column, distance to the nearest edge pixel of that column; one wave per column,
 * 64 rows per step, nearest set bit of the ballot above / below each lane */
__global__ void __launch_bounds__(64)
edt_columns_kernel(const unsigned char *__restrict__ edge, size_t edge_stride, int rows, int cols, int *__restrict__ g) {
    const int xx = blockIdx.x, lane = threadIdx.x;
    edge += (size_t)blockIdx.y * edge_stride;
    g += (size_t)blockIdx.y * rows * cols;
    const size_t base = (size_t)xx * rows;
    const int INF = DVO_EDT_INF(rows, cols);
    const int nchunk = (rows + 63) / 64;
    int carry = INF;                                    /* distance from the row above this chunk to the nearest edge above it */
    for (int c = 0; c < nchunk; c++) {
        const int yy = c * 64 + lane;
        const bool e = (yy < rows) && (edge[base + yy] != 0);
        const unsigned long long m = __ballot(e);
        const unsigned long long low = m & ((lane == 63) ? ~0ull : ((2ull << lane) - 1ull));   /* bits 0..lane */
        int da;
        if (low) da = lane - (63 - __clzll((long long)low));
        else da = (carry >= INF) ? INF : carry + lane + 1;
        if (yy < rows) g[base + yy] = da;
        if (m) carry = 63 - (63 - __clzll((long long)m));                 /* from lane 63 up to the highest edge */
        else carry = (carry >= INF) ? INF : carry + 64;
    }
    carry = INF;                                        /* distance from the row below this chunk to the nearest edge below it */
    for (int c = nchunk - 1; c >= 0; c--) {
        const int yy = c * 64 + lane;
        const bool e = (yy < rows) && (edge[base + yy] != 0);
        const unsigned long long m = __ballot(e);
        const unsigned long long high = m & (~0ull << lane);                                     /* bits lane..63 */
        int db;
        if (high) db = (__ffsll((long long)high) - 1) - lane;
        else db = (carry >= INF) ? INF : carry + (63 - lane) + 1;
        if (yy < rows) { const int da = g[base + yy]; int v = da < db ? da : db; if (v > INF) v = INF; g[base + yy] = v; }
        if (m) carry = __ffsll((long long)m) - 1;                          /* from lane 0 down to the lowest edge */
        else carry = (carry >= INF) ? INF : carry + 64;
    }
}

/* phase 2: d2(x,y) = min_i (x-i)^2 + g(i,y)^2 along the row, exactly, in integers.  Each pixel scans outwards
 * while i^2 < best: with edges every few pixels that is a few dozen steps, far cheaper on a GPU than the sequential
 * lower-envelope scan (Meijster) a CPU would use -- same minimum.  A workgroup stages R whole rows of g in LDS
 * (16-bit: g <= rows+cols+1) so that the scan runs out of LDS, not L2; per-block maxima go to `partial`. */
template <int R>
__global__ void __launch_bounds__(256)
edt_rows_lds_kernel(const int *__restrict__ g, int rows, int cols, int *__restrict__ d2, int *__restrict__ partial) {
    extern __shared__ unsigned short tile[];                  /* [cols][R] */
    const size_t n = (size_t)rows * cols;
    g += (size_t)blockIdx.y * n; d2 += (size_t)blockIdx.y * n;
    const int y0 = blockIdx.x * R;
    const int total = cols * R;
    for (int idx = threadIdx.x; idx < total; idx += 256) {
        const int xx = idx / R, r = idx - xx * R, yy = y0 + r;
        tile[idx] = (unsigned short)((yy < rows) ? g[(size_t)xx * rows + yy] : 0);
    }
    __syncthreads();
    int mx = 0;
    for (int idx = threadIdx.x; idx < total; idx += 256) {
        const int xx = idx / R, r = idx - xx * R, yy = y0 + r;
        if (yy >= rows) continue;
        const int g0 = tile[idx];
        int best = g0 * g0;                                   /* (rows+cols+1)^2 < 2^31 for every supported size */
        /* Branch-free steps: an index that leaves the row is clamped to its end.  The clamped candidate
         * i^2 + g(end)^2 can only exceed the one the end pixel produced at its true distance, so the minimum is
         * unchanged, and the loop needs no per-side exec masking. */
        const int imax = (xx > cols - 1 - xx) ? xx : cols - 1 - xx;
        const int base = idx - xx * R;                        /* LDS index of (column 0, this row) */
        int i2 = 1;
        for (int i = 1; i2 < best && i <= imax; i++) {
            const int xl = (xx - i > 0) ? xx - i : 0, xr = (xx + i < cols - 1) ? xx + i : cols - 1;
            const int gl = tile[base + xl * R], gr = tile[base + xr * R];
            const int cl = i2 + gl * gl, cr = i2 + gr * gr;
            best = cl < best ? cl : best;
            best = cr < best ? cr : best;
            i2 += 2 * i + 1;
        }
        d2[(size_t)xx * rows + yy] = best;
        mx = best > mx ? best : mx;
    }
    mx = block_reduce_256<true>(mx);
    if (threadIdx.x == 0) partial[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = mx;
}

/* fallback for rows too long for LDS: the same scan out of global memory */
__global__ void __launch_bounds__(256)
edt_rows_kernel(const int *__restrict__ g, int rows, int cols, int *__restrict__ d2, int *__restrict__ partial) {
    const size_t n = (size_t)rows * cols;
    g += (size_t)blockIdx.y * n; d2 += (size_t)blockIdx.y * n;
    int mx = 0;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (size_t)gridDim.x * blockDim.x) {
        const int xx = (int)(idx / rows);
        const int g0 = g[idx];
        int best = g0 * g0;
        for (int i = 1; i * i < best; i++) {
            const bool l = xx - i >= 0, r = xx + i < cols;
            if (!l && !r) break;
            if (l) { const int gl = g[idx - (size_t)i * rows]; const int c = i * i + gl * gl; best = c < best ? c : best; }
            if (r) { const int gr = g[idx + (size_t)i * rows]; const int c = i * i + gr * gr; best = c < best ? c : best; }
        }
        d2[idx] = best;
        mx = best > mx ? best : mx;
    }
    mx = block_reduce_256<true>(mx);
    if (threadIdx.x == 0) partial[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = mx;
}

DVO_DEV int reflect101(int i, int n) { return (n == 1) ? 0 : (i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i)); }

/* squared distances -> texels in one pass: min-max normalise to [0,255] (cv::normalize NORM_MINMAX, :1774; an image
 * without any edge pixel -- every distance "infinite" -- normalises to all zeros, as a constant image does), central
 * differences with a reflect-101 border (:1077-1090), weight, tiled store.  One 64 (yy) x 16 (xx) tile per workgroup:
 * the normalised values of the tile and its 1-pixel halo are computed once into LDS (one double sqrt per pixel); the
 * lanes are then mapped so that 8 consecutive lanes write one whole 128-byte texel tile. */
constexpr int NP_TY = 64, NP_TX = 16;
static_assert(DVO_TILE_Y_LOG2 == 2 && DVO_TILE_X_LOG2 == 1, "the store mapping below assumes 4 x 2 texel tiles");
__global__ void __launch_bounds__(256)
dt_normalize_gradient_pack_kernel(const int *__restrict__ d2, int rows, int cols, int tiles_y,
                                  const int *__restrict__ partial, int n_partial,
                                  float4 *__restrict__ out, size_t tex_stride) {
    constexpr int SH = NP_TY + 2, SW = NP_TX + 2;
    __shared__ float sn[SW * SH];                            /* [x][y], halo 1 */
    __shared__ int s_max;
    const size_t n = (size_t)rows * cols;
    d2 += (size_t)blockIdx.y * n;
    partial += (size_t)blockIdx.y * n_partial;
    out += (size_t)blockIdx.y * tex_stride;
    int m = 0;
    for (int k = threadIdx.x; k < n_partial; k += 256) { const int v = partial[k]; m = v > m ? v : m; }
    m = block_reduce_256<true>(m);
    if (threadIdx.x == 0) s_max = m;
    __syncthreads();
    const int INF = DVO_EDT_INF(rows, cols);
    const int m2 = s_max;
    const float mxf = (float)sqrt((double)m2), mnf = 0.0f;
    /* cv::normalize(0, 255, NORM_MINMAX) with OpenCV 2.4's arithmetic: scale and shift in double, then convertTo's 32F -> 32F
     * kernel (cvtScale32f, core/src/convert.cpp) in FLOAT: dst = src*(float)scale + (float)shift */
    const double smin = (double)mnf, smax = (double)mxf;
    const double scale_d = (m2 < INF * INF) ? 255.0 * ((smax - smin > 2.2204460492503131e-16) ? 1. / (smax - smin) : 0.) : 0.;
    const float scale_f = (float)scale_d, shift_f = (float)(0.0 - smin * scale_d);
    const int y0 = (blockIdx.x % tiles_y) * NP_TY, x0 = (blockIdx.x / tiles_y) * NP_TX;
    for (int idx = threadIdx.x; idx < SW * SH; idx += 256) {
        const int lx = idx / SH, ly = idx - lx * SH;
        int yy = y0 + ly - 1, xx = x0 + lx - 1;
        float v = 0.0f;
        if (yy <= rows && xx <= cols) {                      /* one pixel beyond the image is the reflected neighbour */
            yy = reflect101(yy, rows); xx = reflect101(xx, cols);
            const float raw = (float)sqrt((double)d2[(size_t)xx * rows + yy]);
            v = raw * scale_f + shift_f;
        }
        sn[idx] = v;
    }
    __syncthreads();
    const int tpc = texel_tiles_per_col(rows);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int it = 0; it < 4; it++) {
        const int t = it * 32 + wave * 8 + (lane >> 3);     /* texel tile of this 8-lane group: 8 column pairs x 16 row groups */
        const int cp = t >> 4, rg = t & 15;
        const int ly = rg * 4 + (lane & 3), lx = cp * 2 + ((lane >> 2) & 1);
        const int yy = y0 + ly, xx = x0 + lx;
        if (yy < rows && xx < cols) {
            const float *c = sn + (lx + 1) * SH + (ly + 1);
            const float v = c[0];
            out[texel_index(yy, xx, tpc)] = make_float4(v, 0.5f * c[SH] - 0.5f * c[-SH], 0.5f * c[1] - 0.5f * c[-1], weight_of(v));
        }
    }
}

static int edt_rows_per_block(int rows, int cols) {   /* LDS rows per workgroup; 0 = row too long, use the global-memory scan */
    if (rows + cols + 1 > 65535) return 0;            /* g would not fit 16 bits */
    if ((size_t)cols * 16 * 2 <= 64 * 1024) return 16;
    if ((size_t)cols * 8 * 2 <= 64 * 1024) return 8;
    if ((size_t)cols * 4 * 2 <= 64 * 1024) return 4;
    return 0;
}
static unsigned edt_row_blocks(int rows, int cols) {
    const int R = edt_rows_per_block(rows, cols);
    return R ? (unsigned)((rows + R - 1) / R) : grid_x((size_t)rows * cols);
}
size_t edt_work_ints(int rows, int cols, int count) { return (2 * (size_t)rows * cols + edt_row_blocks(rows, cols)) * count; }

hipError_t launch_edges_to_texels(const unsigned char *edge, size_t edge_stride, ImgBatch gb, int *work,
                                  float4 *tex_out, size_t tex_stride, hipStream_t s) {
    const size_t n = (size_t)gb.rows * gb.cols, nb = n * gb.count;
    int *g = work, *d2 = work + nb, *partial = work + 2 * nb;
    const dim3 grid(grid_x(n), gb.count);
    const int R = edt_rows_per_block(gb.rows, gb.cols);
    const unsigned nblk = edt_row_blocks(gb.rows, gb.cols);
    hipLaunchKernelGGL(edt_columns_kernel, dim3(gb.cols, gb.count), dim3(64), 0, s, edge, edge_stride, gb.rows, gb.cols, g);
    const size_t lds = (size_t)gb.cols * R * 2;
    if (R == 16) hipLaunchKernelGGL(edt_rows_lds_kernel<16>, dim3(nblk, gb.count), dim3(256), lds, s, g, gb.rows, gb.cols, d2, partial);
    else if (R == 8) hipLaunchKernelGGL(edt_rows_lds_kernel<8>, dim3(nblk, gb.count), dim3(256), lds, s, g, gb.rows, gb.cols, d2, partial);
    else if (R == 4) hipLaunchKernelGGL(edt_rows_lds_kernel<4>, dim3(nblk, gb.count), dim3(256), lds, s, g, gb.rows, gb.cols, d2, partial);
    else hipLaunchKernelGGL(edt_rows_kernel, grid, dim3(256), 0, s, g, gb.rows, gb.cols, d2, partial);
    const int tiles_y = (gb.rows + NP_TY - 1) / NP_TY, tiles_x = (gb.cols + NP_TX - 1) / NP_TX;
    hipLaunchKernelGGL(dt_normalize_gradient_pack_kernel, dim3(tiles_y * tiles_x, gb.count), dim3(256), 0, s, d2, gb.rows, gb.cols,
                       tiles_y, partial, (int)nblk, tex_out, tex_stride);
    return hipGetLastError();
}
hipError_t launch_now_level_from_edges(const unsigned char *edge, int rows, int cols, int *work, float4 *tex_out, hipStream_t s) {
    return launch_edges_to_texels(edge, 0, ImgBatch{rows, cols, 1}, work, tex_out, 0, s);
}

/* ------------------------------------------------------------------------- */
/* selectedPts + enlistRefEdgePts  (SolveDVO.cpp:1230-1264, :224-264)          */
/* Column-major scan order (xx outer, yy inner): one wave per image column.     */
/*                                                                             */
/* The 3 x N float list (and uv) come out in the reference's order.  The        */
/* compact 8-byte twin {xx | yy << 16, Z} carries its pixel in itself, so its   */
/* order is free: it is written in BLOCK order -- 16 x 16-pixel blocks, block   */
/* columns outer; inside a block column-major -- so that the 64 consecutive     */
/* points of one gather instruction are neighbours in the image and share       */
/* texel lines (modelled on the bench scenes: 124 k -> 108 k L2 requests per    */
/* 640x480x4 alignment).  Sums over points are order-independent up to the      */
/* double rounding the numerics contract already allows.  */
/* ------------------------------------------------------------------------- */
DVO_DEV bool ref_selected(int e, float d) { return (e > 0) && (d > 100.0f); }   /* :1251 */

/* entry of (pixel column xx, block row by) in the block-order count array: ((xx/16)*nby + by)*16 + xx%16 */
DVO_DEV int blk_entry(int xx, int by, int nby) { return ((xx >> 4) * nby + by) * 16 + (xx & 15); }

template <typename E>
__global__ void __launch_bounds__(64)
enlist_count_kernel(const E *__restrict__ edge, size_t edge_stride, const float *__restrict__ depth, size_t depth_stride,
                    int rows, int cols, int *__restrict__ col_counts, int *__restrict__ blk_counts, int nby) {
    const int xx = blockIdx.x, lane = threadIdx.x;
    edge += (size_t)blockIdx.y * edge_stride; depth += (size_t)blockIdx.y * depth_stride;
    col_counts += (size_t)blockIdx.y * (cols + 2);
    const int n_blk = ((cols + 15) >> 4) * nby * 16;
    if (blk_counts) blk_counts += (size_t)blockIdx.y * (n_blk + 2);
    const size_t base = (size_t)xx * rows;
    int cnt = 0;
    for (int y0 = 0; y0 < rows; y0 += 64) {
        const int yy = y0 + lane;
        const bool sel = (yy < rows) && ref_selected((int)edge[base + yy], depth[base + yy]);
        const unsigned long long m = __ballot(sel);
        cnt += __popcll(m);
        const int by = (y0 >> 4) + lane;                     /* lanes 0..3: the four 16-row segments of this chunk */
        if (blk_counts && lane < 4 && by < nby) blk_counts[blk_entry(xx, by, nby)] = __popcll((m >> (16 * lane)) & 0xffffull);
    }
    if (lane == 0) col_counts[xx] = cnt;
    /* the last block column may be narrower than 16: its missing pixel columns count zero */
    if (blk_counts && xx == cols - 1)
        for (int x2 = cols; x2 < ((cols + 15) & ~15); x2++)
            for (int by = lane; by < nby; by += 64) blk_counts[blk_entry(x2, by, nby)] = 0;
}

/* exclusive scan of col_counts[0..cols) in place; col_counts[cols] = col_counts[cols+1] = total */
__global__ void __launch_bounds__(1024)
enlist_scan_kernel(int *__restrict__ col_counts, int cols) {
    __shared__ int part[1024];
    col_counts += (size_t)blockIdx.y * (cols + 2);
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
    if (tid == 1023) { col_counts[cols] = part[1023]; col_counts[cols + 1] = part[1023]; }
}

template <typename E>
__global__ void __launch_bounds__(64)
enlist_write_kernel(const E *__restrict__ edge, size_t edge_stride, const float *__restrict__ depth, size_t depth_stride,
                    int rows, int cols, int level, Intrinsics K, const int *__restrict__ col_offsets,
                    const int *__restrict__ blk_offsets, int nby,
                    float *__restrict__ xyz, size_t xyz_stride, uint2 *__restrict__ compact,
                    float *__restrict__ uv, int capacity, int *__restrict__ N_dst) {
    const int xx = blockIdx.x, lane = threadIdx.x;
    edge += (size_t)blockIdx.y * edge_stride; depth += (size_t)blockIdx.y * depth_stride;
    col_offsets += (size_t)blockIdx.y * (cols + 2);
    if (blk_offsets) blk_offsets += (size_t)blockIdx.y * (((cols + 15) >> 4) * nby * 16 + 2);
    xyz += (size_t)blockIdx.y * xyz_stride;
    if (compact) compact += (size_t)blockIdx.y * (xyz_stride / 3);
    const int Nall = col_offsets[cols];
    if (N_dst && xx == 0 && lane == 0) N_dst[blockIdx.y] = Nall < capacity ? Nall : capacity;
    /* a truncated list (never with the engine's own capacity management) keeps the reference order in the compact twin
     * too: the block order would hold another subset */
    const bool blocked = compact && blk_offsets && Nall <= capacity;
    const size_t base = (size_t)xx * rows;
    const float scaleFac = pow2_neg_f(level);                           /* :231 */
    const float tmpfx = (float)(1. / (double)(scaleFac * K.fx));        /* :232 double division */
    const float tmpfy = (float)(1. / (double)(scaleFac * K.fy));        /* :233 */
    const float tmpcx = scaleFac * K.cx;                                /* :234 */
    const float tmpcy = scaleFac * K.cy;                                /* :235 */
    int run = col_offsets[xx];
    for (int y0 = 0; y0 < rows; y0 += 64) {
        const int yy = y0 + lane;
        float d = 0.0f;
        bool sel = false;
        if (yy < rows) { d = depth[base + yy]; sel = ref_selected((int)edge[base + yy], d); }
        const unsigned long long m = __ballot(sel);
        if (sel) {
            const int nC = run + __popcll(m & ((1ull << lane) - 1ull));
            if (nC < capacity) {
                const float Z = d / 1000.0f;                            /* :248 */
                const float X = Z * ((float)xx - tmpcx) * tmpfx;        /* :249 */
                const float Y = Z * ((float)yy - tmpcy) * tmpfy;        /* :250 */
                xyz[3 * nC] = X; xyz[3 * nC + 1] = Y; xyz[3 * nC + 2] = Z;   /* :254-256 */
                if (compact) {
                    int bC = nC;
                    if (blocked) {
                        const int seg = lane >> 4;
                        bC = blk_offsets[blk_entry(xx, (y0 >> 4) + seg, nby)] +
                             __popcll((m >> (16 * seg)) & ((1ull << (lane & 15)) - 1ull));
                    }
                    compact[bC] = make_uint2((unsigned)xx | ((unsigned)yy << 16), __float_as_uint(Z));
                }
                if (uv) { uv[2 * nC] = (float)xx; uv[2 * nC + 1] = (float)yy; }   /* :244-245 */
            }
        }
        run += __popcll(m);
    }
}

size_t enlist_block_ints(int rows, int cols) { return (size_t)((cols + 15) >> 4) * ((rows + 15) >> 4) * 16 + 2; }

template <typename E>
static hipError_t enlist_count_t(const E *edge, size_t edge_stride, const float *depth, size_t depth_stride, ImgBatch g,
                                 int *col_counts, int *blk_counts, hipStream_t s) {
    const int nby = (g.rows + 15) >> 4;
    hipLaunchKernelGGL(enlist_count_kernel<E>, dim3(g.cols, g.count), dim3(64), 0, s, edge, edge_stride, depth, depth_stride,
                       g.rows, g.cols, col_counts, blk_counts, nby);
    hipLaunchKernelGGL(enlist_scan_kernel, dim3(1, g.count), dim3(1024), 0, s, col_counts, g.cols);
    if (blk_counts)
        hipLaunchKernelGGL(enlist_scan_kernel, dim3(1, g.count), dim3(1024), 0, s, blk_counts, (int)enlist_block_ints(g.rows, g.cols) - 2);
    return hipGetLastError();
}
template <typename E>
static hipError_t enlist_write_t(const E *edge, size_t edge_stride, const float *depth, size_t depth_stride, ImgBatch g,
                                 int level, const Intrinsics &K, const int *col_counts, const int *blk_counts, float *xyz,
                                 size_t xyz_stride, uint2 *compact, float *uv, int capacity, int *N_dst, hipStream_t s) {
    hipLaunchKernelGGL(enlist_write_kernel<E>, dim3(g.cols, g.count), dim3(64), 0, s, edge, edge_stride, depth, depth_stride,
                       g.rows, g.cols, level, K, col_counts, blk_counts, (g.rows + 15) >> 4, xyz, xyz_stride, compact, uv, capacity, N_dst);
    return hipGetLastError();
}

hipError_t launch_enlist_count(const void *edge, int edge_is_u8, size_t edge_stride, const float *depth, size_t depth_stride,
                               ImgBatch g, int *col_counts, int *blk_counts, hipStream_t s) {
    return edge_is_u8 ? enlist_count_t((const unsigned char *)edge, edge_stride, depth, depth_stride, g, col_counts, blk_counts, s)
                      : enlist_count_t((const int32_t *)edge, edge_stride, depth, depth_stride, g, col_counts, blk_counts, s);
}
hipError_t launch_enlist_write(const void *edge, int edge_is_u8, size_t edge_stride, const float *depth, size_t depth_stride,
                               ImgBatch g, int level, const Intrinsics &K, const int *col_counts, const int *blk_counts, float *xyz,
                               size_t xyz_stride, uint2 *compact, float *uv, int capacity, int *N_dst, hipStream_t s) {
    return edge_is_u8 ? enlist_write_t((const unsigned char *)edge, edge_stride, depth, depth_stride, g, level, K, col_counts, blk_counts,
                                       xyz, xyz_stride, compact, uv, capacity, N_dst, s)
                      : enlist_write_t((const int32_t *)edge, edge_stride, depth, depth_stride, g, level, K, col_counts, blk_counts,
                                       xyz, xyz_stride, compact, uv, capacity, N_dst, s);
}

}  // namespace dvo
