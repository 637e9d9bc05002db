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
#include "dvo_palette.h"
#include <stdlib.h>
#include <stdio.h>
#include <string.h>

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
/* image `by` of a launch: at base + by * stride (the landing buffer), or -- camera frames that already sit in HBM, round 6 -- wherever the
 * caller's pointer table says: the frames are read where they are, no landing copy (236 MB each way per 256 VGA frames) */
DVO_DEV void camera_sources(const SrcTab &tab, int by, const unsigned char *__restrict__ &bgr, size_t bgr_stride,
                            const float *__restrict__ &depth_m, size_t depth_stride) {
    if (tab.bgr) bgr = static_cast<const unsigned char *>(tab.bgr[by]); else bgr += (size_t)by * bgr_stride;
    if (tab.depth) depth_m = static_cast<const float *>(tab.depth[by]); else if (depth_m) depth_m += (size_t)by * depth_stride;
}
DVO_DEV void camera_level_body(const int bx, const int by, const unsigned char *__restrict__ bgr, size_t bgr_stride,
                    const float *__restrict__ depth_m, size_t depth_stride,
                    int src_rows, int src_cols, int shift, int tiles_y, UndistortMaps um,
                    unsigned char *__restrict__ grey, float *__restrict__ depth, size_t stride, int rows, int cols, const SrcTab tab) {
    __shared__ unsigned char sg[CAM_TX][CAM_TY + 4];
    __shared__ float sd[CAM_TX][CAM_TY + 1];
    camera_sources(tab, by, bgr, bgr_stride, depth_m, depth_stride);
    grey += (size_t)by * stride;
    if (depth_m) depth += (size_t)by * stride;
    const int y0 = (bx % tiles_y) * CAM_TY, x0 = (bx / tiles_y) * CAM_TX;
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

__global__ void __launch_bounds__(256)
camera_level_kernel(const unsigned char *__restrict__ bgr, size_t bgr_stride, const float *__restrict__ depth_m, size_t depth_stride,
                    int src_rows, int src_cols, int shift, int tiles_y, UndistortMaps um,
                    unsigned char *__restrict__ grey, float *__restrict__ depth, size_t stride, int rows, int cols, const SrcTab tab) {
    camera_level_body(blockIdx.x, blockIdx.y, bgr, bgr_stride, depth_m, depth_stride, src_rows, src_cols, shift, tiles_y, um, grey, depth, stride, rows, cols, tab);
}
/* several pyramid levels of the same camera frames in one launch (see CannyLevels in the Canny section) */
struct CameraLevels {
    int n, src_rows, src_cols;
    int shift[DVO_LEVELS], rows[DVO_LEVELS], cols[DVO_LEVELS];
    unsigned first[DVO_LEVELS + 1];
    unsigned char *grey[DVO_LEVELS]; float *depth[DVO_LEVELS]; size_t stride[DVO_LEVELS];
};
__global__ void __launch_bounds__(256)
camera_levels_kernel(const unsigned char *__restrict__ bgr, size_t bgr_stride, const float *__restrict__ depth_m, size_t depth_stride,
                     UndistortMaps um, const CameraLevels t, const SrcTab tab) {
    int l = 0;
    while (l + 1 < t.n && blockIdx.x >= t.first[l + 1]) l++;
    camera_level_body((int)(blockIdx.x - t.first[l]), blockIdx.y, bgr, bgr_stride, depth_m, depth_stride, t.src_rows, t.src_cols, t.shift[l],
                      (t.rows[l] + CAM_TY - 1) / CAM_TY, um, t.grey[l], t.depth[l], t.stride[l], t.rows[l], t.cols[l], tab);
}

/* levels 1 .. n-1 FROM LEVEL 0 (round 6): nearest-neighbour decimation commutes with every per-pixel step of the level kernel
 * (BGR2GRAY, metres -> millimetres, the undistortion remap), so pixel (yy, xx) of level l IS pixel (yy << l, xx << l) of level 0
 * whenever that index needs no clamping -- the host checks ((rows_l - 1) << l <= rows_0 - 1, columns alike; every camera format).
 * One byte (+ one float) per pixel read from the column-major level 0 instead of three BGR bytes of the row-major source at a
 * stride: 61 -> ~20 us per 256 four-level VGA frames. */
struct DecimateLevels {
    int n, rows0, cols0;
    int rows[DVO_LEVELS], cols[DVO_LEVELS];                     /* levels 1 .. n (index 0 = level 1) */
    unsigned first[DVO_LEVELS + 1];                             /* first workgroup of every level */
    const unsigned char *grey0; const float *depth0; size_t stride0;
    unsigned char *grey[DVO_LEVELS]; float *depth[DVO_LEVELS]; size_t stride[DVO_LEVELS];
};
/* QUAD: four consecutive rows per thread (rows of every level a multiple of four, images 16-byte aligned: every camera format) -- the
 * 4 << l source bytes of a column as dword loads, one dword store; else one pixel per thread */
template <bool QUAD>
__global__ void __launch_bounds__(256) camera_decimate_levels_kernel(const DecimateLevels t) {
    int l = 0;
    while (l + 1 < t.n && blockIdx.x >= t.first[l + 1]) l++;
    const int rows = t.rows[l], cols = t.cols[l], sh = l + 1;
    const unsigned p = ((blockIdx.x - t.first[l]) * 256u + threadIdx.x) * (QUAD ? 4u : 1u);
    if (p >= (unsigned)(rows * cols)) return;
    const unsigned xx = p / (unsigned)rows, yy = p - xx * (unsigned)rows;
    const size_t src = (size_t)(xx << sh) * t.rows0 + (yy << sh);
    const size_t img0 = (size_t)blockIdx.y * t.stride0, img = (size_t)blockIdx.y * t.stride[l];
    if constexpr (QUAD) {
        const unsigned *g = reinterpret_cast<const unsigned *>(t.grey0 + img0 + src);      /* 4 << sh bytes: 2, 4 or 8 ... dwords */
        const unsigned nd = 1u << sh;                                                      /* dwords per output quad */
        unsigned out = 0u;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const unsigned byte = (unsigned)k << sh;                                       /* source byte of output k */
            out |= ((g[(byte >> 2) < nd ? (byte >> 2) : 0] >> (8u * (byte & 3u))) & 0xffu) << (8 * k);
        }
        *reinterpret_cast<unsigned *>(t.grey[l] + img + p) = out;
        if (t.depth0) {
            const float *d = t.depth0 + img0 + src;
            float4 o;
            o.x = d[0]; o.y = d[1u << sh]; o.z = d[2u << sh]; o.w = d[3u << sh];
            *reinterpret_cast<float4 *>(t.depth[l] + img + p) = o;
        }
    } else {
        t.grey[l][img + p] = t.grey0[img0 + src];
        if (t.depth0) t.depth[l][img + p] = t.depth0[img0 + src];
    }
}
/* false: some level would need the clamp of resizeNN -- the caller keeps launch_camera_levels */
bool camera_levels_decimate_ok(int n_levels, const int *rows, const int *cols) {
    for (int l = 1; l < n_levels; l++)
        if (((long long)(rows[l] - 1) << l) > rows[0] - 1 || ((long long)(cols[l] - 1) << l) > cols[0] - 1) return false;
    return n_levels > 1;
}
hipError_t launch_camera_decimate_levels(const unsigned char *grey0, const float *depth0, size_t stride0, int rows0, int cols0, int n,
                                         const int *rows, const int *cols, unsigned char *const *grey, float *const *depth, const size_t *stride,
                                         int count, hipStream_t s) {
    if (n < 1 || n > DVO_LEVELS) return hipErrorInvalidValue;
    DecimateLevels t;
    t.n = n; t.rows0 = rows0; t.cols0 = cols0; t.grey0 = grey0; t.depth0 = depth0; t.stride0 = stride0;
    t.first[0] = 0;
    bool quad = (rows0 & 3) == 0 && (stride0 & 3) == 0 && (reinterpret_cast<size_t>(grey0) & 3) == 0 &&
                (!depth0 || (reinterpret_cast<size_t>(depth0) & 15) == 0);
    for (int l = 0; l < n; l++)
        quad = quad && (rows[l] & 3) == 0 && (stride[l] & 3) == 0 && (reinterpret_cast<size_t>(grey[l]) & 3) == 0 &&
               (!depth0 || (reinterpret_cast<size_t>(depth[l]) & 15) == 0);
    for (int l = 0; l < n; l++) {
        t.rows[l] = rows[l]; t.cols[l] = cols[l]; t.grey[l] = grey[l]; t.depth[l] = depth[l]; t.stride[l] = stride[l];
        t.first[l + 1] = t.first[l] + (unsigned)(((size_t)rows[l] * cols[l] / (quad ? 4 : 1) + 255) / 256);
    }
    if (quad) hipLaunchKernelGGL(camera_decimate_levels_kernel<true>, dim3(t.first[n], count), dim3(256), 0, s, t);
    else hipLaunchKernelGGL(camera_decimate_levels_kernel<false>, dim3(t.first[n], count), dim3(256), 0, s, t);
    return hipGetLastError();
}

/* The full-resolution level of an undistortion-free camera frame (shift 0, no map: every pixel is read once): four pixels per
 * lane -- twelve contiguous BGR bytes as three dwords, four depth floats as one 16-byte load -- 64 x 64 tiles, the transpose to
 * the column-major result through LDS with 4-byte stores along yy.  Needs cols and rows in multiples of four (every camera
 * format); everything else takes camera_level_kernel. */
constexpr int CF_T = 64;
__global__ void __launch_bounds__(256)
camera_level0_kernel(const unsigned char *__restrict__ bgr, size_t bgr_stride, const float *__restrict__ depth_m, size_t depth_stride,
                     int rows, int cols, int tiles_y, int depth_raw,
                     unsigned char *__restrict__ grey, float *__restrict__ depth, size_t stride, const SrcTab tab) {
    __shared__ unsigned sg4[CF_T][CF_T / 4 + 1];                /* [x][y / 4]: grey bytes */
    __shared__ float sd[CF_T][CF_T + 1];
    unsigned char (*sg)[CF_T + 4] = reinterpret_cast<unsigned char (*)[CF_T + 4]>(sg4);
    camera_sources(tab, blockIdx.y, bgr, bgr_stride, depth_m, depth_stride);
    grey += (size_t)blockIdx.y * stride;
    if (depth_m) depth += (size_t)blockIdx.y * stride;
    const int y0 = (blockIdx.x % tiles_y) * CF_T, x0 = (blockIdx.x / tiles_y) * CF_T;
    auto to_grey = [](unsigned b, unsigned g, unsigned r) { return (unsigned char)((1868u * b + 9617u * g + 4899u * r + (1u << 13)) >> 14); };   /* BGR2GRAY 8u */
#pragma unroll
    for (int k = 0; k < CF_T * CF_T / 4 / 256; k++) {
        const int p = threadIdx.x + k * 256;
        const int ly = p >> 4, lx = (p & 15) * 4;
        const int yy = y0 + ly, xx = x0 + lx;
        if (yy < rows && xx < cols) {
            const unsigned sp = (unsigned)(yy * cols + xx);
            const unsigned *src = reinterpret_cast<const unsigned *>(bgr + (size_t)sp * 3);
            const unsigned u0 = src[0], u1 = src[1], u2 = src[2];
            sg[lx][ly] = to_grey(u0 & 255u, (u0 >> 8) & 255u, (u0 >> 16) & 255u);
            sg[lx + 1][ly] = to_grey(u0 >> 24, u1 & 255u, (u1 >> 8) & 255u);
            sg[lx + 2][ly] = to_grey((u1 >> 16) & 255u, u1 >> 24, u2 & 255u);
            sg[lx + 3][ly] = to_grey((u2 >> 8) & 255u, (u2 >> 16) & 255u, u2 >> 24);
            if (depth_m) {
                const float4 d = *reinterpret_cast<const float4 *>(depth_m + sp);
                sd[lx][ly] = depth_raw ? d.x : depth_m_to_mm(d.x); sd[lx + 1][ly] = depth_raw ? d.y : depth_m_to_mm(d.y);
                sd[lx + 2][ly] = depth_raw ? d.z : depth_m_to_mm(d.z); sd[lx + 3][ly] = depth_raw ? d.w : depth_m_to_mm(d.w);
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < CF_T * CF_T / 4 / 256; k++) {
        const int p = threadIdx.x + k * 256;
        const int ly4 = p & 15, lx = p >> 4;
        const int yy = y0 + 4 * ly4, xx = x0 + lx;
        if (yy < rows && xx < cols) *reinterpret_cast<unsigned *>(grey + (size_t)xx * rows + yy) = sg4[lx][ly4];
    }
    if (depth_m) {
#pragma unroll
        for (int k = 0; k < CF_T * CF_T / 256; k++) {
            const int p = threadIdx.x + k * 256;
            const int ly = p & 63, lx = p >> 6;
            const int yy = y0 + ly, xx = x0 + lx;
            if (yy < rows && xx < cols) depth[(size_t)xx * rows + yy] = sd[lx][ly];
        }
    }
}

/* DVO_UPLOAD_DEVICE: camera images that already sit in HBM, one pointer each -> the landing buffer (image i at dst + i*stride).
 * One launch per 32 images (the pointers travel as a kernel argument) instead of one copy call per image. */
struct GatherPack { const void *src[32]; };
__global__ void __launch_bounds__(256)
gather_images_kernel(GatherPack pk, unsigned char *__restrict__ dst, size_t bytes, size_t stride) {
    const unsigned char *src = static_cast<const unsigned char *>(pk.src[blockIdx.y]);
    unsigned char *d = dst + (size_t)blockIdx.y * stride;
    const size_t t0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nt = (size_t)gridDim.x * blockDim.x;
    if (((reinterpret_cast<size_t>(src) | reinterpret_cast<size_t>(d)) & 15) == 0) {
        const size_t n16 = bytes / 16;
        for (size_t i = t0; i < n16; i += nt) reinterpret_cast<uint4 *>(d)[i] = reinterpret_cast<const uint4 *>(src)[i];
        for (size_t i = n16 * 16 + t0; i < bytes; i += nt) d[i] = src[i];
    } else {
        for (size_t i = t0; i < bytes; i += nt) d[i] = src[i];
    }
}
hipError_t launch_gather_images(const void *const *src, int count, void *dst, size_t bytes, size_t stride, hipStream_t s, int max_wgs_per_image) {
    for (int b = 0; b < count; b += 32) {
        GatherPack pk;
        const int nc = count - b < 32 ? count - b : 32;
        for (int i = 0; i < 32; i++) pk.src[i] = src[b + (i < nc ? i : 0)];
        /* HBM sources: up to 64 workgroups per image.  Mapped host memory: a few -- the link needs ~100 KB in flight, and a
         * grid that fills every wave slot with lanes waiting on PCIe would lock the preprocessing kernels of the previous chunk out */
        const size_t want = (bytes / 16 + 255) / 256;
        const unsigned gx = (unsigned)(want < 1 ? 1 : (want > (size_t)max_wgs_per_image ? (size_t)max_wgs_per_image : want));
        hipLaunchKernelGGL(gather_images_kernel, dim3(gx, nc), dim3(256), 0, s, pk, static_cast<unsigned char *>(dst) + (size_t)b * stride, bytes, stride);
    }
    return hipGetLastError();
}

hipError_t launch_camera_level(const unsigned char *bgr, size_t bgr_stride, const float *depth_m, size_t depth_stride,
                               int src_rows, int src_cols, int shift, const short2 *umap_xy, const unsigned short *umap_frac,
                               int depth_raw, unsigned char *grey, float *depth_mm, size_t stride, ImgBatch g, hipStream_t s, SrcTab tab) {
    /* a pointer table: the caller has checked every image's alignment (4 bytes for BGR, 16 for depth) and passes depth_m != NULL iff
     * the table has depth images */
    const bool src_ok = tab.bgr ? true : (((reinterpret_cast<size_t>(bgr) | bgr_stride) & 3) == 0 &&
                                          (!depth_m || ((reinterpret_cast<size_t>(depth_m) | (depth_stride * 4)) & 15) == 0));
    if (shift == 0 && !umap_xy && g.rows == src_rows && g.cols == src_cols && (g.rows & 3) == 0 && (g.cols & 3) == 0 && src_ok &&
        ((reinterpret_cast<size_t>(grey) | stride) & 3) == 0) {
        const int ty = (g.rows + CF_T - 1) / CF_T, tx = (g.cols + CF_T - 1) / CF_T;
        hipLaunchKernelGGL(camera_level0_kernel, dim3(ty * tx, g.count), dim3(256), 0, s, bgr, bgr_stride, depth_m, depth_stride,
                           g.rows, g.cols, ty, depth_raw, grey, depth_mm, stride, tab);
        return hipGetLastError();
    }
    const int tiles_y = (g.rows + CAM_TY - 1) / CAM_TY, tiles_x = (g.cols + CAM_TX - 1) / CAM_TX;
    UndistortMaps um{umap_xy, umap_frac, depth_raw};
    hipLaunchKernelGGL(camera_level_kernel, dim3(tiles_y * tiles_x, g.count), dim3(256), 0, s, bgr, bgr_stride, depth_m,
                       depth_stride, src_rows, src_cols, shift, tiles_y, um, grey, depth_mm, stride, g.rows, g.cols, tab);
    return hipGetLastError();
}
/* levels first_level .. n-1 of the same camera frames in one launch (the full-resolution level keeps its own kernel) */
hipError_t launch_camera_levels(const unsigned char *bgr, size_t bgr_stride, const float *depth_m, size_t depth_stride, int src_rows, int src_cols,
                                int n, const int *shift, const int *rows, const int *cols, const short2 *umap_xy, const unsigned short *umap_frac,
                                int depth_raw, unsigned char *const *grey, float *const *depth_mm, const size_t *stride, int count, hipStream_t s, SrcTab tab) {
    if (n < 1 || n > DVO_LEVELS) return hipErrorInvalidValue;
    CameraLevels t;
    t.n = n; t.src_rows = src_rows; t.src_cols = src_cols;
    t.first[0] = 0;
    for (int l = 0; l < n; l++) {
        t.shift[l] = shift[l]; t.rows[l] = rows[l]; t.cols[l] = cols[l]; t.grey[l] = grey[l]; t.depth[l] = depth_mm[l]; t.stride[l] = stride[l];
        t.first[l + 1] = t.first[l] + (unsigned)(((rows[l] + CAM_TY - 1) / CAM_TY) * ((cols[l] + CAM_TX - 1) / CAM_TX));
    }
    UndistortMaps um{umap_xy, umap_frac, depth_raw};
    hipLaunchKernelGGL(camera_levels_kernel, dim3(t.first[n], count), dim3(256), 0, s, bgr, bgr_stride, depth_m, depth_stride, um, t, tab);
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
/* find with path halving for the tile-local (LDS) forest: every node passed on the way is re-hung below its grandparent, so
 * the chains that column-by-column unions leave behind (as long as the tile is wide) are walked at full length once, not by
 * every later find.  atomicMin keeps the invariant that labels only decrease, whatever races with it. */
DVO_DEV int uf_find_halving(int *L, int a) {
    int p = uf_load(L, a);
    while (p != a) {
        const int gp = uf_load(L, p);
        if (gp != p) atomicMin(L + a, gp);
        a = p; p = gp;
    }
    return a;
}
template <bool HALVE = false>
DVO_DEV void uf_union(int *L, int a, int b) {
    for (;;) {
        a = HALVE ? uf_find_halving(L, a) : uf_find(L, a);
        b = HALVE ? uf_find_halving(L, b) : uf_find(L, b);
        if (a == b) return;
        if (a < b) { const int t = a; a = b; b = t; }            /* a is the larger root: hang it below b */
        const int old = atomicMin(L + a, b);
        if (old == a) return;
        a = old;                                                  /* a had been re-parented meanwhile: go on from there */
    }
}

/* Canny front end, one 64 x 32 pixel tile per workgroup, everything between the grey load and the candidate
 * map in LDS: 3x3 Sobel (BORDER_REPLICATE) -> squared magnitude -> sector non-maximum suppression ->
 * union-find of the tile's candidates -> hysteresis INSIDE the tile.  Out, per pixel, cand:
 *     0                 suppressed
 *     CAND_WEAK (1)     candidate whose tile-local component holds no pixel above `high`: an edge only if a neighbouring
 *                       tile's part of the component does (border / flag / final kernels decide)
 *     CAND_SURE (2)     candidate of a component that holds a pixel above `high` in this tile: an edge, nothing to look up
 *     | CAND_ROOT (16)  the pixel is the root of its tile-local component
 * and for candidates label = global index of the tile-local root; flag = 0 at the roots.  Pairs of candidates in different
 * tiles are joined by canny_border_kernel.
 * Threads walk the tile as (row = tid & 63, column = tid >> 6 + 4k): no divisions; the gradient of a pixel is recomputed from
 * the grey tile for the few pixels above `low` instead of being parked in LDS for all of them (22 KB of LDS per workgroup). */
constexpr int CT_Y = 64, CT_X = 32;
enum { CAND_WEAK = 1, CAND_SURE = 2, CAND_KIND = 3, CAND_ROOT = 16 };
/* LISTS (round 6, the all-levels launch): the tile writes the EDGE MAP itself -- 0 none, 255 a sure edge, EDGE_WEAK a weak candidate still
 * to be decided -- and keeps a RECORD of 256 ints per tile: its weak candidates and the roots of its strong components (global pixel
 * indices, up to CT_WEAK / CT_STRONG of them) plus their two counts.  The passes that follow walk the records (a few per cent of the
 * pixels) instead of reading every pixel's candidate byte and writing every pixel's edge byte.  A tile with more entries than its
 * record holds is looked at pixel by pixel by both passes (correct for any image; never seen on camera frames).  `cand` then IS the
 * edge map (`cand_stride` bytes per image): the border pass only asks whether a pixel is a candidate at all.  (A first form with one
 * list per image -- staging in LDS, a global atomic WITH return per tile -- made the tile kernel 18 % slower: every tile waited a
 * memory round trip at its end.) */
constexpr unsigned char EDGE_WEAK = 1;
constexpr int CT_REC = 256, CT_WEAK = 192, CT_STRONG = CT_REC - CT_WEAK;
struct CannyLists {
    int *cnt;          /* [image][tile][2]: {weak, strong} entries (written by every tile) */
    int *ent;          /* [image][tile][CT_REC]: weak[CT_WEAK], strong[CT_STRONG] */
    int tiles;         /* per image */
    int weak_cap, strong_cap;
};
template <bool LISTS = false>
DVO_DEV void canny_tile_body(const int bx, const int by, const unsigned char *__restrict__ grey, size_t stride, int rows, int cols, int tiles_y,
                             int low, int high, unsigned char *__restrict__ cand, unsigned char *__restrict__ flag, int *__restrict__ label,
                             size_t cand_stride = 0, CannyLists lists = CannyLists{}) {
    constexpr int GH = CT_Y + 4, GW = CT_X + 4, MH = CT_Y + 2, MW = CT_X + 2, NT = CT_Y * CT_X;
    static_assert(CT_Y == 64 && 4 * GW <= 256 && 2 * MW <= 256, "thread mapping: 64 rows per step, the extra halo rows in one more");
    __shared__ unsigned char sg[GW * GH];         /* grey, halo 2, [x][y] */
    __shared__ int smag[MW * MH];                 /* squared magnitude, halo 1 (0 outside the image) */
    __shared__ int slab[NT];
    __shared__ unsigned scand32[NT / 4];          /* one byte per pixel: 0 / 1 candidate / 2 candidate above `high`; bit 2 at a root: strong */
    unsigned char *scand = reinterpret_cast<unsigned char *>(scand32);
    const size_t n = (size_t)rows * cols;
    grey += (size_t)by * stride;
    cand += (size_t)by * (LISTS ? cand_stride : n); flag += (size_t)by * n; label += (size_t)by * n;
    const int ty = bx % tiles_y, tx = bx / tiles_y;
    const int y0 = ty * CT_Y, x0 = tx * CT_X;
    const int tid = threadIdx.x, ry = tid & 63, cx = tid >> 6;
    __shared__ int s_cnt[2];                        /* LISTS: {weak, strong} entries of this tile so far */
    if (LISTS && tid < 2) s_cnt[tid] = 0;

    /* tiles whose halo lies inside the image (all but the rim: 84 % of a VGA frame's) skip the clamps and the bounds tests: this
     * kernel is bound by its vector instructions (round 6: 1128 per wave, the vector unit ~90 % busy) */
    const bool halo_inside = x0 >= 2 && x0 + CT_X + 2 <= cols && y0 >= 2 && y0 + CT_Y + 2 <= rows;      /* workgroup-uniform */
    auto load_grey = [&](int lx, int ly) {
        int gy = y0 + ly - 2, gx = x0 + lx - 2;
        if (!halo_inside) {
            gy = gy < 0 ? 0 : (gy > rows - 1 ? rows - 1 : gy);                  /* BORDER_REPLICATE */
            gx = gx < 0 ? 0 : (gx > cols - 1 ? cols - 1 : gx);
        }
        sg[lx * GH + ly] = grey[(unsigned)(gx * rows + gy)];
    };
#pragma unroll
    for (int k = 0; k < GW / 4; k++) load_grey(cx + 4 * k, ry);
    if (tid < 4 * GW) load_grey(tid >> 2, 64 + (tid & 3));                      /* rows 64..67 of the halo'd tile */
    __syncthreads();
    auto sobel = [&](int lx, int ly, int &dx, int &dy) {                        /* (lx, ly) in smag coordinates (halo 1) */
        const unsigned char *c = sg + (lx + 1) * GH + (ly + 1);                 /* the pixel itself */
        const int a = c[-GH - 1], b = c[-1], cc = c[GH - 1];                    /* row above: x-1, x, x+1 */
        const int d = c[-GH], f = c[GH];
        const int g = c[-GH + 1], h = c[1], i = c[GH + 1];
        dx = (cc - a) + 2 * (f - d) + (i - g);
        dy = (g - a) + 2 * (h - b) + (i - cc);
    };
    /* the tile's own 64 x 32 pixels: gradient kept in registers for the suppression below (round 6; it used to be recomputed there --
     * eight LDS byte reads and fourteen operations per pixel of every wave that holds a candidate, i.e. of every wave) */
    int gdx[CT_X / 4], gdy[CT_X / 4];
#pragma unroll
    for (int k = 0; k < CT_X / 4; k++) {
        const int lx = cx + 4 * k, ly = ry;
        sobel(lx + 1, ly + 1, gdx[k], gdy[k]);
        int m = gdx[k] * gdx[k] + gdy[k] * gdy[k];
        if (!halo_inside && !(y0 + ly < rows && x0 + lx < cols)) m = 0;         /* outside the image (y0 + ly, x0 + lx are never negative) */
        smag[(lx + 1) * MH + (ly + 1)] = m;
    }
    /* the ring around them (magnitudes only): columns -1 and 32 (66 rows each), rows -1 and 64 (32 columns each) */
    if (tid < 2 * MH + 2 * CT_X) {
        int lx, ly;
        if (tid < MH) { lx = 0; ly = tid; }
        else if (tid < 2 * MH) { lx = MW - 1; ly = tid - MH; }
        else if (tid < 2 * MH + CT_X) { lx = tid - 2 * MH + 1; ly = 0; }
        else { lx = tid - 2 * MH - CT_X + 1; ly = MH - 1; }
        const int py = y0 + ly - 1, px = x0 + lx - 1;
        int m = 0;
        if (halo_inside || (py >= 0 && py < rows && px >= 0 && px < cols)) { int dx, dy; sobel(lx, ly, dx, dy); m = dx * dx + dy * dy; }
        smag[lx * MH + ly] = m;
    }
    __syncthreads();
    constexpr int SHIFT = 15;
    constexpr int TG22 = 13573;                                   /* round(tan(22.5 deg) * 2^15) */
#pragma unroll
    for (int k = 0; k < CT_X / 4; k++) {
        const int lx = cx + 4 * k, ly = ry, idx = lx * CT_Y + ly;
        const int mi = (lx + 1) * MH + (ly + 1);
        const int m = smag[mi];
        bool keep = false;
        if (m > low) {                                            /* pixels outside the image carry m = 0 <= low */
            const int xs = gdx[k], ys = gdy[k];
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
        /* a wave holds the 64 rows of one tile column: the vertical runs of candidates are read off the ballot, and a pixel
         * starts out labelled with the top pixel of its run -- no union inside a column ever happens */
        const unsigned long long col = __builtin_amdgcn_ballot_w64(keep);
        const unsigned long long gaps_above = ~col & ((1ull << ly) - 1ull);
        const int top = gaps_above ? 64 - __clzll((long long)gaps_above) : 0;
        slab[idx] = keep ? lx * CT_Y + top : -1;
    }
    __syncthreads();
    /* join the runs of a column with those of the column to its left.  8-connectivity: (lx-1, ly-1 .. ly+1).  One union per pair
     * of touching runs is enough, so a pixel leaves the union to its upper (lower) neighbour in the run whenever that one sees the
     * same left run.  The unions a thread owes are first collected in a bit mask and then worked off in ONE loop: a wave pays
     * the latency of a union as often as its busiest lane has one, not once per (column, case) slot */
    unsigned todo = 0u;
#pragma unroll
    for (int k = 0; k < CT_X / 4; k++) {
        const int lx = cx + 4 * k, ly = ry, idx = lx * CT_Y + ly;
        if (!scand[idx] || lx == 0) continue;
        const int q = idx - CT_Y;
        const bool up = ly > 0 && scand[idx - 1], down = ly < CT_Y - 1 && scand[idx + 1];
        const bool cq = scand[q], cqm = ly > 0 && scand[q - 1], cqp = ly < CT_Y - 1 && scand[q + 1];
        if (cq) {
            if (!(up && cqm)) todo |= 1u << (3 * k);              /* q-1, q, q+1 are one run */
        } else {
            if (cqm && !up) todo |= 2u << (3 * k);                /* `up` has q-1 straight to its left */
            if (cqp && !down) todo |= 4u << (3 * k);              /* `down` has q+1 straight to its left */
        }
    }
    while (todo) {
        const int b = __ffs((int)todo) - 1;
        todo &= todo - 1u;
        const int k = b / 3, which = b - 3 * k;
        const int idx = (cx + 4 * k) * CT_Y + ry;
        uf_union<true>(slab, idx, idx - CT_Y + (which == 0 ? 0 : (which == 1 ? -1 : 1)));
    }
    __syncthreads();
    unsigned strong = 0u;                                         /* a pixel above `high` makes its component's root strong */
#pragma unroll
    for (int k = 0; k < CT_X / 4; k++) if ((scand[(cx + 4 * k) * CT_Y + ry] & 3) == 2) strong |= 1u << k;
    while (strong) {
        const int k = __ffs((int)strong) - 1;
        strong &= strong - 1u;
        const int r = uf_find_halving(slab, (cx + 4 * k) * CT_Y + ry);
        atomicOr(&scand32[r >> 2], 4u << (8 * (r & 3)));
    }
    __syncthreads();
    unsigned wbits = 0u, rbits = 0u;                                  /* LISTS: which of the lane's eight pixels are weak candidates / strong roots */
#pragma unroll
    for (int k = 0; k < CT_X / 4; k++) {
        const int lx = cx + 4 * k, ly = ry, idx = lx * CT_Y + ly;
        const int py = y0 + ly, px = x0 + lx;
        if (py >= rows || px >= cols) continue;
        const unsigned p = (unsigned)(px * rows + py);
        unsigned char c = 0;
        if (scand[idx] & 3) {
            const int r = uf_find_halving(slab, idx);
            const int rx = r >> 6, rr = r & 63;
            const bool sure = (scand[r] & 4) != 0;
            c = (unsigned char)((sure ? CAND_SURE : CAND_WEAK) | (r == idx ? CAND_ROOT : 0));
            label[p] = (x0 + rx) * rows + (y0 + rr);
            if (r == idx) flag[p] = 0;
            if constexpr (LISTS) {
                c = sure ? 255 : EDGE_WEAK;
                wbits |= (sure ? 0u : 1u) << k;
                rbits |= ((sure && r == idx) ? 1u : 0u) << k;
            }
        }
        cand[p] = c;
    }
    if constexpr (LISTS) {
        /* a lane's entries go to the tile's record at places it draws from two LDS counters -- one atomic per lane THAT HAS entries and list
         * (a first form drew places per column from wave ballots: +41 us on the 256-frame step; this kernel is bound by its vector
         * instructions) -- and the counts are written by one lane behind a barrier: the record needs no zeroing and no global atomic */
        int *rec = lists.ent + ((size_t)by * lists.tiles + bx) * CT_REC;
        if (wbits) {
            int at = atomicAdd(&s_cnt[0], __popc(wbits));
            for (unsigned m = wbits; m; m &= m - 1u, at++) {
                const int k = __ffs((int)m) - 1;
                if (at < lists.weak_cap) rec[at] = (x0 + cx + 4 * k) * rows + (y0 + ry);
            }
        }
        if (rbits) {
            int at = atomicAdd(&s_cnt[1], __popc(rbits));
            for (unsigned m = rbits; m; m &= m - 1u, at++) {
                const int k = __ffs((int)m) - 1;
                if (at < lists.strong_cap) rec[CT_WEAK + at] = (x0 + cx + 4 * k) * rows + (y0 + ry);
            }
        }
        __syncthreads();
        if (tid < 2) lists.cnt[((size_t)by * lists.tiles + bx) * 2 + tid] = s_cnt[tid];
    }
}

/* LISTS: the two passes after the border unions, one wave per tile record. */
DVO_DEV void canny_flag_list_body(const int bx, const int by, const unsigned char *__restrict__ edge, size_t edge_stride, int rows, int cols,
                                  const int *__restrict__ label, unsigned char *__restrict__ flag, CannyLists lists) {
    const size_t n = (size_t)rows * cols;
    label += (size_t)by * n; flag += (size_t)by * n;
    const int tile = bx * 4 + (int)(threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (tile >= lists.tiles) return;
    const int *rec = lists.ent + ((size_t)by * lists.tiles + tile) * CT_REC;
    const int cnt = lists.cnt[((size_t)by * lists.tiles + tile) * 2 + 1];
    if (cnt <= lists.strong_cap) {
        for (int i = lane; i < cnt; i += 64) flag[uf_find(label, rec[CT_WEAK + i])] = 1;
    } else {                                                       /* every sure edge of the tile speaks for its component */
        edge += (size_t)by * edge_stride;
        const int tiles_y = (rows + CT_Y - 1) / CT_Y, y0 = (tile % tiles_y) * CT_Y, x0 = (tile / tiles_y) * CT_X;
        for (int lx = 0; lx < CT_X; lx++) {
            const int py = y0 + lane, px = x0 + lx;
            if (py < rows && px < cols && edge[(size_t)px * rows + py] == 255) flag[uf_find(label, px * rows + py)] = 1;
        }
    }
}
DVO_DEV void canny_weak_list_body(const int bx, const int by, unsigned char *__restrict__ edge, size_t edge_stride, int rows, int cols,
                                  const int *__restrict__ label, const unsigned char *__restrict__ flag, CannyLists lists) {
    const size_t n = (size_t)rows * cols;
    label += (size_t)by * n; flag += (size_t)by * n; edge += (size_t)by * edge_stride;
    const int tile = bx * 4 + (int)(threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (tile >= lists.tiles) return;
    const int *rec = lists.ent + ((size_t)by * lists.tiles + tile) * CT_REC;
    const int cnt = lists.cnt[((size_t)by * lists.tiles + tile) * 2 + 0];
    if (cnt <= lists.weak_cap) {
        for (int i = lane; i < cnt; i += 64) {
            const int p = rec[i];
            edge[p] = flag[uf_find(label, p)] ? 255 : 0;
        }
    } else {
        const int tiles_y = (rows + CT_Y - 1) / CT_Y, y0 = (tile % tiles_y) * CT_Y, x0 = (tile / tiles_y) * CT_X;
        for (int lx = 0; lx < CT_X; lx++) {
            const int py = y0 + lane, px = x0 + lx;
            if (py >= rows || px >= cols) continue;
            const size_t p = (size_t)px * rows + py;
            if (edge[p] == EDGE_WEAK) edge[p] = flag[uf_find(label, (int)p)] ? 255 : 0;
        }
    }
}

/* candidate pairs that straddle a tile boundary: rows r = 64, 128, ... looking up, columns c = 32, 64, ... looking left */
DVO_DEV void canny_border_body(const int bx, const int gx, const int by, const unsigned char *__restrict__ cand, int rows, int cols, int *__restrict__ label,
                               size_t cand_stride = 0 /* bytes per image when the candidates are read off the edge map (LISTS) */) {
    const size_t n = (size_t)rows * cols;
    cand += (size_t)by * (cand_stride ? cand_stride : n); label += (size_t)by * n;
    const int nA = ((rows - 1) / CT_Y) * cols, nB = ((cols - 1) / CT_X) * rows;
    for (int i = bx * blockDim.x + threadIdx.x; i < nA + nB; i += gx * blockDim.x) {
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
        if (!(c & CAND_ROOT)) continue;     /* tile-local roots speak for their components */
        const int r = uf_find(label, (int)p);
        label[p] = r;                       /* racing writers only ever store ancestors: find() stays correct */
        if ((c & CAND_KIND) == CAND_SURE) flag[r] = 1;
    }
}

__global__ void __launch_bounds__(256)
canny_final_kernel(const unsigned char *__restrict__ cand, const int *__restrict__ label, const unsigned char *__restrict__ flag,
                   size_t n, unsigned char *__restrict__ edge, size_t edge_stride) {
    cand += (size_t)blockIdx.y * n; label += (size_t)blockIdx.y * n; flag += (size_t)blockIdx.y * n;
    edge += (size_t)blockIdx.y * edge_stride;
    for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (size_t)gridDim.x * blockDim.x) {
        const unsigned char c = cand[p];
        bool e = (c & CAND_KIND) == CAND_SURE;
        if ((c & CAND_KIND) == CAND_WEAK) e = flag[uf_find(label, (int)p)] != 0;
        edge[p] = e ? 255 : 0;
    }
}

/* the same two passes, four pixels per thread: candidates are a few percent of the pixels, so most threads see one zero word
 * (images whose pixel count is a multiple of four -- every camera format) */
DVO_DEV void canny_flag4_body(const int bx, const int gx, const int by, const unsigned char *__restrict__ cand, size_t n, int *__restrict__ label, unsigned char *__restrict__ flag) {
    cand += (size_t)by * n; label += (size_t)by * n; flag += (size_t)by * n;
    const unsigned *cand4 = reinterpret_cast<const unsigned *>(cand);
    for (size_t q = (size_t)bx * blockDim.x + threadIdx.x; q < n / 4; q += (size_t)gx * blockDim.x) {
        unsigned w = cand4[q];
        for (int k = 0; w; k++, w >>= 8) {
            const unsigned c = w & 0xffu;
            if (!(c & CAND_ROOT)) continue; /* tile-local roots speak for their components */
            const int p = (int)(4 * q) + k;
            const int r = uf_find(label, p);
            label[p] = r;                   /* racing writers only ever store ancestors: find() stays correct */
            if ((c & CAND_KIND) == CAND_SURE) flag[r] = 1;
        }
    }
}
DVO_DEV void canny_final4_body(const int bx, const int gx, const int by, const unsigned char *__restrict__ cand, const int *__restrict__ label,
                               const unsigned char *__restrict__ flag, size_t n, unsigned char *__restrict__ edge, size_t edge_stride) {
    cand += (size_t)by * n; label += (size_t)by * n; flag += (size_t)by * n;
    edge += (size_t)by * edge_stride;
    const unsigned *cand4 = reinterpret_cast<const unsigned *>(cand);
    unsigned *edge4 = reinterpret_cast<unsigned *>(edge);
    for (size_t q = (size_t)bx * blockDim.x + threadIdx.x; q < n / 4; q += (size_t)gx * blockDim.x) {
        unsigned w = cand4[q], out = 0u;
        for (int k = 0; w; k++, w >>= 8) {
            const unsigned kind = w & CAND_KIND;
            if (kind == CAND_SURE || (kind == CAND_WEAK && flag[uf_find(label, (int)(4 * q) + k)] != 0)) out |= 0xffu << (8 * k);
        }
        edge4[q] = out;
    }
}

/* round 6: the same two passes, SIXTEEN pixels per thread (one 16-byte load; images whose pixel count is a multiple of sixteen).
 * The four-pixel forms issued one dword load per thread and took 84 + 144 us per 256 four-level 640x480 frames -- 1.4 TB/s for
 * passes that move 1 and 2 bytes per pixel. */
DVO_DEV void canny_flag16_body(const int bx, const int gx, const int by, const unsigned char *__restrict__ cand, size_t n, int *__restrict__ label, unsigned char *__restrict__ flag) {
    cand += (size_t)by * n; label += (size_t)by * n; flag += (size_t)by * n;
    const uint4 *cand16 = reinterpret_cast<const uint4 *>(cand);
    for (size_t q = (size_t)bx * blockDim.x + threadIdx.x; q < n / 16; q += (size_t)gx * blockDim.x) {
        const uint4 v = cand16[q];
        const unsigned ws[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; j++) {
            unsigned w = ws[j];
            if (!(w & (0x01010101u * CAND_ROOT))) continue;
            for (int k = 0; w; k++, w >>= 8) {
                const unsigned c = w & 0xffu;
                if (!(c & CAND_ROOT)) continue; /* tile-local roots speak for their components */
                const int p = (int)(16 * q) + 4 * j + k;
                const int r = uf_find(label, p);
                label[p] = r;                   /* racing writers only ever store ancestors: find() stays correct */
                if ((c & CAND_KIND) == CAND_SURE) flag[r] = 1;
            }
        }
    }
}
DVO_DEV void canny_final16_body(const int bx, const int gx, const int by, const unsigned char *__restrict__ cand, const int *__restrict__ label,
                                const unsigned char *__restrict__ flag, size_t n, unsigned char *__restrict__ edge, size_t edge_stride) {
    cand += (size_t)by * n; label += (size_t)by * n; flag += (size_t)by * n;
    edge += (size_t)by * edge_stride;
    const uint4 *cand16 = reinterpret_cast<const uint4 *>(cand);
    uint4 *edge16 = reinterpret_cast<uint4 *>(edge);
    for (size_t q = (size_t)bx * blockDim.x + threadIdx.x; q < n / 16; q += (size_t)gx * blockDim.x) {
        const uint4 v = cand16[q];
        const unsigned ws[4] = {v.x, v.y, v.z, v.w};
        unsigned o[4], weak = 0u;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const unsigned w = ws[j];
            o[j] = ((w >> 1) & ~w & 0x01010101u) * 0xffu;          /* kind 2: surely an edge */
            const unsigned t = w & ~(w >> 1) & 0x01010101u;         /* kind 1: ask the component */
            weak |= ((t | (t >> 7) | (t >> 14) | (t >> 21)) & 0xfu) << (4 * j);
        }
        if (weak) {
            /* After the flag pass a candidate's label is its tile-local root and THAT pixel's label the root of the whole component:
             * the answer is flag[label[label[p]]], three loads deep for every weak candidate of the thread AT ONCE (the loop of the
             * four-pixel form walked the positions one after the other: a wave paid a dependent chain per position that held a weak
             * candidate in any of its lanes). */
            const int p0 = (int)(16 * q);
            int a[16];
#pragma unroll
            for (int i = 0; i < 16; i++) a[i] = label[((weak >> i) & 1u) ? p0 + i : p0];
#pragma unroll
            for (int i = 0; i < 16; i++) a[i] = label[((weak >> i) & 1u) ? a[i] : p0];
            unsigned e = 0u;
#pragma unroll
            for (int i = 0; i < 16; i++) e |= (unsigned)(flag[((weak >> i) & 1u) ? a[i] : p0] != 0) << i;
            e &= weak;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const unsigned b4 = (e >> (4 * j)) & 0xfu;
                o[j] |= ((b4 & 1u) | ((b4 & 2u) << 7) | ((b4 & 4u) << 14) | ((b4 & 8u) << 21)) * 0xffu;
            }
        }
        edge16[q] = make_uint4(o[0], o[1], o[2], o[3]);
    }
}

__global__ void __launch_bounds__(256)
canny_tile_kernel(const unsigned char *__restrict__ grey, size_t stride, int rows, int cols, int tiles_y, int low, int high,
                  unsigned char *__restrict__ cand, unsigned char *__restrict__ flag, int *__restrict__ label) {
    canny_tile_body(blockIdx.x, blockIdx.y, grey, stride, rows, cols, tiles_y, low, high, cand, flag, label);
}
__global__ void __launch_bounds__(256)
canny_border_kernel(const unsigned char *__restrict__ cand, int rows, int cols, int *__restrict__ label) {
    canny_border_body(blockIdx.x, gridDim.x, blockIdx.y, cand, rows, cols, label);
}
__global__ void __launch_bounds__(256)
canny_flag4_kernel(const unsigned char *__restrict__ cand, size_t n, int *__restrict__ label, unsigned char *__restrict__ flag) {
    canny_flag4_body(blockIdx.x, gridDim.x, blockIdx.y, cand, n, label, flag);
}
__global__ void __launch_bounds__(256)
canny_final4_kernel(const unsigned char *__restrict__ cand, const int *__restrict__ label, const unsigned char *__restrict__ flag,
                    size_t n, unsigned char *__restrict__ edge, size_t edge_stride) {
    canny_final4_body(blockIdx.x, gridDim.x, blockIdx.y, cand, label, flag, n, edge, edge_stride);
}

/* ALL PYRAMID LEVELS OF A STAGE IN ONE LAUNCH.  The levels below the first hold a quarter, a sixteenth, ... of its pixels: their
 * own launches cost the fixed ~10 us each and leave the GPU mostly empty, which a single camera stream pays per frame and the
 * chunked upload pipeline per chunk.  The grid is the concatenation of the levels' grids (first[l] = first block of level l); a
 * workgroup looks its level up in the table it gets as kernel argument and runs that level's unchanged body. */
struct CannyLevels {
    int n, low, high;
    int rows[DVO_LEVELS], cols[DVO_LEVELS];
    unsigned first[DVO_LEVELS + 1];
    const unsigned char *grey[DVO_LEVELS]; size_t grey_stride[DVO_LEVELS];
    unsigned char *cand[DVO_LEVELS], *flag[DVO_LEVELS]; int *label[DVO_LEVELS];
    unsigned char *edge[DVO_LEVELS]; size_t edge_stride[DVO_LEVELS];
    CannyLists lists[DVO_LEVELS];
};
DVO_DEV int level_of_block(const unsigned *first, int n, unsigned bx) {
    int l = 0;
    while (l + 1 < n && bx >= first[l + 1]) l++;
    return l;
}
__global__ void __launch_bounds__(256) canny_tile_levels_kernel(const CannyLevels t) {
    const int l = level_of_block(t.first, t.n, blockIdx.x);
    canny_tile_body((int)(blockIdx.x - t.first[l]), blockIdx.y, t.grey[l], t.grey_stride[l], t.rows[l], t.cols[l], (t.rows[l] + CT_Y - 1) / CT_Y,
                    t.low, t.high, t.cand[l], t.flag[l], t.label[l]);
}
__global__ void __launch_bounds__(256) canny_border_levels_kernel(const CannyLevels t) {
    const int l = level_of_block(t.first, t.n, blockIdx.x);
    canny_border_body((int)(blockIdx.x - t.first[l]), (int)(t.first[l + 1] - t.first[l]), blockIdx.y, t.cand[l], t.rows[l], t.cols[l], t.label[l]);
}
__global__ void __launch_bounds__(256) canny_flag4_levels_kernel(const CannyLevels t) {
    const int l = level_of_block(t.first, t.n, blockIdx.x);
    canny_flag4_body((int)(blockIdx.x - t.first[l]), (int)(t.first[l + 1] - t.first[l]), blockIdx.y, t.cand[l], (size_t)t.rows[l] * t.cols[l], t.label[l], t.flag[l]);
}
__global__ void __launch_bounds__(256) canny_final4_levels_kernel(const CannyLevels t) {
    const int l = level_of_block(t.first, t.n, blockIdx.x);
    canny_final4_body((int)(blockIdx.x - t.first[l]), (int)(t.first[l + 1] - t.first[l]), blockIdx.y, t.cand[l], t.label[l], t.flag[l],
                      (size_t)t.rows[l] * t.cols[l], t.edge[l], t.edge_stride[l]);
}

__global__ void __launch_bounds__(256) canny_tile_lists_levels_kernel(const CannyLevels t) {
    const int l = level_of_block(t.first, t.n, blockIdx.x);
    canny_tile_body<true>((int)(blockIdx.x - t.first[l]), blockIdx.y, t.grey[l], t.grey_stride[l], t.rows[l], t.cols[l], (t.rows[l] + CT_Y - 1) / CT_Y,
                          t.low, t.high, t.edge[l], t.flag[l], t.label[l], t.edge_stride[l], t.lists[l]);
}
__global__ void __launch_bounds__(256) canny_border_lists_levels_kernel(const CannyLevels t) {
    const int l = level_of_block(t.first, t.n, blockIdx.x);
    canny_border_body((int)(blockIdx.x - t.first[l]), (int)(t.first[l + 1] - t.first[l]), blockIdx.y, t.edge[l], t.rows[l], t.cols[l], t.label[l], t.edge_stride[l]);
}
__global__ void __launch_bounds__(256) canny_flag_list_levels_kernel(const CannyLevels t) {
    const int l = level_of_block(t.first, t.n, blockIdx.x);
    canny_flag_list_body((int)(blockIdx.x - t.first[l]), blockIdx.y, t.edge[l], t.edge_stride[l], t.rows[l], t.cols[l], t.label[l], t.flag[l], t.lists[l]);
}
__global__ void __launch_bounds__(256) canny_weak_list_levels_kernel(const CannyLevels t) {
    const int l = level_of_block(t.first, t.n, blockIdx.x);
    canny_weak_list_body((int)(blockIdx.x - t.first[l]), blockIdx.y, t.edge[l], t.edge_stride[l], t.rows[l], t.cols[l], t.label[l], t.flag[l], t.lists[l]);
}

__global__ void __launch_bounds__(256) canny_flag16_levels_kernel(const CannyLevels t) {
    const int l = level_of_block(t.first, t.n, blockIdx.x);
    canny_flag16_body((int)(blockIdx.x - t.first[l]), (int)(t.first[l + 1] - t.first[l]), blockIdx.y, t.cand[l], (size_t)t.rows[l] * t.cols[l], t.label[l], t.flag[l]);
}
__global__ void __launch_bounds__(256) canny_final16_levels_kernel(const CannyLevels t) {
    const int l = level_of_block(t.first, t.n, blockIdx.x);
    canny_final16_body((int)(blockIdx.x - t.first[l]), (int)(t.first[l + 1] - t.first[l]), blockIdx.y, t.cand[l], t.label[l], t.flag[l],
                       (size_t)t.rows[l] * t.cols[l], t.edge[l], t.edge_stride[l]);
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
    if ((n & 3) == 0 && (edge_stride & 3) == 0 && (reinterpret_cast<size_t>(edge) & 3) == 0) {
        const dim3 grid4(grid_x(n / 4), g.count);
        hipLaunchKernelGGL(canny_flag4_kernel, grid4, blk, 0, s, cand, n, label, flag);
        hipLaunchKernelGGL(canny_final4_kernel, grid4, blk, 0, s, cand, label, flag, n, edge, edge_stride);
    } else {
        hipLaunchKernelGGL(canny_flag_kernel, grid, blk, 0, s, cand, n, label, flag);
        hipLaunchKernelGGL(canny_final_kernel, grid, blk, 0, s, cand, label, flag, n, edge, edge_stride);
    }
    return hipGetLastError();
}

/* Canny of `n` levels of the same `count` images in four launches.  work: the levels' scratch back to back (canny_work_ints each,
 * rounded to 4 ints).  False if a level does not meet the four-pixels-per-thread conditions: the caller then launches per level. */
/* + per level and image: a record of CT_REC entries and two counters per tile (the all-levels launch's lists) */
static inline size_t canny_tiles(int rows, int cols) { return (size_t)((rows + CT_Y - 1) / CT_Y) * ((cols + CT_X - 1) / CT_X); }
size_t canny_levels_work_ints(int n, const int *rows, const int *cols, int count) {
    size_t t = 0;
    for (int l = 0; l < n; l++) t += (canny_work_ints(rows[l], cols[l], count) + 3) / 4 * 4 + canny_tiles(rows[l], cols[l]) * (CT_REC + 2) * (size_t)count;
    return t;
}
bool canny_levels_ok(int n, const int *rows, const int *cols, unsigned char *const *edge, const size_t *edge_stride) {
    if (n < 2 || n > DVO_LEVELS) return false;
    for (int l = 0; l < n; l++) {
        const size_t px = (size_t)rows[l] * cols[l];
        if ((px & 3) || (edge_stride[l] & 3) || (reinterpret_cast<size_t>(edge[l]) & 3)) return false;
    }
    return true;
}
hipError_t launch_canny_levels(int n, const int *rows, const int *cols, const unsigned char *const *grey, const size_t *grey_stride,
                               unsigned char *const *edge, const size_t *edge_stride, int count, int low, int high, int *work, hipStream_t s) {
    CannyLevels t;
    t.n = n; t.low = low; t.high = high;
    int *w = work;
    for (int l = 0; l < n; l++) {
        const size_t nb = (size_t)rows[l] * cols[l] * count;
        t.rows[l] = rows[l]; t.cols[l] = cols[l];
        t.grey[l] = grey[l]; t.grey_stride[l] = grey_stride[l]; t.edge[l] = edge[l]; t.edge_stride[l] = edge_stride[l];
        t.label[l] = w;
        t.cand[l] = reinterpret_cast<unsigned char *>(w + nb);
        t.flag[l] = t.cand[l] + ((nb + 3) / 4) * 4;
        w += (canny_work_ints(rows[l], cols[l], count) + 3) / 4 * 4;
        /* LISTS: the tiles' records behind the level's scratch; the counters of ALL levels together at the very end (one fill) */
        const char *shrink_env = getenv("DVO_CANNY_LIST_SHRINK");      /* tests; read at every call: capacities divided by k, so that
                                                                          ordinary tiles overflow their records and take the dense form */
        const int shrink = shrink_env && atoi(shrink_env) > 1 ? atoi(shrink_env) : 1;
        t.lists[l].tiles = (int)canny_tiles(rows[l], cols[l]);
        t.lists[l].ent = w;
        t.lists[l].weak_cap = CT_WEAK / shrink;
        t.lists[l].strong_cap = CT_STRONG / shrink;
        w += canny_tiles(rows[l], cols[l]) * CT_REC * (size_t)count;
    }
    size_t cnt_ints = 0;
    for (int l = 0; l < n; l++) { t.lists[l].cnt = w + cnt_ints; cnt_ints += canny_tiles(rows[l], cols[l]) * 2 * (size_t)count; }
    auto prefix = [&](auto blocks_of) { t.first[0] = 0; for (int l = 0; l < n; l++) t.first[l + 1] = t.first[l] + blocks_of(l); return t.first[n]; };
    const dim3 blk(256);
    static const bool lists_off = [] { const char *e = getenv("DVO_CANNY_LISTS"); return e && !strcmp(e, "off"); }();
    if (!lists_off) {
        unsigned g = prefix([&](int l) { return (unsigned)(((rows[l] + CT_Y - 1) / CT_Y) * ((cols[l] + CT_X - 1) / CT_X)); });
        hipLaunchKernelGGL(canny_tile_lists_levels_kernel, dim3(g, count), blk, 0, s, t);
        g = prefix([&](int l) { const int nb = ((rows[l] - 1) / CT_Y) * cols[l] + ((cols[l] - 1) / CT_X) * rows[l]; return nb > 0 ? grid_x((size_t)nb) : 0u; });
        if (g) hipLaunchKernelGGL(canny_border_lists_levels_kernel, dim3(g, count), blk, 0, s, t);
        g = prefix([&](int l) { return (unsigned)((canny_tiles(rows[l], cols[l]) + 3) / 4); });      /* one wave per tile record */
        hipLaunchKernelGGL(canny_flag_list_levels_kernel, dim3(g, count), blk, 0, s, t);
        hipLaunchKernelGGL(canny_weak_list_levels_kernel, dim3(g, count), blk, 0, s, t);
        return hipGetLastError();
    }
    unsigned g = prefix([&](int l) { return (unsigned)(((rows[l] + CT_Y - 1) / CT_Y) * ((cols[l] + CT_X - 1) / CT_X)); });
    hipLaunchKernelGGL(canny_tile_levels_kernel, dim3(g, count), blk, 0, s, t);
    g = prefix([&](int l) { const int nb = ((rows[l] - 1) / CT_Y) * cols[l] + ((cols[l] - 1) / CT_X) * rows[l]; return nb > 0 ? grid_x((size_t)nb) : 0u; });
    if (g) hipLaunchKernelGGL(canny_border_levels_kernel, dim3(g, count), blk, 0, s, t);
    bool wide = true;                                          /* sixteen pixels per thread where every level allows it */
    for (int l = 0; l < n; l++) {
        const size_t px = (size_t)rows[l] * cols[l];
        wide = wide && (px & 15) == 0 && (edge_stride[l] & 15) == 0 && (reinterpret_cast<size_t>(edge[l]) & 15) == 0 && (reinterpret_cast<size_t>(t.cand[l]) & 15) == 0;
    }
    if (wide) {
        g = prefix([&](int l) { return grid_x((size_t)rows[l] * cols[l] / 16); });
        hipLaunchKernelGGL(canny_flag16_levels_kernel, dim3(g, count), blk, 0, s, t);
        hipLaunchKernelGGL(canny_final16_levels_kernel, dim3(g, count), blk, 0, s, t);
        return hipGetLastError();
    }
    g = prefix([&](int l) { return grid_x((size_t)rows[l] * cols[l] / 4); });
    hipLaunchKernelGGL(canny_flag4_levels_kernel, dim3(g, count), blk, 0, s, t);
    hipLaunchKernelGGL(canny_final4_levels_kernel, dim3(g, count), blk, 0, s, t);
    return hipGetLastError();
}

/* ------------------------------------------------------------------------- */
/* now-frame preprocessing after Canny: computeDistTransfrmOfNow (SolveDVO.cpp:1768-1795) +
 * imageGradient (:1063-1098).  edge mask -> exact squared EDT in integers (two separable passes)
 * -> the level's COMPACT form (dvo_palette.h), natively:
 *
 *   DT = (float)sqrt(d2) * scale (cv::normalize NORM_MINMAX, :1774) is a non-decreasing function of the integer squared
 *   distance d2, gx / gy are central differences of DT with a reflect-101 border (:1077-1090) and w = getWeightOf(DT)
 *   (:1047-1053).  So the sorted list of the d2 values PRESENT in the image is the palette, and a pixel is the rank of its
 *   d2 in that list.  The row pass records the values it produces in a presence bitmap (LDS, merged into HBM per workgroup);
 *   the pack pass turns bitmap + prefix popcounts into ranks by one table look-up per pixel and writes the 4-byte rank words.
 *   No hashing, no sorting, no verification pass -- the palette value, the weight and the gradients are produced by the very
 *   expressions the 16-byte texel path uses (0.5*a - 0.5*b == 0.5*(a - b) bit for bit: scaling by 0.5 is exact), and the
 *   16-byte texels {DT, gx, gy, w} are not written at all (17 instead of 45 bytes of HBM traffic per pixel); the inspection and
 *   host-driven paths decode them from the compact form on demand (p4_decode_texels_kernel).
 *
 *   Images the compact form cannot hold completely -- more than DVO_PAL_MAX - 2 distinct distances, a pixel further than 511
 *   pixels from every edge, a rank step between horizontal neighbours beyond +-127 -- get a PARTIAL compact form (round 5,
 *   dvo_palette.h: the pixels it cannot express carry the rank of a NaN palette entry) AND their 16-byte texels from the same d2
 *   (dt_normalize_gradient_pack_kernel, launched with a per-image predicate); pal_n = count | DVO_PAL_PARTIAL.  Float images
 *   handed in directly keep the all-or-nothing rule (pal_n = -reason).                                                       */
/* ------------------------------------------------------------------------- */
#define DVO_EDT_INF(rows, cols) ((rows) + (cols) + 1)
enum { EDT_FLAG_FAR = 1, EDT_FLAG_STEP = 2, EDT_FLAG_BAD = 4 /* float images only: not an exact distance transform */,
       EDT_FLAG_PARTIAL = 8 /* the rank-pack pass wrote the NaN rank somewhere: a partial compact form (dvo_palette.h) */ };

/* 32-bit words of one image's presence bitmap: every possible d2 of a small image, distances below 512 pixels otherwise */
static inline int edt_bitmap_words(int rows, int cols) {
    const long long inf = DVO_EDT_INF(rows, cols);
    long long bits = inf * inf + 1;
    if (bits > DVO_EDT_BITMAP_BITS) bits = DVO_EDT_BITMAP_BITS;
    return (int)((bits + 31) / 32);
}

/* g, the intermediate between the two passes, is stored in ROW BLOCKS of R rows (R = the row pass's rows per workgroup):
 * [block][column][R rows] -- the column pass writes 16 contiguous bytes per lane, and a workgroup of the row pass reads its
 * whole tile as one contiguous chunk in the very layout its LDS tile has.  Rows past the image (the last block's padding) hold 0. */
__host__ __device__ inline size_t edt_g_index(int xx, int yy, int cols, int R) { return ((size_t)(yy / R) * cols + xx) * R + (yy & (R - 1)); }   /* R: a power of two */
__host__ __device__ inline size_t edt_g_count(int rows, int cols, int R) { return (size_t)((rows + R - 1) / R) * R * cols; }

/* phase 1: per column, distance to the nearest edge pixel of that column (16 bits: <= rows+cols+1 < 46341).  One wave per
 * column, EIGHT rows per lane (512 rows per step: a whole column of every ordinary image at once): each lane turns its eight
 * edge bytes into a bit mask; the nearest edge above / below a lane's rows outside the lane comes from the ballot of the
 * non-empty masks and one cross-lane read of that lane's mask; inside the lane the distances are counted along the eight rows.
 * The downward pass parks its result in LDS (each lane reads back what it wrote itself), so HBM sees one byte read and two
 * bytes written per pixel.  The launch also clears the image's presence bitmap and flags for the row pass. */
DVO_DEV unsigned edt_nonzero_bytes(unsigned v) {           /* bit k = (byte k of v != 0), k = 0..3 */
    const unsigned t = ((((v & 0x7f7f7f7fu) + 0x7f7f7f7fu) | v) & 0x80808080u) >> 7;
    return (t | (t >> 7) | (t >> 14) | (t >> 21)) & 0xfu;
}
/* per 8-bit edge mask of a lane: the distance of each of its eight rows to the nearest set bit of the mask, 16 bits each (rows j,
 * j + 1 share a dword), 0xffff = no edge among the eight rows */
struct EdtColLut { unsigned v[256][4]; };
constexpr EdtColLut edt_col_lut_make() {
    EdtColLut t{};
    for (int m = 0; m < 256; m++)
        for (int r = 0; r < 8; r++) {
            unsigned best = 0xffffu;
            for (int k = 0; k < 8; k++) if ((m >> k) & 1) { const unsigned a = (unsigned)(r > k ? r - k : k - r); best = a < best ? a : best; }
            t.v[m][r >> 1] |= best << ((r & 1) * 16);
        }
    return t;
}
__device__ const EdtColLut EDT_COL_LUT = edt_col_lut_make();
typedef unsigned short edt_col_us2 __attribute__((ext_vector_type(2)));
DVO_DEV edt_col_us2 edt_col_pair(unsigned v) { return __builtin_bit_cast(edt_col_us2, v); }
DVO_DEV unsigned edt_col_word(edt_col_us2 v) { return __builtin_bit_cast(unsigned, v); }
template <int WAVES>
DVO_DEV void edt_columns8_body(const int bx, const int gx, const int by, const unsigned char *__restrict__ edge, size_t edge_stride, int rows, int cols, int R, unsigned short *__restrict__ g,
                    unsigned *__restrict__ bitmap, int bm_words, int *__restrict__ flags, const bool keep_flags = false /* round 6: the band stage set them */) {
    /* Round 4: ONE pass per 512-row chunk.  The distance of a lane's row j to the nearest edge of its column is the minimum of
     * three: the nearest edge among the lane's own eight rows (a 256-entry table in LDS: per mask, eight 16-bit distances, 0xffff
     * = none), the nearest edge above the lane's rows (d_up + 1 + j) and the nearest below them (d_dn + 8 - j) -- the last two as
     * packed 16-bit additions / minima, two rows per instruction, already in the layout of the store.  (Rounds 1-3 walked the
     * eight rows twice with a running counter, up and down, parking the upward pass in LDS: 312 vector instructions per wave and
     * chunk, the kernel was bound by their issue.)  Columns taller than 512 rows first collect, per chunk, the distance from the
     * chunk's borders to the nearest edge outside it (two sweeps of ballots, wave-uniform). */
    extern __shared__ uint4 s_da8[];                    /* per wave: 2 * nchunk ints, rounded up to 16 bytes (the launchers size it) */
    __shared__ uint4 s_lut[256];
    __shared__ uint4 s_tile[WAVES == 8 ? 4 * 64 * 9 : 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    {
        unsigned *bm = bitmap + (size_t)by * bm_words;
        for (int i = bx * (WAVES * 64) + threadIdx.x; i < bm_words; i += gx * (WAVES * 64)) bm[i] = 0u;
        if (bx == 0 && threadIdx.x == 0 && !keep_flags) flags[by] = 0;
    }
    for (int m = threadIdx.x; m < 256; m += WAVES * 64)
        s_lut[m] = make_uint4(EDT_COL_LUT.v[m][0], EDT_COL_LUT.v[m][1], EDT_COL_LUT.v[m][2], EDT_COL_LUT.v[m][3]);
    __syncthreads();
    g += (size_t)by * edt_g_count(rows, cols, R);
    const int rows_pad = ((rows + R - 1) / R) * R;
    const int nchunk = (rows + 511) / 512;
    int *carries = reinterpret_cast<int *>(s_da8) + (size_t)wave * (((size_t)2 * nchunk + 3) & ~(size_t)3);
    const int INF = DVO_EDT_INF(rows, cols);
    const unsigned inf2 = (unsigned)INF | ((unsigned)INF << 16);
    const unsigned char *img = edge + (size_t)by * edge_stride;
    auto load_mask = [&](const unsigned char *col, bool vec, int y0) -> unsigned {             /* bit j = edge at row y0 + j */
        if (y0 >= rows) return 0u;
        if (vec) {
            const uint2 v = *reinterpret_cast<const uint2 *>(col + y0);
            return edt_nonzero_bytes(v.x) | (edt_nonzero_bytes(v.y) << 4);
        }
        unsigned m = 0;
        for (int j = 0; j < 8; j++) if (y0 + j < rows && col[y0 + j] != 0) m |= 1u << j;
        return m;
    };
    /* the eight rows of this lane in chunk c of column xx from their mask: three candidates per row, packed minima, one store */
    auto emit = [&](int xx, int c, unsigned m8, uint4 *to_lds) {
        const int y0 = c * 512 + lane * 8;
        const unsigned long long bal = __ballot(m8 != 0u);
        const unsigned long long lower = bal & ((1ull << lane) - 1ull);                          /* lanes above these rows */
        const unsigned long long upper = (lane == 63) ? 0ull : (bal & (~0ull << (lane + 1)));  /* lanes below these rows */
        const int lpu = lower ? 63 - __clzll((long long)lower) : lane;
        const int lpd = upper ? __ffsll((long long)upper) - 1 : lane;
        const unsigned mlu = (unsigned)__shfl((int)m8, lpu);   /* every lane takes part: no cross-lane read under a branch */
        const unsigned mld = (unsigned)__shfl((int)m8, lpd);
        const int cu = nchunk > 1 ? carries[c] : INF, cd = nchunk > 1 ? carries[nchunk + c] : INF;
        int d_up, d_dn;                                     /* from row y0 - 1 upwards / from row y0 + 8 downwards to the nearest edge */
        if (lower) d_up = (lane * 8 - 1) - (lpu * 8 + (31 - __clz((int)mlu)));
        else d_up = (cu + lane * 8 > INF) ? INF : cu + lane * 8;
        if (upper) d_dn = (lpd * 8 + (__ffs((int)mld) - 1)) - (lane * 8 + 8);
        else d_dn = (cd - 1 + (63 - lane) * 8 > INF) ? INF : cd - 1 + (63 - lane) * 8;
        const uint4 in = s_lut[m8];
        const edt_col_us2 up2 = edt_col_pair((unsigned)d_up | ((unsigned)d_up << 16)), dn2 = edt_col_pair((unsigned)d_dn | ((unsigned)d_dn << 16));
        const edt_col_us2 lim = edt_col_pair(inf2);
        auto rows2 = [&](unsigned inside, unsigned up_off, unsigned dn_off) -> unsigned {
            const edt_col_us2 a = up2 + edt_col_pair(up_off), b = dn2 + edt_col_pair(dn_off);
            return edt_col_word(__builtin_elementwise_min(__builtin_elementwise_min(edt_col_pair(inside), a), __builtin_elementwise_min(b, lim)));
        };
        uint4 o;                                            /* rows j, j + 1 of a dword: d_up + 1 + j, d_dn + 8 - j */
        o.x = rows2(in.x, 0x00020001u, 0x00070008u);
        o.y = rows2(in.y, 0x00040003u, 0x00050006u);
        o.z = rows2(in.z, 0x00060005u, 0x00030004u);
        o.w = rows2(in.w, 0x00080007u, 0x00010002u);
        if (y0 + 8 > rows) {                                /* rows past the image hold 0 */
            const int n = rows - y0;                        /* rows of this lane inside the image: <= 7 here, possibly <= 0 */
            auto keep = [&](unsigned w, int j) -> unsigned { return (j + 1 < n) ? w : ((j < n) ? (w & 0xffffu) : 0u); };
            o.x = keep(o.x, 0); o.y = keep(o.y, 2); o.z = keep(o.z, 4); o.w = keep(o.w, 6);
        }
        if (to_lds) { *to_lds = o; return; }                /* the workgroup stores whole 128-byte lines afterwards */
        if (y0 < rows_pad) {
            if ((R & 7) == 0) {                             /* the eight rows are contiguous inside their block */
                *reinterpret_cast<uint4 *>(g + edt_g_index(xx, y0, cols, R)) = o;
            } else {
                const unsigned ow[4] = {o.x, o.y, o.z, o.w};
                for (int j = 0; j < 8; j++) if (y0 + j < rows_pad) g[edt_g_index(xx, y0 + j, cols, R)] = (unsigned short)((ow[j >> 1] >> ((j & 1) * 16)) & 0xffffu);
            }
        }
    };
    /* A workgroup takes the column groups bx, bx + gx, ... (the table and the launch are paid once for several columns), FOUR at a
     * time when a column is one chunk: the four loads are in flight together -- a wave's work per column is a memory latency
     * followed by ~100 instructions, so columns in flight, not instructions, set the pace. */
    constexpr int KF = 4;
    const int stride = gx * WAVES;
    if (WAVES == 8 && nchunk == 1 && R == 8) {
        /* The throughput shape (every level of a 640x480 pyramid): the workgroup's eight waves are eight ADJACENT columns, and a
         * 128-byte line of g is one 8-row block of exactly those eight columns.  Each wave parks its column's blocks in an LDS tile,
         * then the waves write the tile out line by line (eight lanes = one line): whole-line stores instead of sixty 16-byte pieces
         * of sixty different lines per wave. */
        uint4 *tile = s_tile;                               /* [KF][64 blocks][8 columns + 1 pad] */
        const int ngroups = (cols + 7) >> 3;
        for (int grp0 = bx; grp0 < ngroups; grp0 += KF * gx) {        /* workgroup-uniform: barriers inside */
            unsigned m8[KF];
#pragma unroll
            for (int k = 0; k < KF; k++) {
                const int xx = (grp0 + k * gx) * 8 + wave;
                const int xc = xx < cols ? xx : 0;
                const unsigned char *col = img + (size_t)xc * rows;
                m8[k] = load_mask(col, ((rows & 7) == 0) && ((reinterpret_cast<size_t>(col) & 7) == 0), lane * 8);
            }
#pragma unroll
            for (int k = 0; k < KF; k++) {
                const int xx = (grp0 + k * gx) * 8 + wave;
                if (grp0 + k * gx < ngroups) emit(xx, 0, (xx < cols) ? m8[k] : 0u, tile + ((size_t)k * 64 + lane) * 9 + wave);      /* workgroup-uniform */
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < KF; k++) {
                const int b = wave * 8 + (lane >> 3), cix = lane & 7;
                const int xx = (grp0 + k * gx) * 8 + cix;
                if (grp0 + k * gx < ngroups && xx < cols && b * 8 < rows_pad)
                    *reinterpret_cast<uint4 *>(g + edt_g_index(xx, b * 8, cols, 8)) = tile[((size_t)k * 64 + b) * 9 + cix];
            }
            __syncthreads();
        }
        return;
    }
    for (int xx0 = bx * WAVES + wave; xx0 < cols; xx0 += KF * stride) {
        if (nchunk == 1) {
            unsigned m8[KF];
#pragma unroll
            for (int k = 0; k < KF; k++) {
                const int xx = xx0 + k * stride;
                const int xc = xx < cols ? xx : xx0;        /* a column past the image: the first one again (not emitted) */
                const unsigned char *col = img + (size_t)xc * rows;
                m8[k] = load_mask(col, ((rows & 7) == 0) && ((reinterpret_cast<size_t>(col) & 7) == 0), lane * 8);
            }
#pragma unroll
            for (int k = 0; k < KF; k++) {
                const int xx = xx0 + k * stride;
                if (xx < cols) emit(xx, 0, m8[k], nullptr);          /* wave-uniform */
            }
            continue;
        }
        for (int k = 0; k < KF; k++) {
            const int xx = xx0 + k * stride;
            if (xx >= cols) break;
            const unsigned char *col = img + (size_t)xx * rows;
            const bool vec = ((rows & 7) == 0) && ((reinterpret_cast<size_t>(col) & 7) == 0);
            int cu = INF;                                   /* distance from the row above chunk c to the nearest edge at or above it */
            for (int c = 0; c < nchunk; c++) {
                if (lane == 0) carries[c] = cu;
                const unsigned m = load_mask(col, vec, c * 512 + lane * 8);
                const unsigned long long bal = __ballot(m != 0u);
                const int L = bal ? 63 - __clzll((long long)bal) : 0;
                const unsigned mL = (unsigned)__shfl((int)m, L);
                if (bal) cu = 511 - (L * 8 + (31 - __clz((int)mL)));
                else cu = (cu + 512 > INF) ? INF : cu + 512;
            }
            int cd = INF;                                   /* distance from the LAST row of chunk c to the nearest edge below the chunk */
            for (int c = nchunk - 1; c >= 0; c--) {
                if (lane == 0) carries[nchunk + c] = cd;
                const unsigned m = load_mask(col, vec, c * 512 + lane * 8);
                const unsigned long long bal = __ballot(m != 0u);
                const int L = bal ? __ffsll((long long)bal) - 1 : 0;
                const unsigned mL = (unsigned)__shfl((int)m, L);
                if (bal) cd = L * 8 + (__ffs((int)mL) - 1) + 1;          /* from the last row of chunk c - 1 */
                else cd = (cd + 512 > INF) ? INF : cd + 512;
            }
            /* the wave reads back what its lane 0 wrote: LDS operations of one wave complete in order */
            for (int c = 0; c < nchunk; c++) emit(xx, c, load_mask(col, vec, c * 512 + lane * 8), nullptr);
        }
    }
}

/* phase 2: d2(x,y) = min_i (x-i)^2 + g(i,y)^2 along the row, exactly, in integers (< 2^32 for every supported size).  Each
 * pixel scans outwards while i^2 < best: with edges every few pixels that is a few dozen steps, far cheaper on a GPU than
 * the sequential lower-envelope scan (Meijster) a CPU would use -- same minimum.  A workgroup stages R whole rows of g^2
 * (32 bit: no multiply in the scan) in LDS so that the scan runs out of LDS, not L2.  The scan has two phases: while both
 * neighbours i columns away are inside the image, four steps per trip with the LDS addresses in the instructions' offset
 * fields (a step past i^2 >= best only adds candidates that cannot win, so the exit test runs once per trip); then, for the
 * pixels that have not finished when one side hits the border, the remaining side alone.  Per-block maxima go to `partial`;
 * the values produced are recorded in the image's presence bitmap (an LDS copy first -- a bit is only set if it is not there
 * yet, so the atomics die out after the first few pixels -- merged into HBM once per workgroup). */
constexpr int EDT_LBITS_WORDS = 1024;               /* d2 < 32768 go through the LDS copy of the bitmap */
/* the workgroup's presence bits -> the image's bitmap: only the bits the image does not have yet (most are there already, set by the
 * workgroups before this one).  The four words of a thread are looked up TOGETHER (round 5: they were four dependent memory round
 * trips at the end of every workgroup -- 16 % of the row pass's wave time, tools/experiments/r05_edt_stamps.py). */
DVO_DEV void edt_flush_lbits(const unsigned *lbits, unsigned *bm, int bm_words) {
    static_assert(EDT_LBITS_WORDS == 4 * 256, "four words per thread of the 256");
    unsigned v[4], have[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int w = (int)threadIdx.x + 256 * q;
        v[q] = (w < bm_words) ? lbits[w] : 0u;
    }
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int w = (int)threadIdx.x + 256 * q;
        have[q] = v[q] ? __hip_atomic_load(bm + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0xffffffffu;
    }
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int w = (int)threadIdx.x + 256 * q;
        if ((have[q] & v[q]) != v[q]) atomicOr(bm + w, v[q]);
    }
}
/* T = unsigned: the tile holds g^2; T = unsigned short (rows too long for that, beyond 16 K columns): g, squared at use */
template <typename T> DVO_DEV unsigned edt_sq(T v) { return (sizeof(T) == 2) ? (unsigned)v * (unsigned)v : (unsigned)v; }
template <int R, typename T>
__global__ void __launch_bounds__(256)
edt_rows16_kernel(const unsigned short *__restrict__ g, int rows, int cols, unsigned *__restrict__ d2, int *__restrict__ partial,
                  unsigned *__restrict__ bitmap, int bm_words, int *__restrict__ flags) {
    extern __shared__ unsigned char tile_raw[];
    T *tile = reinterpret_cast<T *>(tile_raw);                /* [cols][R]: g^2 (or g) */
    __shared__ unsigned lbits[EDT_LBITS_WORDS];
    g += (size_t)blockIdx.y * edt_g_count(rows, cols, R) + (size_t)blockIdx.x * cols * R;       /* this workgroup's row block: [cols][R] */
    d2 += (size_t)blockIdx.y * edt_g_count(rows, cols, R);
    unsigned *bm = bitmap + (size_t)blockIdx.y * bm_words;
    const int y0 = blockIdx.x * R;
    const int total = cols * R;
    for (int i = threadIdx.x; i < EDT_LBITS_WORDS; i += 256) lbits[i] = 0u;
    for (int idx = threadIdx.x; idx < total; idx += 256) {
        const unsigned gv = (unsigned)g[idx];               /* rows past the image hold 0 */
        tile[idx] = (T)((sizeof(T) == 2) ? gv : gv * gv);
    }
    __syncthreads();
    unsigned mx = 0;
    bool far = false;
    const int lane = threadIdx.x & 63;
    const int wave_base = __builtin_amdgcn_readfirstlane((int)(threadIdx.x & ~63u));
    for (int base = wave_base; base < total; base += 256) {         /* wave-uniform: 64 consecutive pixels of the tile = 64/R columns */
        const int idx = base + lane;
        const int cidx = idx < total ? idx : total - 1;
        const int xx = cidx / R, r = cidx - xx * R, yy = y0 + r;
        const bool live = idx < total && yy < rows;
        unsigned best = live ? edt_sq<T>(tile[cidx]) : 0u;          /* 0: finished before it starts */
        const int near_side = (xx < cols - 1 - xx) ? xx : cols - 1 - xx;      /* steps for which both neighbours exist */
        const int far_side = (xx > cols - 1 - xx) ? xx : cols - 1 - xx;
        const T *pl = tile + cidx, *pr = tile + cidx;
        /* the wave's step counter lives in scalar registers: every lane is at the same distance i, and a lane that has already
         * finished (i^2 >= best) only sees candidates that cannot win -- no masking, no per-lane loop state.  lim = the steps
         * for which both neighbours exist for EVERY column of the wave (min(xx, cols-1-xx) is concave: the end columns decide) */
        const int xf = base / R, xl = ((base + 63 < total) ? base + 63 : total - 1) / R;
        const int nf = (xf < cols - 1 - xf) ? xf : cols - 1 - xf, nl = (xl < cols - 1 - xl) ? xl : cols - 1 - xl;
        const int lim = __builtin_amdgcn_readfirstlane(nf < nl ? nf : nl);
        int i = 1;
        unsigned i2 = 1;
        /* the left side is addressed from the FAR end of a trip (LDS instructions take unsigned offsets only); the empty asm
         * keeps the compiler from re-deriving four negative offsets from the loop-carried address */
        typedef __attribute__((address_space(3))) const T lds_ct;
        unsigned la = (unsigned)(size_t)(lds_ct *)tile + (unsigned)((cidx - 4 * R) * (int)sizeof(T));
        while (i + 3 <= lim && __builtin_amdgcn_ballot_w64(i2 < best) != 0ull) {
            asm volatile("" : "+v"(la));
            lds_ct *ql = (lds_ct *)(size_t)la;
            const T a0 = ql[3 * R], b0 = pr[1 * R], a1 = ql[2 * R], b1 = pr[2 * R];
            const T a2 = ql[1 * R], b2 = pr[3 * R], a3 = ql[0], b3 = pr[4 * R];
            const unsigned s1 = i2 + 2u * i + 1u, s2 = s1 + 2u * i + 3u, s3 = s2 + 2u * i + 5u;      /* (i+1)^2, (i+2)^2, (i+3)^2 */
            unsigned c0 = i2 + edt_sq<T>((T)(a0 < b0 ? a0 : b0)), c1 = s1 + edt_sq<T>((T)(a1 < b1 ? a1 : b1));      /* squaring is monotone: min first */
            unsigned c2 = s2 + edt_sq<T>((T)(a2 < b2 ? a2 : b2)), c3 = s3 + edt_sq<T>((T)(a3 < b3 ? a3 : b3));
            c0 = c0 < c1 ? c0 : c1; c2 = c2 < c3 ? c2 : c3;
            c0 = c0 < c2 ? c0 : c2;
            best = c0 < best ? c0 : best;
            la -= 4 * R * (unsigned)sizeof(T); pr += 4 * R;
            i += 4; i2 = s3 + 2u * i - 1u;                    /* (i+4)^2 = (i+3)^2 + 2(i+3)+1, with i already advanced */
        }
        pl -= (i - 1) * R;
        /* the lanes still open when the wave's common range ends (pixels near the left / right border, or far from every
         * edge): per-lane loops from here.  First the remaining two-sided steps */
        for (; i <= near_side && i2 < best; i++) {
            pl -= R; pr += R;
            const T a = *pl, b = *pr;
            const unsigned c = i2 + edt_sq<T>((T)(a < b ? a : b));
            best = c < best ? c : best;
            i2 += 2u * i + 1u;
        }
        /* one side left (the pixel sits closer to the other border than its nearest edge found so far) */
        if (i <= far_side && i2 < best) {
            const int dir = (xx < cols - 1 - xx) ? R : -R;    /* the side that still has columns */
            const T *p = tile + cidx + dir * (i - 1);
            for (; i <= far_side && i2 < best; i++) {
                p += dir;
                const unsigned c = i2 + edt_sq<T>(*p);
                best = c < best ? c : best;
                i2 += 2u * i + 1u;
            }
        }
        if (!live) continue;
        d2[(size_t)blockIdx.x * cols * R + idx] = best;              /* the same row-block layout as g */
        mx = best > mx ? best : mx;
        const unsigned w = best >> 5, bit = 1u << (best & 31u);
        if (w < (unsigned)EDT_LBITS_WORDS) {
            if (!(*(volatile unsigned *)&lbits[w] & bit)) atomicOr(&lbits[w], bit);
        } else if (w < (unsigned)bm_words) {
            if (!(__hip_atomic_load(bm + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & bit)) atomicOr(bm + w, bit);
        } else {
            far = true;
        }
    }
    const int m = block_reduce_256<true>((int)(mx > 0x7fffffffu ? 0x7fffffffu : mx));    /* d2 < 2^31 (rows + cols < 46340) */
    if (threadIdx.x == 0) partial[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = m;
    __syncthreads();
    edt_flush_lbits(lbits, bm, bm_words);
    if (__syncthreads_or(far ? 1 : 0) && threadIdx.x == 0) atomicOr(flags + blockIdx.y, (int)EDT_FLAG_FAR);
}

/* phase 2, the kernel every ordinary image takes: the same scan in PACKED 16-bit arithmetic, two image rows per lane.
 * Squared distances below 65535 -- nearest edge closer than 256 pixels -- are what real frames have, so the tile holds
 * min(g^2, 65535) as 16-bit halves (rows 2m / 2m+1 of a column share a dword) and a step of the scan is, for TWO pixels, two
 * LDS dwords, v_pk_min_u16 (left against right), a saturating v_pk_add_u16 of the step's i^2 and v_pk_min_u16 into the best:
 * a third of the vector instructions and half the LDS traffic per pixel of the 32-bit scan.  The step counter is the wave's
 * (scalar registers: every lane is at the same distance, a lane that has finished only sees candidates that cannot win);
 * EDT_PK_PAD columns of "infinity" either side of the tile let it run past the image border.  Saturation is harmless while
 * the true minimum is below 65535: a clamped candidate is >= 65535 and cannot be it.  What the packed scan cannot finish --
 * a pixel whose best is still 65535, or still open when the wave's common range (the pad, 252 steps) ends -- is finished by
 * the exact 32-bit per-lane scan over the second tile (g itself), from scratch if it saturated. */
/* Round 5, measured (tools/experiments/r05_edt_rows_ab.sh, 256 camera frames, per launch): pad 32: 153 us, 64: 158, 128: 158, 256: 176 -- the
 * exact finish that a small pad sends border pixels to is not what the pass spends its time on.  Also measured: the finish reading g from
 * HBM instead of a second LDS tile (15 instead of 26 KB per workgroup, 10 instead of 6 workgroups per CU): 170 us. */
#ifndef DVO_EDT_PAD
#define DVO_EDT_PAD 32
#endif
constexpr int EDT_PK_PAD = DVO_EDT_PAD;
typedef unsigned short edt_us2 __attribute__((ext_vector_type(2)));
DVO_DEV edt_us2 edt_as_us2(unsigned v) { return __builtin_bit_cast(edt_us2, v); }
DVO_DEV unsigned edt_as_u32(edt_us2 v) { return __builtin_bit_cast(unsigned, v); }

/* exact 32-bit finish of pixel (xx, r) of the tile `tg` ([cols][R], g): steps i.. with `best` found so far */
template <int R>
DVO_DEV unsigned edt_finish32(const unsigned short *tg, int cols, int xx, int r, int i, unsigned best) {
    const int near_side = (xx < cols - 1 - xx) ? xx : cols - 1 - xx;
    const int far_side = (xx > cols - 1 - xx) ? xx : cols - 1 - xx;
    const unsigned short *pc = tg + xx * R + r;
    unsigned i2 = (unsigned)i * (unsigned)i;
    for (; i <= near_side && i2 < best; i++) {
        const unsigned a = pc[-i * R], b = pc[i * R];
        const unsigned m = a < b ? a : b;
        const unsigned c = i2 + m * m;
        best = c < best ? c : best;
        i2 += 2u * i + 1u;
    }
    if (i <= far_side && i2 < best) {
        const int dir = (xx < cols - 1 - xx) ? R : -R;
        for (; i <= far_side && i2 < best; i++) {
            const unsigned a = pc[dir * i];
            const unsigned c = i2 + a * a;
            best = c < best ? c : best;
            i2 += 2u * i + 1u;
        }
    }
    return best;
}

typedef unsigned edt_u2 __attribute__((ext_vector_type(2)));
/* the 8 + 8 look-ups of one trip of the four-rows-per-lane scan: a[j-1] = 8 bytes at la + (8 - j) * CB, b[j-1] = at ra + j * CB
 * (CB = bytes per tile column), as sixteen ds_read_b64 with immediate offsets and one wait */
template <int CB>
DVO_DEV void edt_rows_read16(unsigned la, unsigned ra, edt_u2 (&a)[8], edt_u2 (&b)[8]) {
    asm volatile("ds_read_b64 %0, %16 offset:%18\n\tds_read_b64 %8, %17 offset:%26\n\t"
                 "ds_read_b64 %1, %16 offset:%19\n\tds_read_b64 %9, %17 offset:%27\n\t"
                 "ds_read_b64 %2, %16 offset:%20\n\tds_read_b64 %10, %17 offset:%28\n\t"
                 "ds_read_b64 %3, %16 offset:%21\n\tds_read_b64 %11, %17 offset:%29\n\t"
                 "ds_read_b64 %4, %16 offset:%22\n\tds_read_b64 %12, %17 offset:%30\n\t"
                 "ds_read_b64 %5, %16 offset:%23\n\tds_read_b64 %13, %17 offset:%31\n\t"
                 "ds_read_b64 %6, %16 offset:%24\n\tds_read_b64 %14, %17 offset:%32\n\t"
                 "ds_read_b64 %7, %16 offset:%25\n\tds_read_b64 %15, %17 offset:%33\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(a[0]), "=&v"(a[1]), "=&v"(a[2]), "=&v"(a[3]), "=&v"(a[4]), "=&v"(a[5]), "=&v"(a[6]), "=&v"(a[7]),
                   "=&v"(b[0]), "=&v"(b[1]), "=&v"(b[2]), "=&v"(b[3]), "=&v"(b[4]), "=&v"(b[5]), "=&v"(b[6]), "=&v"(b[7])
                 : "v"(la), "v"(ra),
                   "n"(7 * CB), "n"(6 * CB), "n"(5 * CB), "n"(4 * CB), "n"(3 * CB), "n"(2 * CB), "n"(1 * CB), "n"(0 * CB),
                   "n"(1 * CB), "n"(2 * CB), "n"(3 * CB), "n"(4 * CB), "n"(5 * CB), "n"(6 * CB), "n"(7 * CB), "n"(8 * CB)
                 : "memory");
}
/* Round 5, MEASURED AND NOT TAKEN: FOUR rows per lane (NW = 2 dwords, R >= 4).  The premise: the scan's look-ups are ds_read_b32, which
 * the LDS delivers at half the bytes per cycle of ds_read_b64 (MI355X_MICROARCH.md, LDS table: 128 against 256 B/clk per CU), so the
 * same look-ups as 8-byte reads should halve the LDS time.  Result over 256 camera frames (tools/experiments/r05_edt_rows_ab.sh,
 * profiles/r05_experiments/edt_rows_ab.txt): 157 us per launch against 154 us for two rows per lane (164 us when the compiler is left to
 * pair the reads into ds_read2_b64: edt_rows_read16 keeps them single).  The counters say why (profiles/r05_frames/pmc_rows.txt): the
 * kernel issues 0.2 instructions per cycle and SIMD -- 36 k packed vector, 25 k scalar and 7 k LDS instructions per SIMD in 330 k cycles
 * -- so no unit is near its rate; a wave's trip is a chain (look-ups, wait, 48 dependent packed operations at the single-wave cadence
 * of 6 cycles, the scalar step counters with their hazard no-ops) and three waves per SIMD do not cover it.
 * make EXP=rowsb64 EXPDEFS=-DDVO_EDT_ROWS_B64=1 builds this form. */
#ifdef DVO_EDT_ROWS_B64
#define DVO_EDT_NW(R) ((R) >= 4 ? 2 : 1)
#else
#define DVO_EDT_NW(R) 1
#endif
#ifdef DVO_EDT_STAMPS
/* diagnostic build (make EXP=edtstamps EXPDEFS=-DDVO_EDT_STAMPS=1; tools/experiments/r05_edt_stamps.py): where the waves of the row pass spend
 * their cycles -- s_memtime sums over all waves: [0] staging incl. its barrier, [1] scan trips, [2] exact finish, [3] stores + presence
 * bitmap, [4] tail (block maximum, bitmap flush), [5] waves, [6] trips, [7] whole kernel */
constexpr int EDT_STAMP_SLOTS = 1 << 18;                 /* one slot per wave: plain stores, no contention added to what is measured */
__device__ unsigned long long g_edt_stamp[EDT_STAMP_SLOTS][8];
DVO_DEV unsigned long long edt_now() {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
#define EDT_T(v) const unsigned long long v = edt_now()
#define EDT_ACC(k, a, b) acc_t[k] += (b) - (a)
#else
#define EDT_T(v) do {} while (0)
#define EDT_ACC(k, a, b) do {} while (0)
#endif
template <int R, int NW = DVO_EDT_NW(R)>
DVO_DEV void edt_rows_pk_body(const int bx, const int gx, const int by, const unsigned short *__restrict__ g, int rows, int cols, unsigned *__restrict__ d2, int *__restrict__ partial,
                   unsigned *__restrict__ bitmap, int bm_words, int *__restrict__ flags) {
    static_assert(R >= 2 && (R & 1) == 0, "two rows per dword");
    constexpr int RP = R / 2;                                  /* row pairs = dwords per tile column */
    static_assert(NW == 1 || (NW == 2 && RP % 2 == 0), "a lane's dwords are one aligned 8-byte piece of a column");
    constexpr int LP = RP / NW;                                /* lanes per tile column */
    constexpr int NR = 2 * NW;                                 /* rows per lane */
    extern __shared__ unsigned char tile_raw[];
    unsigned *tq = reinterpret_cast<unsigned *>(tile_raw);     /* [PAD + cols + PAD][RP]: min(g^2, 65535), two rows per dword */
    unsigned short *tg = reinterpret_cast<unsigned short *>(tq + (size_t)(cols + 2 * EDT_PK_PAD) * RP);     /* [cols][R]: g */
    __shared__ unsigned lbits[EDT_LBITS_WORDS];
    /* this workgroup's row block of g is one contiguous chunk, [cols][R] like the tile (rows past the image hold 0) */
    const unsigned *gblk = reinterpret_cast<const unsigned *>(g + (size_t)by * edt_g_count(rows, cols, R) + (size_t)bx * cols * R);
    d2 += (size_t)by * edt_g_count(rows, cols, R) + (size_t)bx * cols * R;      /* d2 too is written in row blocks */
    unsigned *bm = bitmap + (size_t)by * bm_words;
#ifdef DVO_EDT_STAMPS
    unsigned long long acc_t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    EDT_T(t_begin);
    const int y0 = bx * R;
    const int totalp = cols * RP;                              /* dwords of the tile */
    const int totall = cols * LP;                              /* lane items: NW dwords each */
    for (int i = threadIdx.x; i < EDT_LBITS_WORDS; i += 256) lbits[i] = 0u;
    for (int i = threadIdx.x; i < EDT_PK_PAD * RP; i += 256) {
        tq[i] = 0xffffffffu;
        tq[(size_t)(EDT_PK_PAD + cols) * RP + i] = 0xffffffffu;
    }
    auto stage = [&](int p, unsigned gg) {
        const unsigned g0 = gg & 0xffffu, g1 = gg >> 16;
        const unsigned q0 = g0 > 255u ? 65535u : g0 * g0, q1 = g1 > 255u ? 65535u : g1 * g1;       /* 255^2 = 65025 */
        reinterpret_cast<unsigned *>(tg)[p] = gg;
        tq[EDT_PK_PAD * RP + p] = q0 | (q1 << 16);
    };
    const int n4 = totalp >> 2;
    for (int q = threadIdx.x; q < n4; q += 256) {
        const uint4 v = reinterpret_cast<const uint4 *>(gblk)[q];
        stage(4 * q, v.x); stage(4 * q + 1, v.y); stage(4 * q + 2, v.z); stage(4 * q + 3, v.w);
    }
    for (int p = 4 * n4 + threadIdx.x; p < totalp; p += 256) stage(p, gblk[p]);
    __syncthreads();
    EDT_T(t_staged);
    EDT_ACC(0, t_begin, t_staged);
    typedef __attribute__((address_space(3))) const unsigned lds_cu;

    const unsigned tq_lds = (unsigned)(size_t)(lds_cu *)tq + (unsigned)(EDT_PK_PAD * RP * 4);         /* LDS byte address of column 0 */
    unsigned mx = 0;
    bool far = false;
    const int lane = threadIdx.x & 63;
    /* (round 5, measured: handing the chunks out dynamically through an LDS counter instead of this static interleave: 161 against 153 us
     * per launch -- the workgroup's tail is not wave imbalance) */
    const int wave_base = __builtin_amdgcn_readfirstlane((int)(threadIdx.x & ~63u));
    for (int base = wave_base; base < totall; base += 256) {    /* wave-uniform: 64 lane items = 64/LP columns */
        const int p = base + lane;
        const int cp = p < totall ? p : totall - 1;
        const int xx = cp / LP, h = cp - xx * LP;
        const int dq = xx * RP + h * NW;                        /* the lane's first dword of the tile */
        const int r0 = 2 * h * NW, yy = y0 + r0;                /* its first row: of the block, of the image */
        bool live[NR];
#pragma unroll
        for (int k = 0; k < NR; k++) live[k] = p < totall && yy + k < rows;
        edt_us2 best[NW];
#pragma unroll
        for (int w = 0; w < NW; w++) best[w] = edt_as_us2(p < totall ? tq[EDT_PK_PAD * RP + dq + w] : 0u);      /* rows past the image hold 0: finished */
        /* steps every column of the wave can take inside image + pad (min(xx, cols-1-xx) is concave: the end columns decide),
         * capped where i^2 still fits 16 bits */
        const int xf = base / LP, xl = ((base + 63 < totall) ? base + 63 : totall - 1) / LP;
        const int nf = (xf < cols - 1 - xf) ? xf : cols - 1 - xf, nl = (xl < cols - 1 - xl) ? xl : cols - 1 - xl;
        int lim = __builtin_amdgcn_readfirstlane((nf < nl ? nf : nl) + EDT_PK_PAD);
        lim = lim < 255 ? lim : 255;
        int i = 1;
        /* i^2 and 2i+1 in both halves of a scalar register each: the next step's pair is two scalar additions away */
        unsigned S = 0x00010001u, D = 0x00030003u;
        unsigned la = tq_lds + (unsigned)((dq - 8 * RP) * 4);   /* far end of a trip's left side: LDS offsets are unsigned */
        unsigned ra = tq_lds + (unsigned)(dq * 4);
        EDT_T(t_s0);
        while (i + 7 <= lim) {                                  /* eight steps per trip, one exit test */
            unsigned open = edt_as_u32(__builtin_elementwise_sub_sat(best[0], edt_as_us2(S)));
            if (NW == 2) open |= edt_as_u32(__builtin_elementwise_sub_sat(best[NW - 1], edt_as_us2(S)));
            if (__builtin_amdgcn_ballot_w64(open != 0u) == 0ull) break;      /* i^2 >= best everywhere */
            asm volatile("" : "+v"(la), "+v"(ra));             /* keep the two addresses as they are: offsets go into the instructions */
            edt_us2 acc[NW];
#pragma unroll
            for (int w = 0; w < NW; w++) acc[w] = best[w];
            if constexpr (NW == 1) {
                lds_cu *ql = (lds_cu *)(size_t)la, *qr = (lds_cu *)(size_t)ra;
#pragma unroll
                for (int j = 1; j <= 8; j++) {
                    const edt_us2 a = edt_as_us2(ql[(8 - j) * RP]), b = edt_as_us2(qr[j * RP]);
                    acc[0] = __builtin_elementwise_min(acc[0], __builtin_elementwise_add_sat(__builtin_elementwise_min(a, b), edt_as_us2(S)));
                    S += D; D += 0x00020002u;
                }
            } else {
                /* sixteen ds_read_b64 and their wait as ONE statement: left to itself the compiler pairs them into ds_read2_b64 */
                edt_u2 av[8], bv2[8];
                edt_rows_read16<RP * 4>(la, ra, av, bv2);
#pragma unroll
                for (int j = 1; j <= 8; j++) {
                    const edt_u2 a = av[j - 1], b = bv2[j - 1];                        /* four rows of columns x - j, x + j */
                    acc[0] = __builtin_elementwise_min(acc[0], __builtin_elementwise_add_sat(__builtin_elementwise_min(edt_as_us2(a.x), edt_as_us2(b.x)), edt_as_us2(S)));
                    acc[NW - 1] = __builtin_elementwise_min(acc[NW - 1], __builtin_elementwise_add_sat(__builtin_elementwise_min(edt_as_us2(a.y), edt_as_us2(b.y)), edt_as_us2(S)));
                    S += D; D += 0x00020002u;
                }
            }
#pragma unroll
            for (int w = 0; w < NW; w++) best[w] = acc[w];
            la -= 8u * RP * 4u; ra += 8u * RP * 4u;
            i += 8;
        }
        EDT_T(t_s1);
        EDT_ACC(1, t_s0, t_s1);
#ifdef DVO_EDT_STAMPS
        acc_t[6] += (unsigned long long)(i >> 3);
#endif
        const unsigned i2 = S & 0xffffu;
        /* the exact finish of what is still open: rare (a pixel further than the pad from the border AND from every edge found
         * so far, or further than 255 pixels from every edge) */
        unsigned bv[NR];
#pragma unroll
        for (int w = 0; w < NW; w++) { bv[2 * w] = best[w].x; bv[2 * w + 1] = best[w].y; }
#pragma unroll
        for (int k = 0; k < NR; k++) {
            if (live[k] && i2 < bv[k]) {
                const unsigned gk = tg[xx * R + r0 + k];
                bv[k] = (bv[k] == 65535u) ? edt_finish32<R>(tg, cols, xx, r0 + k, 1, gk * gk) : edt_finish32<R>(tg, cols, xx, r0 + k, i, bv[k]);
            }
        }
        EDT_T(t_s2);
        EDT_ACC(2, t_s1, t_s2);
        if (p < totall) {
            /* all presence words are requested before anything waits on them; one store for the lane's rows (rows past the image:
             * 0, never read) */
            unsigned wd[NR], bit[NR], have[NR];
            bool in[NR];
#pragma unroll
            for (int k = 0; k < NR; k++) {
                wd[k] = bv[k] >> 5; bit[k] = 1u << (bv[k] & 31u); in[k] = wd[k] < (unsigned)EDT_LBITS_WORDS;
                have[k] = in[k] ? __hip_atomic_load(&lbits[wd[k]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : 0u;
            }
#pragma unroll
            for (int k = 0; k < NR; k++) if (!live[k]) bv[k] = 0u;
            if constexpr (NW == 1) reinterpret_cast<uint2 *>(d2)[cp] = make_uint2(bv[0], bv[1]);
            else reinterpret_cast<uint4 *>(d2)[cp] = make_uint4(bv[0], bv[1], bv[2], bv[3]);
#pragma unroll
            for (int k = 0; k < NR; k++) {
                mx = bv[k] > mx ? bv[k] : mx;
                if (live[k]) {
                    if (in[k]) { if (!(have[k] & bit[k])) atomicOr(&lbits[wd[k]], bit[k]); }
                    else if (wd[k] < (unsigned)bm_words) { if (!(__hip_atomic_load(bm + wd[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & bit[k])) atomicOr(bm + wd[k], bit[k]); }
                    else far = true;
                }
            }
        }
        EDT_T(t_s3);
        EDT_ACC(3, t_s2, t_s3);
    }
    EDT_T(t_loop);

    const int m = block_reduce_256<true>((int)(mx > 0x7fffffffu ? 0x7fffffffu : mx));    /* d2 < 2^31 (rows + cols < 46340) */
    if (threadIdx.x == 0) partial[(size_t)by * gx + bx] = m;
    __syncthreads();
    edt_flush_lbits(lbits, bm, bm_words);
    if (__syncthreads_or(far ? 1 : 0) && threadIdx.x == 0) atomicOr(flags + by, (int)EDT_FLAG_FAR);
#ifdef DVO_EDT_STAMPS
    {
        EDT_T(t_end);
        EDT_ACC(4, t_loop, t_end);
        EDT_ACC(7, t_begin, t_end);
        acc_t[5] = 1;
        const unsigned slot = ((blockIdx.y * gridDim.x + blockIdx.x) * 4u + (threadIdx.x >> 6)) & (unsigned)(EDT_STAMP_SLOTS - 1);
        if ((threadIdx.x & 63) == 0)
            for (int k = 0; k < 8; k++) g_edt_stamp[slot][k] = acc_t[k];
    }
#endif
}

DVO_DEV int reflect101(int i, int n) { return (n == 1) ? 0 : (i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i)); }

/* cv::normalize(0, 255, NORM_MINMAX) of sqrt(d2) with OpenCV 2.4's arithmetic: min = 0 (an edge pixel), max = sqrt(m2);
 * scale and shift in double, then convertTo's 32F -> 32F kernel (cvtScale32f, core/src/convert.cpp) in FLOAT:
 * dst = src*(float)scale + (float)shift.  An image without any edge pixel -- every distance "infinite" -- normalises to all
 * zeros, as a constant image does. */
struct EdtScale { float scale_f, shift_f; };
DVO_DEV EdtScale edt_scale(unsigned m2, int rows, int cols) {
    const long long INF = DVO_EDT_INF(rows, cols);
    const float mxf = (float)sqrt((double)m2), mnf = 0.0f;
    const double smin = (double)mnf, smax = (double)mxf;
    const double scale_d = ((long long)m2 < INF * INF) ? 255.0 * ((smax - smin > 2.2204460492503131e-16) ? 1. / (smax - smin) : 0.) : 0.;
    EdtScale s;
    s.scale_f = (float)scale_d; s.shift_f = (float)(0.0 - smin * scale_d);
    return s;
}
DVO_DEV float edt_value(unsigned d2v, const EdtScale &s) {
    const float raw = (float)sqrt((double)d2v);
    return raw * s.scale_f + s.shift_f;
}

/* phase 3: squared distances -> rank words + palette.  A workgroup takes a strip of tiles (PK_LC x PK_LR lines of the compact
 * image = 64 x 48 pixels each, `strip` of them along yy).  It rebuilds the image's rank table from the presence bitmap in LDS
 * once (a few hundred words for ordinary images: prefix popcounts), then per tile looks up the rank of every pixel and of a
 * one-pixel halo (reflect-101 at the image border: cv::filter2D's default) and writes the tile's words -- own rank, signed rank
 * steps to the horizontal neighbours, the apron rows above and below.  Workgroup 0 of an image also writes the palette {P, W}
 * (the 16-byte path's expressions: edt_value, weight_of), the sentinel entry and line, and pal_n.
 * Two instantiations: BM_WORDS = 2048 (squared distances below 65536: 19 KB of LDS, many workgroups per CU -- every ordinary
 * image) and the full bitmap (54 KB); each skips the images that belong to the other. */
constexpr int PK_LC = 16, PK_LR = 8;
constexpr int PK_W = PK_LC * 4 + 2, PK_H = PK_LR * DVO_P4_ROWS + 2;
constexpr int PK_SMALL_WORDS = 2048;
static_assert(PK_LC * 4 == 64 && (PK_LR * DVO_P4_ROWS) % 8 == 0 && 2 * PK_W + 2 * (PK_H - 2) <= 256, "the rank look-up walks the tile as 2 x 32 columns by groups of 8 rows, the halo in one step of 256 threads");
template <int BM_WORDS>
DVO_DEV void edt_rank_pack_body(const int bx, const int gx, const int by, const unsigned *__restrict__ d2, int rows, int cols, int R, int tiles_y, int strip, const int *__restrict__ partial, int n_partial,
                     const unsigned *__restrict__ bitmap, int bm_words, int *__restrict__ flags,
                     unsigned *__restrict__ p4, size_t p4_stride, float2 *__restrict__ pal, int *__restrict__ pal_n, int first_pair,
                     const unsigned *__restrict__ unit_bits /* NULL, or per image: ~bits of the value of distance 1 (float images) */) {
    __shared__ unsigned lbm[BM_WORDS];
    __shared__ unsigned short lpre[BM_WORDS];
    __shared__ unsigned short rk[PK_W * PK_H];                  /* [x][y], halo 1 */
    __shared__ int s_wave[4];
    __shared__ int s_max;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int pair = first_pair + by;
    d2 += (size_t)by * edt_g_count(rows, cols, R);      /* row blocks of R rows, like g */
    partial += (size_t)by * n_partial;
    const unsigned *bm = bitmap + (size_t)by * bm_words;
    p4 += (size_t)pair * p4_stride;
    pal += (size_t)pair * DVO_PAL_MAX;
    const bool first_wg = bx == 0;
    const bool small = BM_WORDS == PK_SMALL_WORDS;
    if (rows < 2 || cols < 2) { if (first_wg && tid == 0 && small) pal_n[pair] = -(int)PAL_SHAPE; return; }
    if (flags[by] & EDT_FLAG_BAD) { if (first_wg && tid == 0 && small) pal_n[pair] = -(int)PAL_BAD_VALUE; return; }
    /* round 5: a pixel beyond the bitmap's range (EDT_FLAG_FAR), too many distinct distances, a rank step that does not fit: the
     * image gets a PARTIAL compact form (dvo_palette.h) instead of none.  Float images handed in directly (unit_bits) keep the
     * all-or-nothing rule: their planes are verified pixel by pixel against the decode (compact_verify_planes_kernel). */
    const bool far_img = (flags[by] & EDT_FLAG_FAR) != 0;
    if (far_img && unit_bits) { if (first_wg && tid == 0 && small) pal_n[pair] = -(int)PAL_FAR; return; }
    int m = 0;
    for (int k = tid; k < n_partial; k += 256) { const int v = partial[k]; m = v > m ? v : m; }
    m = block_reduce_256<true>(m);
    if (tid == 0) s_max = m;
    __syncthreads();
    const unsigned m2 = (unsigned)s_max;
    int nw = (int)(m2 >> 5) + 1;
    if (nw > bm_words) nw = bm_words;                           /* EDT_FLAG_FAR: the distances beyond the bitmap take the NaN rank */
    if ((nw <= PK_SMALL_WORDS) != small) return;                /* the other instantiation's image */
    if (nw > BM_WORDS) nw = BM_WORDS;
    const int per = (nw + 255) / 256;
    const int w0 = tid * per, w1 = (w0 + per < nw) ? w0 + per : nw;
    int cnt = 0;
    for (int w = w0; w < w1; w++) { const unsigned v = bm[w]; lbm[w] = v; cnt += __popc(v); }
    int incl = cnt;                                             /* inclusive scan over the workgroup: wave shuffles + 4 wave totals */
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const int v = __shfl_up(incl, off, 64); if (lane >= off) incl += v; }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    int wave_off = 0, n_pal = 0;
#pragma unroll
    for (int w = 0; w < 4; w++) { const int t = s_wave[w]; n_pal += t; if (w < wave) wave_off += t; }
    if (n_pal > DVO_PAL_MAX - 2 && unit_bits) { if (first_wg && tid == 0) pal_n[pair] = -(int)PAL_TOO_MANY; return; }
    /* ranks kept: all of them, or -- too many for the 13-bit field / the LDS -- the lowest DVO_PAL_CAP_PARTIAL.  Palette layout:
     * [0, cap) real entries, [cap] the zero sentinel (as ever), [cap + 1] the NaN entry (read by partial forms only) */
    const int cap = (n_pal > DVO_PAL_MAX - 2) ? DVO_PAL_CAP_PARTIAL : n_pal;
    const bool partial_known = far_img || cap < n_pal;
    {
        int run = wave_off + incl - cnt;
        for (int w = w0; w < w1; w++) { lpre[w] = (unsigned short)run; run += __popc(lbm[w]); }
    }
    __syncthreads();
    if (first_wg) {
        EdtScale sc = edt_scale(m2, rows, cols);
        if (unit_bits) { sc.scale_f = unit_bits[by] ? __uint_as_float(~unit_bits[by]) : 0.0f; sc.shift_f = 0.0f; }
        for (int w = w0; w < w1; w++) {
            unsigned v = lbm[w];
            int r = lpre[w];
            while (v) {
                const int b = __ffs((int)v) - 1;
                v &= v - 1u;
                const float P = edt_value((unsigned)(w * 32 + b), sc);
                if (r < cap) pal[r] = make_float2(P, weight_of(P));         /* getWeightOf, SolveDVO.cpp:1047-1053 */
                r++;
            }
        }
        if (tid == 0) {
            const float qnan = __uint_as_float(0x7fc00000u);
            pal[cap] = make_float2(0.0f, 0.0f);                  /* the sentinel entry */
            pal[cap + 1] = make_float2(qnan, qnan);              /* the NaN entry (partial forms; never referenced otherwise) */
            if (partial_known) atomicOr(flags + by, (int)EDT_FLAG_PARTIAL);
        }
        if (tid < 32) p4[tid] = (unsigned)cap << 3;              /* the sentinel line */
    }
    const int rshift = 31 - __clz(R);                           /* R is a power of two */
    const unsigned v_end = (unsigned)nw << 5;                    /* squared distances the bitmap holds */
    const int nanr = cap + 1;                                    /* the NaN rank */
    auto rank_of = [&](unsigned v) -> int {
        if (v >= v_end) return nanr;                             /* beyond the bitmap */
        const int r = (int)lpre[v >> 5] + __popc(lbm[v >> 5] & ((1u << (v & 31u)) - 1u));
        return r < cap ? r : nanr;
    };
    const int tpc = p4_tiles_per_col(rows);
    const int n_tcols = (cols + 3) >> 2;
    const int n_strips = (tiles_y + strip - 1) / strip;
    /* a workgroup takes the strips bx, bx + gx, ...: the small-bitmap launch has one workgroup per strip; the large-bitmap launch,
     * which nearly every image leaves at the test above, only a handful per image (its 54 KB of LDS made a full grid cost 17-25 us per
     * 256 frames just to leave) */
    const int n_work = n_strips * ((n_tcols + PK_LC - 1) / PK_LC);
    int sy = 0, tc0 = 0, x0 = 0;
    bool bad_step = false, nan_rank = false;
    /* A thread's share of a tile's pixels + one-pixel halo: 12 interior pixels and one of the halo.  d2 lies in row blocks of R
     * rows ([block][column][R]), so the 64 x 48 interior is walked in groups of 8 rows x 32 columns -- eight consecutive lanes
     * read one 32-byte run of a column, a wave 8 adjacent columns -- and the halo (2 rows, 2 columns) takes one more step.
     * The squared distances of the NEXT tile of the strip are fetched into registers before the current tile's words are
     * assembled and stored, so that their latency hides behind that work.  32-bit index arithmetic throughout. */
    constexpr int PK_NV = 2 * (PK_LR * DVO_P4_ROWS / 8) + 1;
    constexpr unsigned PK_NONE = 0xffffffffu;                    /* "outside image + halo": rank 0 (never read) */
    const int r8 = tid & 7, cg = tid >> 3;
    int hlx = -1, hly = 0;                                       /* this thread's halo pixel */
    if (tid < 2 * PK_W) { hlx = tid < PK_W ? tid : tid - PK_W; hly = tid < PK_W ? 0 : PK_H - 1; }
    else if (tid - 2 * PK_W < 2 * (PK_H - 2)) { const int k = tid - 2 * PK_W; hlx = k < PK_H - 2 ? 0 : PK_W - 1; hly = 1 + (k < PK_H - 2 ? k : k - (PK_H - 2)); }
    auto fetch_at = [&](int y0, int lx, int ly) -> unsigned {
        int yy = y0 + ly, xx = x0 + lx;
        if (yy > rows || xx > cols) return PK_NONE;              /* one pixel beyond the image is the reflected neighbour */
        yy = reflect101(yy, rows); xx = reflect101(xx, cols);
        return d2[(unsigned)(((yy >> rshift) * cols + xx) << rshift) + (unsigned)(yy & (R - 1))];
    };
    auto fetch_tile = [&](int t, unsigned (&v)[PK_NV]) {
        const int y0 = t * PK_LR * DVO_P4_ROWS - 1;
#pragma unroll
        for (int rb = 0; rb < PK_LR * DVO_P4_ROWS / 8; rb++) {
            v[2 * rb] = fetch_at(y0, 1 + cg, 1 + rb * 8 + r8);
            v[2 * rb + 1] = fetch_at(y0, 33 + cg, 1 + rb * 8 + r8);
        }
        v[PK_NV - 1] = hlx >= 0 ? fetch_at(y0, hlx, hly) : PK_NONE;
    };
    for (int b = bx; b < n_work; b += gx) {
    sy = b % n_strips; tc0 = (b / n_strips) * PK_LC; x0 = tc0 * 4 - 1;
    const int t_first = sy * strip, t_end = (tiles_y < (sy + 1) * strip) ? tiles_y : (sy + 1) * strip;
    unsigned cur[PK_NV];
    if (t_first < t_end) fetch_tile(t_first, cur);
    for (int t = t_first; t < t_end; t++) {
        const int ty0 = t * PK_LR;
        const int y0 = ty0 * DVO_P4_ROWS - 1;                   /* image coordinates of rk[0][0]: (y0, x0) */
#pragma unroll
        for (int rb = 0; rb < PK_LR * DVO_P4_ROWS / 8; rb++) {
            rk[(1 + cg) * PK_H + 1 + rb * 8 + r8] = (unsigned short)(cur[2 * rb] == PK_NONE ? 0 : rank_of(cur[2 * rb]));
            rk[(33 + cg) * PK_H + 1 + rb * 8 + r8] = (unsigned short)(cur[2 * rb + 1] == PK_NONE ? 0 : rank_of(cur[2 * rb + 1]));
        }
        if (hlx >= 0) rk[hlx * PK_H + hly] = (unsigned short)(cur[PK_NV - 1] == PK_NONE ? 0 : rank_of(cur[PK_NV - 1]));
        if (t + 1 < t_end) fetch_tile(t + 1, cur);               /* in flight across the barrier and the stores below */
        __syncthreads();
        for (int s = tid; s < PK_LC * PK_LR * 32; s += 256) {
            /* consecutive lanes -> consecutive words of a line, consecutive lines of a column of lines: contiguous in memory */
            const int lc = s / (PK_LR * 32), rem = s - lc * (PK_LR * 32);
            const int lr = rem >> 5, wd = rem & 31, xl = wd >> 3, srow = wd & 7;
            const int tc = tc0 + lc, ty = ty0 + lr;
            if (tc >= n_tcols || ty >= tpc) continue;
            const int xx = tc * 4 + xl, ys = ty * DVO_P4_ROWS + srow - 1;    /* image row this slot stands for (-1 / rows: reflected) */
            unsigned word = 0u;
            if (xx < cols && ys <= rows) {
                const int lx = xx - x0, ly = ys - y0;
                int c = rk[lx * PK_H + ly];
                if (srow >= 1 && srow <= DVO_P4_ROWS && ys < rows) {
                    int dr = (int)rk[(lx + 1) * PK_H + ly] - c, dl = (int)rk[(lx - 1) * PK_H + ly] - c;
                    if (dr < -127 || dr > 127 || dl < -127 || dl > 127) {
                        if (unit_bits) bad_step = true;          /* float images: all or nothing */
                        else { c = nanr; dr = 0; dl = 0; }       /* this pixel is looked up in the 16-byte texels (NaN rank) */
                    }
                    word = (((unsigned)dr & 0xffu) << 16) | (((unsigned)dl & 0xffu) << 24);
                }
                if (c == nanr) nan_rank = true;
                word |= (unsigned)c << 3;
            }
            p4[32u + ((size_t)tc * tpc + ty) * 32u + wd] = word;
        }
        __syncthreads();
    }
    }
    if (__syncthreads_or(bad_step ? 1 : 0) && tid == 0) atomicOr(flags + by, (int)EDT_FLAG_STEP);
    if (__syncthreads_or(nan_rank ? 1 : 0) && tid == 0) atomicOr(flags + by, (int)EDT_FLAG_PARTIAL);
    /* EDT_FLAG_STEP / EDT_FLAG_PARTIAL are settled by the texel pass that follows (a launch boundary later: every workgroup's flag is in) */
    if (first_wg && tid == 0) pal_n[pair] = cap | (partial_known ? DVO_PAL_PARTIAL : 0);
}

/* squared distances -> 16-byte texels in one pass (images without a compact form; everything when the caller wants no compact
 * form at all): min-max normalise, central differences with a reflect-101 border (:1077-1090), weight, tiled store.  One
 * 64 (yy) x 16 (xx) tile per workgroup: the normalised values of the tile and its 1-pixel halo are computed once into LDS
 * (one double sqrt per pixel); the lanes are then mapped so that 8 consecutive lanes write one whole 128-byte texel tile. */
constexpr int NP_TY = 64, NP_TX = 16;
static_assert(DVO_TILE_Y_LOG2 == 2 && DVO_TILE_X_LOG2 == 1, "the store mapping below assumes 4 x 2 texel tiles");
/* bx / gx: this workgroup's index among the gx workgroups of its image (and level); it takes tiles bx, bx + gx, ... of the
 * n_tiles the image has.  With a compact form in play the launch gives every image a handful of workgroups only (round 4): all
 * but a rare image leave at the first test, and a grid of one workgroup per tile spent 63 us per 256 frames on leaving. */
DVO_DEV void dt_normalize_gradient_pack_body(const int bx, const int gx, const int by, const unsigned *__restrict__ d2, int rows, int cols, int R, int tiles_y,
                                  const int *__restrict__ partial, int n_partial,
                                  float4 *__restrict__ out, size_t tex_stride,
                                  int *__restrict__ pal_n /* NULL: every image */, const int *__restrict__ flags, int first_pair, int n_tiles) {
    constexpr int SH = NP_TY + 2, SW = NP_TX + 2;
    __shared__ float sn[SW * SH];                            /* [x][y], halo 1 */
    __shared__ int s_max;
    if (pal_n) {                                             /* only the images the compact form could not hold */
        const int pair = first_pair + by;
        const bool step = (flags[by] & EDT_FLAG_STEP) != 0;
        const bool part = (flags[by] & EDT_FLAG_PARTIAL) != 0;     /* a partial compact form: the texels are the image's complete form */
        if (pal_n[pair] > 0 && !step && !part) return;
        if (step && bx == 0 && threadIdx.x == 0) pal_n[pair] = -(int)PAL_STEP;   /* the other workgroups read the flag, not this */
        else if (part && bx == 0 && threadIdx.x == 0 && pal_n[pair] > 0) pal_n[pair] |= DVO_PAL_PARTIAL;
    }
    /* sparse texel slabs (round 4): the host has not mapped texel memory for these pairs yet -- this pass only settles pal_n;
     * the host reads the palette sizes back, maps the texels of the (rare) images that need them and runs this pass again */
    if (!out) return;
    d2 += (size_t)by * edt_g_count(rows, cols, R);      /* row blocks of R rows, like g */
    partial += (size_t)by * n_partial;
    out += (size_t)by * tex_stride;
    int m = 0;
    for (int k = threadIdx.x; k < n_partial; k += 256) { const int v = partial[k]; m = v > m ? v : m; }
    m = block_reduce_256<true>(m);
    if (threadIdx.x == 0) s_max = m;
    __syncthreads();
    const EdtScale sc = edt_scale((unsigned)s_max, rows, cols);
    for (int tile = bx; tile < n_tiles; tile += gx) {
    const int y0 = (tile % tiles_y) * NP_TY, x0 = (tile / tiles_y) * NP_TX;
    for (int idx = threadIdx.x; idx < SW * SH; idx += 256) {
        const int lx = idx / SH, ly = idx - lx * SH;
        int yy = y0 + ly - 1, xx = x0 + lx - 1;
        float v = 0.0f;
        if (yy <= rows && xx <= cols) {                      /* one pixel beyond the image is the reflected neighbour */
            yy = reflect101(yy, rows); xx = reflect101(xx, cols);
            v = edt_value(d2[edt_g_index(xx, yy, cols, R)], sc);
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
    __syncthreads();                                         /* the tile's values are consumed: the next tile may overwrite them */
    }
}

/* the kernels of the distance-transform stage: one level per launch ... */
template <int WAVES>
__global__ void __launch_bounds__(WAVES * 64)
edt_columns8_kernel(const unsigned char *__restrict__ edge, size_t edge_stride, int rows, int cols, int R, unsigned short *__restrict__ g,
                    unsigned *__restrict__ bitmap, int bm_words, int *__restrict__ flags) {
    edt_columns8_body<WAVES>(blockIdx.x, gridDim.x, blockIdx.y, edge, edge_stride, rows, cols, R, g, bitmap, bm_words, flags);
}
template <int R>
__global__ void __launch_bounds__(256)
edt_rows_pk_kernel(const unsigned short *__restrict__ g, int rows, int cols, unsigned *__restrict__ d2, int *__restrict__ partial,
                   unsigned *__restrict__ bitmap, int bm_words, int *__restrict__ flags) {
    edt_rows_pk_body<R>(blockIdx.x, gridDim.x, blockIdx.y, g, rows, cols, d2, partial, bitmap, bm_words, flags);
}
template <int BM_WORDS>
__global__ void __launch_bounds__(256)
edt_rank_pack_kernel(const unsigned *__restrict__ d2, int rows, int cols, int R, int tiles_y, int strip, const int *__restrict__ partial, int n_partial,
                     const unsigned *__restrict__ bitmap, int bm_words, int *__restrict__ flags,
                     unsigned *__restrict__ p4, size_t p4_stride, float2 *__restrict__ pal, int *__restrict__ pal_n, int first_pair,
                     const unsigned *__restrict__ unit_bits) {
    edt_rank_pack_body<BM_WORDS>(blockIdx.x, gridDim.x, blockIdx.y, d2, rows, cols, R, tiles_y, strip, partial, n_partial, bitmap, bm_words, flags,
                                 p4, p4_stride, pal, pal_n, first_pair, unit_bits);
}
__global__ void __launch_bounds__(256)
dt_normalize_gradient_pack_kernel(const unsigned *__restrict__ d2, int rows, int cols, int R, int tiles_y,
                                  const int *__restrict__ partial, int n_partial, float4 *__restrict__ out, size_t tex_stride,
                                  int *__restrict__ pal_n, const int *__restrict__ flags, int first_pair, int n_tiles) {
    dt_normalize_gradient_pack_body(blockIdx.x, gridDim.x, blockIdx.y, d2, rows, cols, R, tiles_y, partial, n_partial, out, tex_stride, pal_n, flags, first_pair, n_tiles);
}

/* ... and all pyramid levels per launch (see CannyLevels): the level table of the distance-transform stage.  One R (rows per
 * workgroup of the row pass) and one workgroup shape of the column pass for all levels: those of the largest. */
struct EdtLevels {
    int n, R, first_pair;
    int rows[DVO_LEVELS], cols[DVO_LEVELS], bm_words[DVO_LEVELS], n_partial[DVO_LEVELS], ptiles_y[DVO_LEVELS], strip[DVO_LEVELS], ntiles_y[DVO_LEVELS];
    unsigned first[DVO_LEVELS + 1];
    const unsigned char *edge[DVO_LEVELS]; size_t edge_stride[DVO_LEVELS];
    unsigned short *g[DVO_LEVELS]; unsigned *d2[DVO_LEVELS]; int *partial[DVO_LEVELS]; unsigned *bitmap[DVO_LEVELS]; int *flags[DVO_LEVELS];
    float4 *tex[DVO_LEVELS]; size_t tex_stride[DVO_LEVELS];
    unsigned *p4[DVO_LEVELS]; size_t p4_stride[DVO_LEVELS]; float2 *pal[DVO_LEVELS]; int *pal_n[DVO_LEVELS];
};
template <int WAVES>
__global__ void __launch_bounds__(WAVES * 64) edt_columns8_levels_kernel(const EdtLevels t) {
    const int l = level_of_block(t.first, t.n, blockIdx.x);
    edt_columns8_body<WAVES>((int)(blockIdx.x - t.first[l]), (int)(t.first[l + 1] - t.first[l]), blockIdx.y, t.edge[l], t.edge_stride[l], t.rows[l], t.cols[l],
                             t.R, t.g[l], t.bitmap[l], t.bm_words[l], t.flags[l]);
}
template <int R>
__global__ void __launch_bounds__(256) edt_rows_pk_levels_kernel(const EdtLevels t) {
    const int l = level_of_block(t.first, t.n, blockIdx.x);
    edt_rows_pk_body<R>((int)(blockIdx.x - t.first[l]), (int)(t.first[l + 1] - t.first[l]), blockIdx.y, t.g[l], t.rows[l], t.cols[l], t.d2[l], t.partial[l],
                        t.bitmap[l], t.bm_words[l], t.flags[l]);
}
template <int BM_WORDS>
__global__ void __launch_bounds__(256) edt_rank_pack_levels_kernel(const EdtLevels t) {
    const int l = level_of_block(t.first, t.n, blockIdx.x);
    edt_rank_pack_body<BM_WORDS>((int)(blockIdx.x - t.first[l]), (int)(t.first[l + 1] - t.first[l]), blockIdx.y, t.d2[l], t.rows[l], t.cols[l], t.R,
                                 t.ptiles_y[l], t.strip[l], t.partial[l], t.n_partial[l], t.bitmap[l], t.bm_words[l], t.flags[l],
                                 t.p4[l], t.p4_stride[l], t.pal[l], t.pal_n[l], t.first_pair, nullptr);
}
__global__ void __launch_bounds__(256) dt_normalize_gradient_pack_levels_kernel(const EdtLevels t, int with_p4) {
    const int l = level_of_block(t.first, t.n, blockIdx.x);
    dt_normalize_gradient_pack_body((int)(blockIdx.x - t.first[l]), (int)(t.first[l + 1] - t.first[l]), blockIdx.y, t.d2[l], t.rows[l], t.cols[l], t.R,
                                    t.ntiles_y[l], t.partial[l], t.n_partial[l], t.tex[l], t.tex_stride[l], with_p4 ? t.pal_n[l] : nullptr, t.flags[l],
                                    t.first_pair, t.ntiles_y[l] * ((t.cols[l] + NP_TX - 1) / NP_TX));
}

#include "dvo_edt_band.h"

/* compact form -> 16-byte texels {DT, gx, gy, w} of the images that have one (pal_n > 0), decoded exactly as the fused kernel
 * decodes a pixel (dvo_fused.hip, p4_decode2).  For the inspection and host-driven paths, which read the texels. */
__global__ void __launch_bounds__(256)
p4_decode_texels_kernel(const unsigned *__restrict__ p4, size_t p4_stride, const float2 *__restrict__ pal, const int *__restrict__ pal_n,
                        float4 *__restrict__ tex, size_t tex_stride, int rows, int cols, int first_pair) {
    const int pair = first_pair + blockIdx.y;
    if (pal_n[pair] <= 0 || pal_partial(pal_n[pair])) return;      /* no compact form / a partial one: the texels were written with it */
    p4 += (size_t)pair * p4_stride;
    pal += (size_t)pair * DVO_PAL_MAX;
    tex += (size_t)pair * tex_stride;
    const int tpc = p4_tiles_per_col(rows), tpc16 = texel_tiles_per_col(rows);
    const size_t n = (size_t)rows * cols;
    for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (size_t)gridDim.x * blockDim.x) {
        const int xx = (int)(p / rows), yy = (int)(p - (size_t)xx * rows);
        const int ty = yy / DVO_P4_ROWS;
        const size_t slot = p4_slot(ty, yy - ty * DVO_P4_ROWS + 1, xx, tpc);
        const unsigned wu = p4[slot - 1], wc = p4[slot], wd = p4[slot + 1];
        const int c = (int)((wc >> 3) & 0x1fffu);
        const int r = c + (int)(signed char)((wc >> 16) & 0xffu), l = c + (int)(signed char)(wc >> 24);
        const float2 pc = pal[c];
        const float pr = pal[r].x, pl = pal[l].x, pu = pal[(wu >> 3) & 0x1fffu].x, pd = pal[(wd >> 3) & 0x1fffu].x;
        tex[texel_index(yy, xx, tpc16)] = make_float4(pc.x, (pr - pl) * 0.5f, (pd - pu) * 0.5f, pc.y);
    }
}
hipError_t launch_p4_decode_texels(const unsigned *p4, size_t p4_stride, const float2 *pal, const int *pal_n, float4 *tex,
                                   size_t tex_stride, int rows, int cols, int first_pair, int count, hipStream_t s) {
    if (count <= 0) return hipSuccess;
    hipLaunchKernelGGL(p4_decode_texels_kernel, dim3(grid_x((size_t)rows * cols, 256), count), dim3(256), 0, s, p4, p4_stride, pal, pal_n,
                       tex, tex_stride, rows, cols, first_pair);
    return hipGetLastError();
}

/* ---- round 6: the band stage (dvo_edt_band.h) ---- */
struct EdtLevelShape { int R, waves; size_t lds_cols, lds_rows; };
static bool edt_band_enabled() {
    static const bool on = [] { const char *e = getenv("DVO_EDT_FUSED"); return !(e && e[0] == '0'); }();
    return on;
}
/* the two shapes of the band kernel: T tile rows per band */
static int edt_band_row_slots(int T) { return (6 * T + 2 + 7) & ~7; }
static size_t edt_band_lds(int cols, int T) { return (size_t)(cols + 2 * EB_PAD) * edt_band_row_slots(T) * 2 + EB_DIRECT * 2; }
static int edt_band_T_for(int count, int tiles) {                 /* tiles: tile rows of all levels of one image */
    static const int forced = [] { const char *e = getenv("DVO_EDT_BAND_T"); const int v = e ? atoi(e) : 0; return (v == 2 || v == 5) ? v : 0; }();
    if (forced) return forced;
    return ((long long)count * tiles >= 5 * 768) ? 5 : 2;         /* T = 5 once its 512-thread workgroups fill the chip (three per CU) */
}
static bool edt_band_ok(int rows, int cols) {
    return edt_band_enabled() && rows >= 2 && cols >= 2 && rows <= EB_MAX_ROWS && cols <= EB_MAX_COLS;
}
/* scratch of the band stage for `count` images of one level, in ints: column words | their carries | image maxima | band counters */
static size_t edt_band_ints(int rows, int cols, int count) {
    return (((size_t)((rows + 31) / 32) * cols * 2 + 2) * count + 3) & ~(size_t)3;
}
static size_t edt_band_list_ints(int count) { return ((size_t)DVO_LEVELS * count + 1 + 3) & ~(size_t)3; }
static int *edt_band_carve(int *w, int rows, int cols, int count, EdtBandLevels &tb, int l) {
    const size_t nw = (size_t)((rows + 31) / 32) * cols * count;
    tb.maskT[l] = reinterpret_cast<unsigned *>(w);
    tb.carryT[l] = tb.maskT[l] + nw;
    tb.imax[l] = tb.carryT[l] + nw;
    tb.done[l] = reinterpret_cast<int *>(tb.imax[l] + count);
    return w + edt_band_ints(rows, cols, count);
}

/* LDS rows per workgroup of the row pass.  R >= 2: the packed kernel (two 16-bit tiles, the scanned one padded); R = 1 (rows of
 * more than ~8 K columns): the 32-bit kernel on g^2, beyond 16 K columns on 16-bit g */
static size_t edt_pk_lds_bytes(int cols, int R) { return (size_t)(cols + 2 * EDT_PK_PAD) * (R / 2) * 4 + (size_t)cols * R * 2; }
static int edt_rows_per_block(int cols) {
    static const int start = [] { const char *e = getenv("DVO_EDT_ROWS"); const int v = e ? atoi(e) : 0; return (v == 16 || v == 4 || v == 2) ? v : 8; }();
    for (int R = start; R > 1; R >>= 1)
        if (edt_pk_lds_bytes(cols, R) <= 64 * 1024) return R;
    return 1;                                         /* cols < 46340: at most 91 KiB of 16-bit g */
}
static unsigned edt_row_blocks(int rows, int cols) { const int R = edt_rows_per_block(cols); return (unsigned)((rows + R - 1) / R); }
/* scratch of one launch_edges_to_now over `count` images, in ints: g (16 bit) | d2 | per-block maxima | bitmaps | flags */
size_t edt_work_ints(int rows, int cols, int count) {
    const size_t ng = edt_g_count(rows, cols, edt_rows_per_block(cols));
    return ((ng + 1) / 2 + 4 + ng + edt_row_blocks(rows, cols) + (size_t)edt_bitmap_words(rows, cols) + 1) * count + 64
           + edt_band_ints(rows, cols, count) + edt_band_list_ints(count) + 8;
}

template <int R, typename T>
static hipError_t edt_rows_launch(const unsigned short *g, ImgBatch gb, unsigned nblk, unsigned *d2, int *partial, unsigned *bitmap,
                                  int bm_words, int *flags, hipStream_t s) {
    const size_t lds = (size_t)gb.cols * R * sizeof(T);
    auto kern = edt_rows16_kernel<R, T>;
    if (lds > 48 * 1024) {
        const hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3(nblk, gb.count), dim3(256), lds, s, g, gb.rows, gb.cols, d2, partial, bitmap, bm_words, flags);
    return hipGetLastError();
}

template <int R>
static hipError_t edt_rows_pk_launch(const unsigned short *g, ImgBatch gb, unsigned nblk, unsigned *d2, int *partial, unsigned *bitmap,
                                     int bm_words, int *flags, hipStream_t s) {
    const size_t lds = edt_pk_lds_bytes(gb.cols, R);
    auto kern = edt_rows_pk_kernel<R>;
    if (lds > 48 * 1024) {
        const hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3(nblk, gb.count), dim3(256), lds, s, g, gb.rows, gb.cols, d2, partial, bitmap, bm_words, flags);
    return hipGetLastError();
}

/* the band stage for the levels of `t` (the three-pass stage's table: its scratch serves the images on the list): column words,
 * bands, the listed images' exact columns + rows, palettes.  tb: scratch pointers carved, everything else filled here. */
template <int WAVES>
static void edt_columns8_list_launch(const EdtLevels &t, const EdtListShape &ls, const int *list, size_t lds, hipStream_t s) {
    hipLaunchKernelGGL(edt_columns8_list_kernel<WAVES>, dim3(64), dim3(WAVES * 64), lds, s, t, ls, list);
}
template <int R>
static hipError_t edt_rows_pk_list_launch(const EdtLevels &t, const EdtListShape &ls, const int *list, size_t lds, hipStream_t s) {
    auto kern = edt_rows_pk_list_kernel<R>;
    if (lds > 48 * 1024) {
        const hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3(64), dim3(256), lds, s, t, ls, list);
    return hipGetLastError();
}
template <int T, int THREADS, int NI>
static hipError_t edt_band_kernel_launch(const EdtBandLevels &tb, unsigned g, int count, size_t lds, hipStream_t s) {
    auto kern = edt_band_levels_kernel<T, THREADS, NI>;
    if (lds > 48 * 1024) {
        const hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3(g, count), dim3(THREADS), lds, s, tb);
    return hipGetLastError();
}
static hipError_t edt_band_run(const EdtLevels &t, const EdtLevelShape &sh, EdtBandLevels &tb, int count, hipStream_t s) {
    const int n = t.n;
    int cmax = 0;
    for (int l = 0; l < n; l++) cmax = t.cols[l] > cmax ? t.cols[l] : cmax;
    int tiles = 0;
    for (int l = 0; l < n; l++) tiles += p4_tiles_per_col(t.rows[l]);
    const int T = edt_band_T_for(count, tiles);
    if (cmax > EB_MAX_COLS || (sh.waves != 8 && sh.waves != 4)) return hipErrorInvalidValue;
    tb.n = n; tb.first_pair = t.first_pair;
    tb.firstA[0] = 0; tb.firstB[0] = 0;
    EdtListShape ls;
    ls.gxmax_cols = 1; ls.gxmax_rows = 1;
    for (int l = 0; l < n; l++) {
        tb.rows[l] = t.rows[l]; tb.cols[l] = t.cols[l];
        tb.nwords[l] = (t.rows[l] + 31) / 32;
        tb.tpc[l] = p4_tiles_per_col(t.rows[l]);
        tb.nbands[l] = (tb.tpc[l] + T - 1) / T;
        tb.n_partial[l] = t.n_partial[l];
        tb.edge[l] = t.edge[l]; tb.edge_stride[l] = t.edge_stride[l];
        tb.flags[l] = t.flags[l]; tb.partial[l] = t.partial[l];
        tb.p4[l] = t.p4[l]; tb.p4_stride[l] = t.p4_stride[l]; tb.pal[l] = t.pal[l]; tb.pal_n[l] = t.pal_n[l];
        tb.firstA[l + 1] = tb.firstA[l] + (unsigned)((t.cols[l] + 31) / 32);
        tb.firstB[l + 1] = tb.firstB[l] + (unsigned)tb.nbands[l];
        const int ngroups = (t.cols[l] + sh.waves - 1) / sh.waves;
        ls.gx_cols[l] = ngroups < 16 ? ngroups : 16;
        ls.gxmax_cols = ls.gx_cols[l] > ls.gxmax_cols ? ls.gx_cols[l] : ls.gxmax_cols;
        ls.gxmax_rows = t.n_partial[l] > ls.gxmax_rows ? t.n_partial[l] : ls.gxmax_rows;
    }
    for (int l = n; l < DVO_LEVELS; l++) ls.gx_cols[l] = 0;
    hipLaunchKernelGGL(edt_colmask_levels_kernel, dim3(tb.firstA[n], count), dim3(512), 0, s, tb);
    hipError_t e;
    const size_t lds = edt_band_lds(cmax, T);
    if (T == 5) e = (cmax <= 640) ? edt_band_kernel_launch<5, 512, 5>(tb, tb.firstB[n], count, lds, s) : edt_band_kernel_launch<5, 512, 8>(tb, tb.firstB[n], count, lds, s);
    else e = (cmax <= 640) ? edt_band_kernel_launch<2, 256, 5>(tb, tb.firstB[n], count, lds, s) : edt_band_kernel_launch<2, 256, 8>(tb, tb.firstB[n], count, lds, s);
    if (e != hipSuccess) return e;
    if (sh.waves == 8) edt_columns8_list_launch<8>(t, ls, tb.list, sh.lds_cols, s);
    else edt_columns8_list_launch<4>(t, ls, tb.list, sh.lds_cols, s);
    switch (sh.R) {
    case 16: e = edt_rows_pk_list_launch<16>(t, ls, tb.list, sh.lds_rows, s); break;
    case 8: e = edt_rows_pk_list_launch<8>(t, ls, tb.list, sh.lds_rows, s); break;
    case 4: e = edt_rows_pk_list_launch<4>(t, ls, tb.list, sh.lds_rows, s); break;
    case 2: e = edt_rows_pk_list_launch<2>(t, ls, tb.list, sh.lds_rows, s); break;
    default: return hipErrorInvalidValue;
    }
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(edt_palette_levels_kernel, dim3(n, count), dim3(256), 0, s, tb);
    return hipGetLastError();
}

/* edge masks -> resident now levels of pairs first_pair .. first_pair + count - 1.  With p4 != NULL the compact form is what is
 * written (16-byte texels only for the images it cannot hold); with p4 == NULL the 16-byte texels of every image. */
hipError_t launch_edges_to_now(const unsigned char *edge, size_t edge_stride, ImgBatch gb, int *work,
                               float4 *tex_out, size_t tex_stride, unsigned *p4, size_t p4_stride, float2 *pal, int *pal_n,
                               int first_pair, hipStream_t s, bool only_texels) {
    const int R = edt_rows_per_block(gb.cols);
    const unsigned nblk = edt_row_blocks(gb.rows, gb.cols);
    const int bm_words = edt_bitmap_words(gb.rows, gb.cols);
    /* carve the scratch: 4-byte aligned sections */
    const size_t ng = edt_g_count(gb.rows, gb.cols, R);       /* a multiple of 8 for every R >= 8; 16-byte aligned images then */
    unsigned short *g = reinterpret_cast<unsigned short *>(work);
    unsigned *d2 = reinterpret_cast<unsigned *>(work) + (((ng * gb.count + 1) / 2 + 3) & ~(size_t)3);
    int *partial = reinterpret_cast<int *>(d2 + ng * gb.count);
    unsigned *bitmap = reinterpret_cast<unsigned *>(partial + (size_t)nblk * gb.count);
    int *flags = reinterpret_cast<int *>(bitmap + (size_t)bm_words * gb.count);
    hipError_t e;
    /* round 6: one pass from the edge mask to the rank words (dvo_edt_band.h); the three-pass stage below for what it does not take */
    const bool band = p4 && !only_texels && R >= 2 && edt_band_ok(gb.rows, gb.cols);
    if (band) {
        EdtLevels t{};
        t.n = 1; t.R = R; t.first_pair = first_pair;
        t.rows[0] = gb.rows; t.cols[0] = gb.cols; t.bm_words[0] = bm_words; t.n_partial[0] = (int)nblk;
        t.edge[0] = edge; t.edge_stride[0] = edge_stride;
        t.g[0] = g; t.d2[0] = d2; t.partial[0] = partial; t.bitmap[0] = bitmap; t.flags[0] = flags;
        t.p4[0] = p4; t.p4_stride[0] = p4_stride; t.pal[0] = pal; t.pal_n[0] = pal_n;
        EdtLevelShape sh;
        const size_t lds_wave = 16;                           /* rows <= 512: one chunk */
        sh.R = R; sh.waves = (R <= 8) ? 8 : 4; sh.lds_cols = lds_wave * sh.waves; sh.lds_rows = edt_pk_lds_bytes(gb.cols, R);
        EdtBandLevels tb{};
        int *wb = edt_band_carve(flags + gb.count, gb.rows, gb.cols, gb.count, tb, 0);
        tb.list = wb;
        if ((e = edt_band_run(t, sh, tb, gb.count, s)) != hipSuccess) return e;
        if (getenv("DVO_EDT_DEBUG")) {
            (void)hipStreamSynchronize(s);
            unsigned im = 0; int fl = 0, pn = 0, dn = 0, li = 0;
            (void)hipMemcpy(&im, tb.imax[0], 4, hipMemcpyDeviceToHost); (void)hipMemcpy(&fl, tb.flags[0], 4, hipMemcpyDeviceToHost);
            (void)hipMemcpy(&dn, tb.done[0], 4, hipMemcpyDeviceToHost); (void)hipMemcpy(&li, tb.list, 4, hipMemcpyDeviceToHost);
            (void)hipMemcpy(&pn, pal_n + first_pair, 4, hipMemcpyDeviceToHost);
            fprintf(stderr, "[edt band] %dx%d imax %u flags %d done %d list %d pal_n %d nbands %d\n", gb.rows, gb.cols, im, fl, dn, li, pn, tb.nbands[0]);
        }
    }
    if (!only_texels && !band) {
        const size_t lds_wave = (((size_t)((gb.rows + 511) / 512) * 2 * sizeof(int)) + 15) & ~(size_t)15;      /* per wave: two border distances per 512-row chunk */
        if (R <= 8 && lds_wave * 8 <= 48 * 1024) {            /* eight adjacent columns complete a 128-byte line of 8-row blocks */
            const int cg = gb.count >= 64 ? 4 : (gb.count >= 16 ? 2 : 1);      /* column groups per workgroup */
            hipLaunchKernelGGL(edt_columns8_kernel<8>, dim3(((gb.cols + 7) / 8 + cg - 1) / cg, gb.count), dim3(512), lds_wave * 8, s, edge, edge_stride,
                               gb.rows, gb.cols, R, g, bitmap, bm_words, flags);
        } else if (lds_wave * 4 <= 48 * 1024) {
            hipLaunchKernelGGL(edt_columns8_kernel<4>, dim3((gb.cols + 3) / 4, gb.count), dim3(256), lds_wave * 4, s, edge, edge_stride,
                               gb.rows, gb.cols, R, g, bitmap, bm_words, flags);
        } else {
            auto kern = edt_columns8_kernel<1>;
            if ((e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_wave)) != hipSuccess) return e;
            hipLaunchKernelGGL(kern, dim3(gb.cols, gb.count), dim3(64), lds_wave, s, edge, edge_stride, gb.rows, gb.cols, R, g, bitmap,
                               bm_words, flags);
        }
    }
    if (only_texels || band) e = hipSuccess;
    else switch (R) {
    case 16: e = edt_rows_pk_launch<16>(g, gb, nblk, d2, partial, bitmap, bm_words, flags, s); break;
    case 8: e = edt_rows_pk_launch<8>(g, gb, nblk, d2, partial, bitmap, bm_words, flags, s); break;
    case 4: e = edt_rows_pk_launch<4>(g, gb, nblk, d2, partial, bitmap, bm_words, flags, s); break;
    case 2: e = edt_rows_pk_launch<2>(g, gb, nblk, d2, partial, bitmap, bm_words, flags, s); break;
    default:
        e = ((size_t)gb.cols * 4 <= 64 * 1024) ? edt_rows_launch<1, unsigned>(g, gb, nblk, d2, partial, bitmap, bm_words, flags, s)
                                               : edt_rows_launch<1, unsigned short>(g, gb, nblk, d2, partial, bitmap, bm_words, flags, s);
        break;
    }
    if (e != hipSuccess) return e;
    if (p4 && !only_texels && !band) {
        const int ptiles_y = (p4_tiles_per_col(gb.rows) + PK_LR - 1) / PK_LR, ptiles_x = (((gb.cols + 3) >> 2) + PK_LC - 1) / PK_LC;
        /* tiles per workgroup: whole tile columns for large batches (the rank table is built once per workgroup), single tiles
         * when the launch would not fill the GPU otherwise (one camera stream) */
        long long strip = (long long)gb.count * ptiles_y * ptiles_x / 2048;
        strip = strip < 1 ? 1 : (strip > ptiles_y ? ptiles_y : strip);
        const int n_strips = (ptiles_y + (int)strip - 1) / (int)strip;
        hipLaunchKernelGGL(edt_rank_pack_kernel<PK_SMALL_WORDS>, dim3(n_strips * ptiles_x, gb.count), dim3(256), 0, s, d2, gb.rows, gb.cols, R, ptiles_y,
                           (int)strip, partial, (int)nblk, bitmap, bm_words, flags, p4, p4_stride, pal, pal_n, first_pair, nullptr);
        if (bm_words > PK_SMALL_WORDS)            /* squared distances of 65536 and more are possible at this size: the full-bitmap twin */
            hipLaunchKernelGGL(edt_rank_pack_kernel<DVO_EDT_BITMAP_BITS / 32>, dim3(std::min(n_strips * ptiles_x, 8), gb.count), dim3(256), 0, s, d2, gb.rows,
                               gb.cols, R, ptiles_y, (int)strip, partial, (int)nblk, bitmap, bm_words, flags, p4, p4_stride, pal, pal_n, first_pair, nullptr);
    }
    const int tiles_y = (gb.rows + NP_TY - 1) / NP_TY, tiles_x = (gb.cols + NP_TX - 1) / NP_TX;
    /* with a compact form nearly every image leaves at once: a few workgroups per image then, one per tile otherwise */
    const int pack_wgs = p4 ? std::min(tiles_y * tiles_x, 8) : tiles_y * tiles_x;
    hipLaunchKernelGGL(dt_normalize_gradient_pack_kernel, dim3(pack_wgs, gb.count), dim3(256), 0, s, d2, gb.rows, gb.cols, R,
                       tiles_y, partial, (int)nblk, tex_out, tex_stride, p4 ? pal_n : nullptr, flags, first_pair, tiles_y * tiles_x);
    return hipGetLastError();
}

/* launch_edges_to_now for all pyramid levels of the same `count` images at once: five launches (six when squared distances of
 * 65536 and more are possible) instead of that many per level.  work: edt_levels_work_ints() ints; p4 all NULL or all set. */
static bool edt_levels_shape(int n, const int *rows, const int *cols, EdtLevelShape &sh) {
    if (n < 2 || n > DVO_LEVELS) return false;
    int r0 = 0, c0 = 0;
    for (int l = 0; l < n; l++) { if (rows[l] < 2 || cols[l] < 2) return false; r0 = rows[l] > r0 ? rows[l] : r0; c0 = cols[l] > c0 ? cols[l] : c0; }
    sh.R = edt_rows_per_block(c0);
    if (sh.R < 2) return false;
    const size_t lds_wave = (((size_t)((r0 + 511) / 512) * 2 * sizeof(int)) + 15) & ~(size_t)15;       /* per wave: two border distances per 512-row chunk */
    sh.waves = (sh.R <= 8 && lds_wave * 8 <= 48 * 1024) ? 8 : ((lds_wave * 4 <= 48 * 1024) ? 4 : 0);
    if (!sh.waves) return false;
    sh.lds_cols = lds_wave * sh.waves;
    sh.lds_rows = edt_pk_lds_bytes(c0, sh.R);
    return true;
}
bool edt_levels_ok(int n, const int *rows, const int *cols) { EdtLevelShape sh; return edt_levels_shape(n, rows, cols, sh); }
static size_t edt_level_ints(int rows, int cols, int R, int count) {
    const size_t ng = edt_g_count(rows, cols, R);
    return (((ng * count + 1) / 2 + 3) & ~(size_t)3) + ng * count + ((size_t)((rows + R - 1) / R) + (size_t)edt_bitmap_words(rows, cols) + 1) * count + 8;
}
size_t edt_levels_work_ints(int n, const int *rows, const int *cols, int count) {
    EdtLevelShape sh;
    if (!edt_levels_shape(n, rows, cols, sh)) return 0;
    size_t t = 0;
    for (int l = 0; l < n; l++) t += ((edt_level_ints(rows[l], cols[l], sh.R, count) + 3) & ~(size_t)3) + edt_band_ints(rows[l], cols[l], count);
    return t + 64 + edt_band_list_ints(count);
}
template <int R>
static hipError_t edt_rows_pk_levels_launch(const EdtLevels &t, unsigned g, int count, size_t lds, hipStream_t s) {
    auto kern = edt_rows_pk_levels_kernel<R>;
    if (lds > 48 * 1024) {
        const hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3(g, count), dim3(256), lds, s, t);
    return hipGetLastError();
}
hipError_t launch_edges_to_now_levels(int n, const int *rows, const int *cols, const unsigned char *const *edge, const size_t *edge_stride, int count,
                                      int *work, float4 *const *tex_out, const size_t *tex_stride, unsigned *const *p4, const size_t *p4_stride,
                                      float2 *const *pal, int *const *pal_n, int first_pair, hipStream_t s, bool only_texels) {
    EdtLevelShape sh;
    if (!edt_levels_shape(n, rows, cols, sh)) return hipErrorInvalidValue;
    EdtLevels t;
    t.n = n; t.R = sh.R; t.first_pair = first_pair;
    const bool with_p4 = p4 && p4[0];
    int *w = work;
    int max_bm = 0;
    for (int l = 0; l < n; l++) {
        const size_t ng = edt_g_count(rows[l], cols[l], sh.R);
        const int nblk = (rows[l] + sh.R - 1) / sh.R;
        t.rows[l] = rows[l]; t.cols[l] = cols[l]; t.bm_words[l] = edt_bitmap_words(rows[l], cols[l]); t.n_partial[l] = nblk;
        max_bm = t.bm_words[l] > max_bm ? t.bm_words[l] : max_bm;
        t.edge[l] = edge[l]; t.edge_stride[l] = edge_stride[l];
        t.g[l] = reinterpret_cast<unsigned short *>(w);
        t.d2[l] = reinterpret_cast<unsigned *>(w) + (((ng * count + 1) / 2 + 3) & ~(size_t)3);
        t.partial[l] = reinterpret_cast<int *>(t.d2[l] + ng * count);
        t.bitmap[l] = reinterpret_cast<unsigned *>(t.partial[l] + (size_t)nblk * count);
        t.flags[l] = reinterpret_cast<int *>(t.bitmap[l] + (size_t)t.bm_words[l] * count);
        w += (edt_level_ints(rows[l], cols[l], sh.R, count) + 3) & ~(size_t)3;
        t.tex[l] = tex_out[l]; t.tex_stride[l] = tex_stride[l];
        t.p4[l] = with_p4 ? p4[l] : nullptr; t.p4_stride[l] = p4_stride[l]; t.pal[l] = pal[l]; t.pal_n[l] = pal_n[l];
        const int pty = (p4_tiles_per_col(rows[l]) + PK_LR - 1) / PK_LR, ptx = (((cols[l] + 3) >> 2) + PK_LC - 1) / PK_LC;
        long long strip = (long long)count * pty * ptx / 2048;
        strip = strip < 1 ? 1 : (strip > pty ? pty : strip);
        t.ptiles_y[l] = pty; t.strip[l] = (int)strip;
        t.ntiles_y[l] = (rows[l] + NP_TY - 1) / NP_TY;
    }
    auto prefix = [&](auto blocks_of) { t.first[0] = 0; for (int l = 0; l < n; l++) t.first[l + 1] = t.first[l] + blocks_of(l); return t.first[n]; };
    hipError_t e = hipSuccess;
    unsigned g = 0;
    /* round 6: one pass from the edge masks to the rank words (dvo_edt_band.h) when every level qualifies */
    bool band = with_p4 && !only_texels;
    for (int l = 0; l < n && band; l++) band = edt_band_ok(rows[l], cols[l]);
    if (band) {
        EdtBandLevels tb{};
        int *wb = w;
        for (int l = 0; l < n; l++) wb = edt_band_carve(wb, rows[l], cols[l], count, tb, l);
        tb.list = wb;
        if ((e = edt_band_run(t, sh, tb, count, s)) != hipSuccess) return e;
    }
    if (!only_texels && !band) {
        /* column groups per workgroup: several for large batches (the workgroup's table and launch are paid once) */
        const int cg = count >= 64 ? 4 : (count >= 16 ? 2 : 1);
        g = prefix([&](int l) { const int ng = (cols[l] + sh.waves - 1) / sh.waves; return (unsigned)((ng + cg - 1) / cg); });
        if (sh.waves == 8) hipLaunchKernelGGL(edt_columns8_levels_kernel<8>, dim3(g, count), dim3(512), sh.lds_cols, s, t);
        else hipLaunchKernelGGL(edt_columns8_levels_kernel<4>, dim3(g, count), dim3(256), sh.lds_cols, s, t);
        g = prefix([&](int l) { return (unsigned)t.n_partial[l]; });
        switch (sh.R) {
        case 16: e = edt_rows_pk_levels_launch<16>(t, g, count, sh.lds_rows, s); break;
        case 8: e = edt_rows_pk_levels_launch<8>(t, g, count, sh.lds_rows, s); break;
        case 4: e = edt_rows_pk_levels_launch<4>(t, g, count, sh.lds_rows, s); break;
        default: e = edt_rows_pk_levels_launch<2>(t, g, count, sh.lds_rows, s); break;
        }
    }
    if (e != hipSuccess) return e;
    if (with_p4 && !only_texels && !band) {
        g = prefix([&](int l) {
            const int ptx = (((cols[l] + 3) >> 2) + PK_LC - 1) / PK_LC, n_strips = (t.ptiles_y[l] + t.strip[l] - 1) / t.strip[l];
            return (unsigned)(n_strips * ptx);
        });
        hipLaunchKernelGGL(edt_rank_pack_levels_kernel<PK_SMALL_WORDS>, dim3(g, count), dim3(256), 0, s, t);
        if (max_bm > PK_SMALL_WORDS) {          /* images with a squared distance >= 65536: rare, a few workgroups per image and level loop over the strips */
            g = prefix([&](int l) {
                const int ptx = (((cols[l] + 3) >> 2) + PK_LC - 1) / PK_LC, n_strips = (t.ptiles_y[l] + t.strip[l] - 1) / t.strip[l];
                return std::min((unsigned)(n_strips * ptx), 8u);
            });
            hipLaunchKernelGGL(edt_rank_pack_levels_kernel<DVO_EDT_BITMAP_BITS / 32>, dim3(g, count), dim3(256), 0, s, t);
        }
    }
    g = prefix([&](int l) { const unsigned nt = (unsigned)(t.ntiles_y[l] * ((cols[l] + NP_TX - 1) / NP_TX)); return with_p4 ? std::min(nt, 8u) : nt; });
    hipLaunchKernelGGL(dt_normalize_gradient_pack_levels_kernel, dim3(g, count), dim3(256), 0, s, t, with_p4 ? 1 : 0);
    return hipGetLastError();
}

/* ------------------------------------------------------------------------- */
/* Caller-supplied float images (dvo_set_now_level: DT, gx, gy as the reference keeps them) -> the compact form, directly.
 * A normalised exact distance transform is DT = (float)sqrt(d2) * s with integer d2, and its smallest positive value is s itself
 * (a pixel next to an edge pixel: sqrt(1) * s).  So: (1) s = min positive DT; (2) d2 = round((DT / s)^2) per pixel, accepted
 * only if (float)sqrt(d2) * s reproduces DT bit for bit, recorded in the presence bitmap like the row pass does; (3) the rank-pack
 * pass of the native builder, with s as the scale of the palette; (4) the caller's gx / gy are compared bit for bit with what the
 * kernel will decode from the rank words.  Any mismatch (another distance transform, gradients that are not imageGradient(DT),
 * too many / too distant values) and pal_n = -reason: the pair keeps its 16-byte texels, which are written in any case.
 * 35 bytes of HBM traffic per pixel in four launches; the generic hash / sort builder (dvo_palette.hip) reads 16-byte texels
 * twice and remains for images that are not exact distance transforms. */
__global__ void __launch_bounds__(256)
float_level_unit_kernel(const float *__restrict__ dt, size_t n, unsigned *__restrict__ unit_bits, int *__restrict__ flags) {
    unsigned lo = 0xffffffffu;
    bool bad = false;
    for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (size_t)gridDim.x * blockDim.x) {
        const float v = dt[p];
        if (!(v >= 0.0f) || v > 3.0e38f) bad = true;                 /* NaN, negative, infinite */
        else if (v > 0.0f) { const unsigned b = __float_as_uint(v); lo = b < lo ? b : lo; }     /* positive floats order like their bits */
    }
    /* max of ~bits = min of bits; the reduction is signed, so the sign bit is flipped on the way in and out.  One atomic per
     * workgroup: the unit value sits next to every edge pixel of the image */
    const int m = block_reduce_256<true>((int)(~lo ^ 0x80000000u));
    if (threadIdx.x == 0 && m != (int)0x80000000u) atomicMax(unit_bits, (unsigned)m ^ 0x80000000u);
    if (__syncthreads_or(bad ? 1 : 0) && threadIdx.x == 0) atomicOr(flags, (int)EDT_FLAG_BAD);
}
__global__ void __launch_bounds__(256)
float_level_d2_kernel(const float *__restrict__ dt, int rows, int cols, int R, const unsigned *__restrict__ unit_bits,
                      unsigned *__restrict__ d2, int *__restrict__ partial, unsigned *__restrict__ bitmap, int bm_words, int *__restrict__ flags) {
    const unsigned ub = *unit_bits;
    const float unit = ub ? __uint_as_float(~ub) : 0.0f;              /* no positive value at all: an all-zero image, scale 0 */
    const EdtScale sc{unit, 0.0f};
    const size_t n = (size_t)rows * cols;
    __shared__ unsigned lbits[EDT_LBITS_WORDS];                     /* the workgroup's copy of the presence bitmap, as in the row pass */
    for (int i = threadIdx.x; i < EDT_LBITS_WORDS; i += 256) lbits[i] = 0u;
    __syncthreads();
    unsigned mx = 0;
    bool bad = false, far = false;
    for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (size_t)gridDim.x * blockDim.x) {
        const float v = dt[p];
        const int xx = (int)(p / rows), yy = (int)(p - (size_t)xx * rows);
        unsigned d = 0u;
        if (v > 0.0f && unit > 0.0f) {
            const double q = (double)v / (double)unit, dd = rint(q * q);
            if (dd < 2147483648.0) d = (unsigned)dd; else bad = true;
        }
        if (__float_as_uint(edt_value(d, sc)) != __float_as_uint(v)) bad = true;
        d2[edt_g_index(xx, yy, cols, R)] = d;
        mx = d > mx ? d : mx;
        const unsigned w = d >> 5, bit = 1u << (d & 31u);
        if (w < (unsigned)EDT_LBITS_WORDS) { if (!(__hip_atomic_load(&lbits[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) & bit)) atomicOr(&lbits[w], bit); }
        else if (w < (unsigned)bm_words) { if (!(__hip_atomic_load(bitmap + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & bit)) atomicOr(bitmap + w, bit); }
        else far = true;
    }
    const int m = block_reduce_256<true>((int)(mx > 0x7fffffffu ? 0x7fffffffu : mx));
    if (threadIdx.x == 0 && m > 0) atomicMax(partial, m);
    __syncthreads();
    edt_flush_lbits(lbits, bitmap, bm_words);
    const int f = (bad ? EDT_FLAG_BAD : 0) | (far ? EDT_FLAG_FAR : 0);
    const int any = __syncthreads_or(f & EDT_FLAG_BAD) ? EDT_FLAG_BAD : 0, anyfar = __syncthreads_or(f & EDT_FLAG_FAR) ? EDT_FLAG_FAR : 0;
    if (threadIdx.x == 0 && (any | anyfar)) atomicOr(flags, any | anyfar);
}
/* what the fused kernel will decode from the rank words against the caller's gradient images (DT was checked value by value) */
__global__ void __launch_bounds__(256)
compact_verify_planes_kernel(const unsigned *__restrict__ p4, const float2 *__restrict__ pal, int *__restrict__ pal_n, int pair,
                             const int *__restrict__ flags, const float *__restrict__ dt, const float *__restrict__ gx,
                             const float *__restrict__ gy, int rows, int cols) {
    if (pal_n[pair] <= 0) return;                                     /* refused by the pack pass already */
    if (*flags & EDT_FLAG_STEP) { if (blockIdx.x == 0 && threadIdx.x == 0) pal_n[pair] = -(int)PAL_STEP; return; }
    const int tpc = p4_tiles_per_col(rows);
    const size_t n = (size_t)rows * cols;
    bool bad = false;
    for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (size_t)gridDim.x * blockDim.x) {
        const int xx = (int)(p / rows), yy = (int)(p - (size_t)xx * rows);
        const int ty = yy / DVO_P4_ROWS;
        const size_t slot = p4_slot(ty, yy - ty * DVO_P4_ROWS + 1, xx, tpc);
        const unsigned wu = p4[slot - 1], wc = p4[slot], wd = p4[slot + 1];
        const int c = (int)((wc >> 3) & 0x1fffu);
        const int r = c + (int)(signed char)((wc >> 16) & 0xffu), l = c + (int)(signed char)(wc >> 24);
        const float pc = pal[c].x, pr = pal[r].x, pl = pal[l].x, pu = pal[(wu >> 3) & 0x1fffu].x, pd = pal[(wd >> 3) & 0x1fffu].x;
        if (__float_as_uint(pc) != __float_as_uint(dt[p]) || __float_as_uint((pr - pl) * 0.5f) != __float_as_uint(gx[p]) ||
            __float_as_uint((pd - pu) * 0.5f) != __float_as_uint(gy[p])) bad = true;
    }
    if (__syncthreads_or(bad ? 1 : 0) && threadIdx.x == 0) pal_n[pair] = -(int)PAL_GRADIENT;
}
/* work: float_level_work_ints() ints.  The level's 16-byte texels are written by the caller as before. */
size_t float_level_work_ints(int rows, int cols) {
    return edt_g_count(rows, cols, edt_rows_per_block(cols)) + (size_t)edt_bitmap_words(rows, cols) + 8;
}
hipError_t launch_float_level_to_compact(const float *dt, const float *gx, const float *gy, int rows, int cols, int *work,
                                         unsigned *p4, size_t p4_stride, float2 *pal, int *pal_n, int pair, hipStream_t s) {
    const int R = edt_rows_per_block(cols);
    const size_t ng = edt_g_count(rows, cols, R), n = (size_t)rows * cols;
    const int bm_words = edt_bitmap_words(rows, cols);
    unsigned *d2 = reinterpret_cast<unsigned *>(work);
    unsigned *bitmap = d2 + ng;
    int *partial = reinterpret_cast<int *>(bitmap + bm_words);       /* [0] max d2, [1] flags, [2] ~bits of the unit value */
    int *flags = partial + 1;
    unsigned *unit_bits = reinterpret_cast<unsigned *>(partial + 2);
    hipError_t e = hipMemsetAsync(bitmap, 0, sizeof(unsigned) * ((size_t)bm_words + 4), s);
    if (e != hipSuccess) return e;
    const unsigned gx_blocks = grid_x(n);
    hipLaunchKernelGGL(float_level_unit_kernel, dim3(gx_blocks), dim3(256), 0, s, dt, n, unit_bits, flags);
    hipLaunchKernelGGL(float_level_d2_kernel, dim3(gx_blocks), dim3(256), 0, s, dt, rows, cols, R, unit_bits, d2, partial, bitmap, bm_words, flags);
    const int ptiles_y = (p4_tiles_per_col(rows) + PK_LR - 1) / PK_LR, ptiles_x = (((cols + 3) >> 2) + PK_LC - 1) / PK_LC;
    hipLaunchKernelGGL(edt_rank_pack_kernel<PK_SMALL_WORDS>, dim3(ptiles_y * ptiles_x, 1), dim3(256), 0, s, d2, rows, cols, R, ptiles_y, 1,
                       partial, 1, bitmap, bm_words, flags, p4, p4_stride, pal, pal_n, pair, unit_bits);
    if (bm_words > PK_SMALL_WORDS)
        hipLaunchKernelGGL(edt_rank_pack_kernel<DVO_EDT_BITMAP_BITS / 32>, dim3(ptiles_y * ptiles_x, 1), dim3(256), 0, s, d2, rows, cols, R,
                           ptiles_y, 1, partial, 1, bitmap, bm_words, flags, p4, p4_stride, pal, pal_n, pair, unit_bits);
    hipLaunchKernelGGL(compact_verify_planes_kernel, dim3(gx_blocks), dim3(256), 0, s, p4 + (size_t)pair * p4_stride, pal + (size_t)pair * DVO_PAL_MAX,
                       pal_n, pair, flags, dt, gx, gy, rows, cols);
    return hipGetLastError();
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
                    float *__restrict__ xyz, size_t xyz_stride, uint2 *__restrict__ compact, unsigned *__restrict__ cidx,
                    float *__restrict__ uv, int capacity, int *__restrict__ N_dst) {
    const int xx = blockIdx.x, lane = threadIdx.x;
    edge += (size_t)blockIdx.y * edge_stride; depth += (size_t)blockIdx.y * depth_stride;
    col_offsets += (size_t)blockIdx.y * (cols + 2);
    if (blk_offsets) blk_offsets += (size_t)blockIdx.y * (((cols + 15) >> 4) * nby * 16 + 2);
    xyz += (size_t)blockIdx.y * xyz_stride;
    if (compact) compact += (size_t)blockIdx.y * (xyz_stride / 3);
    if (cidx) cidx += (size_t)blockIdx.y * (xyz_stride / 3);
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
                    if (cidx) cidx[bC] = (unsigned)nC;          /* where this point sits in the reference's list */
                }
                if (uv) { uv[2 * nC] = (float)xx; uv[2 * nC + 1] = (float)yy; }   /* :244-245 */
            }
        }
        run += __popcll(m);
    }
}

/* round 6: a caller's 3xN float list -> its compact twin, if it IS an enlistRefEdgePts list for these intrinsics.
 * X = (Z * ((float)xx - tmpcx)) * tmpfx (:249) is inverted to a candidate column, the three nearest integers are put through the very
 * expression enlist_write_kernel uses and compared bit for bit (Y and the row alike); a point whose X or Y no integer pixel
 * reproduces -- or whose pixel does not fit the 16-bit fields -- raises `fail`, and the list keeps the one-point-per-lane kernel.
 * Lossless by verification, like the compact form of caller-supplied now images: the packed kernel recomputes X, Y from {xx, yy, Z}
 * with the same expression, so it sees the caller's bits.  The twin is in the caller's order (cidx = identity). */
__global__ void __launch_bounds__(256)
points_recover_compact_kernel(const float *__restrict__ xyz, int N, int level, Intrinsics K, uint2 *__restrict__ compact,
                              unsigned *__restrict__ cidx, int *__restrict__ fail) {
    const float scaleFac = pow2_neg_f(level);
    const float tmpfx = (float)(1. / (double)(scaleFac * K.fx)), tmpfy = (float)(1. / (double)(scaleFac * K.fy));
    const float tmpcx = scaleFac * K.cx, tmpcy = scaleFac * K.cy;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x) {
        const float X = xyz[3 * i], Y = xyz[3 * i + 1], Z = xyz[3 * i + 2];
        auto recover = [&](float V, float tc, float tf) -> int {
            const double guess = (double)V / ((double)Z * (double)tf) + (double)tc;
            if (!(guess > -2.0 && guess < 65537.0)) return -1;
            const int g0 = (int)__double2int_rn(guess);
            for (int d = 0; d < 3; d++) {
                const int q = g0 + (d == 0 ? 0 : (d == 1 ? -1 : 1));
                if (q < 0 || q > 65535) continue;
                const float v = Z * ((float)q - tc) * tf;                 /* :249 / :250, the order of enlist_write_kernel */
                if (__float_as_uint(v) == __float_as_uint(V)) return q;
            }
            return -1;
        };
        const int xx = recover(X, tmpcx, tmpfx), yy = recover(Y, tmpcy, tmpfy);
        if (xx < 0 || yy < 0) { *fail = 1; continue; }
        compact[i] = make_uint2((unsigned)xx | ((unsigned)yy << 16), __float_as_uint(Z));
        cidx[i] = (unsigned)i;
    }
}
hipError_t launch_points_recover_compact(const float *xyz, int N, int level, const Intrinsics &K, uint2 *compact, unsigned *cidx, int *fail, hipStream_t s) {
    if (N < 1) return hipSuccess;
    hipLaunchKernelGGL(points_recover_compact_kernel, dim3(grid_x((size_t)N, 1024)), dim3(256), 0, s, xyz, N, level, K, compact, cidx, fail);
    return hipGetLastError();
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
                                 size_t xyz_stride, uint2 *compact, unsigned *cidx, float *uv, int capacity, int *N_dst, hipStream_t s) {
    hipLaunchKernelGGL(enlist_write_kernel<E>, dim3(g.cols, g.count), dim3(64), 0, s, edge, edge_stride, depth, depth_stride,
                       g.rows, g.cols, level, K, col_counts, blk_counts, (g.rows + 15) >> 4, xyz, xyz_stride, compact, cidx, uv, capacity, N_dst);
    return hipGetLastError();
}

hipError_t launch_enlist_count(const void *edge, int edge_is_u8, size_t edge_stride, const float *depth, size_t depth_stride,
                               ImgBatch g, int *col_counts, int *blk_counts, hipStream_t s) {
    return edge_is_u8 ? enlist_count_t((const unsigned char *)edge, edge_stride, depth, depth_stride, g, col_counts, blk_counts, s)
                      : enlist_count_t((const int32_t *)edge, edge_stride, depth, depth_stride, g, col_counts, blk_counts, s);
}
hipError_t launch_enlist_write(const void *edge, int edge_is_u8, size_t edge_stride, const float *depth, size_t depth_stride,
                               ImgBatch g, int level, const Intrinsics &K, const int *col_counts, const int *blk_counts, float *xyz,
                               size_t xyz_stride, uint2 *compact, unsigned *cidx, float *uv, int capacity, int *N_dst, hipStream_t s) {
    return edge_is_u8 ? enlist_write_t((const unsigned char *)edge, edge_stride, depth, depth_stride, g, level, K, col_counts, blk_counts,
                                       xyz, xyz_stride, compact, cidx, uv, capacity, N_dst, s)
                      : enlist_write_t((const int32_t *)edge, edge_stride, depth, depth_stride, g, level, K, col_counts, blk_counts,
                                       xyz, xyz_stride, compact, cidx, uv, capacity, N_dst, s);
}

}  // namespace dvo

#ifdef DVO_EDT_STAMPS
extern "C" int dvo_debug_edt_stamps(unsigned long long *out, int reset) {
    static std::vector<unsigned long long> h((size_t)dvo::EDT_STAMP_SLOTS * 8);
    if (out) {
        if (hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(dvo::g_edt_stamp), sizeof(unsigned long long) * h.size()) != hipSuccess) return -1;
        for (int k = 0; k < 8; k++) out[k] = 0;
        for (size_t i = 0; i < h.size(); i++) out[i & 7] += h[i];
    }
    if (reset) {
        std::fill(h.begin(), h.end(), 0ull);
        if (hipMemcpyToSymbol(HIP_SYMBOL(dvo::g_edt_stamp), h.data(), sizeof(unsigned long long) * h.size()) != hipSuccess) return -1;
    }
    return 0;
}
#endif
