/*
 * dvo_oracle_frames.cpp -- CPU ORACLE for the per-frame preprocessing either side of the hot path
 * (SURVEY.md section 8 rows f1 and f2).  TEST INFRASTRUCTURE, NOT PRODUCT CODE (see dvo_oracle.h).
 *
 * PARITY UNPINNED.  Every function here restates arithmetic of OpenCV 2.4.x (the reference links
 * OpenCV 2.4, `opencv2/nonfree` in include/PnPOdometry.h:11, "2.4.9" in src/PnPOdometry.cpp:165), which is
 * neither in /root/reference nor in this image.  What is restated is the published definition of each
 * operation as the reference calls it:
 *
 *   cv::Canny(src8u, dst, 150, 100, 3, true)      src/SolveDVO.cpp:1704 (ref), :1764 (now)
 *       3x3 Sobel derivatives (16-bit, BORDER_REPLICATE); squared L2 magnitude in 32-bit integers compared
 *       with the squared thresholds (the two thresholds are swapped when given high-first);
 *       non-maximum suppression in four sectors decided with the fixed-point constant
 *       round(tan(22.5 deg) * 2^15) and the comparisons (> previous, >= next) along x and y, (>,>) along
 *       the diagonals; hysteresis = every candidate 8-connected (through candidates) to a candidate
 *       above the high threshold.  The result does not depend on the order in which pixels are visited.
 *   cv::cvtColor(BGR2GRAY) on 8-bit              src/camTopic2PublisherPyD.cpp:347
 *       (1868*B + 9617*G + 4899*R + 2^13) >> 14
 *   cv::resize(INTER_NEAREST, scale 1/2^k)        src/camTopic2PublisherPyD.cpp:344-345
 *       dst(y,x) = src(min(floor(y/s), H-1), min(floor(x/s), W-1)), dsize = round-half-even(size*s)
 *   depth metres(32F) -> *1000 -> 16U -> 0 becomes 1   src/camTopic2PublisherPyD.cpp:73-77 and
 *       src/SolveDVO.cpp:514 (the node repeats the 0 -> 1 step)
 *
 * All images in this file are ROW-major (OpenCV's layout); the node converts to column-major Eigen
 * matrices afterwards (cv::cv2eigen, src/SolveDVO.cpp:518-519) -- tests do that with a numpy transpose.
 */
#include "dvo_oracle.h"

#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace {

inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* cvRound on x86 (cvtsd2si, round-half-even); NaN and out-of-range give INT_MIN ("integer indefinite") */
inline int cv_round(double v) {
    if (!(v > -2147483648.5 && v < 2147483647.5)) return INT32_MIN;
    return (int)std::nearbyint(v);
}

}  // namespace

extern "C" {

void dvo_oracle_sobel3(const unsigned char *src, int rows, int cols, short *dx, short *dy) {
    for (int y = 0; y < rows; y++) {
        const int ym = clampi(y - 1, 0, rows - 1), yp = clampi(y + 1, 0, rows - 1);     /* BORDER_REPLICATE */
        for (int x = 0; x < cols; x++) {
            const int xm = clampi(x - 1, 0, cols - 1), xp = clampi(x + 1, 0, cols - 1);
            const int a = src[(size_t)ym * cols + xm], b = src[(size_t)ym * cols + x], c = src[(size_t)ym * cols + xp];
            const int d = src[(size_t)y * cols + xm], f = src[(size_t)y * cols + xp];
            const int g = src[(size_t)yp * cols + xm], h = src[(size_t)yp * cols + x], i = src[(size_t)yp * cols + xp];
            dx[(size_t)y * cols + x] = (short)((c - a) + 2 * (f - d) + (i - g));
            dy[(size_t)y * cols + x] = (short)((g - a) + 2 * (h - b) + (i - c));
        }
    }
}

/* stage outputs (any may be NULL): mag int32 HxW, cand u8 HxW (0 suppressed, 1 candidate, 2 candidate above high) */
void dvo_oracle_canny_stages(const unsigned char *src, int rows, int cols, double threshold1, double threshold2,
                             int *mag_out, unsigned char *cand_out, unsigned char *dst) {
    const size_t n = (size_t)rows * cols;
    std::vector<short> dx(n), dy(n);
    dvo_oracle_sobel3(src, rows, cols, dx.data(), dy.data());
    double lo = threshold1, hi = threshold2;
    if (lo > hi) { const double t = lo; lo = hi; hi = t; }
    lo = lo < 32767.0 ? lo : 32767.0;                     /* L2gradient: thresholds are squared */
    hi = hi < 32767.0 ? hi : 32767.0;
    if (lo > 0) lo *= lo;
    if (hi > 0) hi *= hi;
    const int low = (int)std::floor(lo), high = (int)std::floor(hi);

    std::vector<int> mag(n);
    for (size_t k = 0; k < n; k++) mag[k] = (int)dx[k] * dx[k] + (int)dy[k] * dy[k];
    auto M = [&](int y, int x) -> int { return (y < 0 || y >= rows || x < 0 || x >= cols) ? 0 : mag[(size_t)y * cols + x]; };

    const int SHIFT = 15;
    const int TG22 = (int)(0.4142135623730950488016887242097 * (1 << SHIFT) + 0.5);
    std::vector<unsigned char> cand(n, 0);
    for (int y = 0; y < rows; y++)
        for (int x = 0; x < cols; x++) {
            const int m = mag[(size_t)y * cols + x];
            bool keep = false;
            if (m > low) {
                const int xs = dx[(size_t)y * cols + x], ys = dy[(size_t)y * cols + x];
                const int ax = std::abs(xs), ay = std::abs(ys) << SHIFT;
                const int tg22x = ax * TG22;
                if (ay < tg22x) keep = m > M(y, x - 1) && m >= M(y, x + 1);
                else {
                    const int tg67x = tg22x + (ax << (SHIFT + 1));
                    if (ay > tg67x) keep = m > M(y - 1, x) && m >= M(y + 1, x);
                    else {
                        const int s = ((xs ^ ys) < 0) ? -1 : 1;
                        keep = m > M(y - 1, x - s) && m > M(y + 1, x + s);
                    }
                }
            }
            cand[(size_t)y * cols + x] = keep ? (m > high ? 2 : 1) : 0;
        }
    if (mag_out) std::memcpy(mag_out, mag.data(), sizeof(int) * n);
    if (cand_out) std::memcpy(cand_out, cand.data(), n);
    if (!dst) return;

    /* hysteresis: breadth-first from every strong candidate through 8-connected candidates */
    std::vector<unsigned char> edge(n, 0);
    std::vector<size_t> queue;
    queue.reserve(n / 8 + 16);
    for (size_t k = 0; k < n; k++) if (cand[k] == 2) { edge[k] = 1; queue.push_back(k); }
    for (size_t head = 0; head < queue.size(); head++) {
        const int y = (int)(queue[head] / cols), x = (int)(queue[head] % cols);
        for (int dy_ = -1; dy_ <= 1; dy_++)
            for (int dx_ = -1; dx_ <= 1; dx_++) {
                const int yy = y + dy_, xx = x + dx_;
                if ((dy_ | dx_) == 0 || yy < 0 || yy >= rows || xx < 0 || xx >= cols) continue;
                const size_t k = (size_t)yy * cols + xx;
                if (cand[k] && !edge[k]) { edge[k] = 1; queue.push_back(k); }
            }
    }
    for (size_t k = 0; k < n; k++) dst[k] = edge[k] ? 255 : 0;
}

void dvo_oracle_canny(const unsigned char *src, int rows, int cols, double threshold1, double threshold2,
                      unsigned char *dst) {
    dvo_oracle_canny_stages(src, rows, cols, threshold1, threshold2, nullptr, nullptr, dst);
}

void dvo_oracle_bgr2gray(const unsigned char *bgr, size_t npx, unsigned char *grey) {
    for (size_t k = 0; k < npx; k++)
        grey[k] = (unsigned char)((1868 * bgr[3 * k] + 9617 * bgr[3 * k + 1] + 4899 * bgr[3 * k + 2] + (1 << 13)) >> 14);
}

void dvo_oracle_resize_nn_size(int rows, int cols, double scale, int *drows, int *dcols) {
    *dcols = cv_round(cols * scale);
    *drows = cv_round(rows * scale);
}

void dvo_oracle_resize_nn(const void *src, int rows, int cols, int elem_bytes, double scale, void *dst) {
    int drows, dcols;
    dvo_oracle_resize_nn_size(rows, cols, scale, &drows, &dcols);
    const double inv = 1. / scale;
    const unsigned char *s = (const unsigned char *)src;
    unsigned char *d = (unsigned char *)dst;
    for (int y = 0; y < drows; y++) {
        int sy = (int)std::floor(y * inv); if (sy > rows - 1) sy = rows - 1;
        for (int x = 0; x < dcols; x++) {
            int sx = (int)std::floor(x * inv); if (sx > cols - 1) sx = cols - 1;
            std::memcpy(d + ((size_t)y * dcols + x) * elem_bytes, s + ((size_t)sy * cols + sx) * elem_bytes, elem_bytes);
        }
    }
}

void dvo_oracle_depth_m_to_mm16(const float *depth_m, size_t npx, unsigned short *out) {
    for (size_t k = 0; k < npx; k++) {
        const float mm = depth_m[k] * 1000.0f;            /* Mat * 1000.0 on CV_32F: float multiply */
        const int iv = cv_round((double)mm);              /* convertTo(CV_16U): saturate_cast<ushort>(cvRound) */
        unsigned short v = (unsigned short)(iv < 0 ? 0 : (iv > 65535 ? 65535 : iv));
        if (v == 0) v = 1;                                /* setTo(1, depth16==0) */
        out[k] = v;
    }
}

/* ---- cv::undistort(src, dst, cameraMatrix, distCoeffs) of the pyramid publisher (camTopic2PublisherPyD.cpp:88-107,
 * :306-308), OpenCV 2.4 (imgproc/src/undistort.cpp, imgwarp.cpp), restated -- PARITY UNPINNED like the other OpenCV steps:
 *   undistort()               processes the image in horizontal stripes of stripe0 = min(max(1, 4096/cols), rows) rows; for the
 *                             stripe starting at row y0 it builds a fixed-point map with newCameraMatrix = cameraMatrix but
 *                             cy' = cy - y0, R = I, and remaps with INTER_LINEAR, BORDER_CONSTANT (value 0)
 *   initUndistortRectifyMap() iR = inv(Ar*R) -- for a 3x3 cv::invert(DECOMP_LU) is the adjugate times 1/det --; per map row i:
 *                             _x = i*ir[1]+ir[2], _y = i*ir[4]+ir[5], _w = i*ir[7]+ir[8], advanced by (ir[0], ir[3], ir[6]) per
 *                             column (running sums, in double); x = _x/_w ...; radial/tangential model with k1 k2 p1 p2 k3
 *                             (k4..k6 = 0 for the 5-coefficient camera_info D); u, v in double;
 *                             CV_16SC2 map: iu = cvRound(u*32), m1 = (short)(iu >> 5), fraction index (iv & 31)*32 + (iu & 31)
 *   remap() INTER_LINEAR      8-bit: weights from BilinearTab_i = saturate_cast<short>(w*32768) -- exact for 1/32 fractions,
 *                             except weight 1.0 -> 32767 with the missing 1 added to tap (1,1) --, pixel = (sum + 2^14) >> 15;
 *                             16-bit: float weights, v0*w0 + v1*w1 + v2*w2 + v3*w3 in float left to right, cvRound, saturate;
 *                             a tap outside the image contributes the border value 0 */
namespace {
struct UndistortMap { std::vector<short> sx, sy; std::vector<unsigned short> frac; };

void invert3_adjugate(const double *S, double *t) {      /* cv::invert, n == 3, CV_64F; row-major */
    auto m = [&](int i, int j) { return S[i * 3 + j]; };
    double d = m(0, 0) * (m(1, 1) * m(2, 2) - m(1, 2) * m(2, 1)) - m(0, 1) * (m(1, 0) * m(2, 2) - m(1, 2) * m(2, 0)) +
               m(0, 2) * (m(1, 0) * m(2, 1) - m(1, 1) * m(2, 0));
    d = 1. / d;
    t[0] = (m(1, 1) * m(2, 2) - m(1, 2) * m(2, 1)) * d;
    t[1] = (m(0, 2) * m(2, 1) - m(0, 1) * m(2, 2)) * d;
    t[2] = (m(0, 1) * m(1, 2) - m(0, 2) * m(1, 1)) * d;
    t[3] = (m(1, 2) * m(2, 0) - m(1, 0) * m(2, 2)) * d;
    t[4] = (m(0, 0) * m(2, 2) - m(0, 2) * m(2, 0)) * d;
    t[5] = (m(0, 2) * m(1, 0) - m(0, 0) * m(1, 2)) * d;
    t[6] = (m(1, 0) * m(2, 1) - m(1, 1) * m(2, 0)) * d;
    t[7] = (m(0, 1) * m(2, 0) - m(0, 0) * m(2, 1)) * d;
    t[8] = (m(0, 0) * m(1, 1) - m(0, 1) * m(1, 0)) * d;
}

void build_undistort_map(int rows, int cols, const double *K4, const double *D5, UndistortMap &M) {
    const double fx = K4[0], fy = K4[1], u0 = K4[2], v0 = K4[3];
    const double k1 = D5[0], k2 = D5[1], p1 = D5[2], p2 = D5[3], k3 = D5[4], k4 = 0, k5 = 0, k6 = 0;
    M.sx.assign((size_t)rows * cols, 0); M.sy.assign((size_t)rows * cols, 0); M.frac.assign((size_t)rows * cols, 0);
    int stripe0 = (1 << 12) / (cols > 1 ? cols : 1);
    if (stripe0 < 1) stripe0 = 1;
    if (stripe0 > rows) stripe0 = rows;
    for (int y0 = 0; y0 < rows; y0 += stripe0) {
        const int stripe = (stripe0 < rows - y0) ? stripe0 : rows - y0;
        const double Ar[9] = {fx, 0, u0, 0, fy, v0 - y0, 0, 0, 1};           /* Ar(1,2) = v0 - y; Ar*I is Ar exactly */
        double ir[9];
        invert3_adjugate(Ar, ir);
        for (int i = 0; i < stripe; i++) {
            double _x = i * ir[1] + ir[2], _y = i * ir[4] + ir[5], _w = i * ir[7] + ir[8];
            for (int j = 0; j < cols; j++, _x += ir[0], _y += ir[3], _w += ir[6]) {
                const double w = 1. / _w, x = _x * w, y = _y * w;
                const double x2 = x * x, y2 = y * y;
                const double r2 = x2 + y2, _2xy = 2 * x * y;
                const double kr = (1 + ((k3 * r2 + k2) * r2 + k1) * r2) / (1 + ((k6 * r2 + k5) * r2 + k4) * r2);
                const double u = fx * (x * kr + p1 * _2xy + p2 * (r2 + 2 * x2)) + u0;
                const double v = fy * (y * kr + p1 * (r2 + 2 * y2) + p2 * _2xy) + v0;
                const int iu = cv_round(u * 32), iv = cv_round(v * 32);        /* saturate_cast<int>(double) */
                const size_t o = (size_t)(y0 + i) * cols + j;
                M.sx[o] = (short)(iu >> 5);
                M.sy[o] = (short)(iv >> 5);
                M.frac[o] = (unsigned short)((iv & 31) * 32 + (iu & 31));
            }
        }
    }
}

void bilinear_itab(int fi, short *w) {                  /* BilinearTab_i[fi], fi = fy*32 + fx */
    const int fy = fi >> 5, fx = fi & 31;
    const float ty[2] = {1.f - fy * (1.f / 32), fy * (1.f / 32)}, tx[2] = {1.f - fx * (1.f / 32), fx * (1.f / 32)};
    int isum = 0;
    for (int a = 0; a < 2; a++)
        for (int b = 0; b < 2; b++) {
            const float v = ty[a] * tx[b];
            int iv = cv_round((double)(v * 32768.f));
            if (iv > 32767) iv = 32767;                  /* saturate_cast<short> */
            w[a * 2 + b] = (short)iv;
            isum += iv;
        }
    if (isum != 32768) w[3] = (short)(w[3] - (isum - 32768));   /* only fi == 0: {32767,0,0,0} -> the missing 1 lands on tap (1,1) */
}
}  // namespace

void dvo_oracle_undistort_bgr8(const unsigned char *src, int rows, int cols, const double *K4, const double *D5, unsigned char *dst) {
    UndistortMap M;
    build_undistort_map(rows, cols, K4, D5, M);
    for (int y = 0; y < rows; y++)
        for (int x = 0; x < cols; x++) {
            const size_t o = (size_t)y * cols + x;
            const int sx = M.sx[o], sy = M.sy[o];
            short w[4];
            bilinear_itab(M.frac[o], w);
            for (int ch = 0; ch < 3; ch++) {
                int sum = 0;
                for (int a = 0; a < 2; a++)
                    for (int b = 0; b < 2; b++) {
                        const int yy = sy + a, xx = sx + b;
                        const int v = (yy >= 0 && yy < rows && xx >= 0 && xx < cols) ? src[((size_t)yy * cols + xx) * 3 + ch] : 0;
                        sum += v * w[a * 2 + b];
                    }
                int r = (sum + (1 << 14)) >> 15;          /* FixedPtCast<int, uchar, 15> */
                dst[o * 3 + ch] = (unsigned char)(r < 0 ? 0 : (r > 255 ? 255 : r));
            }
        }
}

void dvo_oracle_undistort_u16(const unsigned short *src, int rows, int cols, const double *K4, const double *D5, unsigned short *dst) {
    UndistortMap M;
    build_undistort_map(rows, cols, K4, D5, M);
    for (int y = 0; y < rows; y++)
        for (int x = 0; x < cols; x++) {
            const size_t o = (size_t)y * cols + x;
            const int sx = M.sx[o], sy = M.sy[o];
            const int fy = M.frac[o] >> 5, fx = M.frac[o] & 31;
            const float ty[2] = {1.f - fy * (1.f / 32), fy * (1.f / 32)}, tx[2] = {1.f - fx * (1.f / 32), fx * (1.f / 32)};
            float acc = 0.f;
            for (int a = 0; a < 2; a++)
                for (int b = 0; b < 2; b++) {
                    const int yy = sy + a, xx = sx + b;
                    const float v = (yy >= 0 && yy < rows && xx >= 0 && xx < cols) ? (float)src[(size_t)yy * cols + xx] : 0.f;
                    const float wt = ty[a] * tx[b];       /* BilinearTab_f */
                    acc = (a == 0 && b == 0) ? v * wt : acc + v * wt;
                }
            const int r = cv_round((double)acc);
            dst[o] = (unsigned short)(r < 0 ? 0 : (r > 65535 ? 65535 : r));
        }
}

}  // extern "C"
