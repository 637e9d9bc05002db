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

}  // extern "C"
