/*
 * dvo_synth.cpp -- seeded synthetic RGB-D edge scene generator (host, no deps).
 *
 * Produces exactly the inputs that sit at the hot-path boundary of
 * SolveDVO::runIterations (reference src/SolveDVO.cpp:619-1017):
 *   ref side : per-level edge mask (int32) + depth in mm (f32), i.e. what
 *              selectedPts/enlistRefEdgePts consume (:1230-1264, :224-264);
 *   now side : per-level distance transform, min-max normalised to [0,255]
 *              (:1771-1774), and its central-difference gradients with a
 *              reflect-101 border (:1077-1090).
 * The reference gets these from OpenCV (Canny, distanceTransform, normalize,
 * filter2D); here they come from a closed-form scene so that no image library
 * is needed and the same bits are produced in the build container and on the
 * GPU box.  Recipe: SURVEY.md section 8(d).  The contract is the seed plus the
 * array hashes pinned in tests/golden, not any library's bit stream.
 *
 * All images are COLUMN-major (Eigen::MatrixXf layout): (yy,xx) at yy+xx*rows.
 */
#include "dvo_synth.h"

#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <vector>

namespace {

struct SplitMix64 {
    uint64_t s;
    explicit SplitMix64(uint64_t seed) : s(seed) {}
    uint64_t next() {
        uint64_t z = (s += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    double uniform() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }   /* [0,1) */
    double uniform(double a, double b) { return a + (b - a) * uniform(); }
};

/* cvRound: round half to even (OpenCV uses lrint / cvtsd2si). */
int cv_round(double v) { return (int)std::nearbyint(v); }

struct Level {
    int rows = 0, cols = 0;
    std::vector<int32_t> ref_edge;      /* 0/255 like cv::Canny output after cv2eigen */
    std::vector<float> ref_depth;       /* mm */
    std::vector<int32_t> now_edge;
    std::vector<float> now_dt, now_gx, now_gy;
};

/* Exact squared Euclidean distance transform (Meijster, Roerdink, Hesselink 2000),
 * integer arithmetic only.  mask!=0 marks the zero set (edge pixels). */
void edt_squared(const std::vector<int32_t> &mask, int rows, int cols, std::vector<int64_t> &d2) {
    const int64_t INF = (int64_t)rows + cols + 1;
    std::vector<int64_t> g((size_t)rows * cols);
    /* phase 1: along each column (yy direction, contiguous in column-major) */
    for (int x = 0; x < cols; x++) {
        const int32_t *m = &mask[(size_t)x * rows];
        int64_t *gc = &g[(size_t)x * rows];
        gc[0] = m[0] ? 0 : INF;
        for (int y = 1; y < rows; y++) gc[y] = m[y] ? 0 : (gc[y - 1] >= INF ? INF : gc[y - 1] + 1);
        for (int y = rows - 2; y >= 0; y--)
            if (gc[y + 1] < gc[y]) gc[y] = gc[y + 1] + 1;
    }
    /* phase 2: along each row, lower envelope of parabolas f(x,i) = (x-i)^2 + g(i)^2 */
    d2.assign((size_t)rows * cols, 0);
    std::vector<int> s(cols), t(cols);
    auto f = [&](int64_t x, int64_t i, int64_t gi) { return (x - i) * (x - i) + gi * gi; };
    auto sep = [&](int64_t i, int64_t u, int64_t gi, int64_t gu) {
        /* first integer x where parabola u is <= parabola i  (i<u) */
        const int64_t num = u * u - i * i + gu * gu - gi * gi, den = 2 * (u - i);   /* den > 0 */
        int64_t qd = num / den;
        if ((num % den != 0) && (num < 0)) qd--;                     /* floor division */
        return qd;
    };
    for (int y = 0; y < rows; y++) {
        int q = 0;
        s[0] = 0; t[0] = 0;
        auto G = [&](int x) { return g[(size_t)x * rows + y]; };
        for (int u = 1; u < cols; u++) {
            while (q >= 0 && f(t[q], s[q], G(s[q])) > f(t[q], u, G(u))) q--;
            if (q < 0) { q = 0; s[0] = u; }
            else {
                int64_t w = 1 + sep(s[q], u, G(s[q]), G(u));
                if (w < cols) { q++; s[q] = u; t[q] = (int)w; }
            }
        }
        for (int u = cols - 1; u >= 0; u--) {
            d2[(size_t)u * rows + y] = f(u, s[q], G(s[q]));
            if (u == t[q]) q--;
        }
    }
}

inline int reflect101(int i, int n) {
    if (n == 1) return 0;
    if (i < 0) return -i;
    if (i >= n) return 2 * n - 2 - i;
    return i;
}

}  // namespace

struct dvo_synth_scene {
    int W, H, n_levels;
    uint64_t seed;
    float fx, fy, cx, cy;
    double R_true[9], t_true[3];
    std::vector<Level> lv;
};

extern "C" {

int dvo_synth_level_rows(int H, int level) { return cv_round((double)H * std::ldexp(1.0, -level)); }
int dvo_synth_level_cols(int W, int level) { return cv_round((double)W * std::ldexp(1.0, -level)); }

dvo_synth_scene *dvo_synth_create(int W, int H, int n_levels, uint64_t seed) { return dvo_synth_create_ex(W, H, n_levels, seed, 0, 1.0); }

/* n_seg <= 0 and x_frac >= 1: the scene of SURVEY.md section 8d (60 W/320 segments anywhere: 5-6 % edge pixels).  Otherwise a SPARSE
 * scene (round 5): n_seg segments drawn only inside the columns [0, x_frac W) -- the rest of the frame has no edge at all, which is
 * what camera images with sky / walls look like and what the bench's standard scenes never have (pixels hundreds of pixels from
 * every edge: many distinct distances, far-from-edge tiles). */
dvo_synth_scene *dvo_synth_create_ex(int W, int H, int n_levels, uint64_t seed, int n_seg_in, double x_frac) {
    if (W < 16 || H < 16 || n_levels < 1 || n_levels > 12) return nullptr;
    if (!(x_frac > 0.0) || x_frac > 1.0) x_frac = 1.0;
    dvo_synth_scene *sc = new dvo_synth_scene();
    sc->W = W; sc->H = H; sc->n_levels = n_levels; sc->seed = seed;
    /* TUM / ROS-default intrinsics scaled to the level-0 size */
    sc->fx = (float)(525.0 * W / 640.0);
    sc->fy = (float)(525.0 * W / 640.0);
    sc->cx = (float)(319.5 * W / 640.0);
    sc->cy = (float)(239.5 * H / 480.0);
    SplitMix64 rng(seed);

    /* ---- level-0 reference edge mask: random segments, 0.5 px steps ---- */
    std::vector<int32_t> edge0((size_t)W * H, 0);
    const double sW = (double)W / 320.0;
    const int n_seg = n_seg_in > 0 ? n_seg_in : cv_round(60.0 * sW);
    const int x_end = (x_frac >= 1.0) ? W : std::max(8, (int)(x_frac * W));
    for (int k = 0; k < n_seg; k++) {
        const double x0 = rng.uniform(0.0, (double)x_end);
        const double y0 = rng.uniform(0.0, (double)H);
        const double ang = rng.uniform(0.0, M_PI);
        const double len = rng.uniform(20.0, 120.0) * sW;
        const double dx = std::cos(ang), dy = std::sin(ang);
        for (double s = 0.0; s <= len; s += 0.5) {
            const int x = (int)std::floor(x0 + s * dx);
            const int y = (int)std::floor(y0 + s * dy);
            if (x >= 0 && x < x_end && y >= 0 && y < H) edge0[(size_t)x * H + y] = 255;
        }
    }
    /* ---- level-0 reference depth (mm, quantised like a u16 sensor image) ---- */
    std::vector<float> depth0((size_t)W * H);
    for (int x = 0; x < W; x++)
        for (int y = 0; y < H; y++) {
            const double d = 2000.0 + 300.0 * std::sin((double)x / 40.0 * 320.0 / W) +
                             200.0 * std::cos((double)y / 30.0 * 240.0 / H);
            depth0[(size_t)x * H + y] = (float)(uint16_t)cv_round(d);
        }

    /* ---- true motion (pose of NOW in REF: P_now = R^T (P_ref - t)) ---- */
    double w[3], tt[3];
    for (int k = 0; k < 3; k++) w[k] = rng.uniform(-0.02, 0.02);
    for (int k = 0; k < 3; k++) tt[k] = rng.uniform(-0.02, 0.02);
    {   /* Rodrigues */
        const double th = std::sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
        double K[9] = {0, w[2], -w[1], -w[2], 0, w[0], w[1], -w[0], 0};   /* col-major hat(w) */
        double K2[9];
        for (int j = 0; j < 3; j++) for (int i = 0; i < 3; i++) {
            double s = 0; for (int m = 0; m < 3; m++) s += K[i + 3 * m] * K[m + 3 * j];
            K2[i + 3 * j] = s;
        }
        const double a = th > 1e-12 ? std::sin(th) / th : 1.0;
        const double b = th > 1e-12 ? (1.0 - std::cos(th)) / (th * th) : 0.5;
        for (int k = 0; k < 9; k++) sc->R_true[k] = ((k % 4 == 0) ? 1.0 : 0.0) + a * K[k] + b * K2[k];
        for (int k = 0; k < 3; k++) sc->t_true[k] = tt[k];
    }

    /* ---- level-0 now edge pixels: project the level-0 ref edge points ---- */
    std::vector<int32_t> nowedge0((size_t)W * H, 0);
    {
        const double fx = sc->fx, fy = sc->fy, cx = sc->cx, cy = sc->cy;
        const double *R = sc->R_true;
        for (int x = 0; x < W; x++)
            for (int y = 0; y < H; y++) {
                if (!edge0[(size_t)x * H + y]) continue;
                const double Z = (double)depth0[(size_t)x * H + y] / 1000.0;
                const double X = Z * ((double)x - cx) / fx;
                const double Y = Z * ((double)y - cy) / fy;
                const double d0 = X - sc->t_true[0], d1 = Y - sc->t_true[1], d2 = Z - sc->t_true[2];
                const double px = R[0] * d0 + R[1] * d1 + R[2] * d2;     /* R^T d */
                const double py = R[3] * d0 + R[4] * d1 + R[5] * d2;
                const double pz = R[6] * d0 + R[7] * d1 + R[8] * d2;
                if (pz <= 0.1) continue;
                const int u = (int)std::floor(fx * px / pz + cx);
                const int v = (int)std::floor(fy * py / pz + cy);
                if (u >= 0 && u < W && v >= 0 && v < H) nowedge0[(size_t)u * H + v] = 255;
            }
    }

    /* ---- per-level data ---- */
    sc->lv.resize(n_levels);
    for (int l = 0; l < n_levels; l++) {
        Level &L = sc->lv[l];
        L.rows = dvo_synth_level_rows(H, l);
        L.cols = dvo_synth_level_cols(W, l);
        const int rows = L.rows, cols = L.cols;
        const size_t n = (size_t)rows * cols;
        L.ref_edge.assign(n, 0);
        L.now_edge.assign(n, 0);
        L.ref_depth.assign(n, 0.f);
        /* masks: unique floor(coord * 2^-l) */
        for (int x = 0; x < W; x++)
            for (int y = 0; y < H; y++) {
                const int xl = x >> l, yl = y >> l;
                if (xl >= cols || yl >= rows) continue;
                if (edge0[(size_t)x * H + y]) L.ref_edge[(size_t)xl * rows + yl] = 255;
                if (nowedge0[(size_t)x * H + y]) L.now_edge[(size_t)xl * rows + yl] = 255;
            }
        /* depth: INTER_NEAREST decimation of the level-0 image (camTopic2PublisherPyD.cpp:338-345) */
        for (int xl = 0; xl < cols; xl++)
            for (int yl = 0; yl < rows; yl++) {
                int x0 = xl << l, y0 = yl << l;
                if (x0 > W - 1) x0 = W - 1;
                if (y0 > H - 1) y0 = H - 1;
                L.ref_depth[(size_t)xl * rows + yl] = depth0[(size_t)x0 * H + y0];
            }
        /* now DT: exact EDT, min-max normalised to [0,255] (SolveDVO.cpp:1771-1774) */
        std::vector<int64_t> d2;
        edt_squared(L.now_edge, rows, cols, d2);
        L.now_dt.resize(n);
        float mn = std::numeric_limits<float>::infinity(), mx = 0.f;
        for (size_t i = 0; i < n; i++) {
            L.now_dt[i] = (float)std::sqrt((double)d2[i]);
            if (L.now_dt[i] < mn) mn = L.now_dt[i];
            if (L.now_dt[i] > mx) mx = L.now_dt[i];
        }
        /* cv::normalize(0,255,NORM_MINMAX) as OpenCV 2.4 evaluates it: scale, shift in double, the conversion in float */
        const double smin = (double)mn, smax = (double)mx;
        const double scale = 255.0 * ((smax - smin > 2.2204460492503131e-16) ? 1. / (smax - smin) : 0.);
        const float scale_f = (float)scale, shift_f = (float)(0.0 - smin * scale);
        for (size_t i = 0; i < n; i++) L.now_dt[i] = L.now_dt[i] * scale_f + shift_f;
        /* gradients: kernels [-.5 0 .5], BORDER_REFLECT_101 (SolveDVO.cpp:1077-1090) */
        L.now_gx.resize(n);
        L.now_gy.resize(n);
        for (int x = 0; x < cols; x++)
            for (int y = 0; y < rows; y++) {
                const float l_ = L.now_dt[(size_t)reflect101(x - 1, cols) * rows + y];
                const float r_ = L.now_dt[(size_t)reflect101(x + 1, cols) * rows + y];
                const float u_ = L.now_dt[(size_t)x * rows + reflect101(y - 1, rows)];
                const float d_ = L.now_dt[(size_t)x * rows + reflect101(y + 1, rows)];
                L.now_gx[(size_t)x * rows + y] = 0.5f * r_ - 0.5f * l_;
                L.now_gy[(size_t)x * rows + y] = 0.5f * d_ - 0.5f * u_;
            }
    }
    return sc;
}

void dvo_synth_destroy(dvo_synth_scene *sc) { delete sc; }

int dvo_synth_rows(const dvo_synth_scene *sc, int level) { return sc->lv[level].rows; }
int dvo_synth_cols(const dvo_synth_scene *sc, int level) { return sc->lv[level].cols; }
const int32_t *dvo_synth_ref_edge(const dvo_synth_scene *sc, int level) { return sc->lv[level].ref_edge.data(); }
const float *dvo_synth_ref_depth(const dvo_synth_scene *sc, int level) { return sc->lv[level].ref_depth.data(); }
const int32_t *dvo_synth_now_edge(const dvo_synth_scene *sc, int level) { return sc->lv[level].now_edge.data(); }
const float *dvo_synth_now_dt(const dvo_synth_scene *sc, int level) { return sc->lv[level].now_dt.data(); }
const float *dvo_synth_now_gx(const dvo_synth_scene *sc, int level) { return sc->lv[level].now_gx.data(); }
const float *dvo_synth_now_gy(const dvo_synth_scene *sc, int level) { return sc->lv[level].now_gy.data(); }
void dvo_synth_intrinsics(const dvo_synth_scene *sc, float *k4) {
    k4[0] = sc->fx; k4[1] = sc->fy; k4[2] = sc->cx; k4[3] = sc->cy;
}
void dvo_synth_true_pose(const dvo_synth_scene *sc, double *R9, double *t3) {
    std::memcpy(R9, sc->R_true, sizeof(double) * 9);
    std::memcpy(t3, sc->t_true, sizeof(double) * 3);
}

}  // extern "C"
