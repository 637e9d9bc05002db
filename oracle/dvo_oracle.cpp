/*
 * dvo_oracle.cpp -- CPU ORACLE (test infrastructure only; see dvo_oracle.h).
 *
 * PARITY UNPINNED: no golden vectors exist in the reference and it cannot be
 * built here; see the header.  Every function cites the reference lines it
 * restates (paths relative to /root/reference).
 *
 * Build: strict IEEE -- g++ -O2 -ffp-contract=off -fno-fast-math (oracle/Makefile).
 * The reference itself is built with -ffast-math -mavx (CMakeLists.txt:106), so
 * its own binary is not bit-reproducible; this file DEFINES the evaluation
 * order (SURVEY.md section 8a "Precision and evaluation-order map").
 *
 * Written matrix-style on purpose, mirroring the Eigen expressions of the
 * reference, so that it is an independent statement from the device kernels
 * (which use an algebraically simplified scalar form).
 */
#include "dvo_oracle.h"

#include <cmath>
#include <cstring>
#include <vector>

namespace {

/* ---- tiny column-major 3x3 helpers: M(i,j) = m[i+3*j] ------------------- */
template <typename T> inline T &M3(T *m, int i, int j) { return m[i + 3 * j]; }
template <typename T> inline const T &M3(const T *m, int i, int j) { return m[i + 3 * j]; }

template <typename T> void mat3_mul(const T *A, const T *B, T *C) {   /* C = A*B */
    T tmp[9];
    for (int j = 0; j < 3; j++)
        for (int i = 0; i < 3; i++) {
            T s = M3(A, i, 0) * M3(B, 0, j);
            s = s + M3(A, i, 1) * M3(B, 1, j);
            s = s + M3(A, i, 2) * M3(B, 2, j);
            tmp[i + 3 * j] = s;
        }
    std::memcpy(C, tmp, sizeof(tmp));
}
template <typename T> void mat3_vec(const T *A, const T *x, T *y) {   /* y = A*x */
    T tmp[3];
    for (int i = 0; i < 3; i++) {
        T s = M3(A, i, 0) * x[0];
        s = s + M3(A, i, 1) * x[1];
        s = s + M3(A, i, 2) * x[2];
        tmp[i] = s;
    }
    y[0] = tmp[0]; y[1] = tmp[1]; y[2] = tmp[2];
}
template <typename T> void mat3_transpose(const T *A, T *At) {
    T tmp[9];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) tmp[j + 3 * i] = M3(A, i, j);
    std::memcpy(At, tmp, sizeof(tmp));
}
template <typename T> void mat3_identity(T *A) {
    for (int k = 0; k < 9; k++) A[k] = T(0);
    A[0] = A[4] = A[8] = T(1);
}

/* SolveDVO::to_se_3  (src/SolveDVO.cpp:1104-1114) */
template <typename T> void to_se_3(const T *w, T *wx) {
    for (int k = 0; k < 9; k++) wx[k] = T(0);
    M3(wx, 1, 2) = -w[0];
    M3(wx, 0, 2) =  w[1];
    M3(wx, 0, 1) = -w[2];
    M3(wx, 2, 1) =  w[0];
    M3(wx, 2, 0) = -w[1];
    M3(wx, 1, 0) =  w[2];
}

inline double norm_n(const double *v, int n) {
    double s = 0.0;
    for (int k = 0; k < n; k++) s += v[k] * v[k];
    return std::sqrt(s);
}

/* ---- Eigen::Quaterniond(Matrix3d)  (Eigen/src/Geometry/Quaternion.h,
 *      quaternionbase_assign_impl<Other,3,3>) -- q = (w,x,y,z) ------------- */
void quat_from_matrix(const double *m, double *q) {
    double t = M3(m, 0, 0) + M3(m, 1, 1) + M3(m, 2, 2);
    if (t > 0.0) {
        t = std::sqrt(t + 1.0);
        q[0] = 0.5 * t;
        t = 0.5 / t;
        q[1] = (M3(m, 2, 1) - M3(m, 1, 2)) * t;
        q[2] = (M3(m, 0, 2) - M3(m, 2, 0)) * t;
        q[3] = (M3(m, 1, 0) - M3(m, 0, 1)) * t;
    } else {
        int i = 0;
        if (M3(m, 1, 1) > M3(m, 0, 0)) i = 1;
        if (M3(m, 2, 2) > M3(m, i, i)) i = 2;
        int j = (i + 1) % 3;
        int k = (j + 1) % 3;
        t = std::sqrt(M3(m, i, i) - M3(m, j, j) - M3(m, k, k) + 1.0);
        q[1 + i] = 0.5 * t;
        t = 0.5 / t;
        q[0] = (M3(m, k, j) - M3(m, j, k)) * t;
        q[1 + j] = (M3(m, j, i) + M3(m, i, j)) * t;
        q[1 + k] = (M3(m, k, i) + M3(m, i, k)) * t;
    }
}
void quat_normalize(double *q) {
    double n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    q[0] /= n; q[1] /= n; q[2] /= n; q[3] /= n;
}
/* Eigen::QuaternionBase::toRotationMatrix */
void quat_to_matrix(const double *q, double *R) {
    const double w = q[0], x = q[1], y = q[2], z = q[3];
    const double tx = 2.0 * x, ty = 2.0 * y, tz = 2.0 * z;
    const double twx = tx * w, twy = ty * w, twz = tz * w;
    const double txx = tx * x, txy = ty * x, txz = tz * x;
    const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
    M3(R, 0, 0) = 1.0 - (tyy + tzz);
    M3(R, 0, 1) = txy - twz;
    M3(R, 0, 2) = txz + twy;
    M3(R, 1, 0) = txy + twz;
    M3(R, 1, 1) = 1.0 - (txx + tzz);
    M3(R, 1, 2) = tyz - twx;
    M3(R, 2, 0) = txz - twy;
    M3(R, 2, 1) = tyz + twx;
    M3(R, 2, 2) = 1.0 - (txx + tyy);
}

const double kSophusEps = 1e-10;   /* SophusConstants<double>::epsilon() */

/* Sophus SO3Group::logAndTheta (atan-based, Hertzberg et al.) */
void so3_log_and_theta(const double *q, double *omega, double *theta) {
    const double squared_n = q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
    const double n = std::sqrt(squared_n);
    const double w = q[0];
    double two_atan_nbyw_by_n;
    if (n < kSophusEps) {
        const double squared_w = w * w;
        two_atan_nbyw_by_n = 2.0 / w - 2.0 * (squared_n) / (w * squared_w);
    } else {
        if (std::fabs(w) < kSophusEps) {
            if (w > 0.0) two_atan_nbyw_by_n = M_PI / n;
            else         two_atan_nbyw_by_n = -M_PI / n;
        } else {
            two_atan_nbyw_by_n = 2.0 * std::atan(n / w) / n;
        }
    }
    *theta = two_atan_nbyw_by_n * n;
    omega[0] = two_atan_nbyw_by_n * q[1];
    omega[1] = two_atan_nbyw_by_n * q[2];
    omega[2] = two_atan_nbyw_by_n * q[3];
}

/* Sophus SO3Group::expAndTheta -> unit quaternion (normalised by the SO3 ctor) */
void so3_exp_and_theta(const double *omega, double *q, double *theta) {
    const double theta_sq = omega[0] * omega[0] + omega[1] * omega[1] + omega[2] * omega[2];
    *theta = std::sqrt(theta_sq);
    const double half_theta = 0.5 * (*theta);
    double imag_factor, real_factor;
    if ((*theta) < kSophusEps) {
        const double theta_po4 = theta_sq * theta_sq;
        imag_factor = 0.5 - (1.0 / 48.0) * theta_sq + (1.0 / 3840.0) * theta_po4;
        real_factor = 1.0 - 0.5 * theta_sq + (1.0 / 384.0) * theta_po4;
    } else {
        const double sin_half_theta = std::sin(half_theta);
        imag_factor = sin_half_theta / (*theta);
        real_factor = std::cos(half_theta);
    }
    q[0] = real_factor;
    q[1] = imag_factor * omega[0];
    q[2] = imag_factor * omega[1];
    q[3] = imag_factor * omega[2];
    quat_normalize(q);
}

/* ---- Eigen JacobiSVD<Matrix3d, NoQRPreconditioner> (two-sided Jacobi) ---- */
struct Jrot { double c, s; };

/* JacobiRotation::makeJacobi(x, y, z) for the real symmetric 2x2 [[x,y],[y,z]] */
Jrot make_jacobi(double x, double y, double z) {
    Jrot r;
    const double deno = 2.0 * std::fabs(y);
    if (deno < 2.2250738585072014e-308) {
        r.c = 1.0; r.s = 0.0;
    } else {
        const double tau = (x - z) / deno;
        const double w = std::sqrt(tau * tau + 1.0);
        double t;
        if (tau > 0.0) t = 1.0 / (tau + w);
        else           t = 1.0 / (tau - w);
        const double sign_t = t > 0.0 ? 1.0 : -1.0;
        const double n = 1.0 / std::sqrt(t * t + 1.0);
        r.s = -sign_t * (y / std::fabs(y)) * std::fabs(t) * n;
        r.c = n;
    }
    return r;
}
/* Eigen's convention (Eigen/src/Jacobi/Jacobi.h): a JacobiRotation is
 * J = [[c, s], [-s, c]]; apply_rotation_in_the_plane(x, y, j) computes
 * x' = c x + s y ; y' = -s x + c y.  applyOnTheLeft(p,q,j) does that on rows
 * p,q (B = J*B); applyOnTheRight(p,q,j) does it on columns with j.transpose()
 * (B = B*J). */
void apply_left(double *A, int p, int q, Jrot j) {      /* B = J*B on rows p,q, J=[[c,s],[-s,c]] */
    const double c = j.c, s = j.s;
    for (int k = 0; k < 3; k++) {
        const double xp = M3(A, p, k), xq = M3(A, q, k);
        M3(A, p, k) = c * xp + s * xq;
        M3(A, q, k) = -s * xp + c * xq;
    }
}
void apply_right(double *A, int p, int q, Jrot j) {     /* columns p,q, with j.transpose() */
    const double c = j.c, s = -j.s;
    for (int k = 0; k < 3; k++) {
        const double xp = M3(A, k, p), xq = M3(A, k, q);
        M3(A, k, p) = c * xp + s * xq;
        M3(A, k, q) = -s * xp + c * xq;
    }
}
Jrot jrot_mul(Jrot a, Jrot b) {          /* JacobiRotation::operator* */
    Jrot r; r.c = a.c * b.c - a.s * b.s; r.s = a.c * b.s + a.s * b.c; return r;
}
Jrot jrot_transpose(Jrot a) { Jrot r; r.c = a.c; r.s = -a.s; return r; }

/* internal::real_2x2_jacobi_svd */
void real_2x2_jacobi_svd(const double *W, int p, int q, Jrot *j_left, Jrot *j_right) {
    double m00 = M3(W, p, p), m01 = M3(W, p, q), m10 = M3(W, q, p), m11 = M3(W, q, q);
    Jrot rot1;
    const double t = m00 + m11;
    const double d = m10 - m01;
    if (d == 0.0) {
        rot1.s = 0.0; rot1.c = 1.0;
    } else {
        const double u = t / d;
        const double tmp = std::sqrt(1.0 + u * u);
        rot1.s = 1.0 / tmp;
        rot1.c = u / tmp;
    }
    /* m.applyOnTheLeft(0,1,rot1) */
    {
        const double c = rot1.c, s = rot1.s;
        const double a00 = c * m00 + s * m10, a01 = c * m01 + s * m11;
        const double a10 = -s * m00 + c * m10, a11 = -s * m01 + c * m11;
        m00 = a00; m01 = a01; m10 = a10; m11 = a11;
    }
    *j_right = make_jacobi(m00, m01, m11);
    *j_left = jrot_mul(rot1, jrot_transpose(*j_right));
}

void jacobi_svd3(const double *Ain, double *U, double *S, double *V) {
    double W[9];
    double scale = 0.0;
    for (int k = 0; k < 9; k++) if (std::fabs(Ain[k]) > scale) scale = std::fabs(Ain[k]);
    if (scale == 0.0) scale = 1.0;
    for (int k = 0; k < 9; k++) W[k] = Ain[k] / scale;
    mat3_identity(U);
    mat3_identity(V);
    const double precision = 2.0 * 2.220446049250313e-16;
    const double considerAsZero = 2.0 * 2.2250738585072014e-308;
    bool finished = false;
    int sweeps = 0;
    while (!finished && sweeps < 100) {
        finished = true;
        sweeps++;
        for (int p = 1; p < 3; p++)
            for (int q = 0; q < p; q++) {
                double md = std::fabs(M3(W, p, p));
                if (std::fabs(M3(W, q, q)) > md) md = std::fabs(M3(W, q, q));
                double threshold = precision * md;
                if (considerAsZero > threshold) threshold = considerAsZero;
                double off = std::fabs(M3(W, p, q));
                if (std::fabs(M3(W, q, p)) > off) off = std::fabs(M3(W, q, p));
                if (off > threshold) {
                    finished = false;
                    Jrot jl, jr;
                    real_2x2_jacobi_svd(W, p, q, &jl, &jr);
                    apply_left(W, p, q, jl);
                    apply_right(U, p, q, jrot_transpose(jl));
                    apply_right(W, p, q, jr);
                    apply_right(V, p, q, jr);
                }
            }
    }
    /* singular values = |diag|, flip U columns for negative entries */
    for (int i = 0; i < 3; i++) {
        const double a = M3(W, i, i);
        S[i] = std::fabs(a) * scale;
        if (a < 0.0) for (int k = 0; k < 3; k++) M3(U, k, i) = -M3(U, k, i);
    }
    /* sort descending, swapping columns of U and V */
    for (int i = 0; i < 3; i++) {
        int pos = i;
        for (int k = i + 1; k < 3; k++) if (S[k] > S[pos]) pos = k;
        if (pos != i) {
            double ts = S[i]; S[i] = S[pos]; S[pos] = ts;
            for (int k = 0; k < 3; k++) {
                double tu = M3(U, k, i); M3(U, k, i) = M3(U, k, pos); M3(U, k, pos) = tu;
                double tv = M3(V, k, i); M3(V, k, i) = M3(V, k, pos); M3(V, k, pos) = tv;
            }
        }
    }
}

/* per-level projection constants (src/SolveDVO.cpp:334-337,344): M = diag(s,s,1)*K */
struct LevelCam {
    float scaleFac;
    float M[9];     /* column-major */
};
LevelCam make_level_cam(int level, float fx, float fy, float cx, float cy) {
    LevelCam lc;
    lc.scaleFac = (float)std::pow(2.0, (double)(-level));          /* :334 */
    float S[9], K[9];
    mat3_identity(S);
    M3(S, 0, 0) = lc.scaleFac;                                      /* :336 */
    M3(S, 1, 1) = lc.scaleFac;                                      /* :337 */
    mat3_identity(K);
    M3(K, 0, 0) = fx; M3(K, 1, 1) = fy; M3(K, 0, 2) = cx; M3(K, 1, 2) = cy;
    mat3_mul(S, K, lc.M);                                           /* (scaleMatrix*K), :344 */
    return lc;
}

/* visibility test of :371 / :435 with the half-open fix of SURVEY Q3:
 * reference skips when u<0 || u>nCols || v<0 || v>nRows; u==nCols would index
 * out of range (Eigen assert, NDEBUG undefined).  Oracle: visible iff
 * 0<=u<nCols && 0<=v<nRows (false for NaN). */
inline bool is_visible(float u, float v, int nRows, int nCols) {
    return (u >= 0.0f) && (u < (float)nCols) && (v >= 0.0f) && (v < (float)nRows);
}

}  // namespace

extern "C" {

void dvo_oracle_params_default(dvo_oracle_params *p) {
    p->beta = 0.5;                  /* :653 */
    p->precond_rot = .5;            /* :725 */
    p->reg_lambda = 0.05;           /* :742 */
    p->step_a = 9.0;                /* :773 */
    p->step_b = 1.0E-2;             /* :773 */
    p->step_decay_after = 5;        /* :773 */
    p->step_decay_offset = 4;       /* :773 */
    p->trust_radius = 0.003;        /* :25, float member (double literal narrowed) */
    p->psi_norm_stop = 1.0E-7;      /* :24, float member */
    p->enable_rotationize = 1;      /* SolveDVO.h:107 */
    p->enable_l2_reg = 1;           /* SolveDVO.h:112 */
    p->interpolate_dt = 0;          /* SolveDVO.h:97 (commented out) */
}

/* SolveDVO::getWeightOf  (src/SolveDVO.cpp:1047-1053): r*r in float, the rest
 * in double (6.0 and .25 are double literals), narrowed on return. */
float dvo_oracle_weight(float r) {
    return (float)(6.0 / (6.0 + (double)(r * r) / .25));
}

/* SolveDVO::interpolate (src/SolveDVO.cpp:1285-1308).  The reference would
 * index one past the end when ceil() hits rows/cols; the oracle clamps. */
float dvo_oracle_interpolate(const float *F, int rows, int cols, float ry, float rx) {
    int ry_d = (int)std::floor((double)ry);
    int rx_d = (int)std::floor((double)rx);
    int ry_u = (int)std::ceil((double)ry);
    int rx_u = (int)std::ceil((double)rx);
    float inc_x = rx - (float)rx_d;
    float inc_y = ry - (float)ry_d;
    if (ry_u > rows - 1) ry_u = rows - 1;
    if (rx_u > cols - 1) rx_u = cols - 1;
#define F_(y, x) F[(y) + (x) * rows]
    float a = (1.0f - inc_x) * F_(ry_d, rx_d) * F_(ry_d, rx_d) + (inc_x) * F_(ry_d, rx_u) * F_(ry_d, rx_u);
    float f_xdyd_xuyd = (float)std::sqrt((double)a);
    float b = (1.0f - inc_x) * F_(ry_u, rx_d) * F_(ry_u, rx_d) + (inc_x) * F_(ry_u, rx_u) * F_(ry_u, rx_u);
    float f_xdyu_xuyu = (float)std::sqrt((double)b);
#undef F_
    float c = (1.0f - inc_y) * f_xdyd_xuyd * f_xdyd_xuyd + inc_y * f_xdyu_xuyu * f_xdyu_xuyu;
    return (float)std::sqrt((double)c);
}

/* selectedPts (:1230-1264) + enlistRefEdgePts (:224-264). */
int dvo_oracle_enlist_ref_points(int level, const int *edge, const float *depth_mm,
                                 int rows, int cols,
                                 float fx, float fy, float cx, float cy,
                                 float *xyz, float *uv, int capacity) {
    int nC = 0;
    float scaleFac = (float)std::pow(2.0, (double)(-level));       /* :231 */
    float tmpfx = (float)(1. / (double)(scaleFac * fx));           /* :232, 1./ is a double division */
    float tmpfy = (float)(1. / (double)(scaleFac * fy));           /* :233 */
    float tmpcx = scaleFac * cx;                                   /* :234 */
    float tmpcy = scaleFac * cy;                                   /* :235 */
    for (int xx = 0; xx < cols; xx++) {                            /* :237  column-major scan */
        for (int yy = 0; yy < rows; yy++) {                        /* :239 */
            const int e = edge[yy + xx * rows];
            const float d = depth_mm[yy + xx * rows];
            if ((e > 0) && (d > 100.0f)) {                         /* :1251 */
                if (nC >= capacity) return -1;
                float Z = d / 1000.0f;                             /* :248 */
                float X = Z * ((float)xx - tmpcx) * tmpfx;         /* :249 */
                float Y = Z * ((float)yy - tmpcy) * tmpfy;         /* :250 */
                if (uv) { uv[2 * nC + 0] = (float)xx; uv[2 * nC + 1] = (float)yy; }   /* :244-245 */
                xyz[3 * nC + 0] = X;                               /* :254-256 */
                xyz[3 * nC + 1] = Y;
                xyz[3 * nC + 2] = Z;
                nC++;
            }
        }
    }
    return nC;
}

/* computeJacobianOfNowFrame (:306-414) + getReprojectedEpsilons (:425-462) */
void dvo_oracle_eval_points(const dvo_oracle_params *prm, int level,
                            const float *xyz, int N,
                            const float *dt, const float *gx, const float *gy,
                            int rows, int cols,
                            float fx, float fy, float cx, float cy,
                            const float *cR, const float *cT,
                            float *reproj, float *Jout, float *eps_out, float *w_out, int *vis_out) {
    const LevelCam lc = make_level_cam(level, fx, fy, cx, cy);
    const float scaleFac = lc.scaleFac;
    float cRt[9];
    mat3_transpose(cR, cRt);                                       /* cR.transpose() */
    const int nCols = cols, nRows = rows;

    for (int i = 0; i < N; i++) {
        /* Step 2 (:328-330): _3d_transformed = cR^T * (_3d - cTRep) */
        float d[3];
        d[0] = xyz[3 * i + 0] - cT[0];
        d[1] = xyz[3 * i + 1] - cT[1];
        d[2] = xyz[3 * i + 2] - cT[2];
        float p[3];
        mat3_vec(cRt, d, p);
        /* Step 3 (:339-341): all three rows *= 1/z */
        const float inv = 1.0f / p[2];
        p[0] = p[0] * inv;
        p[1] = p[1] * inv;
        p[2] = p[2] * inv;
        /* (:344) uv1 = (scaleMatrix*K) * P'; the structurally-zero terms of M
         * are dropped (identical for finite operands). */
        const float u = M3(lc.M, 0, 0) * p[0] + M3(lc.M, 0, 2) * p[2];
        const float v = M3(lc.M, 1, 1) * p[1] + M3(lc.M, 1, 2) * p[2];
        const float w3 = p[2];
        if (reproj) { reproj[3 * i + 0] = u; reproj[3 * i + 1] = v; reproj[3 * i + 2] = w3; }

        float Ji[6] = {0, 0, 0, 0, 0, 0};                          /* rows of invisible points stay 0 (:671) */
        float e = 0.0f, wgt = 0.0f;                                /* :429-430 */
        const bool vis = is_visible(u, v, nRows, nCols);           /* :371, :435 */
        if (vis) {
            const int xx = (int)u;                                 /* :376 */
            const int yy = (int)v;                                 /* :377 */
            const float X = p[0], Y = p[1], Z = p[2];              /* :379-381 (dehomogenised: quirk Q1) */
            float G[2];
            G[0] = gx[yy + xx * rows];                             /* :384 */
            G[1] = gy[yy + xx * rows];                             /* :385 */
            float A1[2][3];
            A1[0][0] = scaleFac * fx / Z;                          /* :388 */
            A1[0][1] = 0.f;
            A1[0][2] = -scaleFac * fx * X / (Z * Z);               /* :390 */
            A1[1][0] = 0.f;
            A1[1][1] = scaleFac * fy / Z;                          /* :392 */
            A1[1][2] = -scaleFac * fy * Y / (Z * Z);               /* :393 */
            float A2[3][6];
            for (int r = 0; r < 3; r++)
                for (int c = 0; c < 3; c++) A2[r][c] = -M3(cRt, r, c);   /* :397 */
            float tmp[3];
            mat3_vec(cRt, p, tmp);                                 /* :399 (cR^T applied again: quirk Q2) */
            float wx[9];
            to_se_3(tmp, wx);                                      /* :401 */
            for (int r = 0; r < 3; r++)
                for (int c = 0; c < 3; c++) A2[r][3 + c] = M3(wx, r, c);  /* :402 */
            /* :405  J_i = (G*A1)*A2, inner sums in index order */
            float GA[3];
            for (int c = 0; c < 3; c++) GA[c] = G[0] * A1[0][c] + G[1] * A1[1][c];
            for (int c = 0; c < 6; c++) {
                float s = GA[0] * A2[0][c];
                s = s + GA[1] * A2[1][c];
                s = s + GA[2] * A2[2][c];
                Ji[c] = s;
            }
            /* :446 / :444 */
            if (prm && prm->interpolate_dt) {
                e = dvo_oracle_interpolate(dt, rows, cols, v, u);
            } else {
                const int fy_ = (int)std::floor((double)v);
                const int fx_ = (int)std::floor((double)u);
                e = dt[fy_ + fx_ * rows];
            }
            wgt = dvo_oracle_weight(e);                            /* :450 */
        }
        if (Jout) for (int c = 0; c < 6; c++) Jout[6 * i + c] = Ji[c];
        if (eps_out) eps_out[i] = e;
        if (w_out) w_out[i] = wgt;
        if (vis_out) vis_out[i] = vis ? 1 : 0;
    }
}

/* Sophus SE3d::exp (tangent = [upsilon(3), omega(3)]) */
void dvo_oracle_se3_exp(const double *psi, double *R, double *t) {
    const double *upsilon = psi, *omega = psi + 3;
    double q[4], theta;
    so3_exp_and_theta(omega, q, &theta);
    quat_to_matrix(q, R);
    double Omega[9], Omega_sq[9], V[9];
    to_se_3(omega, Omega);                              /* SO3::hat */
    mat3_mul(Omega, Omega, Omega_sq);
    if (theta < kSophusEps) {
        std::memcpy(V, R, sizeof(V));
    } else {
        const double theta_sq = theta * theta;
        const double a = (1.0 - std::cos(theta)) / (theta_sq);
        const double b = (theta - std::sin(theta)) / (theta_sq * theta);
        double I[9]; mat3_identity(I);
        for (int k = 0; k < 9; k++) V[k] = (I[k] + a * Omega[k]) + b * Omega_sq[k];
    }
    mat3_vec(V, upsilon, t);
}

/* Sophus SE3d::log, after cGrp.setRotationMatrix(cR); cGrp.translation()=cT (:736-739) */
void dvo_oracle_se3_log(const double *R, const double *t, double *psi) {
    double q[4];
    quat_from_matrix(R, q);
    quat_normalize(q);
    double omega[3], theta;
    so3_log_and_theta(q, omega, &theta);
    double Omega[9], Omega_sq[9], V_inv[9], I[9];
    to_se_3(omega, Omega);
    mat3_mul(Omega, Omega, Omega_sq);
    mat3_identity(I);
    if (std::fabs(theta) < kSophusEps) {
        for (int k = 0; k < 9; k++) V_inv[k] = (I[k] - 0.5 * Omega[k]) + (1. / 12.) * Omega_sq[k];
    } else {
        const double c = (1.0 - theta / (2.0 * std::tan(theta / 2.0))) / (theta * theta);
        for (int k = 0; k < 9; k++) V_inv[k] = (I[k] - 0.5 * Omega[k]) + c * Omega_sq[k];
    }
    mat3_vec(V_inv, t, psi);
    psi[3] = omega[0]; psi[4] = omega[1]; psi[5] = omega[2];
}

void dvo_oracle_svd3(const double *A, double *U, double *S, double *V) { jacobi_svd3(A, U, S, V); }

/* SolveDVO::rotationize (:1269-1282): R = U * diag(sign(sigma)) * V^T */
void dvo_oracle_rotationize(double *R) {
    double U[9], S[3], V[9], Sm[9], Vt[9], US[9];
    jacobi_svd3(R, U, S, V);
    mat3_identity(Sm);
    M3(Sm, 0, 0) = (S[0] > 0) ? 1.0 : -1.0;
    M3(Sm, 1, 1) = (S[1] > 0) ? 1.0 : -1.0;
    M3(Sm, 2, 2) = (S[2] > 0) ? 1.0 : -1.0;
    mat3_transpose(V, Vt);
    mat3_mul(U, Sm, US);
    mat3_mul(US, Vt, R);
}

static void accumulate_from(const float *J, const float *eps, const float *w, const int *vis, int n,
                            double *acc29);

/* ---- sum of eps^2 without an order (round 6) -------------------------------------------------------------------------------
 * aggregateEpsilons (:1310-1312) is epsilon.norm(): a float sum whose order is Eigen's business.  The oracle's definition since round 1
 * is E = (float)sqrt(S), S = sum of (double)eps_i^2; rounds 1-5 added the terms one by one, so S -- and, once in ~50 000 energies, the
 * float E -- depended on the order, which no parallel sum can reproduce.  Since round 6 S is the CORRECTLY ROUNDED double of the exact
 * sum: every eps^2 is a 48-bit integer times a power of four; on the grid of 2^-68 the terms of |eps| in [2^-11, 2^12) (every
 * normalised distance, :1063-1098) are integers below 2^92, summed exactly in three 32-bit limbs (kept as doubles: < 2^53 for up to 2^21
 * points, so partial sums of shards add exactly in ANY order -- tiled mode all-reduces them) and rounded once.  A value outside that
 * range (never produced by the pipeline) marks limb 2 with 2^50 and the sequential sum is used as before. */
static const int kE2Exp0 = 116, kE2MaxBinades = 22;
static const double kE2Bad = 1125899906842624.0;      /* 2^50 */
void dvo_oracle_e2_limbs(const float *eps, int n, double *limbs3) {
    unsigned long long L[3] = {0ull, 0ull, 0ull};
    double bad = 0.0;
    for (int i = 0; i < n; i++) {
        unsigned bits;
        std::memcpy(&bits, &eps[i], 4);
        bits &= 0x7fffffffu;
        if (bits == 0u) continue;
        const int e = (int)(bits >> 23);
        if (e < kE2Exp0 || e > kE2Exp0 + kE2MaxBinades) { bad += 1.0; continue; }
        const unsigned long long m = (bits & 0x7fffffu) | 0x800000u;
        const unsigned __int128 v = (unsigned __int128)(m * m) << (2 * (e - kE2Exp0));
        L[0] += (unsigned long long)(v & 0xffffffffu);
        L[1] += (unsigned long long)((v >> 32) & 0xffffffffu);
        L[2] += (unsigned long long)(v >> 64);
    }
    limbs3[0] = (double)L[0];
    limbs3[1] = (double)L[1];
    limbs3[2] = (double)L[2] + bad * kE2Bad;
}
/* limbs (possibly sums of several shards' limbs) -> the correctly rounded double of the exact sum; `fallback` when a term was out of range */
double dvo_oracle_e2_from_limbs(const double *limbs3, double fallback) {
    if (!(limbs3[2] < kE2Bad)) return fallback;
    unsigned __int128 S = (unsigned __int128)(unsigned long long)limbs3[0] + ((unsigned __int128)(unsigned long long)limbs3[1] << 32) +
                          ((unsigned __int128)(unsigned long long)limbs3[2] << 64);
    if (S == 0) return 0.0;
    int p = 127;
    while (!((S >> p) & 1)) p--;
    if (p <= 52) return std::ldexp((double)(unsigned long long)S, -68);
    const int r = p - 52;
    unsigned long long q = (unsigned long long)(S >> r);
    const unsigned __int128 rem = S & ((((unsigned __int128)1) << r) - 1), half = ((unsigned __int128)1) << (r - 1);
    if (rem > half || (rem == half && (q & 1ull))) q++;
    return std::ldexp((double)q, r - 68);
}
static double sum_eps2_exact(const float *eps, int n) {
    double seq = 0.0, limbs[3];
    for (int i = 0; i < n; i++) seq += (double)eps[i] * (double)eps[i];
    dvo_oracle_e2_limbs(eps, n, limbs);
    return dvo_oracle_e2_from_limbs(limbs, seq);
}

/* The 29 sums of one evaluation over points [first, first+n) at a float pose:
 * acc[0..20] upper triangle of sum w J J^T, acc[21..26] g = (J^T W) eps (:714-720, :777),
 * acc[27] sum eps^2 (:1312; correctly rounded exact sum), acc[28] number of visible points.  The others: sequential order. */
void dvo_oracle_accumulate(const dvo_oracle_params *prm, int level, const float *xyz, int first, int n,
                           const float *dt, const float *gx, const float *gy, int rows, int cols,
                           float fx, float fy, float cx, float cy,
                           const float *cR_32, const float *cT_32, double *acc29) {
    std::vector<float> J(6 * (size_t)n), eps(n), w(n);
    std::vector<int> vis(n);
    dvo_oracle_eval_points(prm, level, xyz + 3 * (size_t)first, n, dt, gx, gy, rows, cols, fx, fy, cx, cy,
                           cR_32, cT_32, nullptr, J.data(), eps.data(), w.data(), vis.data());
    accumulate_from(J.data(), eps.data(), w.data(), vis.data(), n, acc29);
}
/* the same with the three limbs of the range's exact sum of eps^2 in acc32[29..31] (tiled mode: the ranks' limbs add exactly, the
 * reduced acc32[27] is then replaced by dvo_oracle_e2_from_limbs(acc32 + 29, acc32[27])) */
void dvo_oracle_accumulate32(const dvo_oracle_params *prm, int level, const float *xyz, int first, int n,
                             const float *dt, const float *gx, const float *gy, int rows, int cols,
                             float fx, float fy, float cx, float cy,
                             const float *cR_32, const float *cT_32, double *acc32) {
    std::vector<float> J(6 * (size_t)n), eps(n), w(n);
    std::vector<int> vis(n);
    dvo_oracle_eval_points(prm, level, xyz + 3 * (size_t)first, n, dt, gx, gy, rows, cols, fx, fy, cx, cy,
                           cR_32, cT_32, nullptr, J.data(), eps.data(), w.data(), vis.data());
    accumulate_from(J.data(), eps.data(), w.data(), vis.data(), n, acc32);
    dvo_oracle_e2_limbs(eps.data(), n, acc32 + 29);
}

static void accumulate_from(const float *J, const float *eps, const float *w, const int *vis, int n,
                            double *acc29) {
    for (int k = 0; k < 29; k++) acc29[k] = 0.0;
    for (int i = 0; i < n; i++) {
        float jw[6];
        for (int k = 0; k < 6; k++) jw[k] = J[6 * i + k] * w[i];                    /* :716 */
        for (int k = 0; k < 6; k++) acc29[21 + k] += (double)jw[k] * (double)eps[i];   /* :719-720, :777 */
        int h = 0;
        for (int a_ = 0; a_ < 6; a_++)
            for (int b_ = a_; b_ < 6; b_++) acc29[h++] += (double)jw[a_] * (double)J[6 * i + b_];
        if (vis[i]) acc29[28] += 1.0;
    }
    acc29[27] = sum_eps2_exact(eps, n);                 /* the correctly rounded exact sum, see above */
}

/* ---- runIterations as an explicit state machine (same arithmetic as :642-1005) ---- */
void dvo_oracle_state_begin(dvo_oracle_state *s, const double *R, const double *t) {
    std::memcpy(s->R, R, sizeof(double) * 9);
    std::memcpy(s->t, t, sizeof(double) * 3);
    for (int k = 0; k < 6; k++) s->d[k] = 0.0;                      /* :654 */
    mat3_identity(s->bestR);                                        /* :646 */
    s->bestT[0] = s->bestT[1] = s->bestT[2] = 0.0;                  /* :647 */
    s->bestE = 1.0E10;                                              /* :644 */
    s->bestRatio = 1.0f;                                            /* :645 */
    s->bestItr = -1;                                                /* :648 */
    s->stop = 0;
}

/* everything after the per-point phase of iteration itr (:689-920); returns 1 on early termination */
int dvo_oracle_state_update(const dvo_oracle_params *prm, dvo_oracle_state *s, int itr, int N,
                            const double *g_in, double sum_eps2, int n_vis, float *energy_out,
                            double *psi_out) {
    double *cR = s->R, *cT = s->t;
    /* aggregateEpsilons (:1310-1312) = epsilon.norm().  Oracle definition:
     * (float)sqrt(S), S = the correctly rounded exact sum of (double)eps^2 (sum_eps2_exact above): no order. */
    const float currentTotalEpsilon = (float)std::sqrt(sum_eps2);
    *energy_out = currentTotalEpsilon;                              /* :690 */
    const float ratio_of_visible_pts = (float)n_vis / (float)N;     /* :457 */
    if (currentTotalEpsilon <= s->bestE) {                          /* :696 */
        s->bestE = currentTotalEpsilon;
        s->bestRatio = ratio_of_visible_pts;
        std::memcpy(s->bestR, cR, sizeof(double) * 9);
        std::memcpy(s->bestT, cT, sizeof(double) * 3);
        s->bestItr = itr;
    }
    double g[6];
    for (int k = 0; k < 6; k++) g[k] = g_in[k];
    if (psi_out) for (int k = 0; k < 6; k++) psi_out[k] = 0.0;

    /* :724-730 pre-conditioner */
    double PVec[6] = {1.0, 1.0, 1.0, prm->precond_rot, prm->precond_rot, prm->precond_rot};

    /* :734-743 L2 regulariser direction */
    double cPsi[6] = {0, 0, 0, 0, 0, 0};
    if (prm->enable_l2_reg) {
        dvo_oracle_se3_log(cR, cT, cPsi);
        const double n = norm_n(cPsi, 6);
        if (n > 0) for (int k = 0; k < 6; k++) cPsi[k] = cPsi[k] / n;
    }

    /* :773 */
    const double stepLength = prm->step_a * prm->step_b /
                              ((itr > prm->step_decay_after) ? (double)(itr - prm->step_decay_offset) : 1.0);

    if (prm->enable_l2_reg)
        for (int k = 0; k < 6; k++) g[k] += prm->reg_lambda * cPsi[k];            /* :796 */

    for (int k = 0; k < 6; k++)                                     /* :799, 1.0f-BETA promotes to double */
        s->d[k] = (1.0f - prm->beta) * g[k] + prm->beta * s->d[k];

    double psi[6];
    for (int k = 0; k < 6; k++) psi[k] = -stepLength * PVec[k] * s->d[k];          /* :821 */

    const double norm = norm_n(psi, 6);                             /* :832 */
    if (norm > (double)prm->trust_radius) {                         /* :835 */
        for (int k = 0; k < 6; k++) psi[k] = psi[k] / norm * (double)prm->trust_radius;   /* :837 */
    }
    /* :840 dangling else binds to :872; behaviourally "clamp, then test" (quirk Q5) */
    if (norm_n(psi, 6) < (double)prm->psi_norm_stop) {              /* :872 */
        s->stop = 1;
        return 1;                                                   /* :877 */
    }

    double xRot[9], xTrans[3];
    dvo_oracle_se3_exp(psi, xRot, xTrans);                          /* :905-907 */
    double dT[3];
    mat3_vec(cR, xTrans, dT);
    for (int k = 0; k < 3; k++) cT[k] += dT[k];                     /* :916 */
    mat3_mul(cR, xRot, cR);                                         /* :917 */
    if (prm->enable_rotationize) dvo_oracle_rotationize(cR);        /* :919 */
    if (psi_out) std::memcpy(psi_out, psi, sizeof(psi));
    return 0;
}

/* :997-1001 */
void dvo_oracle_state_finish(const dvo_oracle_params *prm, dvo_oracle_state *s, double *R, double *t) {
    std::memcpy(R, s->bestR, sizeof(double) * 9);                   /* :997 */
    if (prm->enable_rotationize) dvo_oracle_rotationize(R);         /* :999 */
    std::memcpy(t, s->bestT, sizeof(double) * 3);                   /* :1001 */
}

/* SolveDVO::runIterations (:619-1017) */
int dvo_oracle_run_iterations(const dvo_oracle_params *prm_in, int level, int maxIterations,
                              const float *xyz, int N,
                              const float *dt, const float *gx, const float *gy,
                              int rows, int cols,
                              float fx, float fy, float cx, float cy,
                              double *cR, double *cT,
                              float *energyAtEachIteration, float *finalEpsilons, float *finalReprojections,
                              int *bestEnergyIndex, float *finalVisibleRatio,
                              dvo_oracle_iter_trace *trace) {
    dvo_oracle_params defaults;
    dvo_oracle_params_default(&defaults);
    const dvo_oracle_params *prm = prm_in ? prm_in : &defaults;

    for (int k = 0; k < maxIterations; k++) energyAtEachIteration[k] = 0.0f;     /* :634 */

    dvo_oracle_state st;
    dvo_oracle_state_begin(&st, cR, cT);                            /* :642-657 */
    std::vector<float> bestEpsilon, bestReprojections;              /* :649-650 */
    std::vector<float> reprojections(3 * (size_t)N), epsilon(N), Jcbian(6 * (size_t)N), weights(N);
    std::vector<int> visible(N);
    int evaluated = 0;

    for (int itr = 0; itr < maxIterations; itr++) {                /* :658 */
        float cR_32[9], cT_32[3];
        for (int k = 0; k < 9; k++) cR_32[k] = (float)st.R[k];     /* :673 */
        for (int k = 0; k < 3; k++) cT_32[k] = (float)st.t[k];     /* :674 */
        dvo_oracle_eval_points(prm, level, xyz, N, dt, gx, gy, rows, cols, fx, fy, cx, cy,
                               cR_32, cT_32, reprojections.data(), Jcbian.data(),
                               epsilon.data(), weights.data(), visible.data());   /* :675, :687 */
        double acc[29];
        accumulate_from(Jcbian.data(), epsilon.data(), weights.data(), visible.data(), N, acc);   /* :714-720, :777 */
        const int prevBest = st.bestItr;
        float energy;
        double psi[6];
        const int broke = dvo_oracle_state_update(prm, &st, itr, N, acc + 21, acc[27], (int)acc[28], &energy, psi);
        energyAtEachIteration[itr] = energy;                        /* :690 */
        evaluated = itr + 1;
        if (st.bestItr != prevBest) {                               /* :703-704 */
            bestEpsilon = epsilon;
            bestReprojections = reprojections;
        }
        if (trace) {
            std::memcpy(trace[itr].g, acc + 21, sizeof(double) * 6);
            std::memcpy(trace[itr].H, acc, sizeof(double) * 21);
            trace[itr].sum_eps2 = acc[27];
            trace[itr].energy = energy;
            trace[itr].n_visible = (int)acc[28];
            trace[itr].broke = broke;
            std::memcpy(trace[itr].psi, psi, sizeof(psi));
            std::memcpy(trace[itr].R, st.R, sizeof(double) * 9);
            std::memcpy(trace[itr].t, st.t, sizeof(double) * 3);
        }
        if (broke) break;                                           /* :877 */
    }

    dvo_oracle_state_finish(prm, &st, cR, cT);                      /* :997-1001 */
    if (finalEpsilons && !bestEpsilon.empty())
        std::memcpy(finalEpsilons, bestEpsilon.data(), sizeof(float) * (size_t)N);         /* :1002 */
    if (finalReprojections && !bestReprojections.empty())
        std::memcpy(finalReprojections, bestReprojections.data(), sizeof(float) * 3 * (size_t)N);   /* :1003 */
    *bestEnergyIndex = st.bestItr;                                  /* :1004 */
    *finalVisibleRatio = st.bestRatio;                              /* :1005 */
    return evaluated;
}

/* ---- now-frame preprocessing after Canny (computeDistTransfrmOfNow, :1768-1795; imageGradient :1063-1098) ----
 * edge > 0 marks an edge pixel (the reference inverts the Canny output so that edges are the zero set of
 * cv::distanceTransform(CV_DIST_L2, CV_DIST_MASK_PRECISE), :1768-1771).  OpenCV 2.4 is not available here
 * (PARITY UNPINNED for this step as well); the definition restated is: exact Euclidean distance
 * (Felzenszwalb-Huttenlocher lower envelopes, as DIST_MASK_PRECISE does), narrowed to float;
 * cv::normalize(src, dst, 0, 255, NORM_MINMAX) (:1774) with OpenCV 2.4's own arithmetic (core/src/convert.cpp): in double
 *   scale = (255 - 0) * (smax - smin > DBL_EPSILON ? 1./(smax - smin) : 0),  shift = 0 - smin*scale
 * (smin, smax from minMaxLoc, doubles), then Mat::convertTo(CV_32F, scale, shift), whose 32F -> 32F kernel cvtScale32f
 * (DEF_CVT_SCALE_FUNC(32f, float, float, float)) works in FLOAT: dst = src*(float)scale + (float)shift, one rounding per
 * operation.  (Round 1 evaluated (src-min)*(255/(max-min)) in double: differs in the last bit on many pixels.)
 * filter2D with [-.5 0 .5] kernels and the default BORDER_REFLECT_101 (:1077-1090). */
static void fh_1d(const double *f, int n, double *d, int *v, double *z) {
    const double INF = 1e20;
    int k = 0;
    v[0] = 0; z[0] = -INF; z[1] = INF;
    for (int q = 1; q < n; q++) {
        double s;
        for (;;) {
            s = ((f[q] + (double)q * q) - (f[v[k]] + (double)v[k] * v[k])) / (2.0 * q - 2.0 * v[k]);
            if (s <= z[k] && k > 0) k--; else break;
        }
        if (s <= z[k]) { v[k] = q; z[k] = -INF; z[k + 1] = INF; }      /* k == 0 and q dominates */
        else { k++; v[k] = q; z[k] = s; z[k + 1] = INF; }
    }
    k = 0;
    for (int q = 0; q < n; q++) {
        while (z[k + 1] < (double)q) k++;
        d[q] = ((double)q - v[k]) * ((double)q - v[k]) + f[v[k]];
    }
}

void dvo_oracle_now_level_from_edges(const unsigned char *edge, int rows, int cols,
                                     float *dt, float *gx, float *gy) {
    const double INF = 1e20;
    const size_t n = (size_t)rows * cols;
    std::vector<double> d2(n);
    const int m = rows > cols ? rows : cols;
    std::vector<double> f(m), d(m), z(m + 1);
    std::vector<int> v(m);
    for (size_t i = 0; i < n; i++) d2[i] = edge[i] ? 0.0 : INF;
    for (int x = 0; x < cols; x++) {                                 /* along yy (contiguous) */
        for (int y = 0; y < rows; y++) f[y] = d2[(size_t)x * rows + y];
        fh_1d(f.data(), rows, d.data(), v.data(), z.data());
        for (int y = 0; y < rows; y++) d2[(size_t)x * rows + y] = d[y];
    }
    for (int y = 0; y < rows; y++) {                                 /* along xx */
        for (int x = 0; x < cols; x++) f[x] = d2[(size_t)x * rows + y];
        fh_1d(f.data(), cols, d.data(), v.data(), z.data());
        for (int x = 0; x < cols; x++) d2[(size_t)x * rows + y] = d[x];
    }
    float mn = 0.f, mx = 0.f;
    for (size_t i = 0; i < n; i++) {
        dt[i] = (float)std::sqrt(d2[i]);
        if (i == 0 || dt[i] < mn) mn = dt[i];
        if (i == 0 || dt[i] > mx) mx = dt[i];
    }
    const double smin = (double)mn, smax = (double)mx;                                           /* minMaxLoc */
    const double scale = (255.0 - 0.0) * ((smax - smin > 2.2204460492503131e-16) ? 1. / (smax - smin) : 0.);
    const double shift = 0.0 - smin * scale;
    const float scale_f = (float)scale, shift_f = (float)shift;                                 /* cvtScale32f: float working type */
    for (size_t i = 0; i < n; i++) dt[i] = dt[i] * scale_f + shift_f;                           /* :1774 */
    auto r101 = [](int i, int len) { if (len == 1) return 0; if (i < 0) return -i; if (i >= len) return 2 * len - 2 - i; return i; };
    for (int x = 0; x < cols; x++)
        for (int y = 0; y < rows; y++) {
            const float l_ = dt[(size_t)r101(x - 1, cols) * rows + y], r_ = dt[(size_t)r101(x + 1, cols) * rows + y];
            const float u_ = dt[(size_t)x * rows + r101(y - 1, rows)], b_ = dt[(size_t)x * rows + r101(y + 1, rows)];
            gx[(size_t)x * rows + y] = 0.5f * r_ - 0.5f * l_;        /* kernX :1077-1079 */
            gy[(size_t)x * rows + y] = 0.5f * b_ - 0.5f * u_;        /* kernY :1080-1082 */
        }
}

/* level schedule of SolveDVO::loop (:2097-2104) */
int dvo_oracle_align_pyramid(const dvo_oracle_params *prm, int n_levels, const int *iters,
                             const float *const *xyz, const int *N,
                             const float *const *dt, const float *const *gx, const float *const *gy,
                             const int *rows, const int *cols,
                             float fx, float fy, float cx, float cy,
                             double *R, double *t,
                             float *energy_out, int *best_idx_out, float *ratio_out,
                             float *final_eps, float *final_reproj) {
    std::vector<int> off(n_levels + 1, 0);
    for (int l = 0; l < n_levels; l++) off[l + 1] = off[l] + (iters[l] > 0 ? iters[l] : 0);
    for (int l = 0; l < n_levels; l++) { best_idx_out[l] = -1; ratio_out[l] = 0.0f; }
    int last = -1;
    for (int f = n_levels - 1; f >= 0; f--) if (iters[f] > 0) last = f;
    for (int f = n_levels - 1; f >= 0; f--) {                       /* :2097 */
        if (iters[f] > 0) {                                         /* :2099 */
            dvo_oracle_run_iterations(prm, f, iters[f], xyz[f], N[f], dt[f], gx[f], gy[f],
                                      rows[f], cols[f], fx, fy, cx, cy, R, t,
                                      energy_out + off[f],
                                      (f == last) ? final_eps : nullptr,
                                      (f == last) ? final_reproj : nullptr,
                                      &best_idx_out[f], &ratio_out[f], nullptr);   /* :2102 */
        }
    }
    return 0;
}

}  // extern "C"
