/*
 * dvo_oracle_photo.cpp -- CPU oracle of the legacy photometric Gauss-Newton odometry:
 * RGBDOdometry::computeJacobian / computeJacobianAllLevels / gaussNewtonIterations / computeEpsilon / exponentialMap
 * (reference src/RGBDOdometry.cpp:363-398, :407-508, :514-597, :602-700, :713-746).
 *
 * TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline).  PARITY UNPINNED: the reference ships
 * no vectors for this path either and cannot be built here.
 *
 * The reference code has defects.  Decision per defect -- `fixed` = 0 reproduces the reference's
 * arithmetic as written (the parity mode: a user switching from the reference node gets the numbers that node computes);
 * `fixed` = 1 is the corrected estimator:
 *   D1  tJ(0) = fx*fx*invZ                       (:485)  fixed: fx*gx*invZ
 *   D2  tJ(5) = fy*gy*X*invZ - fx*gy*Y*invZ     (:490)  fixed: fy*gy*X*invZ - fx*gx*Y*invZ
 *   D3  rows/columns swapped: X from the ROW index with cx, fx; projection compared with rows (:475-476, :661-662, :683)
 *       kept in both modes: it is a self-consistent transposed image convention (u <-> row), not an arithmetic error
 *   D4  level-0 intrinsics at every pyramid level (:475-476) fixed: fx, fy, cx, cy scaled by 2^-level
 *   D5  depth stays in sensor units (mm); the pose translation is in mm (:185-187 multiplies by 1000 again when publishing)
 *       kept in both modes (a unit convention)
 *   D6  only pixels with gx >= 5 are used (signed, x-gradient only) (:467)   kept: it is the selection policy
 *   D7  exponentialMap returns the identity for |w| < 1e-12, dropping the translation (:727-731)   fixed: V = I
 *   D8  iterations stop when |eps| < 200 (absolute) (:556)   kept: policy
 * Evaluation order of every expression follows the source text; Eigen's matrix products / inverses / QR are restated from
 * their definitions (sums in index order; affine inverse through the 3x3 cofactor inverse; Householder QR with column
 * pivoting): agreement with Eigen to ~1e-15 relative, not bit-exact -- the tests compare poses at 1e-9.
 */
#include "dvo_oracle.h"

#include <cmath>
#include <cstring>
#include <vector>

namespace {

/* cv::filter2D(src, dst, CV_64F, kern) with the 3x3 kernels of :423-428 and the default BORDER_REFLECT_101:
 * gx(i,j) = -I(i,j) + I(i,j+1), gy(i,j) = -I(i,j) + I(i+1,j)  (filter2D is a correlation, anchor at the centre) */
inline int reflect101(int p, int len) {
    if (len == 1) return 0;
    if (p < 0) return -p;
    if (p >= len) return 2 * len - 2 - p;
    return p;
}

/* A x = b for a 6x6 A by Householder QR with column pivoting (what A.colPivHouseholderQr().solve(b) computes);
 * pivots below eps*6*|largest pivot| are treated as zero like Eigen's default threshold */
void solve6_colpiv_qr(const double *A_in /* row-major */, const double *b_in, double *x) {
    const int n = 6;
    double A[6][6], b[6];
    int perm[6];
    for (int i = 0; i < n; i++) { for (int j = 0; j < n; j++) A[i][j] = A_in[i * 6 + j]; b[i] = b_in[i]; perm[i] = i; }
    double maxpivot = 0.0;
    int rank = n;
    for (int k = 0; k < n; k++) {
        int best = k; double bestn = -1.0;
        for (int j = k; j < n; j++) {
            double s = 0.0;
            for (int i = k; i < n; i++) s += A[i][j] * A[i][j];
            if (s > bestn) { bestn = s; best = j; }
        }
        if (best != k) { for (int i = 0; i < n; i++) std::swap(A[i][k], A[i][best]); std::swap(perm[k], perm[best]); }
        double norm = std::sqrt(bestn);
        if (k == 0) maxpivot = norm;
        if (norm <= 2.220446049250313e-16 * 6 * maxpivot || norm == 0.0) { rank = k; break; }
        const double alpha = (A[k][k] > 0.0) ? -norm : norm;
        double v[6] = {0, 0, 0, 0, 0, 0};
        for (int i = k; i < n; i++) v[i] = A[i][k];
        v[k] -= alpha;
        double vnorm2 = 0.0;
        for (int i = k; i < n; i++) vnorm2 += v[i] * v[i];
        if (vnorm2 > 0.0) {
            for (int j = k; j < n; j++) {
                double dot = 0.0;
                for (int i = k; i < n; i++) dot += v[i] * A[i][j];
                const double f = 2.0 * dot / vnorm2;
                for (int i = k; i < n; i++) A[i][j] -= f * v[i];
            }
            double dot = 0.0;
            for (int i = k; i < n; i++) dot += v[i] * b[i];
            const double f = 2.0 * dot / vnorm2;
            for (int i = k; i < n; i++) b[i] -= f * v[i];
        }
    }
    double y[6] = {0, 0, 0, 0, 0, 0};
    for (int k = rank - 1; k >= 0; k--) {
        double s = b[k];
        for (int j = k + 1; j < rank; j++) s -= A[k][j] * y[j];
        y[k] = s / A[k][k];
    }
    for (int k = 0; k < n; k++) x[perm[k]] = y[k];
}

void inv3(const double *m /* row-major */, double *o) {
    const double d = m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
    const double id = 1.0 / d;
    o[0] = (m[4] * m[8] - m[5] * m[7]) * id; o[1] = (m[2] * m[7] - m[1] * m[8]) * id; o[2] = (m[1] * m[5] - m[2] * m[4]) * id;
    o[3] = (m[5] * m[6] - m[3] * m[8]) * id; o[4] = (m[0] * m[8] - m[2] * m[6]) * id; o[5] = (m[2] * m[3] - m[0] * m[5]) * id;
    o[6] = (m[3] * m[7] - m[4] * m[6]) * id; o[7] = (m[1] * m[6] - m[0] * m[7]) * id; o[8] = (m[0] * m[4] - m[1] * m[3]) * id;
}

/* inverse of the affine transform [L t; 0 1] (row-major 4x4): [L^-1, -L^-1 t; 0 1] */
void affine_inverse(const double *T, double *Ti) {
    double L[9] = {T[0], T[1], T[2], T[4], T[5], T[6], T[8], T[9], T[10]}, Li[9];
    inv3(L, Li);
    const double t[3] = {T[3], T[7], T[11]};
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) Ti[i * 4 + j] = Li[i * 3 + j];
        Ti[i * 4 + 3] = -((Li[i * 3] * t[0] + Li[i * 3 + 1] * t[1]) + Li[i * 3 + 2] * t[2]);
    }
    Ti[12] = Ti[13] = Ti[14] = 0.0; Ti[15] = 1.0;
}

void mat4_mul(const double *A, const double *B, double *C) {
    double t[16];
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            double s = 0.0;
            for (int k = 0; k < 4; k++) s += A[i * 4 + k] * B[k * 4 + j];
            t[i * 4 + j] = s;
        }
    std::memcpy(C, t, sizeof(t));
}

/* RGBDOdometry::exponentialMap (:713-746): psi = [t(3), w(3)] -> 4x4 */
void exponential_map(const double *psi, int fixed, double *out) {
    const double *t = psi, *w = psi + 3;
    const double wx[9] = {0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0};           /* to_se_3 :753-764 */
    const double theta = std::sqrt((w[0] * w[0] + w[1] * w[1]) + w[2] * w[2]);
    for (int k = 0; k < 16; k++) out[k] = (k % 5 == 0) ? 1.0 : 0.0;
    if (theta < 1E-12) {                                                               /* :727-731 (D7) */
        if (fixed) { out[3] = t[0]; out[7] = t[1]; out[11] = t[2]; }
        return;
    }
    double wx2[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) wx2[i * 3 + j] = (wx[i * 3] * wx[j] + wx[i * 3 + 1] * wx[3 + j]) + wx[i * 3 + 2] * wx[6 + j];
    const double a = std::sin(theta) / theta, b = (1.0 - std::cos(theta)) / (theta * theta);
    const double c = (theta - std::sin(theta)) / (theta * theta * theta);
    double R[9], V[9];
    for (int k = 0; k < 9; k++) {
        const double I = (k % 4 == 0) ? 1.0 : 0.0;
        R[k] = (I + a * wx[k]) + b * wx2[k];                                           /* :739 */
        V[k] = (I + b * wx[k]) + c * wx2[k];                                           /* :741 */
    }
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) out[i * 4 + j] = R[i * 3 + j];
        out[i * 4 + 3] = (V[i * 3] * t[0] + V[i * 3 + 1] * t[1]) + V[i * 3 + 2] * t[2];
    }
}

}  // namespace

extern "C" {

/* computeJacobian (:407-508) of one pyramid level.  grey: rows x cols uint8 ROW-major (cv::Mat); depth: rows x cols in sensor
 * units (uint16 -> double).  Outputs: J (n x 6 row-major), sel_i / sel_j (the selected pixels in the reference's
 * column-major scan order), A = J^T J (6x6).  Returns n, or -1 if n exceeds capacity (:464 assert). */
int dvo_oracle_photo_jacobian(const unsigned char *grey, const unsigned short *depth, int rows, int cols, int level,
                              double fx, double fy, double cx, double cy, int fixed, double grad_threshold, int capacity,
                              double *J, int *sel_i, int *sel_j, double *A36) {
    if (fixed) { const double s = std::ldexp(1.0, -level); fx *= s; fy *= s; cx *= s; cy *= s; }      /* D4 */
    int xc = 0;
    for (int j = 0; j < cols; j++)                                         /* :460-462: Eigen column-major scan */
        for (int i = 0; i < rows; i++) {
            const double c0 = (double)grey[(size_t)i * cols + j];
            const double gx = -c0 + (double)grey[(size_t)i * cols + reflect101(j + 1, cols)];
            if (gx < grad_threshold) continue;                             /* :467 (D6) */
            if (xc >= capacity) return -1;
            const double gy = -c0 + (double)grey[(size_t)reflect101(i + 1, rows) * cols + j];
            const double Z = (double)depth[(size_t)i * cols + j];
            const double X = Z * (i - cx) / fx;                            /* :475 (D3: the ROW index) */
            const double Y = Z * (j - cy) / fy;                            /* :476 */
            const double invZ = 1 / Z, invZ2 = 1 / (Z * Z);
            double *t = J + (size_t)xc * 6;
            t[0] = fixed ? fx * gx * invZ : fx * fx * invZ;                /* :485 (D1) */
            t[1] = fy * gy * invZ;
            t[2] = -fy * gy * Y * invZ2 - fx * gx * X * invZ2;
            t[3] = gy * (-fy * Y * Y * invZ2 - fy) - fx * gx * X * Y * invZ2;
            t[4] = gx * (fx * X * X * invZ2 + fx) + fx * gy * X * Y * invZ2;
            t[5] = fixed ? fy * gy * X * invZ - fx * gx * Y * invZ : fy * gy * X * invZ - fx * gy * Y * invZ;   /* :490 (D2) */
            sel_i[xc] = i; sel_j[xc] = j;
            xc++;
        }
    for (int a = 0; a < 6; a++)
        for (int b = 0; b < 6; b++) {
            double s = 0.0;
            for (int k = 0; k < xc; k++) s += J[(size_t)k * 6 + a] * J[(size_t)k * 6 + b];       /* A = J^T J :379 */
            A36[a * 6 + b] = s;
        }
    return xc;
}

/* computeEpsilon (:602-700): eps[k] for the n selected pixels under T (4x4 row-major); returns |eps| */
double dvo_oracle_photo_epsilon(const unsigned char *grey_ref, const unsigned short *depth_ref, const unsigned char *grey_now,
                                int rows, int cols, int level, double fx, double fy, double cx, double cy, int fixed,
                                const int *sel_i, const int *sel_j, int n, const double *T16, double *eps) {
    if (fixed) { const double s = std::ldexp(1.0, -level); fx *= s; fy *= s; cx *= s; cy *= s; }
    double Ti[16];
    affine_inverse(T16, Ti);                                               /* T.inverse() :665 */
    double s2 = 0.0;
    for (int k = 0; k < n; k++) {
        const int i = sel_i[k], j = sel_j[k];
        const double Z = (double)depth_ref[(size_t)i * cols + j];
        const double X = Z * (i - cx) / fx, Y = Z * (j - cy) / fy;         /* :660-662 */
        const double o0 = ((Ti[0] * X + Ti[1] * Y) + Ti[2] * Z) + Ti[3];
        const double o1 = ((Ti[4] * X + Ti[5] * Y) + Ti[6] * Z) + Ti[7];
        const double o2 = ((Ti[8] * X + Ti[9] * Y) + Ti[10] * Z) + Ti[11];
        const double outu = o0 * fx / o2 + cx, outv = o1 * fy / o2 + cy;   /* :670-671 */
        double e = 0.0;
        if (outu >= 0 && outu < rows && outv >= 0 && outv < cols)          /* :683 (noe_gray.rows(), .cols()) */
            e = (double)grey_ref[(size_t)i * cols + j] - (double)grey_now[(size_t)(int)std::floor(outu) * cols + (int)std::floor(outv)];
        eps[k] = e;
        s2 += e * e;
    }
    return std::sqrt(s2);
}

/* gaussNewtonIterations (:514-597) on one level: up to max_iters iterations of eps -> b = -J^T eps -> QR solve -> T = T exp(psi)^-1.
 * T16 in/out.  eps_norms[max_iters] receives |eps| of every iteration run (-1 where not run).  Returns iterations that updated T. */
int dvo_oracle_photo_gauss_newton(const unsigned char *grey_ref, const unsigned short *depth_ref, const unsigned char *grey_now,
                                  int rows, int cols, int level, double fx, double fy, double cx, double cy, int fixed,
                                  const double *J, const int *sel_i, const int *sel_j, int n, const double *A36,
                                  int max_iters, double eps_stop, double *T16, double *eps_norms) {
    std::vector<double> eps(n > 0 ? n : 1);
    int updates = 0;
    for (int itr = 0; itr < max_iters; itr++) eps_norms[itr] = -1.0;
    for (int itr = 0; itr < max_iters; itr++) {                            /* :545 */
        const double nrm = dvo_oracle_photo_epsilon(grey_ref, depth_ref, grey_now, rows, cols, level, fx, fy, cx, cy, fixed,
                                                    sel_i, sel_j, n, T16, eps.data());
        eps_norms[itr] = nrm;
        if (nrm < eps_stop) break;                                         /* :556 (D8) */
        double b[6];
        for (int a = 0; a < 6; a++) {
            double s = 0.0;
            for (int k = 0; k < n; k++) s += J[(size_t)k * 6 + a] * eps[k];
            b[a] = -s;                                                     /* :566 */
        }
        double psi[6], outTr[16], inv[16];
        solve6_colpiv_qr(A36, b, psi);                                     /* :568 */
        exponential_map(psi, fixed, outTr);                                /* :575 */
        /* outTr.inverse() (:579) is a general 4x4 inverse of a rigid transform; restated through the affine form */
        affine_inverse(outTr, inv);
        mat4_mul(T16, inv, T16);
        updates++;
    }
    return updates;
}

void dvo_oracle_photo_exponential_map(const double *psi6, int fixed, double *out16) { exponential_map(psi6, fixed, out16); }
void dvo_oracle_photo_solve6(const double *A36, const double *b6, double *x6) { solve6_colpiv_qr(A36, b6, x6); }

}  // extern "C"
