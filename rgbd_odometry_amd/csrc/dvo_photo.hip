/*
 * dvo_photo.hip -- the legacy photometric Gauss-Newton odometry of the reference (RGBDOdometry, src/RGBDOdometry.cpp)
 * as HIP kernels: the engine behind dvo_amd::RGBDOdometry.
 *
 *   photo_count / photo_scan / photo_jacobian   computeJacobian (:407-508): forward-difference gradients (filter2D with
 *                        [0 -1 1] kernels, BORDER_REFLECT_101), the pixels with gx >= threshold in the reference's
 *                        column-major scan order, their 1x6 Jacobian rows in double
 *   photo_ata            A = J^T J (:379), the 21 normal-equation sums in double, fixed-shape reduction
 *   photo_gauss_newton   gaussNewtonIterations (:514-597) of one level in ONE single-workgroup launch: per iteration
 *                        computeEpsilon (:602-700: warp with T^-1, project, nearest lookup in the now image),
 *                        b = -J^T eps, the 6x6 solve (Householder QR with column pivoting, what colPivHouseholderQr
 *                        computes), exponentialMap (:713-746), T = T * exp(psi)^-1
 *
 * Everything is double precision like the reference (Eigen::MatrixXd).  The reference's defects are reproduced by default
 * and corrected with `fixed` (the list and the decisions: include/dvo_amd.h at dvo_photo_params).  Images come from the
 * frame store: grey u8 and depth f32 (sensor units), COLUMN-major -- the linear order of the store is the reference's scan
 * order (j outer over columns, i inner over rows, :460-462).
 */
#include "dvo_kernel_common.h"

namespace dvo {

struct PhotoK { double fx, fy, cx, cy; int fixed; double grad_threshold; };

DVO_DEV int photo_reflect101(int p, int len) { return (len == 1) ? 0 : ((p < 0) ? -p : ((p >= len) ? 2 * len - 2 - p : p)); }
/* gx(i,j) = -I(i,j) + I(i,j+1)   (kernX :423-425) */
DVO_DEV double photo_gx(const unsigned char *g, int rows, int cols, int i, int j) {
    return -(double)g[(size_t)j * rows + i] + (double)g[(size_t)photo_reflect101(j + 1, cols) * rows + i];
}
DVO_DEV double photo_gy(const unsigned char *g, int rows, int cols, int i, int j) {
    return -(double)g[(size_t)j * rows + i] + (double)g[(size_t)j * rows + photo_reflect101(i + 1, rows)];
}

/* one wave per image column: number of selected pixels (gx >= threshold, :467) */
__global__ void __launch_bounds__(64)
photo_count_kernel(const unsigned char *__restrict__ grey, int rows, int cols, double thr, int *__restrict__ col_counts) {
    const int j = blockIdx.x, lane = threadIdx.x;
    int n = 0;
    for (int i0 = 0; i0 < rows; i0 += 64) {
        const int i = i0 + lane;
        const bool sel = (i < rows) && (photo_gx(grey, rows, cols, i, j) >= thr);
        n += __popcll(__builtin_amdgcn_ballot_w64(sel));
    }
    if (lane == 0) col_counts[j] = n;
}

/* exclusive scan of the column counts (single workgroup); offs[cols] = total */
__global__ void __launch_bounds__(1024)
photo_scan_kernel(const int *__restrict__ col_counts, int cols, int *__restrict__ offs) {
    __shared__ int part[1024];
    const int per = (cols + 1023) / 1024;
    const int b0 = threadIdx.x * per;
    int s = 0;
    for (int k = 0; k < per; k++) if (b0 + k < cols) s += col_counts[b0 + k];
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        int run = 0;
        for (int k = 0; k < 1024; k++) { const int v = part[k]; part[k] = run; run += v; }
        offs[cols] = run;
    }
    __syncthreads();
    int run = part[threadIdx.x];
    for (int k = 0; k < per; k++)
        if (b0 + k < cols) { offs[b0 + k] = run; run += col_counts[b0 + k]; }
}

/* one wave per column: ordered write of the selected pixels and their Jacobian rows (:460-495) */
__global__ void __launch_bounds__(64)
photo_jacobian_kernel(const unsigned char *__restrict__ grey, const float *__restrict__ depth, int rows, int cols, int level,
                      PhotoK K, const int *__restrict__ offs, int capacity,
                      double *__restrict__ J, int *__restrict__ sel, double *__restrict__ zref, float *__restrict__ gref) {
    const int j = blockIdx.x, lane = threadIdx.x;
    double fx = K.fx, fy = K.fy, cx = K.cx, cy = K.cy;
    if (K.fixed) { const double s = (double)pow2_neg(level); fx *= s; fy *= s; cx *= s; cy *= s; }       /* D4 */
    int base = offs[j];
    for (int i0 = 0; i0 < rows; i0 += 64) {
        const int i = i0 + lane;
        double gx = 0.0;
        bool s = false;
        if (i < rows) { gx = photo_gx(grey, rows, cols, i, j); s = gx >= K.grad_threshold; }
        const unsigned long long m = __builtin_amdgcn_ballot_w64(s);
        const int rank = __popcll(m & ((1ull << lane) - 1ull));
        const int xc = base + rank;
        if (s && xc < capacity) {
            const double gy = photo_gy(grey, rows, cols, i, j);
            const double Z = (double)depth[(size_t)j * rows + i];
            const double X = Z * (i - cx) / fx;                            /* :475 (D3: the ROW index with cx, fx) */
            const double Y = Z * (j - cy) / fy;                            /* :476 */
            const double invZ = 1 / Z, invZ2 = 1 / (Z * Z);
            double *t = J + (size_t)xc * 6;
            t[0] = K.fixed ? fx * gx * invZ : fx * fx * invZ;              /* :485 (D1) */
            t[1] = fy * gy * invZ;
            t[2] = -fy * gy * Y * invZ2 - fx * gx * X * invZ2;
            t[3] = gy * (-fy * Y * Y * invZ2 - fy) - fx * gx * X * Y * invZ2;
            t[4] = gx * (fx * X * X * invZ2 + fx) + fx * gy * X * Y * invZ2;
            t[5] = K.fixed ? fy * gy * X * invZ - fx * gx * Y * invZ : fy * gy * X * invZ - fx * gy * Y * invZ;   /* :490 (D2) */
            sel[xc] = i | (j << 16);
            zref[xc] = Z;
            gref[xc] = (float)grey[(size_t)j * rows + i];
        }
        base += __popcll(m);
    }
}

/* A = J^T J (:379): 21 sums over n rows, single workgroup, fixed-shape reduction; the full symmetric 6x6 is written */
__global__ void __launch_bounds__(1024)
photo_ata_kernel(const double *__restrict__ J, const int *__restrict__ n_ptr, int capacity, double *__restrict__ A36) {
    __shared__ double red[16][32];
    const int n = (*n_ptr < capacity) ? *n_ptr : capacity;
    double h[32];
#pragma unroll
    for (int k = 0; k < 32; k++) h[k] = 0.0;
    for (int r = threadIdx.x; r < n; r += 1024) {
        double v[6];
#pragma unroll
        for (int k = 0; k < 6; k++) v[k] = J[(size_t)r * 6 + k];
        int q = 0;
#pragma unroll
        for (int a = 0; a < 6; a++)
#pragma unroll
            for (int b = a; b < 6; b++) { h[q] = fma(v[a], v[b], h[q]); q++; }
    }
    wave_reduce_scatter<double, 32>(h);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if ((lane & 1) == 0) red[wave][lane >> 1] = h[0];
    __syncthreads();
    if (threadIdx.x < 21) {
        double s = 0.0;
        for (int w = 0; w < 16; w++) s += red[w][threadIdx.x];
        int q = 0;
        for (int a = 0; a < 6; a++)
            for (int b = a; b < 6; b++) { if (q == (int)threadIdx.x) { A36[a * 6 + b] = s; A36[b * 6 + a] = s; } q++; }
    }
}

/* ---- 6x6 / 4x4 double helpers of the update (one lane) -------------------------------------------------------- */
DVO_DEV void photo_solve6(const double *A_in, const double *b_in, double *x) {      /* Householder QR, column pivoting */
    double A[6][6], b[6], y[6];
    int perm[6];
    for (int i = 0; i < 6; i++) { for (int j = 0; j < 6; j++) A[i][j] = A_in[i * 6 + j]; b[i] = b_in[i]; perm[i] = i; y[i] = 0.0; }
    double maxpivot = 0.0;
    int rank = 6;
    for (int k = 0; k < 6; k++) {
        int best = k; double bestn = -1.0;
        for (int j = k; j < 6; j++) {
            double s = 0.0;
            for (int i = k; i < 6; i++) s += A[i][j] * A[i][j];
            if (s > bestn) { bestn = s; best = j; }
        }
        if (best != k) {
            for (int i = 0; i < 6; i++) { const double tmp = A[i][k]; A[i][k] = A[i][best]; A[i][best] = tmp; }
            const int tp = perm[k]; perm[k] = perm[best]; perm[best] = tp;
        }
        const double norm = sqrt(bestn);
        if (k == 0) maxpivot = norm;
        if (norm <= 2.220446049250313e-16 * 6 * maxpivot || norm == 0.0) { rank = k; break; }
        const double alpha = (A[k][k] > 0.0) ? -norm : norm;
        double v[6] = {0, 0, 0, 0, 0, 0};
        for (int i = k; i < 6; i++) v[i] = A[i][k];
        v[k] -= alpha;
        double vn2 = 0.0;
        for (int i = k; i < 6; i++) vn2 += v[i] * v[i];
        if (vn2 > 0.0) {
            for (int j = k; j < 6; j++) {
                double dot = 0.0;
                for (int i = k; i < 6; i++) dot += v[i] * A[i][j];
                const double f = 2.0 * dot / vn2;
                for (int i = k; i < 6; i++) A[i][j] -= f * v[i];
            }
            double dot = 0.0;
            for (int i = k; i < 6; i++) dot += v[i] * b[i];
            const double f = 2.0 * dot / vn2;
            for (int i = k; i < 6; i++) b[i] -= f * v[i];
        }
    }
    for (int k = rank - 1; k >= 0; k--) {
        double s = b[k];
        for (int j = k + 1; j < rank; j++) s -= A[k][j] * y[j];
        y[k] = s / A[k][k];
    }
    for (int k = 0; k < 6; k++) x[perm[k]] = y[k];
}
DVO_DEV void photo_inv3(const double *m, double *o) {
    const double d = m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
    const double id = 1.0 / d;
    o[0] = (m[4] * m[8] - m[5] * m[7]) * id; o[1] = (m[2] * m[7] - m[1] * m[8]) * id; o[2] = (m[1] * m[5] - m[2] * m[4]) * id;
    o[3] = (m[5] * m[6] - m[3] * m[8]) * id; o[4] = (m[0] * m[8] - m[2] * m[6]) * id; o[5] = (m[2] * m[3] - m[0] * m[5]) * id;
    o[6] = (m[3] * m[7] - m[4] * m[6]) * id; o[7] = (m[1] * m[6] - m[0] * m[7]) * id; o[8] = (m[0] * m[4] - m[1] * m[3]) * id;
}
DVO_DEV void photo_affine_inverse(const double *T, double *Ti) {          /* [L t; 0 1]^-1 = [L^-1, -L^-1 t; 0 1] */
    const double L[9] = {T[0], T[1], T[2], T[4], T[5], T[6], T[8], T[9], T[10]};
    double Li[9];
    photo_inv3(L, Li);
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) Ti[i * 4 + j] = Li[i * 3 + j];
        Ti[i * 4 + 3] = -((Li[i * 3] * T[3] + Li[i * 3 + 1] * T[7]) + Li[i * 3 + 2] * T[11]);
    }
    Ti[12] = Ti[13] = Ti[14] = 0.0; Ti[15] = 1.0;
}
DVO_DEV void photo_exponential_map(const double *psi, int fixed, double *out) {     /* :713-746 */
    const double *t = psi, *w = psi + 3;
    const double wx[9] = {0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0};
    const double theta = sqrt((w[0] * w[0] + w[1] * w[1]) + w[2] * w[2]);
    for (int k = 0; k < 16; k++) out[k] = (k % 5 == 0) ? 1.0 : 0.0;
    if (theta < 1E-12) {                                                               /* :727-731 (D7) */
        if (fixed) { out[3] = t[0]; out[7] = t[1]; out[11] = t[2]; }
        return;
    }
    double wx2[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) wx2[i * 3 + j] = (wx[i * 3] * wx[j] + wx[i * 3 + 1] * wx[3 + j]) + wx[i * 3 + 2] * wx[6 + j];
    const double a = sin(theta) / theta, b = (1.0 - cos(theta)) / (theta * theta);
    const double c = (theta - sin(theta)) / (theta * theta * theta);
    double R[9], V[9];
    for (int k = 0; k < 9; k++) {
        const double I = (k % 4 == 0) ? 1.0 : 0.0;
        R[k] = (I + a * wx[k]) + b * wx2[k];
        V[k] = (I + b * wx[k]) + c * wx2[k];
    }
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) out[i * 4 + j] = R[i * 3 + j];
        out[i * 4 + 3] = (V[i * 3] * t[0] + V[i * 3 + 1] * t[1]) + V[i * 3 + 2] * t[2];
    }
}

/* gaussNewtonIterations (:514-597) of one level: single workgroup, all iterations in one launch */
__global__ void __launch_bounds__(1024)
photo_gauss_newton_kernel(const double *__restrict__ J, const int *__restrict__ sel, const double *__restrict__ zref,
                          const float *__restrict__ gref, const int *__restrict__ n_ptr, const double *__restrict__ A36,
                          const unsigned char *__restrict__ grey_now, int rows, int cols, int level, PhotoK K,
                          int max_iters, double eps_stop, double *__restrict__ T16, double *__restrict__ eps_norms,
                          int *__restrict__ updates, double *__restrict__ eps_dump /* may be NULL: eps of the LAST evaluation */) {
    __shared__ double Ti[16];
    __shared__ double red[16][8];
    __shared__ double tot[8];
    __shared__ int stop;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = *n_ptr;
    double fx = K.fx, fy = K.fy, cx = K.cx, cy = K.cy;
    if (K.fixed) { const double s = (double)pow2_neg(level); fx *= s; fy *= s; cx *= s; cy *= s; }
    if (tid == 0) { *updates = 0; stop = 0; }
    __syncthreads();
    for (int k = tid; k < max_iters; k += 1024) eps_norms[k] = -1.0;
    for (int itr = 0; itr < max_iters; itr++) {                            /* :545 */
        if (tid == 0) photo_affine_inverse(T16, Ti);                       /* T.inverse() :665 */
        __syncthreads();
        double d[8];
#pragma unroll
        for (int k = 0; k < 8; k++) d[k] = 0.0;
        for (int r = tid; r < n; r += 1024) {                              /* computeEpsilon :602-700 */
            const int i = sel[r] & 0xffff, j = sel[r] >> 16;
            const double Z = zref[r];
            const double X = Z * (i - cx) / fx, Y = Z * (j - cy) / fy;
            const double o0 = ((Ti[0] * X + Ti[1] * Y) + Ti[2] * Z) + Ti[3];
            const double o1 = ((Ti[4] * X + Ti[5] * Y) + Ti[6] * Z) + Ti[7];
            const double o2 = ((Ti[8] * X + Ti[9] * Y) + Ti[10] * Z) + Ti[11];
            const double outu = o0 * fx / o2 + cx, outv = o1 * fy / o2 + cy;       /* :670-671 */
            double e = 0.0;
            if (outu >= 0 && outu < rows && outv >= 0 && outv < cols)              /* :683 */
                e = (double)gref[r] - (double)grey_now[(size_t)(int)floor(outv) * rows + (int)floor(outu)];
            if (eps_dump) eps_dump[r] = e;
#pragma unroll
            for (int k = 0; k < 6; k++) d[k] = fma(J[(size_t)r * 6 + k], e, d[k]);  /* J^T eps :566 */
            d[6] = fma(e, e, d[6]);
        }
        wave_reduce_scatter<double, 8>(d);
        if ((lane & 7) == 0) red[wave][lane >> 3] = d[0];
        __syncthreads();
        if (tid < 8) {
            double s = 0.0;
            for (int w = 0; w < 16; w++) s += red[w][tid];
            tot[tid] = s;
        }
        __syncthreads();
        if (tid == 0) {
            const double nrm = sqrt(tot[6]);
            eps_norms[itr] = nrm;
            if (nrm < eps_stop) stop = 1;                                  /* :556 (D8) */
            else {
                double b[6], psi[6], outTr[16], inv[16], Tn[16];
                for (int k = 0; k < 6; k++) b[k] = -tot[k];
                photo_solve6(A36, b, psi);                                 /* :568 */
                photo_exponential_map(psi, K.fixed, outTr);                /* :575 */
                photo_affine_inverse(outTr, inv);                          /* outTr.inverse() :579 */
                for (int r = 0; r < 4; r++)
                    for (int q = 0; q < 4; q++) {
                        double s = 0.0;
                        for (int k = 0; k < 4; k++) s += T16[r * 4 + k] * inv[k * 4 + q];
                        Tn[r * 4 + q] = s;
                    }
                for (int k = 0; k < 16; k++) T16[k] = Tn[k];
                *updates = *updates + 1;
            }
        }
        __syncthreads();
        if (stop) break;
    }
}

hipError_t launch_photo_reference(const unsigned char *grey, const float *depth, int rows, int cols, int level,
                                  double fx, double fy, double cx, double cy, int fixed, double grad_threshold, int capacity,
                                  int *col_work /* cols + 2 ints x 2 */, double *J, int *sel, double *zref, float *gref, double *A36,
                                  int *n_out, hipStream_t s) {
    PhotoK K{fx, fy, cx, cy, fixed, grad_threshold};
    int *counts = col_work, *offs = col_work + cols + 1;
    hipLaunchKernelGGL(photo_count_kernel, dim3(cols), dim3(64), 0, s, grey, rows, cols, grad_threshold, counts);
    hipLaunchKernelGGL(photo_scan_kernel, dim3(1), dim3(1024), 0, s, counts, cols, offs);
    hipLaunchKernelGGL(photo_jacobian_kernel, dim3(cols), dim3(64), 0, s, grey, depth, rows, cols, level, K, offs, capacity, J, sel,
                       zref, gref);
    hipError_t e = hipMemcpyAsync(n_out, offs + cols, sizeof(int), hipMemcpyDeviceToDevice, s);
    if (e != hipSuccess) return e;
    /* rows beyond the capacity were not written: the host rejects such a reference (the reference asserts, :464) */
    hipLaunchKernelGGL(photo_ata_kernel, dim3(1), dim3(1024), 0, s, J, n_out, capacity, A36);
    return hipGetLastError();
}

hipError_t launch_photo_gauss_newton(const double *J, const int *sel, const double *zref, const float *gref, const int *n_dev,
                                     const double *A36, const unsigned char *grey_now, int rows, int cols, int level,
                                     double fx, double fy, double cx, double cy, int fixed, int max_iters, double eps_stop,
                                     double *T16, double *eps_norms, int *updates, double *eps_dump, hipStream_t s) {
    PhotoK K{fx, fy, cx, cy, fixed, 0.0};
    hipLaunchKernelGGL(photo_gauss_newton_kernel, dim3(1), dim3(1024), 0, s, J, sel, zref, gref, n_dev, A36, grey_now, rows, cols,
                       level, K, max_iters, eps_stop, T16, eps_norms, updates, eps_dump);
    return hipGetLastError();
}

}  // namespace dvo
