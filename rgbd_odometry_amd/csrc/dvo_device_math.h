/*
 * dvo_device_math.h -- device-side arithmetic of the edge-alignment hot path.
 *
 * Two parts:
 *  (1) per-point float32 math of computeJacobianOfNowFrame + getReprojectedEpsilons
 *      (reference src/SolveDVO.cpp:306-414, :425-462, :1047-1053) in an
 *      algebraically simplified scalar form whose every rounding matches the
 *      evaluation order fixed in SURVEY.md 8a ("Precision and evaluation-order
 *      map").  Compiled with -ffp-contract=off: the only fused operations are
 *      the explicit fma() calls, all on products that are exact in the wider type.
 *  (2) the double-precision 6-DoF update of runIterations (:724-920): SE(3)
 *      log/exp (what Sophus::SE3d does at :736-739, :905-907), rotationize
 *      (:1269-1282), heavy-ball / trust-region step.
 *
 * 3x3 matrices are column-major: M(i,j) = m[i+3*j].
 */
#ifndef DVO_DEVICE_MATH_H_
#define DVO_DEVICE_MATH_H_

#include <hip/hip_runtime.h>

#define DVO_DEV __device__ __forceinline__
/* the double-precision update runs on one lane per workgroup; keeping it out of
 * line keeps its register appetite away from the per-point loop */
#define DVO_DEV_NOINLINE __device__ __noinline__

namespace dvo {

/* Device copy of dvo_params (doubles widened on the host exactly like the
 * reference widens its float members at use, SolveDVO.cpp:835,872). */
struct DevParams {
    double beta, precond_rot, reg_lambda, step_a, step_b;
    double trust_radius, psi_norm_stop;
    int step_decay_after, step_decay_offset;
    int enable_rotationize, enable_l2_reg;
};

/* Per-iteration, per-level constants: float pose (cast at :673-674) and the
 * non-zero entries of M = diag(s,s,1)*K (:334-337,344).  Wave-uniform. */
struct IterConst {
    float r[9];      /* cR, column-major */
    float t[3];      /* cT */
    float m00, m02, m11, m12;   /* s*fx, s*cx, s*fy, s*cy  (float products) */
    float ncols_f, nrows_f;
    int rows;
};

struct PointEval {
    float u, v, zn;      /* reprojection (3 x N column of `reprojections`, :345) */
    float J[6];          /* Jacobian row (:405), 0 if not visible */
    float eps, w;        /* residual (:446) and weight (:450), 0 if not visible */
    bool vis;
};

/* getWeightOf (:1047-1053): r*r in float; /.25, 6.0+ and 6.0/ in double; narrowed. */
DVO_DEV float weight_of(float r) {
    return (float)(6.0 / (6.0 + (double)(r * r) / .25));
}

/* One reference edge point through :328-345 (warp + project).  Returns visibility
 * (half-open bounds, false for NaN -- SURVEY Q3). */
DVO_DEV bool project_point(const IterConst &c, float X, float Y, float Z,
                           float &xn, float &yn, float &zn, float &u, float &v) {
    const float d0 = X - c.t[0], d1 = Y - c.t[1], d2 = Z - c.t[2];        /* _3d - cTRep */
    /* cR^T * d : row i of cR^T is column i of cR */
    const float p0 = (c.r[0] * d0 + c.r[1] * d1) + c.r[2] * d2;
    const float p1 = (c.r[3] * d0 + c.r[4] * d1) + c.r[5] * d2;
    const float p2 = (c.r[6] * d0 + c.r[7] * d1) + c.r[8] * d2;
    const float inv = 1.0f / p2;                                           /* :339 */
    xn = p0 * inv; yn = p1 * inv; zn = p2 * inv;                            /* :340-341 */
    u = c.m00 * xn + c.m02 * zn;                                            /* :344 */
    v = c.m11 * yn + c.m12 * zn;
    return (u >= 0.0f) && (u < c.ncols_f) && (v >= 0.0f) && (v < c.nrows_f);
}

/* Jacobian row from the gathered gradient (:379-406).  X,Y,Z are the
 * DEHOMOGENISED coordinates (quirk Q1), cR^T is applied a second time (Q2). */
DVO_DEV void jacobian_row(const IterConst &c, float xn, float yn, float zn,
                          float gxv, float gyv, float *J) {
    const float zz = zn * zn;
    const float a00 = c.m00 / zn;                      /* scaleFac*fx/Z            :388 */
    const float a02 = ((-c.m00) * xn) / zz;            /* -scaleFac*fx*X/(Z*Z)     :390 */
    const float a11 = c.m11 / zn;                      /* :392 */
    const float a12 = ((-c.m11) * yn) / zz;            /* :393 */
    const float ga0 = gxv * a00;                       /* G*A1, structural zeros dropped */
    const float ga1 = gyv * a11;
    const float ga2 = gxv * a02 + gyv * a12;
    /* tmp = cR^T * (xn,yn,zn)   :399 */
    const float w0 = (c.r[0] * xn + c.r[1] * yn) + c.r[2] * zn;
    const float w1 = (c.r[3] * xn + c.r[4] * yn) + c.r[5] * zn;
    const float w2 = (c.r[6] * xn + c.r[7] * yn) + c.r[8] * zn;
    /* columns 0..2 of A2 are -cR^T: A2(i,k) = -cR(k,i) = -r[k+3i]  (:397) */
    J[0] = -((ga0 * c.r[0] + ga1 * c.r[3]) + ga2 * c.r[6]);
    J[1] = -((ga0 * c.r[1] + ga1 * c.r[4]) + ga2 * c.r[7]);
    J[2] = -((ga0 * c.r[2] + ga1 * c.r[5]) + ga2 * c.r[8]);
    /* columns 3..5 are to_se_3(tmp) (:401-402, :1104-1114) */
    J[3] = ga1 * w2 - ga2 * w1;
    J[4] = ga2 * w0 - ga0 * w2;
    J[5] = ga0 * w1 - ga1 * w0;
}

/* texel = {DT, dDT/dx, dDT/dy, 0} of the now level; index of pixel (yy,xx) */
DVO_DEV int texel_index(int yy, int xx, int rows) { return yy + xx * rows; }

DVO_DEV PointEval eval_point(const IterConst &c, const float4 *__restrict__ tex,
                             float X, float Y, float Z) {
    PointEval o;
    float xn, yn;
    o.vis = project_point(c, X, Y, Z, xn, yn, o.zn, o.u, o.v);
    o.eps = 0.0f; o.w = 0.0f;
#pragma unroll
    for (int k = 0; k < 6; k++) o.J[k] = 0.0f;
    if (o.vis) {
        const int xx = (int)o.u, yy = (int)o.v;         /* :376-377 == floor for u,v >= 0 (:446) */
        const float4 tx = tex[texel_index(yy, xx, c.rows)];
        jacobian_row(c, xn, yn, o.zn, tx.y, tx.z, o.J);
        o.eps = tx.x;
        o.w = weight_of(tx.x);
    }
    return o;
}

/* ------------------------------------------------------------------------- */
/*  double-precision 3x3 / SE(3) helpers (single lane)                        */
/* ------------------------------------------------------------------------- */
DVO_DEV void m3_mul(const double *A, const double *B, double *C) {
    double tmp[9];
#pragma unroll
    for (int j = 0; j < 3; j++)
#pragma unroll
        for (int i = 0; i < 3; i++)
            tmp[i + 3 * j] = (A[i] * B[3 * j] + A[i + 3] * B[1 + 3 * j]) + A[i + 6] * B[2 + 3 * j];
#pragma unroll
    for (int k = 0; k < 9; k++) C[k] = tmp[k];
}
DVO_DEV void m3_vec(const double *A, const double *x, double *y) {
    const double y0 = (A[0] * x[0] + A[3] * x[1]) + A[6] * x[2];
    const double y1 = (A[1] * x[0] + A[4] * x[1]) + A[7] * x[2];
    const double y2 = (A[2] * x[0] + A[5] * x[1]) + A[8] * x[2];
    y[0] = y0; y[1] = y1; y[2] = y2;
}
DVO_DEV void hat3(const double *w, double *W) {          /* to_se_3 / SO3::hat */
    W[0] = 0.0;   W[3] = -w[2]; W[6] = w[1];
    W[1] = w[2];  W[4] = 0.0;   W[7] = -w[0];
    W[2] = -w[1]; W[5] = w[0];  W[8] = 0.0;
}
DVO_DEV double norm6(const double *v) {
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < 6; k++) s += v[k] * v[k];
    return sqrt(s);
}

#define DVO_SOPHUS_EPS 1e-10

/* unit quaternion (w,x,y,z) of a rotation matrix: what Sophus' setRotationMatrix
 * does through Eigen::Quaterniond(R) followed by normalisation (:737). */
DVO_DEV void quat_of_matrix(const double *m, double *q) {
    double t = m[0] + m[4] + m[8];
    if (t > 0.0) {
        t = sqrt(t + 1.0);
        q[0] = 0.5 * t;
        t = 0.5 / t;
        q[1] = (m[5] - m[7]) * t;      /* m(2,1)-m(1,2) */
        q[2] = (m[6] - m[2]) * t;      /* m(0,2)-m(2,0) */
        q[3] = (m[1] - m[3]) * t;      /* m(1,0)-m(0,1) */
    } else {
        int i = 0;
        if (m[4] > m[0]) i = 1;
        if (m[8] > m[i * 4]) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        t = sqrt(m[i * 4] - m[j * 4] - m[k * 4] + 1.0);
        double qi = 0.5 * t;
        t = 0.5 / t;
        const double qw = (m[k + 3 * j] - m[j + 3 * k]) * t;
        const double qj = (m[j + 3 * i] + m[i + 3 * j]) * t;
        const double qk = (m[k + 3 * i] + m[i + 3 * k]) * t;
        q[0] = qw;
        q[1] = (i == 0) ? qi : ((j == 0) ? qj : qk);
        q[2] = (i == 1) ? qi : ((j == 1) ? qj : qk);
        q[3] = (i == 2) ? qi : ((j == 2) ? qj : qk);
    }
    const double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    q[0] /= n; q[1] /= n; q[2] /= n; q[3] /= n;
}

/* SE(3) logarithm, tangent order [upsilon(3), omega(3)] like Sophus::SE3d::log. */
DVO_DEV_NOINLINE void se3_log(const double *R, const double *t, double *psi) {
    double q[4];
    quat_of_matrix(R, q);
    const double squared_n = q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
    const double n = sqrt(squared_n);
    const double w = q[0];
    double k2;                               /* 2*atan(n/w)/n */
    if (n < DVO_SOPHUS_EPS) {
        k2 = 2.0 / w - 2.0 * squared_n / (w * (w * w));
    } else if (fabs(w) < DVO_SOPHUS_EPS) {
        k2 = (w > 0.0) ? (M_PI / n) : (-M_PI / n);
    } else {
        k2 = 2.0 * atan(n / w) / n;
    }
    const double theta = k2 * n;
    double om[3] = {k2 * q[1], k2 * q[2], k2 * q[3]};
    double W[9], W2[9];
    hat3(om, W);
    m3_mul(W, W, W2);
    double c;
    if (fabs(theta) < DVO_SOPHUS_EPS) c = 1. / 12.;
    else c = (1.0 - theta / (2.0 * tan(theta / 2.0))) / (theta * theta);
    double Vi[9];
#pragma unroll
    for (int k = 0; k < 9; k++) Vi[k] = (((k % 4 == 0) ? 1.0 : 0.0) - 0.5 * W[k]) + c * W2[k];
    m3_vec(Vi, t, psi);
    psi[3] = om[0]; psi[4] = om[1]; psi[5] = om[2];
}

/* SE(3) exponential, Sophus::SE3d::exp: quaternion from the half angle, then
 * V = I + (1-cos)/th^2 W + (th-sin)/th^3 W^2. */
DVO_DEV_NOINLINE void se3_exp(const double *psi, double *R, double *t) {
    const double *om = psi + 3;
    const double theta_sq = om[0] * om[0] + om[1] * om[1] + om[2] * om[2];
    const double theta = sqrt(theta_sq);
    const double half_theta = 0.5 * theta;
    double imag, real;
    if (theta < DVO_SOPHUS_EPS) {
        const double theta_po4 = theta_sq * theta_sq;
        imag = 0.5 - (1.0 / 48.0) * theta_sq + (1.0 / 3840.0) * theta_po4;
        real = 1.0 - 0.5 * theta_sq + (1.0 / 384.0) * theta_po4;
    } else {
        imag = sin(half_theta) / theta;
        real = cos(half_theta);
    }
    double qw = real, qx = imag * om[0], qy = imag * om[1], qz = imag * om[2];
    const double qn = sqrt(qw * qw + qx * qx + qy * qy + qz * qz);
    qw /= qn; qx /= qn; qy /= qn; qz /= qn;
    /* Eigen::Quaternion::toRotationMatrix */
    const double tx = 2.0 * qx, ty = 2.0 * qy, tz = 2.0 * qz;
    const double twx = tx * qw, twy = ty * qw, twz = tz * qw;
    const double txx = tx * qx, txy = ty * qx, txz = tz * qx;
    const double tyy = ty * qy, tyz = tz * qy, tzz = tz * qz;
    R[0] = 1.0 - (tyy + tzz); R[3] = txy - twz;         R[6] = txz + twy;
    R[1] = txy + twz;         R[4] = 1.0 - (txx + tzz); R[7] = tyz - twx;
    R[2] = txz - twy;         R[5] = tyz + twx;         R[8] = 1.0 - (txx + tyy);
    double W[9], W2[9], V[9];
    hat3(om, W);
    m3_mul(W, W, W2);
    if (theta < DVO_SOPHUS_EPS) {
#pragma unroll
        for (int k = 0; k < 9; k++) V[k] = R[k];
    } else {
        const double a = (1.0 - cos(theta)) / theta_sq;
        const double b = (theta - sin(theta)) / (theta_sq * theta);
#pragma unroll
        for (int k = 0; k < 9; k++) V[k] = (((k % 4 == 0) ? 1.0 : 0.0) + a * W[k]) + b * W2[k];
    }
    m3_vec(V, psi, t);
}

/* rotationize (:1269-1282): R <- U V^T of R = U S V^T, i.e. the orthogonal polar
 * factor.  The reference gets it from a Jacobi SVD; the polar factor is unique
 * for a non-singular matrix, so a scaled Newton iteration X <- (g X + X^-T / g)/2
 * converges to the same matrix (quadratically; 1-2 steps for the nearly
 * orthogonal products that reach this function). */
DVO_DEV_NOINLINE void rotationize(double *R) {
    double X[9];
#pragma unroll
    for (int k = 0; k < 9; k++) X[k] = R[k];
    for (int it = 0; it < 32; it++) {
        /* cofactor matrix, C(i,j) at C[i+3j]  (= det * X^-T) */
        double C[9];
        C[0] = X[4] * X[8] - X[7] * X[5];
        C[1] = X[6] * X[5] - X[3] * X[8];
        C[2] = X[3] * X[7] - X[6] * X[4];
        C[3] = X[7] * X[2] - X[1] * X[8];
        C[4] = X[0] * X[8] - X[6] * X[2];
        C[5] = X[6] * X[1] - X[0] * X[7];
        C[6] = X[1] * X[5] - X[4] * X[2];
        C[7] = X[3] * X[2] - X[0] * X[5];
        C[8] = X[0] * X[4] - X[3] * X[1];
        const double det = X[0] * C[0] + X[1] * C[1] + X[2] * C[2];
        if (det == 0.0 || !(det == det)) break;          /* singular / NaN: leave as is */
        const double idet = 1.0 / det;
        /* Frobenius-norm scaling (Higham) */
        double nx = 0.0, ny = 0.0;
#pragma unroll
        for (int k = 0; k < 9; k++) { nx += X[k] * X[k]; const double y = C[k] * idet; ny += y * y; }
        const double gam = sqrt(sqrt(ny / nx));
        double diff = 0.0;
#pragma unroll
        for (int k = 0; k < 9; k++) {
            const double xn_ = 0.5 * (gam * X[k] + (C[k] * idet) / gam);
            const double d = xn_ - X[k];
            diff += d * d;
            X[k] = xn_;
        }
        if (diff <= 1e-30 * nx) break;
    }
#pragma unroll
    for (int k = 0; k < 9; k++) R[k] = X[k];
}

/* Optimiser state of one runIterations call (lives in LDS, touched by one lane). */
struct PoseState {
    double R[9], t[3];           /* cR, cT */
    double d[6];                 /* descentDirection (:654) */
    double bestR[9], bestT[3];   /* :646-647 */
    float bestE, bestRatio;      /* :644-645 */
    int bestItr;                 /* :648 */
    int stop;
    float Rf[9], tf[3];          /* cR_32, cT_32 (:673-674) for the next evaluation */
};

DVO_DEV void pose_state_begin(PoseState &s) {
#pragma unroll
    for (int k = 0; k < 6; k++) s.d[k] = 0.0;
#pragma unroll
    for (int k = 0; k < 9; k++) s.bestR[k] = (k % 4 == 0) ? 1.0 : 0.0;
    s.bestT[0] = s.bestT[1] = s.bestT[2] = 0.0;
    s.bestE = 1.0E10f;
    s.bestRatio = 1.0f;
    s.bestItr = -1;
    s.stop = 0;
#pragma unroll
    for (int k = 0; k < 9; k++) s.Rf[k] = (float)s.R[k];
#pragma unroll
    for (int k = 0; k < 3; k++) s.tf[k] = (float)s.t[k];
}

/* Everything runIterations does after the per-point phase of iteration `itr`
 * (:689-920).  g = J^T W eps (:777), sum_eps2 = sum eps^2, n_vis visible points.
 * Returns the energy; sets s.stop on early termination. */
DVO_DEV_NOINLINE float pose_update(PoseState &s, const DevParams &prm, int itr, int N,
                          const double *g_in, double sum_eps2, int n_vis) {
    const float energy = (float)sqrt(sum_eps2);                          /* :689, :1312 */
    if (energy <= s.bestE) {                                              /* :696 */
        s.bestE = energy;
        s.bestRatio = (float)n_vis / (float)N;                            /* :457 */
#pragma unroll
        for (int k = 0; k < 9; k++) s.bestR[k] = s.R[k];
#pragma unroll
        for (int k = 0; k < 3; k++) s.bestT[k] = s.t[k];
        s.bestItr = itr;
    }
    double g[6];
#pragma unroll
    for (int k = 0; k < 6; k++) g[k] = g_in[k];
    if (prm.enable_l2_reg) {                                              /* :734-743, :796 */
        double cpsi[6];
        se3_log(s.R, s.t, cpsi);
        const double n = norm6(cpsi);
        if (n > 0.0) {
#pragma unroll
            for (int k = 0; k < 6; k++) cpsi[k] = cpsi[k] / n;
        }
#pragma unroll
        for (int k = 0; k < 6; k++) g[k] += prm.reg_lambda * cpsi[k];
    }
    const double step = prm.step_a * prm.step_b /
                        ((itr > prm.step_decay_after) ? (double)(itr - prm.step_decay_offset) : 1.0);  /* :773 */
    double psi[6];
#pragma unroll
    for (int k = 0; k < 6; k++) {
        s.d[k] = (1.0 - prm.beta) * g[k] + prm.beta * s.d[k];             /* :799 */
        const double pk = (k < 3) ? 1.0 : prm.precond_rot;                /* :729 */
        psi[k] = ((-step) * pk) * s.d[k];                                 /* :821 */
    }
    const double nrm = norm6(psi);                                        /* :832 */
    if (nrm > prm.trust_radius) {                                         /* :835 */
#pragma unroll
        for (int k = 0; k < 6; k++) psi[k] = psi[k] / nrm * prm.trust_radius;   /* :837 */
    }
    if (norm6(psi) < prm.psi_norm_stop) {                                 /* :872 */
        s.stop = 1;
        return energy;
    }
    double xR[9], xT[3], dT[3];
    se3_exp(psi, xR, xT);                                                 /* :905-907 */
    m3_vec(s.R, xT, dT);
    s.t[0] += dT[0]; s.t[1] += dT[1]; s.t[2] += dT[2];                    /* :916 */
    double nR[9];
    m3_mul(s.R, xR, nR);                                                  /* :917 */
    if (prm.enable_rotationize) rotationize(nR);                          /* :919 */
#pragma unroll
    for (int k = 0; k < 9; k++) { s.R[k] = nR[k]; s.Rf[k] = (float)nR[k]; }
#pragma unroll
    for (int k = 0; k < 3; k++) s.tf[k] = (float)s.t[k];
    return energy;
}

}  // namespace dvo
#endif
