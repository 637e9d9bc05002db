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

namespace dvo {

/* Device copy of dvo_params (doubles widened on the host exactly like the
 * reference widens its float members at use, SolveDVO.cpp:835,872). */
struct DevParams {
    double beta, precond_rot, reg_lambda, step_a, step_b;
    double trust_radius, psi_norm_stop;
    int step_decay_after, step_decay_offset;
    int enable_rotationize, enable_l2_reg;
    int interpolate_dt;
};

/* Per-iteration, per-level constants: float pose (cast at :673-674) and the
 * non-zero entries of M = diag(s,s,1)*K (:334-337,344).  Wave-uniform. */
struct IterConst {
    float r[9];      /* cR, column-major */
    float t[3];      /* cT */
    float m00, m02, m11, m12;   /* s*fx, s*cx, s*fy, s*cy  (float products) */
    float m00_z1, m11_z1;       /* m00/(1-2^-24), m11/(1-2^-24): A1(0,0), A1(1,1) when Z = 1-2^-24 */
    float ncols_f, nrows_f;
    int rows, cols;
    int tiles_per_col;          /* texel tiles along yy (see texel_index) */
    int interp;                 /* __INTERPOLATE_DISTANCE_TRANSFORM (SolveDVO.h:97): eps from interpolate() */
    float pcx, pfx, pcy, pfy;   /* tmpcx, tmpfx, tmpcy, tmpfy of enlistRefEdgePts (:232-235): rebuild X, Y of compact points */
    unsigned nby;               /* 16-row blocks per block column of the level: 4-byte points (pt4_decode) */
    float inv_nby, half_inv_nby;
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

/* ---- exact float divisions without the IEEE division sequence --------------------------------
 * The five float divisions per point of the reference (:339, :388-393) dominate the instruction
 * count of the per-point loop.  All of them can be produced bit-exactly with a few cheap
 * instructions; every identity below is verified over ALL 2^32 bit patterns on the GPU by
 * tools/exhaustive/div_tricks.hip (run by tests/test_gpu_exact_division.py):
 *   (1) 1.0f/x  ==  rcp(x) refined by one fma Newton step, for 2^-126 <= |x| <= 2^126
 *   (2) for such x,  zn = x * (1.0f/x)  is exactly 1 or 1-2^-24            (quirk Q1's "Z")
 *   (3) n / (1-2^-24)  ==  the next float away from zero (n normal); n itself for 0/subnormal/inf/nan
 *   (4) n / (1-2^-23)  ==  n advanced by 1 ulp if its 24-bit significand <= 12582910, else by 2
 *                          (subnormals: by 1 if the mantissa >= 2^22; overflow saturates to inf)
 *   and (1-2^-24)^2 rounds to 1-2^-23, so zn*zn is 1 or 1-2^-23.
 * Outside the proven range (|x| subnormal, huge, 0, inf, nan) the callers fall back to real divisions. */
DVO_DEV bool rcp_in_proven_range(float x) {
    const float ax = fabsf(x);
    return (ax >= 1.17549435e-38f) && (ax <= 8.50705917e37f);       /* 2^-126 .. 2^126 */
}
DVO_DEV float exact_rcp(float x) {                                    /* identity (1) */
    const float r = __builtin_amdgcn_rcpf(x);
    return __builtin_fmaf(__builtin_fmaf(-x, r, 1.0f), r, r);
}
DVO_DEV float exact_div_z1(float n) {                                 /* identity (3): n / (1-2^-24) */
    const unsigned b = __float_as_uint(n);
    const unsigned ex = b & 0x7f800000u;
    return ((ex != 0u) && (ex != 0x7f800000u)) ? __uint_as_float(b + 1u) : n;
}
DVO_DEV float exact_div_zz1(float n) {                                /* identity (4): n / (1-2^-23) */
    const unsigned b = __float_as_uint(n);
    const unsigned ex = b & 0x7f800000u;
    const unsigned man = b & 0x007fffffu;
    /* normal: significand (man | 2^23) <= 12582910  <=>  man <= 4194302 */
    unsigned inc = (man <= 4194302u) ? 1u : 2u;
    inc = (ex == 0u) ? ((man >= 0x00400000u) ? 1u : 0u) : inc;        /* zero / subnormal */
    inc = (ex == 0x7f800000u) ? 0u : inc;                             /* inf / nan unchanged */
    unsigned r = b + inc;
    r = ((r & 0x7fffffffu) > 0x7f800000u && ex != 0x7f800000u) ? ((b & 0x80000000u) | 0x7f800000u) : r;   /* +-FLT_MAX -> +-inf */
    return __uint_as_float(r);
}
#define DVO_Z1 0x3f7fffffu          /* bits of 1-2^-24 */

/* One reference edge point through :328-345 (warp + project).  Returns visibility
 * (half-open bounds, false for NaN -- SURVEY Q3). */
DVO_DEV bool project_point(const IterConst &c, float X, float Y, float Z,
                           float &xn, float &yn, float &zn, float &u, float &v) {
    const float d0 = X - c.t[0], d1 = Y - c.t[1], d2 = Z - c.t[2];        /* _3d - cTRep */
    /* cR^T * d : row i of cR^T is column i of cR */
    const float p0 = (c.r[0] * d0 + c.r[1] * d1) + c.r[2] * d2;
    const float p1 = (c.r[3] * d0 + c.r[4] * d1) + c.r[5] * d2;
    const float p2 = (c.r[6] * d0 + c.r[7] * d1) + c.r[8] * d2;
    float inv = exact_rcp(p2);                                             /* :339, == 1.0f/p2 (identity 1) */
    const bool odd = !rcp_in_proven_range(p2);                             /* 0, subnormal, huge, nan: never in practice */
    if (__builtin_amdgcn_ballot_w64(odd) != 0ull) {                        /* wave-uniform branch: not if-converted */
        if (odd) inv = 1.0f / p2;
    }
    xn = p0 * inv; yn = p1 * inv; zn = p2 * inv;                            /* :340-341 */
    u = c.m00 * xn + c.m02 * zn;                                            /* :344 */
    v = c.m11 * yn + c.m12 * zn;
    return (u >= 0.0f) && (u < c.ncols_f) && (v >= 0.0f) && (v < c.nrows_f);
}

/* A compact reference point {xx | yy << 16, Z = depth/1000}: X and Y rebuilt with the very operations of
 * enlistRefEdgePts (:249-250), so the bits are those of the 3 x N float list. */
DVO_DEV void expand_compact(const IterConst &c, unsigned pk, float z, float &X, float &Y, float &Z) {
    const float xx = (float)(pk & 0xffffu), yy = (float)(pk >> 16);
    Z = z;
    X = Z * (xx - c.pcx) * c.pfx;                                  /* :249 */
    Y = Z * (yy - c.pcy) * c.pfy;                                  /* :250 */
}

/* A reference point in FOUR bytes (round 3): { pixel inside its 16 x 16 block: xx & 15 | (yy & 15) << 4 ; depth in whole
 * millimetres << 8 ; block index relative to the point's 64-point chunk << 24 }.  The compact lists are in block order, so a
 * chunk of 64 consecutive points spans few blocks; `L0` = linear block index (block column * nby + block row) of the chunk's
 * first point comes from a per-chunk header (one scalar load per wave and round).  Z = depth / 1000.0f (:248) is rebuilt with
 * a multiply and two fmas; like the pixel it is only trusted because the BUILDER decodes every point of a list with this very
 * function and compares the bits with the 8-byte form -- a list with one mismatch (fractional depths, depths beyond 65535 mm,
 * more than 255 blocks inside a chunk) keeps the 8-byte form.  Why: the fused kernel sits on the memory-request ceiling and the
 * per-iteration stream of the points that do not fit in LDS was 15 % (640x480) to 35 % (1920x1080) of its requests. */
DVO_DEV void pt4_decode(unsigned nby, float inv_nby, float half_inv_nby, unsigned w, unsigned L0, float &xxf, float &yyf, float &Z) {
    const unsigned L = L0 + (w >> 24);
    int bx;
    {   /* floor((L + 0.5) / nby): exact for L < 2^21 */
        const float t = __builtin_fmaf((float)L, inv_nby, half_inv_nby);
        asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(bx) : "v"(t));
    }
    const unsigned by = L - __umul24((unsigned)bx, nby);
    xxf = (float)(((unsigned)bx << 4) + (w & 15u));
    yyf = (float)((by << 4) + ((w >> 4) & 15u));
    const float d = (float)((w >> 8) & 0xffffu);
    const float q = d * 0.001f;
    const float r = __builtin_fmaf(-q, 1000.0f, d);
    Z = __builtin_fmaf(r, 0.001f, q);                     /* == d / 1000.0f for every d the builder let through */
}

/* Jacobian row from the gathered gradient (:379-406).  X,Y,Z are the
 * DEHOMOGENISED coordinates (quirk Q1), cR^T is applied a second time (Q2). */
DVO_DEV void jacobian_row(const IterConst &c, float xn, float yn, float zn,
                          float gxv, float gyv, float *J) {
    /* Z is z*(1/z): exactly 1 or 1-2^-24 for every point whose 1/z took the fast path; the four
     * divisions by Z and Z*Z then have closed forms (identities 3 and 4).  Anything else (a point
     * with a degenerate z) takes the literal divisions. */
    /* all four read before they are chosen from: `cond ? c.m00 : c.m00_z1` is a choice between two ADDRESSES to the compiler, which can
     * pin a slice of IterConst to scratch memory (it did when the exact energy sweep added a second user, round 6) */
    const float m00 = c.m00, m11 = c.m11, m00_z1 = c.m00_z1, m11_z1 = c.m11_z1;
    const float n02 = (-m00) * xn, n12 = (-m11) * yn;
    const unsigned zb = __float_as_uint(zn);
    const bool z_is_1 = (zb == 0x3f800000u);           /* Z == 1 */
    const bool z_is_z1 = (zb == DVO_Z1);               /* Z == 1-2^-24, Z*Z == 1-2^-23 */
    float a00 = z_is_1 ? m00 : m00_z1;                 /* scaleFac*fx/Z            :388 */
    float a11 = z_is_1 ? m11 : m11_z1;                 /* :392 */
    float a02 = z_is_1 ? n02 : exact_div_zz1(n02);     /* -scaleFac*fx*X/(Z*Z)     :390 */
    float a12 = z_is_1 ? n12 : exact_div_zz1(n12);     /* :393 */
    const bool odd = !(z_is_1 || z_is_z1);
    if (__builtin_amdgcn_ballot_w64(odd) != 0ull) {    /* wave-uniform; literal divisions for degenerate z */
        if (odd) {
            const float zz = zn * zn;
            a00 = m00 / zn; a02 = n02 / zz; a11 = m11 / zn; a12 = n12 / zz;
        }
    }
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

/* texel = {DT, dDT/dx, dDT/dy, getWeightOf(DT)} of the now level (16 B): the weight (:1047-1053) is a
 * pure function of the pixel's DT value, so it is evaluated once per pixel when the level is
 * packed (same double-precision formula, weight_of above) instead of once per point per iteration.  Texels are stored in
 * tiles of DVO_TILE_Y x DVO_TILE_X pixels (yy fastest inside a tile, tiles in
 * column-major order), so that one 64/128-byte memory request covers a 2-D
 * patch: reprojected contour points that are neighbours in either direction
 * then share requests.  1x1 = the reference's plain column-major layout. */
#ifndef DVO_TILE_Y_LOG2
#define DVO_TILE_Y_LOG2 2
#endif
#ifndef DVO_TILE_X_LOG2
#define DVO_TILE_X_LOG2 1
#endif
#define DVO_TILE_Y (1 << DVO_TILE_Y_LOG2)
#define DVO_TILE_X (1 << DVO_TILE_X_LOG2)
__host__ __device__ inline int texel_tiles_per_col(int rows) { return (rows + DVO_TILE_Y - 1) >> DVO_TILE_Y_LOG2; }
__host__ __device__ inline size_t texel_count(int rows, int cols) {
    return (size_t)texel_tiles_per_col(rows) * (size_t)((cols + DVO_TILE_X - 1) >> DVO_TILE_X_LOG2) *
           (size_t)(DVO_TILE_Y * DVO_TILE_X);
}
__host__ __device__ inline int texel_index(int yy, int xx, int tiles_per_col) {
    const int tile = (xx >> DVO_TILE_X_LOG2) * tiles_per_col + (yy >> DVO_TILE_Y_LOG2);
    return (tile << (DVO_TILE_Y_LOG2 + DVO_TILE_X_LOG2)) + ((xx & (DVO_TILE_X - 1)) << DVO_TILE_Y_LOG2) +
           (yy & (DVO_TILE_Y - 1));
}

/* SolveDVO::interpolate (:1285-1308), the reference's optional residual lookup (call site :443-444,
 * compiled out by default): a "bilinear" interpolation of SQUARES, sqrt((1-a) F0^2 + a F1^2), along x for
 * the floor and ceil rows, then along y.  Only eps uses it; the gradient lookup stays nearest (:376-385).
 * The reference would index one past the end when ceil() reaches rows/cols; clamped like the oracle. */
DVO_DEV float interpolate_dt(const IterConst &c, const float4 *__restrict__ tex, float ry, float rx) {
    const int ry_d = (int)floor((double)ry), rx_d = (int)floor((double)rx);
    int ry_u = (int)ceil((double)ry), rx_u = (int)ceil((double)rx);
    const float inc_x = rx - (float)rx_d, inc_y = ry - (float)ry_d;
    if (ry_u > c.rows - 1) ry_u = c.rows - 1;
    if (rx_u > c.cols - 1) rx_u = c.cols - 1;
    const float f00 = tex[texel_index(ry_d, rx_d, c.tiles_per_col)].x, f01 = tex[texel_index(ry_d, rx_u, c.tiles_per_col)].x;
    const float f10 = tex[texel_index(ry_u, rx_d, c.tiles_per_col)].x, f11 = tex[texel_index(ry_u, rx_u, c.tiles_per_col)].x;
    const float a = (1.0f - inc_x) * f00 * f00 + (inc_x) * f01 * f01;
    const float f_d = (float)sqrt((double)a);
    const float b = (1.0f - inc_x) * f10 * f10 + (inc_x) * f11 * f11;
    const float f_u = (float)sqrt((double)b);
    const float cc = (1.0f - inc_y) * f_d * f_d + inc_y * f_u * f_u;
    return (float)sqrt((double)cc);
}

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
        const float4 tx = tex[texel_index(yy, xx, c.tiles_per_col)];
        jacobian_row(c, xn, yn, o.zn, tx.y, tx.z, o.J);
        o.eps = tx.x;
        o.w = tx.w;             /* getWeightOf(eps), evaluated once per pixel when the texel is packed */
        if (c.interp) {         /* :443-444 */
            o.eps = interpolate_dt(c, tex, o.v, o.u);
            o.w = weight_of(o.eps);
        }
    }
    return o;
}

/* ------------------------------------------------------------------------- */
/*  double-precision helpers of the 6-DoF update (one lane per workgroup)      */
/*                                                                             */
/*  This lane is the serial section of every iteration, so its instruction     */
/*  count is the latency floor of the whole alignment.  Everything here is     */
/*  inlined, keeps its operands in registers and avoids the generic IEEE       */
/*  division / ocml transcendental sequences: reciprocal and square root are   */
/*  hardware seeds + Newton/Goldschmidt steps (<= ~1 ulp), sin/cos/atan are    */
/*  short series after an exact-enough argument reduction (~1e-16 relative).   */
/*  Parity with the CPU oracle does not need bit-equality here: the pose is    */
/*  narrowed to float before it touches the per-point math (:673-674), and     */
/*  1e-16-level differences change that cast with probability ~1e-9.           */
/* ------------------------------------------------------------------------- */
DVO_DEV double d_rcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
}
DVO_DEV double d_div(double a, double b) {
    const double r = d_rcp(b);
    const double q = a * r;
    return fma(fma(-b, q, a), r, q);
}
DVO_DEV double d_sqrt(double x) {            /* x >= 0, normal range */
    if (x == 0.0) return 0.0;
    double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = 0.5 * y;
    double r = fma(-h, g, 0.5);
    g = fma(g, r, g); h = fma(h, r, h);
    r = fma(-h, g, 0.5);
    g = fma(g, r, g); h = fma(h, r, h);
    return fma(fma(-g, g, x), h, g);
}
/* Double constants of the update's polynomials.  Left to itself the compiler hoists every literal of the update into vector
 * registers for the whole kernel (two dozen register pairs pinned across the point loops, the surplus spilled to scratch and
 * re-loaded inside the serial chain).  DVO_K(c) materialises c where it is used, in a scalar register pair (two s_mov_b32 that
 * the optimiser may not move); make EXP=vconst EXPDEFS=-DDVO_CONST_VGPR=1 builds the plain-literal form for the A/B (DESIGN.md
 * section 6).  With it no fused kernel of the 256- or 512-thread shapes touches scratch any more. */
#ifndef DVO_CONST_VGPR
template <unsigned long long BITS> DVO_DEV double kconst_sgpr() {
    unsigned lo, hi;
    asm volatile("s_mov_b32 %0, %2\n\ts_mov_b32 %1, %3" : "=s"(lo), "=s"(hi) : "i"((unsigned)BITS), "i"((unsigned)(BITS >> 32)));
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
#define DVO_K(c) kconst_sgpr<__builtin_bit_cast(unsigned long long, (double)(c))>()
#else
#define DVO_K(c) (c)
#endif
/* 1/sqrt(x), x > 0 in the normal range: hardware seed + two Newton steps y <- y + y (1 - x y^2) / 2 (<= ~1 ulp).  Nine
 * instructions where d_rcp(d_sqrt(x)) takes fifteen: the update is one lane's serial instruction stream (round 5). */
DVO_DEV double d_rsqrt(double x) {
    double y = __builtin_amdgcn_rsq(x);
    double e = fma(-(x * y), y, 1.0);
    y = fma(0.5 * y, e, y);
    e = fma(-(x * y), y, 1.0);
    return fma(0.5 * y, e, y);
}
/* 1/sqrt(n2) for the squared norm of a quaternion that is unit up to rounding (|n2 - 1| ~ 1e-15: a product of two unit
 * quaternions, or {cos, sin * axis}): first order, error 3/8 (n2 - 1)^2 < 1e-30.  Two instructions instead of fifteen. */
DVO_DEV double unit_rnorm(double n2) { return fma(-0.5, n2 - 1.0, 1.0); }
/* sin and cos of |x| <= ~2*pi : quadrant reduction with a two-part pi/2, then
 * Taylor series on [-pi/4, pi/4] (terms below 1e-19). */
DVO_DEV void d_sincos(double x, double &s, double &c) {
    const double k = rint(x * DVO_K(0.63661977236758134308));          /* 2/pi */
    double r = fma(-k, DVO_K(1.57079632679489655800e+00), x);          /* pi/2 hi */
    r = fma(-k, DVO_K(6.12323399573676603587e-17), r);                 /* pi/2 lo */
    const double z = r * r;
    double ps = DVO_K(1.0 / 355687428096000.0);                        /* 1/17! */
    ps = fma(ps, z, DVO_K(-1.0 / 1307674368000.0));                    /* 1/15! */
    ps = fma(ps, z, DVO_K(1.0 / 6227020800.0));                        /* 1/13! */
    ps = fma(ps, z, DVO_K(-1.0 / 39916800.0));                         /* 1/11! */
    ps = fma(ps, z, DVO_K(1.0 / 362880.0));                            /* 1/9!  */
    ps = fma(ps, z, DVO_K(-1.0 / 5040.0));                             /* 1/7!  */
    ps = fma(ps, z, DVO_K(1.0 / 120.0));                               /* 1/5!  */
    ps = fma(ps, z, DVO_K(-1.0 / 6.0));                                /* 1/3!  */
    const double sr = fma(ps * z, r, r);
    double pc = DVO_K(-1.0 / 6402373705728000.0);                      /* 1/18! */
    pc = fma(pc, z, DVO_K(1.0 / 20922789888000.0));                    /* 1/16! */
    pc = fma(pc, z, DVO_K(-1.0 / 87178291200.0));                      /* 1/14! */
    pc = fma(pc, z, DVO_K(1.0 / 479001600.0));                         /* 1/12! */
    pc = fma(pc, z, DVO_K(-1.0 / 3628800.0));                          /* 1/10! */
    pc = fma(pc, z, DVO_K(1.0 / 40320.0));                             /* 1/8!  */
    pc = fma(pc, z, DVO_K(-1.0 / 720.0));                              /* 1/6!  */
    pc = fma(pc, z, DVO_K(1.0 / 24.0));                                /* 1/4!  */
    pc = fma(pc, z, -0.5);
    const double cr = fma(pc, z, 1.0);
    const int q = ((int)k) & 3;
    const double s0 = (q & 1) ? cr : sr;
    const double c0 = (q & 1) ? sr : cr;
    s = (q & 2) ? -s0 : s0;
    c = ((q == 1) || (q == 2)) ? -c0 : c0;
}
/* atan(x) for any finite x: |x|>1 -> pi/2 - atan(1/|x|); two half-angle steps
 * atan(y) = 2 atan(y / (1 + sqrt(1+y^2))) bring |y| <= tan(pi/16); 12-term series. */
DVO_DEV double d_atan_series(double y) {        /* |y| <= 0.25: |y|^31/31 < 1e-20 */
    const double z = y * y;
    double p = DVO_K(1.0 / 29.0);
    p = fma(p, z, DVO_K(-1.0 / 27.0));
    p = fma(p, z, DVO_K(1.0 / 25.0));
    p = fma(p, z, DVO_K(-1.0 / 23.0));
    p = fma(p, z, DVO_K(1.0 / 21.0));
    p = fma(p, z, DVO_K(-1.0 / 19.0));
    p = fma(p, z, DVO_K(1.0 / 17.0));
    p = fma(p, z, DVO_K(-1.0 / 15.0));
    p = fma(p, z, DVO_K(1.0 / 13.0));
    p = fma(p, z, DVO_K(-1.0 / 11.0));
    p = fma(p, z, DVO_K(1.0 / 9.0));
    p = fma(p, z, DVO_K(-1.0 / 7.0));
    p = fma(p, z, DVO_K(1.0 / 5.0));
    p = fma(p, z, DVO_K(-1.0 / 3.0));
    return fma(p * z, y, y);
}
DVO_DEV double d_atan(double x) {
    if (fabs(x) <= 0.25) return d_atan_series(x);        /* plain series (the usual case) */
    const double ax = fabs(x);
    const bool inv = ax > 1.0;
    double y = inv ? d_rcp(ax) : ax;
    y = d_div(y, 1.0 + d_sqrt(fma(y, y, 1.0)));
    y = d_div(y, 1.0 + d_sqrt(fma(y, y, 1.0)));
    const double z = y * y;
    double p = DVO_K(-1.0 / 23.0);
    p = fma(p, z, DVO_K(1.0 / 21.0));
    p = fma(p, z, DVO_K(-1.0 / 19.0));
    p = fma(p, z, DVO_K(1.0 / 17.0));
    p = fma(p, z, DVO_K(-1.0 / 15.0));
    p = fma(p, z, DVO_K(1.0 / 13.0));
    p = fma(p, z, DVO_K(-1.0 / 11.0));
    p = fma(p, z, DVO_K(1.0 / 9.0));
    p = fma(p, z, DVO_K(-1.0 / 7.0));
    p = fma(p, z, DVO_K(1.0 / 5.0));
    p = fma(p, z, DVO_K(-1.0 / 3.0));
    double a = 4.0 * fma(p * z, y, y);
    if (inv) a = DVO_K(1.57079632679489655800e+00) - a + DVO_K(6.12323399573676603587e-17);
    return (x < 0.0) ? -a : a;
}

DVO_DEV void m3_mul(const double *A, const double *B, double *C) {
    double tmp[9];
#pragma unroll
    for (int j = 0; j < 3; j++)
#pragma unroll
        for (int i = 0; i < 3; i++)
            tmp[i + 3 * j] = fma(A[i + 6], B[2 + 3 * j], fma(A[i + 3], B[1 + 3 * j], A[i] * B[3 * j]));
#pragma unroll
    for (int k = 0; k < 9; k++) C[k] = tmp[k];
}
DVO_DEV void m3_vec(const double *A, const double *x, double *y) {
    const double y0 = fma(A[6], x[2], fma(A[3], x[1], A[0] * x[0]));
    const double y1 = fma(A[7], x[2], fma(A[4], x[1], A[1] * x[0]));
    const double y2 = fma(A[8], x[2], fma(A[5], x[1], A[2] * x[0]));
    y[0] = y0; y[1] = y1; y[2] = y2;
}
/* y = (I + a*hat(w) + b*hat(w)^2) x   without forming the matrices:
 * hat(w) x = w cross x ;  hat(w)^2 x = w (w.x) - (w.w) x */
DVO_DEV void apply_I_aW_bW2(const double *w, double a, double b, const double *x, double *y) {
    const double c0 = fma(w[1], x[2], -(w[2] * x[1]));
    const double c1 = fma(w[2], x[0], -(w[0] * x[2]));
    const double c2 = fma(w[0], x[1], -(w[1] * x[0]));
    const double wx = fma(w[2], x[2], fma(w[1], x[1], w[0] * x[0]));
    const double ww = fma(w[2], w[2], fma(w[1], w[1], w[0] * w[0]));
    y[0] = fma(b, fma(w[0], wx, -ww * x[0]), fma(a, c0, x[0]));
    y[1] = fma(b, fma(w[1], wx, -ww * x[1]), fma(a, c1, x[1]));
    y[2] = fma(b, fma(w[2], wx, -ww * x[2]), fma(a, c2, x[2]));
}
DVO_DEV double norm6(const double *v) {
    double s = v[0] * v[0];
#pragma unroll
    for (int k = 1; k < 6; k++) s = fma(v[k], v[k], s);
    return d_sqrt(s);
}

/* ---- constants of the update, as a block in LDS (round 5) ------------------------------------------------------------------
 * The update is ONE wave's serial instruction stream, and a wave alone issues one instruction every 4-5 cycles whatever it is.
 * A polynomial coefficient written as a literal costs such a stream up to six issue slots (two s_mov_b32, two v_mov_b32 and the
 * v_fmac that wants its addend in a vector register) -- or, hoisted by the compiler, a register pair pinned for the whole kernel
 * (the scratch traffic of rounds 2-4's fused kernels was exactly that).  Kept in LDS, two coefficients arrive per ds_read_b128
 * in registers that the v_fmac may overwrite: 1.5 slots per Horner step.  The parameters of dvo_params the update reads are in
 * the same block, so that the kernels need not hold DevParams in scalar registers (they were spilled to vector-register lanes and
 * read back with v_readlane all over the serial chain).  Built once per kernel by upd_const_build(); kernels without LDS state
 * (single-lane helpers) build it on the stack, where it folds back into literals. */
struct __attribute__((aligned(16))) UpdConst {
    double omb, beta;            /* 1 - beta, beta (:799) */
    double ab, lambda;           /* step_a * step_b (:773), reg_lambda (:742) */
    double pk[6];                /* pre-conditioner diagonal (:724-730) */
    double tr, tr2;              /* trust radius (:25, widened like the reference does at :835) and its square */
    double stop2, eps2;          /* psi_norm_stop^2 (:24, :872); DVO_SOPHUS_EPS^2 */
    double ke[12];               /* se3_exp_q, theta < 0.01: sin(x)/x {3}, cos(x) {3}, (th - sin th)/th^3 {4}, threshold, spare */
    double kl[12];               /* se3_log_q, theta < 0.1: atan(y)/y {6}, c {5}, threshold */
    int decay_after, decay_offset, l2_reg, pad_;
};
DVO_DEV void upd_const_build(UpdConst &u, const DevParams &prm) {
    u.omb = 1.0 - prm.beta; u.beta = prm.beta;
    u.ab = prm.step_a * prm.step_b; u.lambda = prm.reg_lambda;
#pragma unroll
    for (int k = 0; k < 6; k++) u.pk[k] = (k < 3) ? 1.0 : prm.precond_rot;
    u.tr = prm.trust_radius; u.tr2 = prm.trust_radius * prm.trust_radius;
    u.stop2 = prm.psi_norm_stop * prm.psi_norm_stop; u.eps2 = 1e-10 * 1e-10;
    /* sin(x)/x = 1 - z/6 + z^2/120 - z^3/5040, z = x^2 <= 2.5e-5 (z^4/9! < 1e-24) */
    u.ke[0] = -1.0 / 5040.0; u.ke[1] = 1.0 / 120.0; u.ke[2] = -1.0 / 6.0;
    /* cos(x) = 1 - z/2 + z^2/24 - z^3/720 + z^4/40320 */
    u.ke[3] = 1.0 / 40320.0; u.ke[4] = -1.0 / 720.0; u.ke[5] = 1.0 / 24.0;
    /* (th - sin th)/th^3 = 1/6 - th^2/120 + th^4/5040 - th^6/362880 (th^8/11! < 3e-24) */
    u.ke[6] = -1.0 / 362880.0; u.ke[7] = 1.0 / 5040.0; u.ke[8] = -1.0 / 120.0; u.ke[9] = 1.0 / 6.0;
    u.ke[10] = 1e-4; u.ke[11] = 0.0;
    /* atan(y)/y = 1 - y^2/3 + ... + y^12/13, y^2 < 2.5e-3 (y^14/15 < 1e-19) */
    u.kl[0] = 1.0 / 13.0; u.kl[1] = -1.0 / 11.0; u.kl[2] = 1.0 / 9.0; u.kl[3] = -1.0 / 7.0; u.kl[4] = 1.0 / 5.0; u.kl[5] = -1.0 / 3.0;
    /* c = (1 - (th/2) cot(th/2)) / th^2 = 1/12 + th^2/720 + th^4/30240 + th^6/1209600 + th^8/47900160 */
    u.kl[6] = 1.0 / 47900160.0; u.kl[7] = 1.0 / 1209600.0; u.kl[8] = 1.0 / 30240.0; u.kl[9] = 1.0 / 720.0; u.kl[10] = 1.0 / 12.0;
    u.kl[11] = 2.5e-3;
    u.decay_after = prm.step_decay_after; u.decay_offset = prm.step_decay_offset;
    u.l2_reg = prm.enable_l2_reg; u.pad_ = 0;
}

#define DVO_SOPHUS_EPS 1e-10

/* unit quaternion (w,x,y,z) of a rotation matrix: what Sophus' setRotationMatrix
 * does through Eigen::Quaterniond(R) followed by normalisation (:737). */
DVO_DEV void quat_of_matrix(const double *m, double *q) {
    double t = m[0] + m[4] + m[8];
    if (t > 0.0) {
        t = d_sqrt(t + 1.0);
        q[0] = 0.5 * t;
        t = 0.5 * d_rcp(t);
        q[1] = (m[5] - m[7]) * t;      /* m(2,1)-m(1,2) */
        q[2] = (m[6] - m[2]) * t;      /* m(0,2)-m(2,0) */
        q[3] = (m[1] - m[3]) * t;      /* m(1,0)-m(0,1) */
    } else {
        /* largest diagonal element first (Eigen's branch for trace <= 0) */
        const bool i1 = m[4] > m[0];
        const bool i2 = m[8] > (i1 ? m[4] : m[0]);
        if (i2) {            /* i=2, j=0, k=1 */
            t = d_sqrt(m[8] - m[0] - m[4] + 1.0);
            q[3] = 0.5 * t; t = 0.5 * d_rcp(t);
            q[0] = (m[1] - m[3]) * t;              /* m(k,j)-m(j,k) = m(1,0)-m(0,1) */
            q[1] = (m[2] + m[6]) * t;              /* m(j,i)+m(i,j) = m(0,2)+m(2,0) */
            q[2] = (m[5] + m[7]) * t;              /* m(k,i)+m(i,k) = m(1,2)+m(2,1) */
        } else if (i1) {     /* i=1, j=2, k=0 */
            t = d_sqrt(m[4] - m[8] - m[0] + 1.0);
            q[2] = 0.5 * t; t = 0.5 * d_rcp(t);
            q[0] = (m[6] - m[2]) * t;              /* m(0,2)-m(2,0) */
            q[3] = (m[5] + m[7]) * t;              /* m(2,1)+m(1,2) */
            q[1] = (m[1] + m[3]) * t;              /* m(0,1)+m(1,0) */
        } else {             /* i=0, j=1, k=2 */
            t = d_sqrt(m[0] - m[4] - m[8] + 1.0);
            q[1] = 0.5 * t; t = 0.5 * d_rcp(t);
            q[0] = (m[5] - m[7]) * t;              /* m(2,1)-m(1,2) */
            q[2] = (m[1] + m[3]) * t;              /* m(1,0)+m(0,1) */
            q[3] = (m[2] + m[6]) * t;              /* m(2,0)+m(0,2) */
        }
    }
    const double n2 = fma(q[3], q[3], fma(q[2], q[2], fma(q[1], q[1], q[0] * q[0])));
    const double rn = d_rcp(d_sqrt(n2));
    q[0] *= rn; q[1] *= rn; q[2] *= rn; q[3] *= rn;
}

/* Eigen::Quaternion::toRotationMatrix (what Sophus' rotationMatrix() returns, :906) */
DVO_DEV void quat_to_matrix(const double *q, double *R) {
    const double qw = q[0], qx = q[1], qy = q[2], qz = q[3];
    const double tx = 2.0 * qx, ty = 2.0 * qy, tz = 2.0 * qz;
    const double twx = tx * qw, twy = ty * qw, twz = tz * qw;
    const double txx = tx * qx, txy = ty * qx, txz = tz * qx;
    const double tyy = ty * qy, tyz = tz * qy, tzz = tz * qz;
    R[0] = 1.0 - (tyy + tzz); R[3] = txy - twz;         R[6] = txz + twy;
    R[1] = txy + twz;         R[4] = 1.0 - (txx + tzz); R[7] = tyz - twx;
    R[2] = txz - twy;         R[5] = tyz + twx;         R[8] = 1.0 - (txx + tyy);
}

/* out = normalise(a (x) b): the rotation cR*xRot of :917 followed by the
 * re-orthogonalisation of :919 -- for unit quaternions the orthogonal polar
 * factor of the product matrix is the matrix of the normalised product. */
DVO_DEV void quat_mul_normalize(const double *a, const double *b, double *o) {
    /* one multiply and three fused multiply-adds per component (round 5: the update is one wave's serial instruction stream;
     * written out with -ffp-contract=off these were 28 instructions) */
    const double w = fma(-a[3], b[3], fma(-a[2], b[2], fma(-a[1], b[1], a[0] * b[0])));
    const double x = fma(-a[3], b[2], fma(a[2], b[3], fma(a[1], b[0], a[0] * b[1])));
    const double y = fma(a[3], b[1], fma(a[2], b[0], fma(-a[1], b[3], a[0] * b[2])));
    const double z = fma(a[3], b[0], fma(-a[2], b[1], fma(a[1], b[2], a[0] * b[3])));
    const double n2 = fma(z, z, fma(y, y, fma(x, x, w * w)));
    /* both factors are unit quaternions up to rounding, so |n2 - 1| ~ 1e-15; anything else (a caller that bypassed
     * pose_state_load) takes the full form */
    const double rn = (fabs(n2 - 1.0) < DVO_K(1e-8)) ? unit_rnorm(n2) : d_rsqrt(n2);
    o[0] = w * rn; o[1] = x * rn; o[2] = y * rn; o[3] = z * rn;
}

/* SE(3) logarithm of (unit quaternion q, translation t); tangent order
 * [upsilon(3), omega(3)] like Sophus::SE3d::log (atan form of SO3::logAndTheta).
 * tan(theta/2) = |q.vec| / q.w for a unit quaternion, so no sin/cos is needed. */
DVO_DEV void se3_log_q(const UpdConst &u, const double *q, const double *t, double *psi) {
    const double squared_n = fma(q[3], q[3], fma(q[2], q[2], q[1] * q[1]));
    const double w = q[0];
    double k2, c;                            /* k2 = 2*atan(n/w)/n ;  c = (1 - theta/(2 tan(theta/2))) / theta^2 */
    if (w > 0.5 && squared_n < u.kl[11] * (w * w)) {
        /* the usual case (round 5): a rotation below ~0.1 rad (down to none: Sophus' own small-angle forms, 2/w - 2 n^2/w^3 and
         * c = 1/12, are the same doubles as these series below n = 1e-10).  With y = n/w = tan(theta/2), y^2 < 2.5e-3:
         * atan(y)/n = P(y^2)/w with P the arctangent series (through y^12/13), so neither n = sqrt(n^2) nor 1/n is needed;
         * theta^2 = k2^2 n^2, and c from its own series (th^2 < 0.01: the term after th^8 is 1e-13 of c, and c multiplies a term
         * th^2 times smaller than psi) -- which also avoids the cancellation of the closed form.  ~40 instructions where the
         * general form below takes ~130. */
        const double rw = d_rcp(w);
        const double y2 = squared_n * (rw * rw);
        double P = u.kl[0];
        P = fma(P, y2, u.kl[1]);
        P = fma(P, y2, u.kl[2]);
        P = fma(P, y2, u.kl[3]);
        P = fma(P, y2, u.kl[4]);
        P = fma(P, y2, u.kl[5]);
        P = fma(P, y2, 1.0);
        k2 = (2.0 * rw) * P;
        const double th2 = (k2 * k2) * squared_n;
        c = u.kl[6];
        c = fma(c, th2, u.kl[7]);
        c = fma(c, th2, u.kl[8]);
        c = fma(c, th2, u.kl[9]);
        c = fma(c, th2, u.kl[10]);
    } else {
        const double n = d_sqrt(squared_n);
        if (n < DVO_K(DVO_SOPHUS_EPS)) {
            const double rw = d_rcp(w);
            k2 = 2.0 * rw - 2.0 * squared_n * (rw * rw * rw);
        } else if (fabs(w) < DVO_K(DVO_SOPHUS_EPS)) {
            k2 = (w > 0.0) ? d_div(DVO_K(M_PI), n) : -d_div(DVO_K(M_PI), n);
        } else {
            k2 = 2.0 * d_atan(n * d_rcp(w)) * d_rcp(n);
        }
        const double theta = k2 * n;
        if (fabs(theta) < DVO_K(DVO_SOPHUS_EPS)) c = DVO_K(1. / 12.);
        else c = (1.0 - 0.5 * theta * w * d_rcp(n)) * d_rcp(theta * theta);
    }
    const double om[3] = {k2 * q[1], k2 * q[2], k2 * q[3]};
    apply_I_aW_bW2(om, -0.5, c, t, psi);     /* V^-1 t = (I - W/2 + c W^2) t */
    psi[3] = om[0]; psi[4] = om[1]; psi[5] = om[2];
}

/* SE(3) exponential, Sophus::SE3d::exp: unit quaternion from the half angle and
 * t = V upsilon with V = I + (1-cos)/th^2 W + (th-sin)/th^3 W^2. */
DVO_DEV void se3_exp_q(const UpdConst &u, const double *psi, double *q, double *t) {
    const double *om = psi + 3;
    const double theta_sq = fma(om[2], om[2], fma(om[1], om[1], om[0] * om[0]));
    double imag, real, a, b;
    const bool small_angle = theta_sq < u.eps2;
    if (small_angle) {
        const double theta_po4 = theta_sq * theta_sq;
        imag = 0.5 - DVO_K(1.0 / 48.0) * theta_sq + DVO_K(1.0 / 3840.0) * theta_po4;
        real = 1.0 - 0.5 * theta_sq + DVO_K(1.0 / 384.0) * theta_po4;
        a = 0.0; b = 0.0;
    } else if (theta_sq < u.ke[10]) {
        /* the usual case (round 5): the step is clamped to the trust radius (0.003 by default, :25), so theta < 0.01.  The
         * coefficients are short series in theta^2 (the terms left out are below 1e-17 relative): no square root, no reciprocal,
         * no argument reduction -- and none of the closed forms' cancellation ((theta - sin theta)/theta^3 loses ten digits at
         * theta = 0.003).  ~20 instructions where the general form below takes ~90. */
        const double z = 0.25 * theta_sq;              /* (theta/2)^2 <= 2.5e-5 */
        double p = u.ke[0];                            /* sin(x)/x, x = theta/2 */
        p = fma(p, z, u.ke[1]);
        p = fma(p, z, u.ke[2]);
        p = fma(p, z, 1.0);
        imag = 0.5 * p;                                /* sin(theta/2)/theta */
        double c = u.ke[3];                            /* cos(x) */
        c = fma(c, z, u.ke[4]);
        c = fma(c, z, u.ke[5]);
        c = fma(c, z, -0.5);
        real = fma(c, z, 1.0);
        a = (2.0 * imag) * imag;                       /* (1 - cos th)/th^2 = 2 sin^2(th/2)/th^2 */
        b = u.ke[6];                                   /* (th - sin th)/th^3 */
        b = fma(b, theta_sq, u.ke[7]);
        b = fma(b, theta_sq, u.ke[8]);
        b = fma(b, theta_sq, u.ke[9]);
    } else {
        const double theta = d_sqrt(theta_sq);
        double sh, ch;
        d_sincos(0.5 * theta, sh, ch);
        const double rth = d_rcp(theta);
        imag = sh * rth;
        real = ch;
        const double rth2 = rth * rth;
        a = (2.0 * sh * sh) * rth2;                       /* (1-cos th)/th^2 */
        b = (theta - 2.0 * sh * ch) * (rth2 * rth);       /* (th-sin th)/th^3 */
    }
    double qw = real, qx = imag * om[0], qy = imag * om[1], qz = imag * om[2];
    /* {cos, sin * axis}: unit up to rounding by construction in all three branches (Sophus normalises here) */
    const double rn = unit_rnorm(fma(qz, qz, fma(qy, qy, fma(qx, qx, qw * qw))));
    q[0] = qw * rn; q[1] = qx * rn; q[2] = qy * rn; q[3] = qz * rn;
    if (small_angle) {                                    /* Sophus: V = so3.matrix() */
        double R[9];
        quat_to_matrix(q, R);
        m3_vec(R, psi, t);
    } else {
        apply_I_aW_bW2(om, a, b, psi, t);
    }
}

/* matrix forms (C ABI helpers and tests) */
DVO_DEV void se3_log(const double *R, const double *t, double *psi) {
    DevParams none = {};
    UpdConst u;
    upd_const_build(u, none);              /* on the stack: folds into literals */
    double q[4];
    quat_of_matrix(R, q);
    se3_log_q(u, q, t, psi);
}
DVO_DEV void se3_exp(const double *psi, double *R, double *t) {
    DevParams none = {};
    UpdConst u;
    upd_const_build(u, none);
    double q[4];
    se3_exp_q(u, psi, q, t);
    quat_to_matrix(q, R);
}

/* rotationize (:1269-1282): R <- U V^T of R = U S V^T, i.e. the orthogonal polar
 * factor.  The reference gets it from a Jacobi SVD; the polar factor is unique
 * for a non-singular matrix, so the Newton iteration X <- (g X + X^-T / g)/2
 * converges to the same matrix.  Only used for caller-supplied matrices: inside
 * the iteration loop the rotation is carried as a unit quaternion, whose
 * normalisation is the same projection (see quat_mul_normalize). */
DVO_DEV void rotationize(double *X) {
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
        const double det = fma(X[2], C[2], fma(X[1], C[1], X[0] * C[0]));
        if (det == 0.0 || !(det == det)) break;          /* singular / NaN: leave as is */
        const double idet = d_rcp(det);
        double nx = 0.0, nc = 0.0;
#pragma unroll
        for (int k = 0; k < 9; k++) { nx = fma(X[k], X[k], nx); nc = fma(C[k], C[k], nc); }
        const double ny = nc * idet * idet;              /* ||X^-T||_F^2 */
        double gx = 0.5, gy = 0.5 * idet;
        if (fabs(ny - nx) > 1e-3 * nx) {                 /* far from orthogonal: scaled step */
            const double gam = d_sqrt(d_sqrt(d_div(ny, nx)));
            gx = 0.5 * gam; gy = 0.5 * idet * d_rcp(gam);
        }
        double diff = 0.0;
#pragma unroll
        for (int k = 0; k < 9; k++) {
            const double xn_ = fma(gx, X[k], gy * C[k]);
            const double d = xn_ - X[k];
            diff = fma(d, d, diff);
            X[k] = xn_;
        }
        if (diff <= 1e-30 * nx) break;
    }
}

/* Optimiser state of one runIterations call (lives in LDS, or in HBM between the launches of the tiled schedule).
 * The rotation cR is carried as a unit quaternion q; R is its matrix (what the reference holds in cR after rotationize, :919).
 *
 * Round 5: the pose iterate {q, t, float casts} exists TWICE (p[0], p[1]).  The update of iteration itr reads `cur` and writes
 * `nxt`; the kernels that run the update on one wave and the best-iterate bookkeeping (:689-705) on another AT THE SAME TIME
 * (dvo_fused.hip) pass p[itr & 1] and p[(itr + 1) & 1], so the bookkeeping copies the current iterate while the next one is
 * being written; everybody else passes p[0] for both (the update reads all it needs before its first store). */
struct PoseCur {
    double q[4];                 /* cR (as quaternion) */
    double t[3], pad_;           /* cT */
    float Rf[9], tf[3];          /* cR_32, cT_32 (:673-674) for the next evaluation */
};
struct __attribute__((aligned(16))) PoseState {
    PoseCur p[2];
    double R[9], e2_fast;        /* matrix of the current q; packed kernel: the iteration's sum of eps^2 as added (its energy is open) */
    double d[6];                 /* descentDirection (:654) */
    double creg[6], creg_scale, pad1_;  /* regulariser of the CURRENT pose, precomputed (fused kernels): log(pose) and lambda/|log| */
    double bq[4], bt[3], pad2_;  /* best iterate (:646-647) */
    float bRf[9], btf[3];        /* cR_32, cT_32 of the best iterate (for finalEpsilons/Reprojections) */
    float bestE, bestRatio;      /* :644-645 */
    int bestItr;                 /* :648 */
    int stop;
    int exact_ran;               /* packed kernel, inspection: a wave took the literal-division fallback during this level */
    int e2_open, e2_nvis;        /* packed kernel: the certificate of this iteration's energy failed -> every wave sweeps the residuals again, exactly */
    int e2_ran;                  /* inspection: iterations of this level that took that sweep */
    UpdConst u;                  /* the update's constants: built once per kernel (or once per level by iter_begin_kernel for the state
                                    that travels through HBM between the launches of the tiled schedule) */
};
static_assert(sizeof(PoseCur) % 16 == 0 && sizeof(PoseState) % 16 == 0, "16-byte LDS accesses");

/* state from a caller-supplied pose (kernel entry) */
DVO_DEV void pose_state_load(PoseState &s, const double *R, const double *t) {
    double Rl[9], q[4];
#pragma unroll
    for (int k = 0; k < 9; k++) { Rl[k] = R[k]; s.R[k] = R[k]; }
    quat_of_matrix(Rl, q);
#pragma unroll
    for (int k = 0; k < 4; k++) s.p[0].q[k] = q[k];
#pragma unroll
    for (int k = 0; k < 3; k++) s.p[0].t[k] = t[k];
}

/* start of one runIterations call (:642-657); the current iterate is p[0] */
DVO_DEV void pose_state_begin(PoseState &s) {
#pragma unroll
    for (int k = 0; k < 6; k++) s.d[k] = 0.0;
    s.bq[0] = 1.0; s.bq[1] = s.bq[2] = s.bq[3] = 0.0;     /* bestcR = I  :646 */
    s.bt[0] = s.bt[1] = s.bt[2] = 0.0;                     /* bestcT = 0  :647 */
    s.bestE = 1.0E10f;
    s.bestRatio = 1.0f;
    s.bestItr = -1;
    s.stop = 0;
#pragma unroll
    for (int k = 0; k < 9; k++) s.p[0].Rf[k] = (float)s.R[k];
#pragma unroll
    for (int k = 0; k < 3; k++) s.p[0].tf[k] = (float)s.p[0].t[k];
}

/* end of one runIterations call (:997-1001): pose <- best iterate (into p[0]) */
DVO_DEV void pose_state_finish(PoseState &s) {
    double q[4], R[9];
#pragma unroll
    for (int k = 0; k < 4; k++) { q[k] = s.bq[k]; s.p[0].q[k] = q[k]; }
    quat_to_matrix(q, R);
#pragma unroll
    for (int k = 0; k < 9; k++) s.R[k] = R[k];
#pragma unroll
    for (int k = 0; k < 3; k++) s.p[0].t[k] = s.bt[k];
}

/* Phase fence for the update: it is written as short phases that hand their results over through LDS (PoseState) or a few
 * registers, and this fence stops the compiler from hoisting the next phase's loads / keeping the previous phase's values
 * alive across it. */
#define DVO_PHASE_FENCE()                       \
    do {                                        \
        asm volatile("" ::: "memory");          \
        __builtin_amdgcn_sched_barrier(0);      \
    } while (0)

/* the L2 regulariser's share that depends on the pose only (:734-743): cpsi = log(pose), returns lambda/|cpsi| (0 if the
 * pose is the identity: nothing is added then).  reg_lambda > 0, so "scale == 0" <=> "n == 0". */
DVO_DEV double pose_regulariser_terms(const PoseCur &cur, const UpdConst &u, double *cpsi) {
    double q[4], t[3];
#pragma unroll
    for (int k = 0; k < 4; k++) q[k] = cur.q[k];
#pragma unroll
    for (int k = 0; k < 3; k++) t[k] = cur.t[k];
    se3_log_q(u, q, t, cpsi);
    double n2 = cpsi[0] * cpsi[0];
#pragma unroll
    for (int k = 1; k < 6; k++) n2 = fma(cpsi[k], cpsi[k], n2);
    return (n2 > 0.0) ? u.lambda * d_rsqrt(n2) : 0.0;
}
/* fused kernels: one lane runs this right after the barrier that releases the other waves into the next iteration */
DVO_DEV void pose_regulariser_precompute(PoseState &s, const PoseCur &cur, const UpdConst &u) {
    if (!u.l2_reg) return;
    double cpsi[6];
    const double sc = pose_regulariser_terms(cur, u, cpsi);
#pragma unroll
    for (int k = 0; k < 6; k++) s.creg[k] = cpsi[k];
    s.creg_scale = sc;
}

/* ---- everything runIterations does after the per-point phase of iteration `itr` (:689-920), in three pieces -------------
 *   pose_bookkeep   energy + best-iterate bookkeeping (:689-705)              needs sum eps^2, the visible count, `cur`
 *   pose_direction  regulariser, heavy ball, step (:724-821) -> psi           needs g = J^T W eps (:777)
 *   pose_apply      trust region, termination, exponential map, compose (:832-919, :673-674): cur -> nxt
 * The pieces are independent but for `cur`, which the first only reads and the last only reads before it writes `nxt`. */

/* ---- the energy without an order of summation (round 6) ---------------------------------------------------------------------
 * E = (float)sqrt(S) (:689, :1310-1312) with S the CORRECTLY ROUNDED double of the exact sum of eps^2 -- a definition no order of
 * additions enters (rounds 1-5: S was whatever the kernel's tree of double additions gave; once in ~50 000 energies that flipped the
 * float).  Two tools:
 *  (a) the certificate.  A sum S' of N non-negative doubles added in ANY order is within N 2^-53 S' of the exact sum, the exact
 *      sum's rounding within 2^-53 more; sqrt and the narrowing are monotone, so if every x in S' (1 -+ (N + 16) 2^-53) gives the same
 *      (float)sqrt(x), that float IS the energy.  It fails for about N 2^-28 of the iterations.
 *  (b) the exact sum.  eps^2 of a float is a 48-bit integer times a power of four: on the grid of 2^-68 every |eps| in
 *      [2^-11, 2^12) (every normalised distance) gives an integer below 2^92, added into three 32-bit limbs.  Limb sums stay below
 *      2^53 for up to 2^21 points, so held in doubles they pass through every reduction the other sums use (DPP, LDS, team
 *      exchange, ncclAllReduce) EXACTLY, in any order; one rounding at the end.  A value outside the range marks limb 2 with 2^50
 *      and the caller keeps S'.
 * The packed fused kernel runs (a) every iteration and (b) -- one more sweep over the points, residual only -- where (a) fails; the
 * kernels whose sums travel between launches or ranks carry the limbs always. */
#define DVO_E2_EXP0 116              /* biased exponent of 2^-11 */
#define DVO_E2_BINADES 22
#define DVO_E2_BAD 1125899906842624.0 /* 2^50 */
struct E2Limbs { unsigned long long l0, l1, l2; };
DVO_DEV void e2_limbs_zero(E2Limbs &a) { a.l0 = a.l1 = a.l2 = 0ull; }
DVO_DEV void e2_limbs_add(E2Limbs &a, float eps) {
    const unsigned bits = __float_as_uint(eps) & 0x7fffffffu;
    const unsigned de = (bits >> 23) - (unsigned)DVO_E2_EXP0;
    const bool ok = de <= (unsigned)DVO_E2_BINADES;
    const unsigned m = (bits & 0x7fffffu) | 0x800000u;
    const unsigned long long sq = (unsigned long long)m * m;             /* 48 bits */
    const unsigned sh = ok ? 2u * de : 0u;                               /* <= 44 */
    const unsigned long long lo = sq << sh, hi = sh ? (sq >> (64u - sh)) : 0ull;
    if (ok) { a.l0 += lo & 0xffffffffull; a.l1 += lo >> 32; a.l2 += hi; }
    else if (bits != 0u) a.l2 += 1ull << 50;
}
/* (sums of) limbs -> the correctly rounded double of the exact sum; `fallback` if a term was out of range */
DVO_DEV double e2_from_limbs(double d0, double d1, double d2, double fallback) {
    if (!(d2 < DVO_E2_BAD)) return fallback;
    const unsigned long long a0 = (unsigned long long)d0, a1 = (unsigned long long)d1, a2 = (unsigned long long)d2;
    const unsigned long long t = a1 << 32;
    unsigned long long lo = a0 + t, hi = a2 + (a1 >> 32) + ((lo < t) ? 1ull : 0ull);      /* a0 + a1 2^32 + a2 2^64 */
    if (hi == 0ull) return (double)lo * 0x1p-68;                         /* u64 -> f64 rounds to nearest even */
    const int n = __builtin_clzll(hi);
    const unsigned long long H = n ? ((hi << n) | (lo >> (64 - n))) : hi, L = lo << n;      /* bit 63 of H set */
    unsigned long long q = H >> 11;
    const unsigned rem = (unsigned)(H & 0x7ffull);
    if (rem > 0x400u || (rem == 0x400u && (L != 0ull || (q & 1ull)))) q++;
    return (double)q * __longlong_as_double((long long)(1023 + 7 - n) << 52);      /* q 2^(11 + 64 - n - 68) */
}
DVO_DEV bool energy_certified(double sum_eps2, int N, float &energy) {
    const float e = (float)sqrt(sum_eps2);
    energy = e;
    if (sum_eps2 == 0.0) return true;
    const unsigned b = __float_as_uint(e);
    if (b - 0x00800000u >= 0x7effffffu) return false;                   /* not a positive normal float below FLT_MAX (NaN, inf, ...) */
    /* (float)sqrt(x) == e for every x strictly between the squares of the midpoints to e's two neighbours; the squares are rounded
     * (2^-53 each, like sqrt itself): the slack of 16 covers that */
    const double m_lo = 0.5 * ((double)e + (double)__uint_as_float(b - 1u)), m_hi = 0.5 * ((double)e + (double)__uint_as_float(b + 1u));
    const double d = sum_eps2 * ((double)(N + 16) * 0x1p-53);
    return (sum_eps2 - d) > m_lo * m_lo && (sum_eps2 + d) < m_hi * m_hi;
}

/* :696-705 with the energy in hand */
DVO_DEV void pose_bookkeep_e(PoseState &s, const PoseCur &cur, int itr, int N, float energy, int n_vis) {
    if (energy <= s.bestE) {                                              /* :696 */
        s.bestE = energy;
        s.bestRatio = (float)n_vis / (float)N;                            /* :457 */
#pragma unroll
        for (int k = 0; k < 4; k++) s.bq[k] = cur.q[k];
#pragma unroll
        for (int k = 0; k < 3; k++) { s.bt[k] = cur.t[k]; s.btf[k] = cur.tf[k]; }
#pragma unroll
        for (int k = 0; k < 9; k++) s.bRf[k] = cur.Rf[k];
        s.bestItr = itr;
    }
}
/* energy narrows to float, so it needs the correctly rounded double sqrt to match the oracle's (float)sqrt(double) bit for bit;
 * sum_eps2 = the correctly rounded exact sum (e2_from_limbs), or a sum the caller has certified */
DVO_DEV float pose_bookkeep(PoseState &s, const PoseCur &cur, int itr, int N, double sum_eps2, int n_vis) {
    const float energy = (float)sqrt(sum_eps2);                           /* :689, :1312 */
    if (energy <= s.bestE) {                                              /* :696 */
        s.bestE = energy;
        s.bestRatio = (float)n_vis / (float)N;                            /* :457 */
#pragma unroll
        for (int k = 0; k < 4; k++) s.bq[k] = cur.q[k];
#pragma unroll
        for (int k = 0; k < 3; k++) { s.bt[k] = cur.t[k]; s.btf[k] = cur.tf[k]; }
#pragma unroll
        for (int k = 0; k < 9; k++) s.bRf[k] = cur.Rf[k];
        s.bestItr = itr;
    }
    return energy;
}

/* -stepLength of iteration itr (:773): a function of the iteration index only */
DVO_DEV double pose_neg_step(const UpdConst &u, int itr) {
    const double step = u.ab * ((itr > u.decay_after) ? d_rcp((double)(itr - u.decay_offset)) : 1.0);
    return -step;
}

/* one lane, all six components (the host-driven / tiled kernels and the one-point-per-lane kernel) */
template <bool PRE_REG>
DVO_DEV void pose_direction(PoseState &s, const PoseCur &cur, const UpdConst &u, int itr, const double *g_in, double *psi) {
    double g[6];
#pragma unroll
    for (int k = 0; k < 6; k++) g[k] = g_in[k];
    if (u.l2_reg) {                                                   /* :734-743, :796 */
        if (PRE_REG) {                                                /* log(pose) was taken while the other waves worked */
            const double lam_n = s.creg_scale;
            if (lam_n != 0.0) {
#pragma unroll
                for (int k = 0; k < 6; k++) g[k] = fma(lam_n, s.creg[k], g[k]);
            }
        } else {
            double cpsi[6];
            const double lam_n = pose_regulariser_terms(cur, u, cpsi);
            if (lam_n != 0.0) {
#pragma unroll
                for (int k = 0; k < 6; k++) g[k] = fma(lam_n, cpsi[k], g[k]);
            }
        }
    }
    const double ns = pose_neg_step(u, itr);
#pragma unroll
    for (int k = 0; k < 6; k++) {
        const double dk = u.omb * g[k] + u.beta * s.d[k];                 /* :799 */
        s.d[k] = dk;
        psi[k] = (ns * u.pk[k]) * dk;                                     /* :821 */
    }
}

/* a double of lane `lane` -> every lane (two v_readlane_b32: the value lands in scalar registers) */
DVO_DEV double readlane_f64(double x, int lane) {
    const long long b = __double_as_longlong(x);
    const int lo = __builtin_amdgcn_readlane((int)b, lane), hi = __builtin_amdgcn_readlane((int)(b >> 32), lane);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
/* The same, executed by a WHOLE WAVE whose lane k < 6 holds g_k (the packed kernel hands the reduced sums over in registers):
 * component k is worked on by lane k -- one instruction where the single-lane form has six, one LDS access for creg / d / the
 * pre-conditioner where it has six each -- and the six results are then broadcast through scalar registers.  Needs the
 * regulariser precomputed (creg finite; creg_scale == 0 adds an exact zero).  neg_step = pose_neg_step(u, itr), which the
 * caller can take before the sums exist. */
DVO_DEV void pose_direction_lanes(PoseState &s, const UpdConst &u, double neg_step, double g_lane, int lane, double *psi) {
    const int k = (lane < 6) ? lane : 5;
    double g = g_lane;
    if (u.l2_reg) g = fma(s.creg_scale, s.creg[k], g);                    /* :734-743, :796 */
    const double dk = u.omb * g + u.beta * s.d[k];                        /* :799 */
    if (lane < 6) s.d[k] = dk;
    const double pl = (neg_step * u.pk[k]) * dk;                          /* :821 */
#pragma unroll
    for (int q = 0; q < 6; q++) psi[q] = readlane_f64(pl, q);
}

/* Trust region, termination test, exponential map, composition (:832-919).  Reads cur (and s.R), writes nxt (and s.R); sets
 * s.stop instead when the step is below the termination threshold (:872).  The norm is only ever compared, so it is compared
 * squared; the clamp's 1/|psi| is one reciprocal square root. */
DVO_DEV void pose_apply(PoseState &s, const PoseCur &cur, PoseCur &nxt, const UpdConst &u, double *psi) {
    double n2 = psi[0] * psi[0];
#pragma unroll
    for (int k = 1; k < 6; k++) n2 = fma(psi[k], psi[k], n2);                 /* :832 */
    const double tr2 = u.tr2;
    if (n2 > tr2) {                                                           /* :835 */
        const double sc = u.tr * d_rsqrt(n2);                                 /* :837 */
#pragma unroll
        for (int k = 0; k < 6; k++) psi[k] *= sc;
        n2 = tr2;                       /* |psi|^2 after the projection; only compared with the threshold below (Q5) */
    }
    if (n2 < u.stop2) {                                                       /* :872 */
        s.stop = 1;
        return;
    }
    double qx[4], xT[3];
    se3_exp_q(u, psi, qx, xT);                                                /* :905-907 */
    DVO_PHASE_FENCE();
    {   /* cT += cR*xTrans (:916), with the OLD cR */
        double R[9], dT[3];
#pragma unroll
        for (int k = 0; k < 9; k++) R[k] = s.R[k];
        m3_vec(R, xT, dT);
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const double tk = cur.t[k] + dT[k];
            nxt.t[k] = tk; nxt.tf[k] = (float)tk;
        }
    }
    DVO_PHASE_FENCE();
    {   /* cR *= xRot; rotationize (:917, :919); casts (:673) */
        double q[4], R[9];
#pragma unroll
        for (int k = 0; k < 4; k++) q[k] = cur.q[k];
        quat_mul_normalize(q, qx, q);
        quat_to_matrix(q, R);
#pragma unroll
        for (int k = 0; k < 4; k++) nxt.q[k] = q[k];
#pragma unroll
        for (int k = 0; k < 9; k++) { s.R[k] = R[k]; nxt.Rf[k] = (float)R[k]; }
    }
}

/* the three pieces in a row on one lane, in place (p[0] is both the iterate read and the iterate written) */
template <bool PRE_REG>
DVO_DEV float pose_update_t(PoseState &s, const UpdConst &u, int itr, int N,
                            const double *g_in, double sum_eps2, int n_vis) {
    const float energy = pose_bookkeep(s, s.p[0], itr, N, sum_eps2, n_vis);
    DVO_PHASE_FENCE();
    double psi[6];
    pose_direction<PRE_REG>(s, s.p[0], u, itr, g_in, psi);
    DVO_PHASE_FENCE();
    pose_apply(s, s.p[0], s.p[0], u, psi);
    return energy;
}
DVO_DEV float pose_update(PoseState &s, const UpdConst &u, int itr, int N, const double *g_in, double sum_eps2, int n_vis) {
    return pose_update_t<false>(s, u, itr, N, g_in, sum_eps2, n_vis);
}

}  // namespace dvo
#endif
