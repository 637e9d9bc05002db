/*
 * dvo_point_pk.h -- the per-point float32 math of computeJacobianOfNowFrame + getReprojectedEpsilons
 * (reference src/SolveDVO.cpp:306-414, :425-462) for TWO reference points per lane, written on 2-wide float
 * vectors so that every multiply / add of the pair is one packed instruction (v_pk_mul_f32 / v_pk_add_f32 /
 * v_pk_fma_f32 with the wave-uniform pose entries as scalar-register operands).
 *
 * Why (measured, tools/exhaustive/valu_rates.hip, profiles/r02_valu_rates.txt): at the two waves per SIMD of the fused
 * kernel a wave issues one vector instruction every 4.4-4.7 cycles whatever it is -- a packed multiply of two points
 * costs what a scalar multiply with a scalar-register operand costs -- so halving the instruction count of the
 * per-point loop nearly halves its time.  Every operation below is the IEEE single operation of the scalar form in
 * dvo_device_math.h (project_point, jacobian_row), applied to each half: same roundings, same order, no contraction
 * (-ffp-contract=off; the explicit fma()s are exact identities, see below), hence the same bits as the oracle.
 *
 * Exact identities used here (all checked over all 2^32 bit patterns by tools/exhaustive/div_tricks.hip):
 *   (1),(2)  1.0f/x == rcp+Newton and zn = x*fl(1/x) in {1, 1-2^-24}   for 2^-126 <= |x| <= 2^126  (dvo_device_math.h)
 *   (5)  n / (1-2^-24) == fma(n, (1+2^-23) 2^-24, n)     for EVERY float n
 *   (6)  n / (1-2^-23) == fma(n, 2^-23 + 2^-46, n)       for EVERY float n
 *        and with dz = 1 - zn in {0, 2^-24} (exact): dz * 2^24 * c is 0 or c exactly, fma(n, 0, n) == n for finite n,
 *        so the divisions by Z and Z*Z of :388-393 need no comparison at all
 *   (7)  u >= 0 && u < C  <=>  (unsigned)floor_i32(max(u,-1)) < C   for integer C in [1, 2^24], any u incl. NaN / inf
 */
#ifndef DVO_POINT_PK_H_
#define DVO_POINT_PK_H_

#include "dvo_device_math.h"

namespace dvo {

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

DVO_DEV v2f pk_fma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
DVO_DEV v2f pk_splat(float s) { v2f r; r.x = s; r.y = s; return r; }

#define DVO_K23 0x1.000002p+1f     /* 2^24 * (2^-23 + 2^-46) = 2 + 2^-22 */
#define DVO_K24 0x1.000002p+0f     /* 2^24 * (1 + 2^-23) 2^-24 = 1 + 2^-23 */
#define DVO_C23 0x1.000002p-23f    /* 2^-23 + 2^-46  (0x34000001) */
#define DVO_C24 0x1.000002p-24f    /* (1 + 2^-23) 2^-24  (0x33800001) */

/* floor to int32 (saturating); NaN -> callers guard with max(u,-1) */
DVO_DEV int cvt_floor_i32(float u) {
    int i;
    asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(i) : "v"(u));
    return i;
}
/* IEEE maxNum in ONE instruction (fmaxf() makes the compiler add canonicalising instructions): NaN -> the other operand */
DVO_DEV float max_num(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
/* identity (7): pixel column/row of a reprojection, and whether it is inside [0, C) */
DVO_DEV bool pixel_in_range(float u, int C, int &px) {
    px = cvt_floor_i32(max_num(u, -1.0f));
    return (unsigned)px < (unsigned)C;
}

/* ---- lane masks as scalar values (round 4 experiment; the product form of the compact-form loops since round 5) --------------------
 * A comparison that feeds a ballot (a visible count, the wave's "some z is degenerate" flag) or a select is a 64-bit lane mask in a
 * scalar register pair by nature: v_cmp_* writes one.  Written as `bool`, the compiler turns every __builtin_amdgcn_ballot_w64 of a
 * combined condition into v_cndmask 0/1 + v_cmp_ne, and keeps an accumulated wave-uniform flag in a vector register: eight vector
 * instructions per round of two points in the packed loop.  These helpers produce and consume the masks directly; and / or /
 * popcount of masks are scalar instructions.  All lanes are active wherever they are used (the packed point loops are not
 * divergent).  Round 4: -2 % alone (not taken); round 5, with the shorter serial chain: +3 % (taken: dvo_fused.hip, round2_issue). */
typedef unsigned long long lanemask;
DVO_DEV float uniform_const_f(float k) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, k))); }
DVO_DEV lanemask mask_lt_u32(unsigned a, unsigned b_uniform) {           /* a < b, b wave-uniform */
    lanemask m;
    asm("v_cmp_gt_u32_e64 %0, %2, %1" : "=s"(m) : "v"(a), "s"(b_uniform));
    return m;
}
DVO_DEV lanemask mask_lt_i32(int a, int b_uniform) {
    lanemask m;
    asm("v_cmp_gt_i32_e64 %0, %2, %1" : "=s"(m) : "v"(a), "s"(b_uniform));
    return m;
}
/* !(lo <= |x| <= hi), NaN included: the complement of rcp_in_proven_range */
DVO_DEV lanemask mask_abs_outside(float x, float lo_uniform, float hi_uniform) {
    lanemask a, b;
    asm("v_cmp_nge_f32_e64 %0, |%1|, %2" : "=s"(a) : "v"(x), "s"(lo_uniform));
    asm("v_cmp_nle_f32_e64 %0, |%1|, %2" : "=s"(b) : "v"(x), "s"(hi_uniform));
    return a | b;
}
DVO_DEV unsigned select_or_zero(lanemask m, unsigned v) {                 /* m ? v : 0 */
    unsigned r;
    asm("v_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(r) : "v"(v), "s"(m));
    return r;
}
DVO_DEV unsigned select_or(lanemask m, unsigned v, unsigned other) {      /* m ? v : other */
    unsigned r;
    asm("v_cndmask_b32_e64 %0, %3, %1, %2" : "=v"(r) : "v"(v), "s"(m), "v"(other));
    return r;
}
/* identity (7) as a mask: pixel column/row of a reprojection, and the lanes where it is inside [0, C) */
DVO_DEV lanemask pixel_in_range_mask(float u, int C_uniform, int &px) {
    px = cvt_floor_i32(max_num(u, -1.0f));
    return mask_lt_u32((unsigned)px, (unsigned)C_uniform);
}

/* :328-345 for two points.  Returns per-half "degenerate z" flags in odd0/odd1: those lanes must be redone with the
 * scalar project_point (literal IEEE divisions). */
DVO_DEV void project_point2(const IterConst &c, v2f X, v2f Y, v2f Z, v2f &xn, v2f &yn, v2f &zn, v2f &u, v2f &v,
                            bool &odd0, bool &odd1) {
    const v2f d0 = X - c.t[0], d1 = Y - c.t[1], d2 = Z - c.t[2];            /* _3d - cTRep          :329-330 */
    const v2f p0 = (c.r[0] * d0 + c.r[1] * d1) + c.r[2] * d2;              /* cR^T * d */
    const v2f p1 = (c.r[3] * d0 + c.r[4] * d1) + c.r[5] * d2;
    const v2f p2 = (c.r[6] * d0 + c.r[7] * d1) + c.r[8] * d2;
    v2f r;
    r.x = __builtin_amdgcn_rcpf(p2.x);
    r.y = __builtin_amdgcn_rcpf(p2.y);
    const v2f inv = pk_fma(pk_fma(-p2, r, pk_splat(1.0f)), r, r);           /* == 1.0f/p2, identity (1)   :339 */
    odd0 = !rcp_in_proven_range(p2.x);
    odd1 = !rcp_in_proven_range(p2.y);
    xn = p0 * inv; yn = p1 * inv; zn = p2 * inv;                            /* :340-341 */
    u = c.m00 * xn + c.m02 * zn;                                            /* :344 */
    v = c.m11 * yn + c.m12 * zn;
}

/* the same with the degenerate-z lanes of both halves as ONE lane mask (DVO_VALU_DIET) */
DVO_DEV void project_point2m(const IterConst &c, v2f X, v2f Y, v2f Z, v2f &xn, v2f &yn, v2f &zn, v2f &u, v2f &v, lanemask &odd) {
    const v2f d0 = X - c.t[0], d1 = Y - c.t[1], d2 = Z - c.t[2];            /* _3d - cTRep          :329-330 */
    const v2f p0 = (c.r[0] * d0 + c.r[1] * d1) + c.r[2] * d2;              /* cR^T * d */
    const v2f p1 = (c.r[3] * d0 + c.r[4] * d1) + c.r[5] * d2;
    const v2f p2 = (c.r[6] * d0 + c.r[7] * d1) + c.r[8] * d2;
    v2f r;
    r.x = __builtin_amdgcn_rcpf(p2.x);
    r.y = __builtin_amdgcn_rcpf(p2.y);
    const v2f inv = pk_fma(pk_fma(-p2, r, pk_splat(1.0f)), r, r);           /* == 1.0f/p2, identity (1)   :339 */
    const float lo = uniform_const_f(1.17549435e-38f), hi = uniform_const_f(8.50705917e37f);       /* 2^-126 .. 2^126 */
    odd = mask_abs_outside(p2.x, lo, hi) | mask_abs_outside(p2.y, lo, hi);
    xn = p0 * inv; yn = p1 * inv; zn = p2 * inv;                            /* :340-341 */
    u = c.m00 * xn + c.m02 * zn;                                            /* :344 */
    v = c.m11 * yn + c.m12 * zn;
}

/* :379-406 for two points whose zn is 1 or 1-2^-24 (every lane that took the fast reciprocal; identity (2)).
 * gx*, gy*, w*: the gathered gradient and weight of point 0 / point 1 -- they arrive in the registers of two separate
 * 16-byte loads, so the products that consume them are written per half (no register shuffles to pair them up).
 * Returns jw[k] = (float)(J_k * w)  (:716): the Jacobian row already scaled by the weight. */
DVO_DEV void jacobian_weighted2(const IterConst &c, v2f xn, v2f yn, v2f zn, float gx0, float gx1, float gy0, float gy1,
                                float wt0, float wt1, v2f *jw, v2f *jout = nullptr /* the unweighted rows (H = sum w J J^T), or NULL */) {
    const v2f n02 = (-c.m00) * xn, n12 = (-c.m11) * yn;
    const v2f dz = pk_splat(1.0f) - zn;                                     /* 0 or 2^-24, exact */
    const v2f cz1 = dz * DVO_K24, cz2 = dz * DVO_K23;                       /* 0 or the constants of (5), (6) */
    const v2f a00 = pk_fma(pk_splat(c.m00), cz1, pk_splat(c.m00));          /* scaleFac*fx/Z          :388 */
    const v2f a11 = pk_fma(pk_splat(c.m11), cz1, pk_splat(c.m11));          /* :392 */
    const v2f a02 = pk_fma(n02, cz2, n02);                                  /* -scaleFac*fx*X/(Z*Z)   :390 */
    const v2f a12 = pk_fma(n12, cz2, n12);                                  /* :393 */
    v2f ga0, ga1, t02, t12;                                                 /* G*A1, structural zeros dropped */
    ga0.x = gx0 * a00.x; ga0.y = gx1 * a00.y;
    ga1.x = gy0 * a11.x; ga1.y = gy1 * a11.y;
    t02.x = gx0 * a02.x; t02.y = gx1 * a02.y;
    t12.x = gy0 * a12.x; t12.y = gy1 * a12.y;
    const v2f ga2 = t02 + t12;
    const v2f w0 = (c.r[0] * xn + c.r[1] * yn) + c.r[2] * zn;              /* tmp = cR^T*(xn,yn,zn)  :399 */
    const v2f w1 = (c.r[3] * xn + c.r[4] * yn) + c.r[5] * zn;
    const v2f w2 = (c.r[6] * xn + c.r[7] * yn) + c.r[8] * zn;
    /* columns 0..2 of A2 are -cR^T (:397); the sign rides on the weight multiply (exact) */
    const v2f j0 = -((ga0 * c.r[0] + ga1 * c.r[3]) + ga2 * c.r[6]);
    const v2f j1 = -((ga0 * c.r[1] + ga1 * c.r[4]) + ga2 * c.r[7]);
    const v2f j2 = -((ga0 * c.r[2] + ga1 * c.r[5]) + ga2 * c.r[8]);
    const v2f j3 = ga1 * w2 - ga2 * w1;                                     /* to_se_3(tmp) :401-402, :1104-1114 */
    const v2f j4 = ga2 * w0 - ga0 * w2;
    const v2f j5 = ga0 * w1 - ga1 * w0;
    jw[0].x = j0.x * wt0; jw[0].y = j0.y * wt1;
    jw[1].x = j1.x * wt0; jw[1].y = j1.y * wt1;
    jw[2].x = j2.x * wt0; jw[2].y = j2.y * wt1;
    jw[3].x = j3.x * wt0; jw[3].y = j3.y * wt1;
    jw[4].x = j4.x * wt0; jw[4].y = j4.y * wt1;
    jw[5].x = j5.x * wt0; jw[5].y = j5.y * wt1;
    if (jout) { jout[0] = j0; jout[1] = j1; jout[2] = j2; jout[3] = j3; jout[4] = j4; jout[5] = j5; }
}

/* the same with the gradients and weights of the two points already paired (they come from LDS lookups, dvo_palette.h): every
 * product is packed.  Same operations in the same order per half -> same bits as jacobian_weighted2. */
DVO_DEV void jacobian_weighted2p(const IterConst &c, v2f xn, v2f yn, v2f zn, v2f gx, v2f gy, v2f wt, v2f *jw, v2f *jout = nullptr) {
    const v2f n02 = (-c.m00) * xn, n12 = (-c.m11) * yn;
    const v2f dz = pk_splat(1.0f) - zn;
    const v2f cz1 = dz * DVO_K24, cz2 = dz * DVO_K23;
    const v2f a00 = pk_fma(pk_splat(c.m00), cz1, pk_splat(c.m00));          /* :388 */
    const v2f a11 = pk_fma(pk_splat(c.m11), cz1, pk_splat(c.m11));          /* :392 */
    const v2f a02 = pk_fma(n02, cz2, n02);                                  /* :390 */
    const v2f a12 = pk_fma(n12, cz2, n12);                                  /* :393 */
    const v2f ga0 = gx * a00, ga1 = gy * a11;
    const v2f ga2 = gx * a02 + gy * a12;
    const v2f w0 = (c.r[0] * xn + c.r[1] * yn) + c.r[2] * zn;              /* :399 */
    const v2f w1 = (c.r[3] * xn + c.r[4] * yn) + c.r[5] * zn;
    const v2f w2 = (c.r[6] * xn + c.r[7] * yn) + c.r[8] * zn;
    const v2f j0 = -((ga0 * c.r[0] + ga1 * c.r[3]) + ga2 * c.r[6]);
    const v2f j1 = -((ga0 * c.r[1] + ga1 * c.r[4]) + ga2 * c.r[7]);
    const v2f j2 = -((ga0 * c.r[2] + ga1 * c.r[5]) + ga2 * c.r[8]);
    const v2f j3 = ga1 * w2 - ga2 * w1;                                     /* :401-402, :1104-1114 */
    const v2f j4 = ga2 * w0 - ga0 * w2;
    const v2f j5 = ga0 * w1 - ga1 * w0;
    jw[0] = j0 * wt; jw[1] = j1 * wt; jw[2] = j2 * wt; jw[3] = j3 * wt; jw[4] = j4 * wt; jw[5] = j5 * wt;     /* :716 */
    if (jout) { jout[0] = j0; jout[1] = j1; jout[2] = j2; jout[3] = j3; jout[4] = j4; jout[5] = j5; }
}

}  // namespace dvo
#endif
