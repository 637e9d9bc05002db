/*
 * dvo_fused.hip -- the throughput form of the edge-alignment hot path (gfx950 / CDNA4, wave64): the whole
 * coarse-to-fine schedule of SolveDVO::loop (reference src/SolveDVO.cpp:2097-2104) over runIterations (:619-1017) for
 * one frame pair per workgroup, with the per-point phase (:306-414, :425-462) on TWO points per lane in packed
 * float32 arithmetic (dvo_point_pk.h).
 *
 * Used for launches whose point lists all have the engine's compact 8-byte form (lists built by the enlist kernels);
 * anything else (caller-supplied 3xN float lists, the optional interpolate() lookup) runs align_fused_kernel of
 * dvo_kernels.hip.  Same results bit for bit -- tests/test_gpu_parity.py runs both against the oracle.
 *
 * Per level: the compact points {xx | yy << 16, Z} are staged once into LDS (those beyond the LDS budget are streamed, fetched
 * one round ahead); per iteration a lane takes points i and i + BLOCK of every round of 2*BLOCK points, rebuilds X, Y
 * (:249-250), warps and projects both (:328-345), looks up {DT, gx, gy, w} of the two pixels they fall on, forms the weighted
 * Jacobian rows (:379-406, :716) and accumulates g = J^T W eps (:777) and sum eps^2 (:1312) in double.  The look-up has
 * three forms, chosen per level and pair (TexSrc below): a 16-byte texel gathered from HBM / L2, the same texel from an LDS copy
 * of the whole level, or -- the throughput form -- ONE 12-byte gather of rank words from the level's compact form
 * (dvo_palette.h: 24 pixels per 128-byte line instead of 8) followed by five palette look-ups in LDS.  Gathers are issued one
 * or two rounds ahead of the arithmetic that consumes them.  Reduction, update and bookkeeping as in dvo_kernels.hip.
 *
 * Launch shapes (chosen by the host, dvo_capi.cpp): two 256-thread workgroups per CU for large batches with a compact form,
 * one 512-thread workgroup per CU otherwise, teams of 2..32 workgroups of one XCD per pair for small batches, teams of
 * 64 / 128 / 256 workgroups over all XCDs for one very large frame.
 *
 * Compile with -ffp-contract=off.
 */
#include "dvo_kernel_common.h"
#include "dvo_point_pk.h"
#include "dvo_palette.h"
#include "dvo_tiled_step.h"

#include <cstdlib>
#include <type_traits>

/* rounds of gathers in flight in the compact-form loop.  Measured (640x480x4x10, 1024 pairs): two 256-thread workgroups per
 * CU 555 k aligns/s at depth 2, 593 k at depth 3; one 512-thread workgroup 517 k / 510 k (its serial phases are exposed, not
 * its gathers) */
#ifndef DVO_WPE256
#define DVO_WPE256 2        /* two waves per SIMD = 256 registers: left to itself (1) the compiler parks two loop-invariant addresses of the
                               final pass in AGPRs (258 in all -> ONE wave per SIMD); forced, it spills those two to scratch outside the hot
                               loop (tests/test_kernel_registers.py pins both numbers) */
#endif
#ifndef DVO_P4_DEPTH
#define DVO_P4_DEPTH(BLOCK) ((BLOCK) == 256 ? 3 : 2)
#endif

namespace dvo {

/* the constant 100 MHz counter: one time base for the whole device (s_memtime counts per XCD) */
DVO_DEV unsigned long long stamp_real() {
    unsigned long long t;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    return t;
}

/* per-lane sums of one iteration (the sub-gradient policy of :724-920 needs g and the energy only).  WITH_H (round 5,
 * DVO_FLAG_NORMAL_MATRIX on the packed kernel): also the 21 entries of H = sum w J J^T (upper triangle, row-major; the pattern of
 * SolvePnP.cpp:168-182) -- exact products of two floats added in double, like g. */
template <bool WITH_H, bool WITH_E2 = false>
struct Acc7T {
    static constexpr bool with_h = WITH_H, with_e2 = WITH_E2;
    double g[6];
    double e2;
    int nvis;          /* wave-uniform: ballots */
    double H[WITH_H ? 21 : 1];
    E2Limbs l;         /* WITH_E2 (the step launches of the tiled / wide schedule, whose sums travel between launches and ranks): the exact
                          sum of eps^2 rides along (dvo_device_math.h: the energy without an order); untouched otherwise */
};
typedef Acc7T<false> Acc7;

template <typename ACC>
DVO_DEV void acc7_zero(ACC &a) {
#pragma unroll
    for (int k = 0; k < 6; k++) a.g[k] = 0.0;
    a.e2 = 0.0;
    a.nvis = 0;
    if constexpr (ACC::with_h) {
#pragma unroll
        for (int k = 0; k < 21; k++) a.H[k] = 0.0;
    }
    if constexpr (ACC::with_e2) e2_limbs_zero(a.l);
}
/* H += jw (x) J for one point: entry (i, j >= i) at i*6 - i*(i-1)/2 + (j-i) */
template <typename ACC>
DVO_DEV void acc_h_add(ACC &a, const double *jwd, const double *Jd) {
#pragma unroll
    for (int i = 0; i < 6; i++)
#pragma unroll
        for (int j = i; j < 6; j++) a.H[i * 6 - (i * (i - 1)) / 2 + (j - i)] = fma(jwd[i], Jd[j], a.H[i * 6 - (i * (i - 1)) / 2 + (j - i)]);
}

/* Where a level's now-frame texels are read from (chosen per level and pair, wave-uniform):
 *   TEX_G16  16-byte texels {DT, gx, gy, w} gathered from HBM / L2
 *   TEX_L16  the same texels, staged once per level into LDS ("LDS-staged image tiles"; the reference re-copies the three
 *            now images every iteration, SolveDVO.cpp:310,316-317,427) -- taken when the whole level fits beside its points
 *   TEX_P4   the compact form of dvo_palette.h: ONE 12-byte gather per point fetches the rank words of the pixel above, the
 *            pixel and the pixel below (24 pixels per 128-byte line instead of 8: half the memory requests); DT, w and the
 *            four neighbour values come from the level's palette in LDS, gx = 0.5*(P[r]-P[l]), gy = 0.5*(P[d]-P[u]) -- the
 *            builder verified per pixel that this reproduces the 16-byte texel bit for bit
 *   TEX_R16  (round 5) a coarse level whose compact form fits the LDS beside its palette and points: the centre ranks of the whole
 *            level, as 16-bit palette byte offsets with a one-pixel reflected border, staged once per level -- a point's five ranks
 *            are five ds_read_u16 at fixed offsets around its pixel and the level's iterations never touch its lines in HBM again
 *            (the reference re-copies the three now images every iteration, SolveDVO.cpp:310,316-317,427).  Reported as TEX_P4 with
 *            the flag DVO_TEXMODE_RANKS_LDS. */
enum { TEX_G16 = DVO_TEXMODE_GLOBAL16, TEX_L16 = DVO_TEXMODE_LDS16, TEX_P4 = DVO_TEXMODE_PAL4, TEX_R16 = 3 };

struct TexSrc {
    const char *g16;           /* this pair's level in HBM, tiled 16-byte texels */
    unsigned tile_col_bytes;   /* tiles_per_col * 128 */
    const char *l16;           /* LDS copy of the same bytes (TEX_L16) */
    const char *p4;            /* this pair's level in HBM, compact form (TEX_P4) */
    unsigned p4_col_bytes;     /* p4_tiles_per_col * 128 */
    /* TEX_R16: LDS byte addresses of the padded rank image -- element (yy, xx) at r16_org + (xx + 1) * r16_col_bytes + (yy + 1) * 2 --
     * and of the centre of the sentinel block (all five reads of a lane without a visible point land on the zero entry's offset) */
    unsigned r16_org, r16_col_bytes, r16_sent;
};

/* three consecutive dwords at any 4-byte boundary (one global_load_dwordx3) */
struct __attribute__((packed, aligned(4))) U3 { unsigned a, b, c; };

/* one round of the software pipeline: two points per lane */
template <int TEX> struct Round2 {
    v2f xn, yn, zn;    /* dehomogenised coordinates (finite dummies where not visible) */
    v4f t0, t1;        /* the two texels {DT, gx, gy, w} as loaded */
    bool vis0, vis1;
};
template <> struct Round2<TEX_P4> {
    v2f xn, yn, zn;
    U3 t0, t1;         /* rank words above / at / below the two pixels (the sentinel line for a lane without a visible point) */
};
template <> struct Round2<TEX_R16> {
    v2f xn, yn, zn;
    unsigned c0, u0, d0, l0, r0, c1, u1, d1, l1, r1;      /* palette byte offsets of the pixel and its four neighbours, per point */
};
typedef const __attribute__((address_space(3))) unsigned short lds_cushort;

/* LDS address of the palette = start of the dynamic LDS = size of the kernel's static block (checked at run time) */
typedef const __attribute__((address_space(3))) float lds_cfloat;
typedef const __attribute__((address_space(3))) v2f lds_cv2f;

/* byte offset of texel (yy, xx) in the tiled 16-byte texel image (texel_index() * 16 in seven instructions) */
DVO_DEV unsigned texel_byte_offset(int yy, int xx, unsigned tile_col_bytes /* tiles_per_col * 128 */) {
    static_assert(DVO_TILE_Y_LOG2 == 2 && DVO_TILE_X_LOG2 == 1, "written for 4x2 tiles");
    /* ((xx>>1)*tpc + (yy>>2))*128 + (xx&1)*64 + (yy&3)*16  ==  (xx>>1)*tpc*128 + yy*32 - (yy&3)*16 + (xx&1)*64 */
    /* 24-bit multiply: full rate (a 32-bit v_mul_lo_u32 costs four instruction slots); tile columns and their byte size fit 24 bits */
    return __umul24((unsigned)(xx >> 1), tile_col_bytes) + ((unsigned)yy << 5) - (((unsigned)yy & 3u) << 4) + (((unsigned)xx & 1u) << 6);
}

/* byte offset of the rank word ABOVE pixel (yy, xx) in the compact image (dvo_palette.h): the 12 bytes from there are
 * above / centre / below.  yy / 6 by multiplication (exact for yy < 98 000). */
DVO_DEV unsigned p4_byte_offset(int yy, int xx, unsigned p4_col_bytes /* p4_tiles_per_col * 128 */) {
    static_assert(DVO_P4_ROWS == 6, "written for 6 interior rows per line");
    /* (xx>>2)*col_bytes + ty*128 + (xx&3)*32 + (yy-6ty)*4  ==  (xx>>2)*col_bytes + (xx&3)*32 + yy*4 + ty*104; 24-bit multiplies (full
     * rate; yy < 2^16, 43691 < 2^16, a level's column of lines < 2^24 bytes) */
    const unsigned ty = __umul24((unsigned)yy, 43691u) >> 18;
    return __umul24((unsigned)(xx >> 2), p4_col_bytes) + 128u /* the sentinel line */ + (((unsigned)xx & 3u) << 5) + ((unsigned)yy << 2) + __umul24(ty, 104u);
}

struct LdsPoints {
    const uint2 *p;        /* {xx | yy << 16, Z}: 8 bytes per point, one ds_read_b64 */
    /* the same list in 4 bytes per point (dvo_device_math.h: pt4_decode), used instead when the level's list has a valid one:
     * twice as many points stay in LDS, the streamed remainder is half as many bytes */
    const unsigned *w4;    /* LDS copy (one ds_read_b32 per point) */
    const unsigned *g4;    /* the whole list in HBM (this workgroup's share) */
    const unsigned *hdr;   /* chunk headers: linear block index of point 64 k (wave-uniform: one scalar load per wave and round) */
};

template <bool LDS_SRC>
DVO_DEV void load_compact(const LdsPoints &lp, const uint2 *__restrict__ gpts, int j, unsigned &k, float &z) {
    const uint2 a = LDS_SRC ? lp.p[j] : gpts[j];
    k = a.x; z = __uint_as_float(a.y);
}

/* points streamed from HBM (beyond the LDS budget) are fetched ONE ROUND AHEAD of the round that projects them: a lane's
 * gather address depends on its point, so a point loaded inside the round would put two memory latencies in a row */
struct PointPf { uint2 p0, p1; unsigned h0, h1; /* 4-byte points: the chunk headers of the round's two point groups, loaded a round ahead */ };

/* stage 1 of a round: load, decode, project, issue the two gathers.  Straight-line code: a point whose z is outside the
 * range in which the fast reciprocal is proven exact (|z| < 2^-126, > 2^126, 0, inf, nan -- never in practice) is
 * treated as not visible here and reported through `any_odd` (wave-uniform); the caller then redoes the wave's whole
 * share of the iteration with the literal-division scalar code (accumulate_points_exact). */
/* chunk header of the 64-point chunk a wave's lanes sit in (i = the lane's point index: 64-aligned per wave, so wave-uniform) */
DVO_DEV unsigned pt4_header(const LdsPoints &lp, int i, int end) {
    const int chunk = __builtin_amdgcn_readfirstlane(min(i, end - 1)) >> 6;
    return lp.hdr[chunk];
}
struct PointPf4 { unsigned w0, w1; };

/* streamed reference points / final outputs: loads and stores of data that is touched once per pass.  DVO_NT_POINTS /
 * DVO_NT_FINAL (experiment builds) mark them non-temporal so that they do not displace the look-up lines in the L2 */
DVO_DEV uint2 stream_point(const uint2 *__restrict__ p) {
#if defined(DVO_NT_POINTS)
    typedef unsigned nt_u2 __attribute__((ext_vector_type(2)));
    const nt_u2 v = __builtin_nontemporal_load(reinterpret_cast<const nt_u2 *>(p));
    return make_uint2(v.x, v.y);
#else
    return *p;
#endif
}
DVO_DEV unsigned stream_word(const unsigned *__restrict__ p) {
#if defined(DVO_NT_POINTS)
    return __builtin_nontemporal_load(p);
#else
    return *p;
#endif
}
/* FULL: the caller knows that every lane of the wave has a point in this round (every round but a wave's last): no "past the end"
 * masks.  (make EXP=nodiet EXPDEFS=-DDVO_NO_VALU_DIET=1 builds the round-4 form of the compact-form issue stage for the A/B.) */
template <bool LDS_SRC, int TEX, bool PT4 = false, bool FULL = false>
DVO_DEV void round2_issue(const IterConst &c, const TexSrc &ts, const LdsPoints &lp, const uint2 *__restrict__ gpts,
                          int i0, int i1, int end, int step, PointPf &pf, Round2<TEX> &b, bool &any_odd, int &nvis) {
    const bool valid0 = FULL || i0 < end, valid1 = FULL || i1 < end;
    v2f xx, yy, Z;
    if constexpr (PT4) {
        unsigned w0, w1;
        if constexpr (LDS_SRC) {
            w0 = lp.w4[valid0 ? i0 : (end - 1)];
            w1 = lp.w4[valid1 ? i1 : (end - 1)];
        } else {
            w0 = pf.p0.x; w1 = pf.p1.x;                         /* fetched while the previous round was worked on */
            pf.p0.x = stream_word(lp.g4 + min(i0 + step, end - 1));
            pf.p1.x = stream_word(lp.g4 + min(i1 + step, end - 1));
        }
        float x0, y0, z0, x1, y1, z1;
        const unsigned L00 = pf.h0, L01 = pf.h1;                /* scalar loads issued a round ago: no wait here */
        pf.h0 = pt4_header(lp, i0 + step, end);
        pf.h1 = pt4_header(lp, i1 + step, end);
        pt4_decode(c.nby, c.inv_nby, c.half_inv_nby, w0, L00, x0, y0, z0);
        pt4_decode(c.nby, c.inv_nby, c.half_inv_nby, w1, L01, x1, y1, z1);
        xx.x = x0; xx.y = x1; yy.x = y0; yy.y = y1; Z.x = z0; Z.y = z1;
    } else {
    unsigned k0, k1;
    float z0, z1;
    if constexpr (LDS_SRC) {
        const int j0 = valid0 ? i0 : (end - 1), j1 = valid1 ? i1 : (end - 1);
        load_compact<true>(lp, gpts, j0, k0, z0);
        load_compact<true>(lp, gpts, j1, k1, z1);
    } else {
        k0 = pf.p0.x; z0 = __uint_as_float(pf.p0.y);            /* fetched while the previous round was worked on */
        k1 = pf.p1.x; z1 = __uint_as_float(pf.p1.y);
        pf.p0 = stream_point(gpts + min(i0 + step, end - 1));                  /* the next round's points (the last point again past the end) */
        pf.p1 = stream_point(gpts + min(i1 + step, end - 1));
    }
    xx.x = (float)(k0 & 0xffffu); xx.y = (float)(k1 & 0xffffu);
    yy.x = (float)(k0 >> 16);     yy.y = (float)(k1 >> 16);
    Z.x = z0; Z.y = z1;
    }
    const v2f X = (Z * (xx - c.pcx)) * c.pfx;                               /* :249 */
    const v2f Y = (Z * (yy - c.pcy)) * c.pfy;                               /* :250 */
    v2f xn, yn, zn, u, v;
#ifndef DVO_NO_VALU_DIET
    if constexpr (TEX == TEX_P4 || TEX == TEX_R16) {
        /* every condition of this path is a lane mask in scalar registers (dvo_point_pk.h): the degenerate-z flag, the visible counts
         * and the selects cost no vector instruction beyond the comparisons themselves.  Round 4 measured this diet as a LOSS (-2 %:
         * at the HBM ceiling a faster issue stage only deepened the queues); with round 5's shorter serial chain it is +3 % at
         * 640x480x4 (762 k -> 787 k aligns/s at 8192 pairs), +1-2 % at 1920x1080x5: profiles/r05_experiments/point_loop_ab.txt */
        lanemask odd;
        project_point2m(c, X, Y, Z, xn, yn, zn, u, v, odd);
        any_odd |= (odd != 0ull);
        int px0, py0, px1, py1;
        const lanemask in0 = pixel_in_range_mask(u.x, c.cols, px0) & pixel_in_range_mask(v.x, c.rows, py0);
        const lanemask in1 = pixel_in_range_mask(u.y, c.cols, px1) & pixel_in_range_mask(v.y, c.rows, py1);
        const lanemask vis0 = FULL ? in0 : (in0 & mask_lt_i32(i0, end)), vis1 = FULL ? in1 : (in1 & mask_lt_i32(i1, end));
        b.xn = xn; b.yn = yn; b.zn = zn;
        nvis += __popcll(vis0) + __popcll(vis1);
        if constexpr (TEX == TEX_R16) {
            const unsigned a0 = select_or(vis0, ts.r16_org + __umul24((unsigned)px0 + 1u, ts.r16_col_bytes) + ((unsigned)py0 << 1), ts.r16_sent);
            const unsigned a1 = select_or(vis1, ts.r16_org + __umul24((unsigned)px1 + 1u, ts.r16_col_bytes) + ((unsigned)py1 << 1), ts.r16_sent);
            b.u0 = *(lds_cushort *)(size_t)(a0);     b.c0 = *(lds_cushort *)(size_t)(a0 + 2u); b.d0 = *(lds_cushort *)(size_t)(a0 + 4u);
            b.l0 = *(lds_cushort *)(size_t)(a0 + 2u - ts.r16_col_bytes); b.r0 = *(lds_cushort *)(size_t)(a0 + 2u + ts.r16_col_bytes);
            b.u1 = *(lds_cushort *)(size_t)(a1);     b.c1 = *(lds_cushort *)(size_t)(a1 + 2u); b.d1 = *(lds_cushort *)(size_t)(a1 + 4u);
            b.l1 = *(lds_cushort *)(size_t)(a1 + 2u - ts.r16_col_bytes); b.r1 = *(lds_cushort *)(size_t)(a1 + 2u + ts.r16_col_bytes);
        } else {
            const unsigned o0 = select_or_zero(vis0, p4_byte_offset(py0, px0, ts.p4_col_bytes));     /* else the sentinel line */
            const unsigned o1 = select_or_zero(vis1, p4_byte_offset(py1, px1, ts.p4_col_bytes));
            b.t0 = *reinterpret_cast<const U3 *>(ts.p4 + o0);
            b.t1 = *reinterpret_cast<const U3 *>(ts.p4 + o1);
        }
        return;
    }
#endif
    bool odd0, odd1;
    project_point2(c, X, Y, Z, xn, yn, zn, u, v, odd0, odd1);
    /* a lane past the end of the list re-reads the last point: if THAT one is degenerate the wave takes the exact path as
     * well -- so nothing non-finite ever sits in a lane of a wave that stays on this path, visible or not */
    any_odd |= (__builtin_amdgcn_ballot_w64(odd0 || odd1) != 0ull);                               /* scalar: no branch */
    int px0, py0, px1, py1;
    const bool inx0 = pixel_in_range(u.x, c.cols, px0), iny0 = pixel_in_range(v.x, c.rows, py0);
    const bool inx1 = pixel_in_range(u.y, c.cols, px1), iny1 = pixel_in_range(v.y, c.rows, py1);
    /* `odd` is left out of the visibility on purpose: a wave with one odd lane discards everything this path computes */
    const bool vis0 = inx0 && iny0 && valid0;
    const bool vis1 = inx1 && iny1 && valid1;
    /* a lane without a visible point keeps its (finite: see any_odd) coordinates: its w and eps are exact zeros, so it adds
     * exact zeros to every sum (tests/test_gpu_packed_kernel.py::test_degenerate_depth_takes_the_exact_fallback, team shares) */
    b.xn = xn; b.yn = yn; b.zn = zn;
    if constexpr (TEX == TEX_R16) {
        nvis += __popcll(__builtin_amdgcn_ballot_w64(vis0)) + __popcll(__builtin_amdgcn_ballot_w64(vis1));
        /* address of the element ABOVE the pixel (the three of a column are 2 bytes apart); the sentinel block for a lane without
         * a visible point */
        unsigned a0 = ts.r16_org + __umul24((unsigned)px0 + 1u, ts.r16_col_bytes) + ((unsigned)py0 << 1);
        unsigned a1 = ts.r16_org + __umul24((unsigned)px1 + 1u, ts.r16_col_bytes) + ((unsigned)py1 << 1);
        a0 = vis0 ? a0 : ts.r16_sent;
        a1 = vis1 ? a1 : ts.r16_sent;
        b.u0 = *(lds_cushort *)(size_t)(a0);     b.c0 = *(lds_cushort *)(size_t)(a0 + 2u); b.d0 = *(lds_cushort *)(size_t)(a0 + 4u);
        b.l0 = *(lds_cushort *)(size_t)(a0 + 2u - ts.r16_col_bytes); b.r0 = *(lds_cushort *)(size_t)(a0 + 2u + ts.r16_col_bytes);
        b.u1 = *(lds_cushort *)(size_t)(a1);     b.c1 = *(lds_cushort *)(size_t)(a1 + 2u); b.d1 = *(lds_cushort *)(size_t)(a1 + 4u);
        b.l1 = *(lds_cushort *)(size_t)(a1 + 2u - ts.r16_col_bytes); b.r1 = *(lds_cushort *)(size_t)(a1 + 2u + ts.r16_col_bytes);
    } else if constexpr (TEX == TEX_P4) {
        nvis += __popcll(__builtin_amdgcn_ballot_w64(vis0)) + __popcll(__builtin_amdgcn_ballot_w64(vis1));
        unsigned o0 = p4_byte_offset(py0, px0, ts.p4_col_bytes);
        unsigned o1 = p4_byte_offset(py1, px1, ts.p4_col_bytes);
        o0 = vis0 ? o0 : 0u;                                   /* the sentinel line: DT = gx = gy = w = 0 */
        o1 = vis1 ? o1 : 0u;
        b.t0 = *reinterpret_cast<const U3 *>(ts.p4 + o0);
        b.t1 = *reinterpret_cast<const U3 *>(ts.p4 + o1);
    } else {
        b.vis0 = vis0; b.vis1 = vis1;
        unsigned o0 = texel_byte_offset(py0, px0, ts.tile_col_bytes);
        unsigned o1 = texel_byte_offset(py1, px1, ts.tile_col_bytes);
        o0 = vis0 ? o0 : 0u;
        o1 = vis1 ? o1 : 0u;
        const char *base = (TEX == TEX_L16) ? ts.l16 : ts.g16;
        b.t0 = *reinterpret_cast<const v4f *>(base + o0);
        b.t1 = *reinterpret_cast<const v4f *>(base + o1);
    }
}

/* rank * 8 + 8 * (signed rank step) in one instruction */
DVO_DEV unsigned lshl3_add(int d, unsigned c) {
    unsigned r;
    asm("v_lshl_add_u32 %0, %1, 3, %2" : "=v"(r) : "v"(d), "v"(c));
    return r;
}
/* TEX_P4: {DT, gx, gy, w} of the two pixels of a round from their rank words and the palette in LDS (at the compile-time LDS
 * address PAL: the look-ups use the instruction's offset field, no address add); the values arrive in free registers, so the
 * two pixels are paired and the gradients are packed subtractions / multiplications */
template <unsigned PAL>
DVO_DEV void p4_decode2(const U3 &t0, const U3 &t1, v2f &dt, v2f &gx, v2f &gy, v2f &w) {
    const unsigned c0 = t0.b & 0xfff8u, c1 = t1.b & 0xfff8u;
    const v2f pw0 = *(lds_cv2f *)(size_t)(PAL + c0);
    const v2f pw1 = *(lds_cv2f *)(size_t)(PAL + c1);
    v2f pu, pd, pr, pl;
    pu.x = *(lds_cfloat *)(size_t)(PAL + (t0.a & 0xfff8u));
    pu.y = *(lds_cfloat *)(size_t)(PAL + (t1.a & 0xfff8u));
    pd.x = *(lds_cfloat *)(size_t)(PAL + (t0.c & 0xfff8u));
    pd.y = *(lds_cfloat *)(size_t)(PAL + (t1.c & 0xfff8u));
    pr.x = *(lds_cfloat *)(size_t)(PAL + lshl3_add(__builtin_amdgcn_sbfe((int)t0.b, 16, 8), c0));
    pr.y = *(lds_cfloat *)(size_t)(PAL + lshl3_add(__builtin_amdgcn_sbfe((int)t1.b, 16, 8), c1));
    pl.x = *(lds_cfloat *)(size_t)(PAL + lshl3_add(((int)t0.b) >> 24, c0));
    pl.y = *(lds_cfloat *)(size_t)(PAL + lshl3_add(((int)t1.b) >> 24, c1));
    dt.x = pw0.x; dt.y = pw1.x; w.x = pw0.y; w.y = pw1.y;
    gx = (pr - pl) * 0.5f;          /* imageGradient, SolveDVO.cpp:1063-1098 */
    gy = (pd - pu) * 0.5f;
}

/* (double)(float)(J_k w) * (double)eps is exact, so fma(a,b,c) == c + a*b bit for bit (:719-720, :777) */
template <typename ACC>
DVO_DEV void acc7_add(ACC &a, const float *jw, const float *J, float eps) {
    const double e = (double)eps;
    double jwd[6];
#pragma unroll
    for (int k = 0; k < 6; k++) { jwd[k] = (double)jw[k]; a.g[k] = fma(jwd[k], e, a.g[k]); }
    a.e2 = fma(e, e, a.e2);
    if constexpr (ACC::with_e2) e2_limbs_add(a.l, eps);
    if constexpr (ACC::with_h) {
        double Jd[6];
#pragma unroll
        for (int k = 0; k < 6; k++) Jd[k] = (double)J[k];
        acc_h_add(a, jwd, Jd);
    }
}

/* the iteration's float pose into the constants, with literal indices: IterConst then splits into registers in the compiler's FIRST
 * pass over the kernel (a loop over k does so only once it is unrolled, and round 6's exact-energy block after the barrier then left a
 * 24-byte slice of it in scratch memory) */
DVO_DEV void iter_const_pose(IterConst &c, const float *Rf, const float *tf) {
    c.r[0] = uniform_f(Rf[0]); c.r[1] = uniform_f(Rf[1]); c.r[2] = uniform_f(Rf[2]);
    c.r[3] = uniform_f(Rf[3]); c.r[4] = uniform_f(Rf[4]); c.r[5] = uniform_f(Rf[5]);
    c.r[6] = uniform_f(Rf[6]); c.r[7] = uniform_f(Rf[7]); c.r[8] = uniform_f(Rf[8]);
    c.t[0] = uniform_f(tf[0]); c.t[1] = uniform_f(tf[1]); c.t[2] = uniform_f(tf[2]);
}

/* the exact sweep of an iteration whose energy the certificate left open (dvo_device_math.h: the energy without an order): the
 * residual alone, into the three limbs */
struct AccE2 {
    static constexpr bool with_h = false;
    E2Limbs l;
    int nvis;
};
DVO_DEV void acc7_add(AccE2 &a, const float *, const float *, float eps) { e2_limbs_add(a.l, eps); }

/* stage 2: weighted Jacobian rows + accumulation */
template <int TEX, unsigned PAL, typename ACC>
DVO_DEV void round2_compute(const IterConst &c, const Round2<TEX> &b, ACC &a) {
    v2f jw[6], J[6];
    float eps0, eps1;
    if constexpr (TEX == TEX_R16) {
        /* the ranks arrive as palette byte offsets: five look-ups per point at the compile-time palette address, no decoding */
        const v2f pw0 = *(lds_cv2f *)(size_t)(PAL + b.c0), pw1 = *(lds_cv2f *)(size_t)(PAL + b.c1);
        v2f pu, pd, pr, pl, dt, wt;
        pu.x = *(lds_cfloat *)(size_t)(PAL + b.u0); pu.y = *(lds_cfloat *)(size_t)(PAL + b.u1);
        pd.x = *(lds_cfloat *)(size_t)(PAL + b.d0); pd.y = *(lds_cfloat *)(size_t)(PAL + b.d1);
        pr.x = *(lds_cfloat *)(size_t)(PAL + b.r0); pr.y = *(lds_cfloat *)(size_t)(PAL + b.r1);
        pl.x = *(lds_cfloat *)(size_t)(PAL + b.l0); pl.y = *(lds_cfloat *)(size_t)(PAL + b.l1);
        dt.x = pw0.x; dt.y = pw1.x; wt.x = pw0.y; wt.y = pw1.y;
        const v2f gx = (pr - pl) * 0.5f, gy = (pd - pu) * 0.5f;          /* imageGradient, SolveDVO.cpp:1063-1098 */
        eps0 = dt.x; eps1 = dt.y;
        jacobian_weighted2p(c, b.xn, b.yn, b.zn, gx, gy, wt, jw, ACC::with_h ? J : nullptr);
    } else if constexpr (TEX == TEX_P4) {
        v2f dt, gx, gy, wt;
        p4_decode2<PAL>(b.t0, b.t1, dt, gx, gy, wt);        /* zeros for a lane without a visible point (sentinel) */
        eps0 = dt.x; eps1 = dt.y;
        jacobian_weighted2p(c, b.xn, b.yn, b.zn, gx, gy, wt, jw, ACC::with_h ? J : nullptr);
    } else {
        a.nvis += __popcll(__builtin_amdgcn_ballot_w64(b.vis0)) + __popcll(__builtin_amdgcn_ballot_w64(b.vis1));
        eps0 = b.vis0 ? b.t0.x : 0.0f; eps1 = b.vis1 ? b.t1.x : 0.0f;
        const float w0 = b.vis0 ? b.t0.w : 0.0f, w1 = b.vis1 ? b.t1.w : 0.0f;
        /* a lane without a visible point gathered texel 0 of the level: its gradient must not reach the sums either (a caller-
         * supplied image may hold Inf / NaN there, and NaN * 0 is NaN; the reference skips such points, :371) */
        const float gx0 = b.vis0 ? b.t0.y : 0.0f, gx1 = b.vis1 ? b.t1.y : 0.0f;
        const float gy0 = b.vis0 ? b.t0.z : 0.0f, gy1 = b.vis1 ? b.t1.z : 0.0f;
        jacobian_weighted2(c, b.xn, b.yn, b.zn, gx0, gx1, gy0, gy1, w0, w1, jw, ACC::with_h ? J : nullptr);
    }
    const double e0 = (double)eps0, e1 = (double)eps1;
    if constexpr (ACC::with_e2) { e2_limbs_add(a.l, eps0); e2_limbs_add(a.l, eps1); }
    if constexpr (!ACC::with_h) {
#pragma unroll
        for (int k = 0; k < 6; k++) {
            a.g[k] = fma((double)jw[k].x, e0, a.g[k]);
            a.g[k] = fma((double)jw[k].y, e1, a.g[k]);
        }
        a.e2 = fma(e0, e0, a.e2);
        a.e2 = fma(e1, e1, a.e2);
    } else {
        /* the other 21 of the "21 + 6" accumulators (a lane without a visible point has jw = 0 and a finite J: exact zeros).  One
         * point after the other, with a scheduling fence between them: interleaved, the 24 widened values of both points are live
         * at once beside the 56 accumulator registers and the loop spills */
        {
            double jd[6], Jd[6];
#pragma unroll
            for (int k = 0; k < 6; k++) { jd[k] = (double)jw[k].x; Jd[k] = (double)J[k].x; a.g[k] = fma(jd[k], e0, a.g[k]); }
            a.e2 = fma(e0, e0, a.e2);
            acc_h_add(a, jd, Jd);
        }
        __builtin_amdgcn_sched_barrier(0);
        {
            double jd[6], Jd[6];
#pragma unroll
            for (int k = 0; k < 6; k++) { jd[k] = (double)jw[k].y; Jd[k] = (double)J[k].y; a.g[k] = fma(jd[k], e1, a.g[k]); }
            a.e2 = fma(e1, e1, a.e2);
            acc_h_add(a, jd, Jd);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

/* the per-point phase of one iteration over points [first, end): rounds of 2*BLOCK points, lane `lane_off` of the
 * round takes points lane_off and BLOCK + lane_off; software-pipelined over rounds with two named buffers */
template <int BLOCK, bool LDS_SRC, int TEX, int DEPTH = 2, unsigned PAL = 0, bool PT4 = false, typename ACC = Acc7>
DVO_DEV void accumulate_points2(const IterConst &c, const TexSrc &ts, const LdsPoints &lp, const uint2 *__restrict__ gpts,
                                int first, int end, int lane_off, ACC &a, bool &any_odd) {
    if (first >= end) return;
    constexpr int STEP = 2 * BLOCK;
    /* rounds in which THIS wave still has a point (wave-uniform) */
    const int wave_off = __builtin_amdgcn_readfirstlane(lane_off - (int)(threadIdx.x & 63));
    const int n_rounds = (end - first - wave_off + STEP - 1) / STEP;
    if (n_rounds <= 0) return;
#define DVO_ISSUE(buf, k) round2_issue<LDS_SRC, TEX, PT4>(c, ts, lp, gpts, base + (k) * STEP, base + (k) * STEP + BLOCK, end, STEP, pf, buf, any_odd, a.nvis)
#ifndef DVO_NO_VALU_DIET
#define DVO_ISSUE_FULL(buf, k) round2_issue<LDS_SRC, TEX, PT4, true>(c, ts, lp, gpts, base + (k) * STEP, base + (k) * STEP + BLOCK, end, STEP, pf, buf, any_odd, a.nvis)
#else
#define DVO_ISSUE_FULL(buf, k) DVO_ISSUE(buf, k)
#endif
#define DVO_COMPUTE(buf) round2_compute<TEX, PAL, ACC>(c, buf, a)
    int base = first + lane_off;
    PointPf pf;
    if constexpr (PT4) { pf.h0 = pt4_header(lp, base, end); pf.h1 = pt4_header(lp, base + BLOCK, end); }
    if constexpr (!LDS_SRC) {
        if constexpr (PT4) { pf.p0.x = stream_word(lp.g4 + min(base, end - 1)); pf.p1.x = stream_word(lp.g4 + min(base + BLOCK, end - 1)); }
        else { pf.p0 = stream_point(gpts + min(base, end - 1)); pf.p1 = stream_point(gpts + min(base + BLOCK, end - 1)); }
    }
    if constexpr (DEPTH == 3) {
        /* gathers issued TWO rounds ahead of the arithmetic that consumes them: with half the requests per point (TEX_P4) the
         * loop is no longer bound by the request rate but by the latency of a gather that misses the L2 (~2 rounds of
         * arithmetic); three named buffers, branch-free steady state (the wait counters stay exact), branches in the tail */
        Round2<TEX> A, B, C;
        DVO_ISSUE(A, 0);
        if (n_rounds > 1) DVO_ISSUE(B, 1);
        int r = 0;
        for (; r + 4 < n_rounds; r += 3) {                     /* rounds r + 2 and r + 3 are not the wave's last */
            DVO_ISSUE_FULL(C, 2); DVO_COMPUTE(A);
            DVO_ISSUE_FULL(A, 3); DVO_COMPUTE(B);
            DVO_ISSUE(B, 4); DVO_COMPUTE(C);
            base += 3 * STEP;
        }
        /* A = round r, B = round r+1 (if any); 1..4 rounds left */
        const int left = n_rounds - r;
        if (left >= 3) DVO_ISSUE(C, 2);
        DVO_COMPUTE(A);
        if (left >= 2) {
            if (left >= 4) DVO_ISSUE(A, 3);
            DVO_COMPUTE(B);
            if (left >= 3) {
                DVO_COMPUTE(C);
                if (left >= 4) DVO_COMPUTE(A);
            }
        }
    } else {
        Round2<TEX> A, B;
        DVO_ISSUE(A, 0);
        int r = 0;
        for (; r + 2 < n_rounds; r += 2) {
            DVO_ISSUE_FULL(B, 1); DVO_COMPUTE(A);
            DVO_ISSUE(A, 2); DVO_COMPUTE(B);
            base += 2 * STEP;
        }
        if (r + 1 < n_rounds) {
            DVO_ISSUE(B, 1);
            DVO_COMPUTE(A);
            DVO_COMPUTE(B);
        } else {
            DVO_COMPUTE(A);
        }
    }
#undef DVO_ISSUE
#undef DVO_ISSUE_FULL
#undef DVO_COMPUTE
}

/* {DT, gx, gy, w} of pixel (yy, xx) from the compact form: the scalar twin of p4_decode2 (cold paths only) */
DVO_DEV float4 p4_texel(const TexSrc &ts, const float2 *pal_lds, int yy, int xx) {
    const U3 w = *reinterpret_cast<const U3 *>(ts.p4 + p4_byte_offset(yy, xx, ts.p4_col_bytes));
    const int c = (int)((w.b >> 3) & 0x1fffu);
    const int cr = c + __builtin_amdgcn_sbfe((int)w.b, 16, 8), cl = c + (((int)w.b) >> 24);
    const float2 pc = pal_lds[c];
    const float pr = pal_lds[cr].x, pl = pal_lds[cl].x, pu = pal_lds[(w.a >> 3) & 0x1fffu].x, pd = pal_lds[(w.c >> 3) & 0x1fffu].x;
    return make_float4(pc.x, (pr - pl) * 0.5f, (pd - pu) * 0.5f, pc.y);
}

/* The same sums with the literal-division scalar code (dvo_device_math.h: project_point, jacobian_row) over THIS wave's
 * points of [first, end) -- taken only when one of them has a degenerate z.  Plain loop, not pipelined: never hot.
 * P4: the level is read through its compact form (the 16-byte texels of such a level may not exist). */
template <int BLOCK, bool LDS_SRC, bool P4, bool PT4 = false, typename ACC = Acc7>
DVO_DEV void accumulate_points_exact(const IterConst &c, const char *__restrict__ tex, const TexSrc &ts, const float2 *pal_lds,
                                     const LdsPoints &lp, const uint2 *__restrict__ gpts, int first, int end, int lane_off, ACC &a) {
    for (int i = first + lane_off; __builtin_amdgcn_ballot_w64(i < end) != 0ull; i += BLOCK) {
        const bool valid = i < end;
        float X, Y, Z, xn, yn, zn, u, v;
        if constexpr (PT4) {
            const int j = valid ? i : (end - 1);
            const unsigned w = LDS_SRC ? lp.w4[j] : lp.g4[j];
            float xf, yf;
            pt4_decode(c.nby, c.inv_nby, c.half_inv_nby, w, pt4_header(lp, i, end), xf, yf, Z);
            X = Z * (xf - c.pcx) * c.pfx;                            /* :249 */
            Y = Z * (yf - c.pcy) * c.pfy;                            /* :250 */
        } else {
        unsigned k; float z;
        load_compact<LDS_SRC>(lp, gpts, valid ? i : (end - 1), k, z);
        expand_compact(c, k, z, X, Y, Z);
        }
        const bool vis = project_point(c, X, Y, Z, xn, yn, zn, u, v) && valid;
        a.nvis += __popcll(__builtin_amdgcn_ballot_w64(vis));
        if (vis) {
            const float4 t = P4 ? p4_texel(ts, pal_lds, (int)v, (int)u)
                                : reinterpret_cast<const float4 *>(tex)[texel_index((int)v, (int)u, c.tiles_per_col)];
            float J[6], jw[6];
            jacobian_row(c, xn, yn, zn, t.y, t.z, J);
#pragma unroll
            for (int q = 0; q < 6; q++) jw[q] = J[q] * t.w;
            acc7_add(a, jw, J, t.x);
        }
    }
}

/* a wave's share of a level through that code, whichever forms the level's points and texels are read in */
template <int BLOCK, typename ACC>
DVO_DEV void sweep_literal(int mode, bool p4_partial, bool pt4, const IterConst &c, const char *__restrict__ tex, const TexSrc &ts, const float2 *pal_lds,
                           const LdsPoints &lp, const uint2 *__restrict__ gpts, int n_lds, int N, int lane_off, ACC &a) {
    if (mode == DVO_TEXMODE_PAL4 && p4_partial && pt4) {          /* partial form: the 16-byte texels are the complete image */
        accumulate_points_exact<BLOCK, true, false, true, ACC>(c, tex, ts, pal_lds, lp, gpts, 0, n_lds, lane_off, a);
        accumulate_points_exact<BLOCK, false, false, true, ACC>(c, tex, ts, pal_lds, lp, gpts, n_lds, N, lane_off, a);
    } else if (mode == DVO_TEXMODE_PAL4 && p4_partial) {
        accumulate_points_exact<BLOCK, true, false, false, ACC>(c, tex, ts, pal_lds, lp, gpts, 0, n_lds, lane_off, a);
        accumulate_points_exact<BLOCK, false, false, false, ACC>(c, tex, ts, pal_lds, lp, gpts, n_lds, N, lane_off, a);
    } else if (mode == DVO_TEXMODE_PAL4 && pt4) {
        accumulate_points_exact<BLOCK, true, true, true, ACC>(c, tex, ts, pal_lds, lp, gpts, 0, n_lds, lane_off, a);
        accumulate_points_exact<BLOCK, false, true, true, ACC>(c, tex, ts, pal_lds, lp, gpts, n_lds, N, lane_off, a);
    } else if (mode == DVO_TEXMODE_PAL4) {
        accumulate_points_exact<BLOCK, true, true, false, ACC>(c, tex, ts, pal_lds, lp, gpts, 0, n_lds, lane_off, a);
        accumulate_points_exact<BLOCK, false, true, false, ACC>(c, tex, ts, pal_lds, lp, gpts, n_lds, N, lane_off, a);
    } else {
        accumulate_points_exact<BLOCK, true, false, false, ACC>(c, tex, ts, pal_lds, lp, gpts, 0, n_lds, lane_off, a);
        accumulate_points_exact<BLOCK, false, false, false, ACC>(c, tex, ts, pal_lds, lp, gpts, n_lds, N, lane_off, a);
    }
}

/* ---- finalEpsilons / finalReprojections of the best iterate (:703-704, :1002-1003) --------------------------------------
 * One more projection of the finest level's points and one DT look-up each, over the COMPACT point list -- the copy the level
 * already holds in LDS (the streamed tail fetched one round ahead), in its 16 x 16-block order, so that the look-ups of a wave
 * share memory lines exactly as in the iterations -- with the look-ups issued one round ahead of the stores that consume them.
 * The two arrays are therefore written in the compact list's order; dvo_get_final_outputs hands them out in the reference's
 * order (LevelSlab.cidx, final_permute_kernel).  (Round 2 walked the 3 x N float list in the reference's order, four points per
 * trip with nothing in flight across trips: 11 % of the whole alignment and 8 k of its 61 k memory requests -- the kernel sits
 * on the request ceiling) */
struct Final2 {
    v2f u, v, zn;
    unsigned w0, w1;        /* TEX_P4: centre rank word; else: bits of DT */
    bool vis0, vis1;
};
template <bool LDS_SRC, int TEX, bool PT4 = false>
DVO_DEV void final2_issue(const IterConst &c, const TexSrc &ts, const LdsPoints &lp, const uint2 *__restrict__ gpts,
                          int i0, int i1, int end, int step, PointPf &pf, Final2 &b) {
    float X0, Y0, Z0, X1, Y1, Z1;
    if constexpr (PT4) {
        unsigned w0, w1;
        if constexpr (LDS_SRC) {
            w0 = lp.w4[min(i0, end - 1)]; w1 = lp.w4[min(i1, end - 1)];
        } else {
            w0 = pf.p0.x; w1 = pf.p1.x;
            pf.p0.x = stream_word(lp.g4 + min(i0 + step, end - 1));
            pf.p1.x = stream_word(lp.g4 + min(i1 + step, end - 1));
        }
        float xf, yf;
        pt4_decode(c.nby, c.inv_nby, c.half_inv_nby, w0, pt4_header(lp, i0, end), xf, yf, Z0);
        X0 = Z0 * (xf - c.pcx) * c.pfx; Y0 = Z0 * (yf - c.pcy) * c.pfy;                 /* :249-250 */
        pt4_decode(c.nby, c.inv_nby, c.half_inv_nby, w1, pt4_header(lp, i1, end), xf, yf, Z1);
        X1 = Z1 * (xf - c.pcx) * c.pfx; Y1 = Z1 * (yf - c.pcy) * c.pfy;
    } else {
        unsigned k0, k1;
        float z0, z1;
        if constexpr (LDS_SRC) {
            load_compact<true>(lp, gpts, min(i0, end - 1), k0, z0);
            load_compact<true>(lp, gpts, min(i1, end - 1), k1, z1);
        } else {
            k0 = pf.p0.x; z0 = __uint_as_float(pf.p0.y);
            k1 = pf.p1.x; z1 = __uint_as_float(pf.p1.y);
            pf.p0 = stream_point(gpts + min(i0 + step, end - 1));                  /* the next round's points */
            pf.p1 = stream_point(gpts + min(i1 + step, end - 1));
        }
        expand_compact(c, k0, z0, X0, Y0, Z0);
        expand_compact(c, k1, z1, X1, Y1, Z1);
    }
    /* one point at a time, scalar-register pose operands: the packed form would want the pose in vector register pairs, and
     * the 256-thread shape has none to spare (tests/test_kernel_registers.py) */
    {
        float xs, ys, zs, us, vs;
        project_point(c, X0, Y0, Z0, xs, ys, zs, us, vs);
        b.u.x = us; b.v.x = vs; b.zn.x = zs;
        project_point(c, X1, Y1, Z1, xs, ys, zs, us, vs);
        b.u.y = us; b.v.y = vs; b.zn.y = zs;
    }
    int px0, py0, px1, py1;
    const bool inx0 = pixel_in_range(b.u.x, c.cols, px0), iny0 = pixel_in_range(b.v.x, c.rows, py0);
    const bool inx1 = pixel_in_range(b.u.y, c.cols, px1), iny1 = pixel_in_range(b.v.y, c.rows, py1);
    b.vis0 = inx0 && iny0; b.vis1 = inx1 && iny1;
    if constexpr (TEX == TEX_P4) {
        const unsigned o0 = b.vis0 ? p4_byte_offset(py0, px0, ts.p4_col_bytes) + 4u : 0u;      /* not visible: the sentinel line (DT = 0) */
        const unsigned o1 = b.vis1 ? p4_byte_offset(py1, px1, ts.p4_col_bytes) + 4u : 0u;
        b.w0 = *reinterpret_cast<const unsigned *>(ts.p4 + o0);
        b.w1 = *reinterpret_cast<const unsigned *>(ts.p4 + o1);
    } else {
        const unsigned o0 = b.vis0 ? texel_byte_offset(py0, px0, ts.tile_col_bytes) : 0u;
        const unsigned o1 = b.vis1 ? texel_byte_offset(py1, px1, ts.tile_col_bytes) : 0u;
        const char *base = (TEX == TEX_L16) ? ts.l16 : ts.g16;
        b.w0 = *reinterpret_cast<const unsigned *>(base + o0);
        b.w1 = *reinterpret_cast<const unsigned *>(base + o1);
    }
}
template <int TEX, unsigned PAL>
DVO_DEV void final2_store(const Final2 &b, int i0, int i1, int end, float *__restrict__ fe, float *__restrict__ fr, const TexSrc &ts, bool partial) {
    float e0, e1;
    if constexpr (TEX == TEX_P4) {
        e0 = *(lds_cfloat *)(size_t)(PAL + (b.w0 & 0xfff8u));
        e1 = *(lds_cfloat *)(size_t)(PAL + (b.w1 & 0xfff8u));
        /* a partial compact form (dvo_palette.h): a pixel it cannot express decodes to NaN -- its DT comes from the image's 16-byte
         * texels (wave-uniform test; the rare lane pays one more load) */
        if (partial && __builtin_amdgcn_ballot_w64((e0 != e0) || (e1 != e1)) != 0ull) {
            if (e0 != e0) e0 = *reinterpret_cast<const float *>(ts.g16 + texel_byte_offset((int)b.v.x, (int)b.u.x, ts.tile_col_bytes));
            if (e1 != e1) e1 = *reinterpret_cast<const float *>(ts.g16 + texel_byte_offset((int)b.v.y, (int)b.u.y, ts.tile_col_bytes));
        }
    } else {
        e0 = b.vis0 ? __uint_as_float(b.w0) : 0.0f;
        e1 = b.vis1 ? __uint_as_float(b.w1) : 0.0f;
    }
#if defined(DVO_NT_FINAL)
    if (i0 < end) {
        __builtin_nontemporal_store(e0, fe + i0);
        float *q = fr + 3 * (size_t)i0;
        __builtin_nontemporal_store(b.u.x, q); __builtin_nontemporal_store(b.v.x, q + 1); __builtin_nontemporal_store(b.zn.x, q + 2);
    }
    if (i1 < end) {
        __builtin_nontemporal_store(e1, fe + i1);
        float *q = fr + 3 * (size_t)i1;
        __builtin_nontemporal_store(b.u.y, q); __builtin_nontemporal_store(b.v.y, q + 1); __builtin_nontemporal_store(b.zn.y, q + 2);
    }
#else
    if (i0 < end) {
        fe[i0] = e0;
        U3 o; o.a = __float_as_uint(b.u.x); o.b = __float_as_uint(b.v.x); o.c = __float_as_uint(b.zn.x);
        *reinterpret_cast<U3 *>(fr + 3 * (size_t)i0) = o;
    }
    if (i1 < end) {
        fe[i1] = e1;
        U3 o; o.a = __float_as_uint(b.u.y); o.b = __float_as_uint(b.v.y); o.c = __float_as_uint(b.zn.y);
        *reinterpret_cast<U3 *>(fr + 3 * (size_t)i1) = o;
    }
#endif
}
/* compact points [first, end) of this workgroup's share; outputs at fe[i], fr[3 i] */
template <int BLOCK, bool LDS_SRC, int TEX, unsigned PAL, bool PT4 = false>
DVO_DEV void final_outputs2(const IterConst &c, const TexSrc &ts, const LdsPoints &lp, const uint2 *__restrict__ gpts, int first, int end,
                            float *__restrict__ fe, float *__restrict__ fr, bool partial = false) {
    constexpr int STEP = 2 * BLOCK;
    const int tid = threadIdx.x;
    const int wave_first = first + __builtin_amdgcn_readfirstlane(tid & ~63);
    if (wave_first >= end) return;                              /* wave-uniform */
    const int n_rounds = (end - wave_first + STEP - 1) / STEP;  /* rounds in which this wave still has a point */
    int base = first + tid;
    PointPf pf;
    if constexpr (!LDS_SRC) {
        if constexpr (PT4) { pf.p0.x = stream_word(lp.g4 + min(base, end - 1)); pf.p1.x = stream_word(lp.g4 + min(base + BLOCK, end - 1)); }
        else { pf.p0 = stream_point(gpts + min(base, end - 1)); pf.p1 = stream_point(gpts + min(base + BLOCK, end - 1)); }
    }
    Final2 A, B;
    final2_issue<LDS_SRC, TEX, PT4>(c, ts, lp, gpts, base, base + BLOCK, end, STEP, pf, A);
    int r = 0;
#pragma clang loop unroll(disable)
    for (; r + 2 < n_rounds; r += 2) {
        final2_issue<LDS_SRC, TEX, PT4>(c, ts, lp, gpts, base + STEP, base + STEP + BLOCK, end, STEP, pf, B);
        final2_store<TEX, PAL>(A, base, base + BLOCK, end, fe, fr, ts, partial);
        final2_issue<LDS_SRC, TEX, PT4>(c, ts, lp, gpts, base + 2 * STEP, base + 2 * STEP + BLOCK, end, STEP, pf, A);
        final2_store<TEX, PAL>(B, base + STEP, base + STEP + BLOCK, end, fe, fr, ts, partial);
        base += 2 * STEP;
    }
    if (r + 1 < n_rounds) {
        final2_issue<LDS_SRC, TEX, PT4>(c, ts, lp, gpts, base + STEP, base + STEP + BLOCK, end, STEP, pf, B);
        final2_store<TEX, PAL>(A, base, base + BLOCK, end, fe, fr, ts, partial);
        final2_store<TEX, PAL>(B, base + STEP, base + STEP + BLOCK, end, fe, fr, ts, partial);
    } else {
        final2_store<TEX, PAL>(A, base, base + BLOCK, end, fe, fr, ts, partial);
    }
}

/* Fixed-shape reduction of the 7 double sums + the visible count, first half: every wave leaves its eight totals in
 * red[wave][0..5] g, [6] sum eps^2, [7] visible points.  After the workgroup barrier lane k < 8 of the waves that go on (wave 0:
 * the update; wave 1: the bookkeeping) adds the waves' rows in wave order into a register (block_sum8) -- round 5: no second
 * pass through LDS, no second barrier. */
template <typename ACC>
DVO_DEV void wave_sums7(const ACC &a, double (*red)[8]) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double d[8];
#pragma unroll
    for (int k = 0; k < 6; k++) d[k] = a.g[k];
    d[6] = a.e2;
    d[7] = 0.0;
#ifndef DVO_NO_DPP_REDUCE
    wave_reduce_scatter8_dpp(d);
    const int idx = reduce_scatter8_dpp_index(lane);
    if (lane < 8 && idx < 7) red[wave][idx] = d[0];
    if (lane == 8) red[wave][7] = (double)a.nvis;
#else
    wave_reduce_scatter<double, 8>(d);
    const int idx = lane >> 3;
    if ((lane & 7) == 0 && idx < 7) red[wave][idx] = d[0];
    if (lane == 1) red[wave][7] = (double)a.nvis;
#endif
}
template <int BLOCK>
DVO_DEV double block_sum8(const double (*red)[8], int k /* 0..7 */) {
    double v[BLOCK / 64];
#pragma unroll
    for (int w = 0; w < BLOCK / 64; w++) v[w] = red[w][k];      /* all reads in flight before the first addition */
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < BLOCK / 64; w++) s += v[w];
    return s;
}

/* the three limbs of a wave -> red[wave][0..2] (integers below 2^53: every addition on the way is exact) */
template <typename ACC>
DVO_DEV void wave_sums_e2(const ACC &a, double (*red)[8]) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double d[8];
    d[0] = (double)a.l.l0; d[1] = (double)a.l.l1; d[2] = (double)a.l.l2;
#pragma unroll
    for (int k = 3; k < 8; k++) d[k] = 0.0;
    wave_reduce_scatter8_dpp(d);
    const int idx = reduce_scatter8_dpp_index(lane);
    if (lane < 8) red[wave][idx] = d[0];
}

/* ---- teams: G workgroups share ONE frame pair (small batches) ------------------------------------------------------
 * With fewer pairs than compute units a launch of one workgroup per pair leaves most of the GPU idle (C4's per-GPU share
 * is 32 pairs on 256 CUs).  In team mode G workgroups take contiguous index ranges of every level's point list; after
 * its own block reduction each member publishes its 8 sums in the team's slot (HBM/L2), a counter tells when all G are
 * there, and EVERY member then adds the G partials in the same fixed order and runs the same double-precision update:
 * identical bits on all members, no broadcast step, one synchronisation per iteration.  The members of a team are placed
 * on ONE XCD (workgroups are dispatched round-robin over the 8 XCDs, so workgroups b, b+8, b+16, ... share an L2): the
 * exchange stays inside that XCD's L2.  Slots are double-buffered by the parity of the exchange index: a member can
 * only reach exchange e+2 after all members have passed e+1, i.e. after everybody has read the slots of e.
 * All G*pairs workgroups must be resident at once (the host launches at most one per CU); a bounded spin turns a broken
 * assumption into an error flag instead of a hang. */
#define DVO_TEAM_MAX 32
/* One exchange, executed by wave 0 of every member.  Each of a member's 8 sums travels as a 16-byte record {value, tag}
 * (tag = exchange index + 1) written with ONE 16-byte store, so a reader that sees the tag sees the value: no separate
 * arrival counter, no wait for store acknowledgements, no fences.  All accesses are agent-scope atomics (sc1: coherent
 * per location across the XCDs' L2s).  Agent-scope release / acquire FENCES are what must be avoided here: on gfx950 they
 * are buffer_wbl2 / buffer_inv of the whole L2, per poll (measured: batch 32 ran 2x slower than without teams, and the
 * other workgroups' texels were thrown out of the L2).  The wave polls all G*8 records at once (lane m*8+k reads record k
 * of member m) until every tag matches (team_exchange below). */
typedef unsigned v4u __attribute__((ext_vector_type(4)));
/* record = {value lo32, tag, value hi32, tag}: each 8-byte half carries the tag, so the record is consistent even if the
 * 16-byte store were performed as two 8-byte pieces */
DVO_DEV void team_store_rec(v4u *p, v4u r) { asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(r) : "memory"); }
/* A record for members that are KNOWN to sit on this workgroup's XCD (same HW_REG_XCC_ID, checked once per launch by a hand-shake
 * over the sc1 form: align_fused2_kernel).  A plain store leaves the line in that XCD's L2, which all its CUs share, and the
 * pollers' sc1 loads (which bypass only their L1) are served from there: 820 instead of 1260 cycles per hop between two
 * workgroups of one XCD -- and NEVER visible to a poller on another XCD (tools/exhaustive/xcd_handoff.hip, profiles/r05_xcd_handoff/:
 * every cross-XCD hop timed out), which is why the placement is verified, not assumed. */
DVO_DEV void team_store_rec_same_xcd(v4u *p, v4u r) { asm volatile("global_store_dwordx4 %0, %1, off" :: "v"(p), "v"(r) : "memory"); }
DVO_DEV unsigned hw_xcc_id() {
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    return x & 0xfu;
}
/* One poll round: the loads of all the lane's records issued together, then ONE wait -- inside one asm statement, so that the
 * compiler never sees (and never copies) a register whose load is still in flight.  Teams of up to 8 members need one record per
 * lane; larger ones poll four (addresses of records beyond the team's are clamped by the caller). */
DVO_DEV void team_poll1(v4u &r0, const v4u *p0) {
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(r0) : "v"(p0) : "memory");
}
DVO_DEV void team_poll4(v4u (&r)[4], const v4u *p0, const v4u *p1, const v4u *p2, const v4u *p3) {
    asm volatile("global_load_dwordx4 %0, %4, off sc1\n\tglobal_load_dwordx4 %1, %5, off sc1\n\t"
                 "global_load_dwordx4 %2, %6, off sc1\n\tglobal_load_dwordx4 %3, %7, off sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]) : "v"(p0), "v"(p1), "v"(p2), "v"(p3) : "memory");
}
DVO_DEV double team_rec_value(const v4u &r) { return __longlong_as_double((long long)(((unsigned long long)r.z << 32) | r.x)); }
/* One exchange, executed by wave 0 of every member (round 5: everything in registers, lane-parallel).  In: lane k < 8 holds this
 * member's sum k.  Out: EVERY lane L holds the team total of sum L & 7.  Lane m*8 + k polls record k of member m (lanes L + 64 q
 * for teams of more than 8), all loads of a round in flight together; once every tag matches, a lane adds its up to four records
 * in order and three lane-permute steps (8, 16, 32) add the members that share L & 7 -- the same fixed order on every member, so
 * identical bits on all of them.  (Rounds 2-4 parked the G x 8 values in LDS and lane 0 added them one by one: 8 x G dependent
 * additions and as many LDS reads in the serial chain of every iteration.) */
DVO_DEV double team_exchange(double mine, v4u *buf /* this pair's [2][DVO_TEAM_MAX][8] records */,
                             int member, int G, unsigned epoch, int *err, bool publish = true, bool same_xcd = false) {
    const int lane = threadIdx.x & 63;
    v4u *base = buf + (size_t)(epoch & 1u) * DVO_TEAM_MAX * 8;
    const unsigned tag = epoch + 1u;
    if (lane < 8 && publish) {
        const unsigned long long bits = (unsigned long long)__double_as_longlong(mine);
        v4u rec;
        rec.x = (unsigned)bits; rec.y = tag; rec.z = (unsigned)(bits >> 32); rec.w = tag;
        if (same_xcd) team_store_rec_same_xcd(base + member * 8 + lane, rec);      /* wave-uniform */
        else team_store_rec(base + member * 8 + lane, rec);
    }
    const int n_rec = G * 8;                       /* <= 256: lane L polls records L, L + 64, L + 128, L + 192 */
    const bool big = n_rec > 64;                   /* wave-uniform */
    v4u r[4];
#pragma unroll
    for (int q = 0; q < 4; q++) r[q] = v4u{0u, 0u, 0u, 0u};
    int spins = 0;
    for (;;) {
        bool ok;
        if (!big) {
            team_poll1(r[0], base + ((lane < n_rec) ? lane : 0));
            ok = (lane >= n_rec) || ((r[0].y == tag) && (r[0].w == tag));
        } else {
            team_poll4(r, base + lane, base + ((lane + 64 < n_rec) ? lane + 64 : 0), base + ((lane + 128 < n_rec) ? lane + 128 : 0),
                       base + ((lane + 192 < n_rec) ? lane + 192 : 0));
            ok = true;
#pragma unroll
            for (int q = 0; q < 4; q++) ok = ok && ((lane + 64 * q >= n_rec) || ((r[q].y == tag) && (r[q].w == tag)));
        }
        if (__builtin_amdgcn_ballot_w64(!ok) == 0ull) break;
        /* A member of this launch already gave up (not co-resident): the results are void and reported as such by every output
         * getter -- do not spin seconds again at each of the remaining exchanges.  Looked at only after 64 fruitless polls (~50
         * us): rounds 2-4 loaded the flag first and WAITED for it before publishing, one memory round trip in the serial chain
         * of every iteration. */
        ++spins;
        if ((spins & 63) == 0 && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
        if (spins > (1 << 21)) { if (lane == 0) *err = 1; break; }       /* seconds: a member is not resident -- report, do not hang */
        __builtin_amdgcn_s_sleep(1);
    }
    double v;
    if (!big) {
        v = (lane < n_rec) ? team_rec_value(r[0]) : 0.0;
    } else {
        v = 0.0;
#pragma unroll
        for (int q = 0; q < 4; q++) v += (lane + 64 * q < n_rec) ? team_rec_value(r[q]) : 0.0;
    }
    v += dpp_xor_row<8>(v);
    v = xor16_sum(v);
    return xor32_sum(v);
}

/* DVO_STAMPS: diagnostic build only (make STAMPS=1 -> libdvo_amd_stamps.so): lane 0 of wave 0 accumulates s_memtime
 * differences of the phases of every iteration into out.dbg[pair*64 + level*8 + {0 points, 1 reduce, 2 update, 3 barrier,
 * 4 iterations, 5 level set-up (staging)}] */
#ifdef DVO_STAMPS
DVO_DEV unsigned long long stamp_now2() {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
#define DVO_STAMP(var) const unsigned long long var = stamp_now2()
#define DVO_STAMP_ADD(slot, a, b) do { if (tid == 0 && out.dbg) out.dbg[(size_t)pair * 64 + l * 8 + (slot)] += (b) - (a); } while (0)
#define DVO_STAMP_T2_DECL() unsigned long long t2 = 0
#define DVO_STAMP_T2() do { t2 = stamp_now2(); } while (0)
#else
#define DVO_STAMP(var) do {} while (0)
#define DVO_STAMP_ADD(slot, a, b) do {} while (0)
#define DVO_STAMP_T2_DECL() do {} while (0)
#define DVO_STAMP_T2() do {} while (0)
#endif

/* WITH_H: DVO_FLAG_NORMAL_MATRIX on the packed kernel (round 5): every lane also accumulates the 21 entries of H = sum w J J^T in
 * double (42 more registers: both one-workgroup-per-pair shapes carry them since round 5, with loop-invariant values in scratch --
 * tests/test_kernel_registers.py), the waves reduce them in three
 * passes of the 8-value DPP reduce-scatter, and wave 2 -- idle during the update -- adds the waves' rows and stores the iterate's H */
template <int BLOCK, bool TEAM, bool WITH_H = false>
__global__ void __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(BLOCK == 256 ? DVO_WPE256 : 1, 8)))
align_fused2_kernel(LevelSet lv, Schedule sc, Intrinsics K, DevParams prm, Outputs out, int first_pair) {
    static_assert(!WITH_H || ((BLOCK == 512 || BLOCK == 256) && !TEAM), "H rides on the one-workgroup-per-pair shapes");
    /* team mode: workgroup b -> XCD b % 8; the G members of pair (q*8 + x) are the workgroups x + 8*(q*G + j) */
    /* teams beyond one XCD (one very large frame, G = 8 x g1 members, g1 of them on every XCD): member = workgroup index;
     * the sums are exchanged in two stages -- inside each XCD as above (slot `xcd` of the team buffer), then the 8 XCD sums
     * through slot 8, published by the first member of every XCD and read by everybody.  Same fixed order of additions on all
     * members -> identical bits; the double-buffering argument above holds per stage. */
    const int G = TEAM ? sc.team : 1;
    const bool super_team = TEAM && G > DVO_TEAM_MAX;
    int pair_local = blockIdx.x, member = 0, xcd = 0, local = 0;
    if (super_team) {
        if ((int)blockIdx.x >= G) return;
        member = blockIdx.x; xcd = blockIdx.x & 7; local = blockIdx.x >> 3; pair_local = 0;
    } else if (TEAM) {
        const int x = blockIdx.x & 7, k = blockIdx.x >> 3, q = k / G;
        member = k - q * G;
        pair_local = q * 8 + x;
        if (pair_local >= sc.n_pairs_launch) return;
    }
    if (!TEAM && out.order) pair_local = out.order[pair_local];      /* longest pairs first: the launch ends with the short ones */
    /* wave-uniform, but loaded through a vector instruction (the launch order): pin it to a scalar register, and with it every
     * per-pair pointer and count derived below -- a dozen 64-bit addresses that would otherwise live in vector registers for
     * the whole kernel (the 256-thread shape has exactly the 256 registers two waves per SIMD allow) */
    const int pair = __builtin_amdgcn_readfirstlane(first_pair + pair_local);
    const int tid = threadIdx.x;
    unsigned epoch = TEAM ? sc.team_epoch0 : 0u;  /* team mode: exchanges done so far, counted on from the launches before (record tag = epoch + 1) */
    /* Team mode, once per launch: do the members that exchange records directly (a team of up to 32, or the members of one XCD
     * of a team over all XCDs) really share an XCD?  The placement (workgroup b on XCD b % 8) is an observation, not a contract:
     * every member publishes its HW_REG_XCC_ID and its square through the sc1 exchange; sum x = G x and sum x^2 = G x^2 hold for
     * one member only if they hold for all (zero variance), so the answer is the same on every member -- and from then on the
     * records of that stage travel as plain stores through the shared L2 (team_store_rec_same_xcd). */
    bool team_same_xcd = false;
    if (TEAM) {
        if (tid < 64) {
            const double x = (double)hw_xcc_id();
            const int k = tid & 7;
            const double mine = (k == 0) ? x : ((k == 1) ? x * x : 0.0);
            v4u *tb = reinterpret_cast<v4u *>(out.team_buf);
            const int Gs = super_team ? (G >> 3) : G;
            const double tot = team_exchange(mine, tb + (size_t)(super_team ? xcd : pair_local) * 2 * DVO_TEAM_MAX * 8, super_team ? local : member, Gs, epoch,
                                             out.team_err);
            const double sx = readlane_f64(tot, 0), sxx = readlane_f64(tot, 1);
            team_same_xcd = !sc.team_no_plain && sx == (double)Gs * x && sxx == (double)Gs * (x * x);
        }
        epoch = sc.team_epoch0 + 1u;
    }
    /* one exchange of wave 0's eight lane values over the whole team (two stages beyond one XCD); `epoch` is advanced by the caller */
    auto team_total = [&](double sl) -> double {
        v4u *tb = reinterpret_cast<v4u *>(out.team_buf);
        if (super_team) {
            sl = team_exchange(sl, tb + (size_t)xcd * 2 * DVO_TEAM_MAX * 8, local, G >> 3, epoch, out.team_err, true, team_same_xcd);
            return team_exchange(sl, tb + (size_t)8 * 2 * DVO_TEAM_MAX * 8, xcd, 8, epoch, out.team_err, local == 0);
        }
        return team_exchange(sl, tb + (size_t)pair_local * 2 * DVO_TEAM_MAX * 8, member, G, epoch, out.team_err, true, team_same_xcd);
    };
    /* static LDS as ONE block of known size, so that the dynamic part -- which starts with the palette of the compact now
     * form -- begins at a compile-time LDS address (kStatic; verified below): palette look-ups then need no address add */
    static_assert(BLOCK >= 128, "wave 0 runs the update, wave 1 the bookkeeping");
    constexpr unsigned kPose = (unsigned)((sizeof(PoseState) + 15) & ~15u);
    constexpr unsigned kRedH = WITH_H ? (BLOCK / 64) * 24 * 8 : 0;             /* the waves' rows of H (24 doubles: 21 + padding) */
    constexpr unsigned kStatic = kPose + (BLOCK / 64) * 64 + 64 + kRedH;
    __shared__ __attribute__((aligned(16))) char s_static[kStatic];
    double (*const redH)[24] = reinterpret_cast<double (*)[24]>(s_static + kPose + (BLOCK / 64) * 64 + 64);
    PoseState &st = *reinterpret_cast<PoseState *>(s_static);
    double (*const red)[8] = reinterpret_cast<double (*)[8]>(s_static + kPose);
    double *const tot = reinterpret_cast<double *>(s_static + kPose + (BLOCK / 64) * 64);      /* team mode: [6], [7] = the team's sum eps^2, visible count */
    UpdConst &uc = st.u;                                         /* the update's constants (dvo_device_math.h) */
    extern __shared__ __attribute__((aligned(16))) float lds_dyn[];   /* sc.lds_bytes: per level [palette |] points [| the now level, when it fits] */
    const bool pal_base_ok = (unsigned)(size_t)(__attribute__((address_space(3))) float *)lds_dyn == kStatic;

#ifdef DVO_STAMPS
    if (tid == 0 && out.dbg) out.dbg[(size_t)pair * 64 + 62] = stamp_real();      /* workgroup start (launch timeline, tools/exp_timeline.py) */
#endif
    if (tid == 0) {
        const double *p = out.poses + (size_t)pair * 12;
        const bool ident = (sc.flags & 2) != 0;                  /* DVO_FLAG_IDENTITY_START (:2210-2211) */
        double R0[9], t0[3];
#pragma unroll
        for (int k = 0; k < 9; k++) R0[k] = ident ? ((k % 4 == 0) ? 1.0 : 0.0) : p[k];
#pragma unroll
        for (int k = 0; k < 3; k++) t0[k] = ident ? 0.0 : p[9 + k];
        pose_state_load(st, R0, t0);
        upd_const_build(uc, prm);
    }
    __syncthreads();

    for (int l = sc.n_levels - 1; l >= 0; --l) {                 /* SolveDVO.cpp:2097 */
        const int iters = sc.iters[l];
        if (iters <= 0) continue;                                 /* :2099 */
        const LevelSlab &L = lv.l[l];
        const int dpair = (sc.alias_mod > 0) ? (pair % sc.alias_mod) : pair;
        const int Nall = __builtin_amdgcn_readfirstlane(L.N[dpair]);
        /* this workgroup's share of the list: all of it, or member `member`'s contiguous even-sized chunk of a team */
        int pfirst = 0, N = Nall;
        /* Solo levels (round 5).  An exchange costs every iteration of a team ~2 us; a level of a few thousand points is finished by ONE
         * workgroup in less than that per iteration.  Such a level is run by member 0 alone (the whole list, no exchange: the serial
         * part of the one-workgroup shape); the other members skip it and pick its result up afterwards -- the pose is all that
         * travels from one runIterations call to the next (:2097-2109) -- through one exchange at the level's end.  Nall and the
         * threshold are the same on every member, so all of them take the same branch.
         * MEASURED AND NOT TAKEN (profiles/r05_experiments/team_solo_ab.txt; default threshold 0 = off, DVO_TEAM_SOLO_MAX=n switches it
         * on): one 640x480x4x10 pair 0.258 ms without, 0.251 with level 3 (230 points) solo, 0.265 with levels 3-2, 0.286 with 3-1;
         * batches of 32 lose 8 %.  The premise was wrong: ONE workgroup's iteration over 230 points takes 3.6 us (1.5 us for the single
         * round of look-ups to come back from the L2, 0.8 us for the waves to meet, 1.2 us update: profiles/r05_final/stamps_anatomy.txt),
         * the team's 3.1 us -- the floor of an iteration is latency either way, and the exchange is the smaller part of it. */
        const bool solo = TEAM && Nall <= sc.team_solo_max;
        const bool works = !solo || member == 0;
        if (TEAM && !solo) {
            const int chunk = (((Nall + G - 1) / G) + 1) & ~1;
            pfirst = min(Nall, member * chunk);
            N = min(Nall, pfirst + chunk) - pfirst;
        }
        if (!works) N = 0;
        const int iters_run = works ? iters : 0;
        const char *__restrict__ tex = reinterpret_cast<const char *>(L.tex + (size_t)dpair * L.tex_stride);
        const uint2 *__restrict__ gpts = L.cpts + (size_t)dpair * L.pt_cap + pfirst;
        float *energy = out.energy + (size_t)pair * sc.e_stride + sc.e_off[l];

        IterConst c;
        level_consts(c, K, l, L.rows, L.cols);
        DVO_STAMP(ts0);

        if (member == 0)
            for (int i = tid; i < iters; i += BLOCK) energy[i] = 0.0f;      /* :634 */
        if (tid == 0) { pose_state_begin(st); pose_regulariser_precompute(st, st.p[0], uc); }   /* :642-657 */

        /* ---- what lives in LDS for this level (wave-uniform decisions) --------------------------------------------
         * Always the compact point list (the reference deep-copies the 3xN list every iteration, :670; here HBM sees it once
         * per level).  If the whole list AND the level's texels fit, the texels are staged too ("LDS-staged image tiles")
         * and the level's iterations never leave the CU again. */
        const int lds_words = sc.lds_bytes >> 2;
        const int n_pad = (N + 3) & ~3;
        const int tex16_words = (int)(L.tex_stride * 4);
        /* the compact form of this pair's level, if the builder could make one (dvo_palette.h): its palette goes first */
        const int pal_n_raw = (!sc.no_p4 && L.pal_n) ? __builtin_amdgcn_readfirstlane(L.pal_n[dpair]) : 0;
        const int n_pal = pal_count(pal_n_raw);
        /* a PARTIAL compact form (dvo_palette.h, round 5): pixels it cannot express decode to NaN; a wave that meets one redoes its
         * share of the iteration on the image's 16-byte texels (the finiteness test below), the final pass patches them per lane */
        const bool p4_partial = pal_partial(pal_n_raw);
        /* the compact form first: a now level written by the engine's own distance-transform stage has no other (its 16-byte
         * texels exist only once something asked for them, dvo_capi.cpp: ensure_tex16) */
        const int mode = (n_pal > 0 && pal_base_ok && 2 * n_pal + 8 <= lds_words) ? TEX_P4
                         : ((!TEAM && !sc.no_lds_tex && 2 * n_pad + tex16_words <= lds_words) ? TEX_L16 : TEX_G16);
        const int n_pal_lds = n_pal + (p4_partial ? 2 : 1);                                   /* + the sentinel entry {0, 0} (+ a partial form's NaN entry) */
        const int pal_words = (mode == TEX_P4) ? ((2 * n_pal_lds + 3) & ~3) : 0;
        /* TEX_R16 (round 5): the level's ranks as 16-bit palette byte offsets in LDS, with a one-pixel reflected border and a block of
         * sentinel elements -- when that fits beside the palette and at least four rounds of points.  Not in team mode (every member
         * would stage the whole level), not on the level whose final outputs are produced (that pass reads the rank words). */
        const int r16_col = L.rows + 2;                                                       /* elements per padded column */
        const int r16_elems = r16_col * (L.cols + 2) + 2 * r16_col + 4;                       /* + the sentinel block */
        const int r16_words = ((r16_elems + 1) / 2 + 3) & ~3;
#ifdef DVO_ENABLE_R16
        constexpr bool kR16 = !WITH_H;
#else
        /* MEASURED AND NOT TAKEN (profiles/r05_experiments/r16_ab.txt): 640x480x4 at 8192 pairs 744 k aligns/s with the ranks of levels 2
         * and 3 in LDS against 765 k without (-2.8 %), 1024 pairs -3 %, 1920x1080x5 -1.6 %, 320x240x4x50 +1 % -- the coarse levels' lines
         * are L2 hits anyway, and staging 25 k elements per pair costs more than their look-ups save.  The code stays for the record:
         * make EXP=r16 EXPDEFS=-DDVO_ENABLE_R16=1 (then DVO_RANKS_LDS=off switches it off at run time). */
        constexpr bool kR16 = false;
#endif
        const bool r16 = kR16 && !TEAM && mode == TEX_P4 && !sc.no_r16 && !((sc.flags & 1) && l == sc.last_level) &&
                         pal_words + r16_words + 8 * BLOCK <= lds_words;
        const int img_words = r16 ? r16_words : 0;
        /* 4-byte points (dvo_device_math.h: pt4_decode): when the builder validated this list's 4-byte twin, the throughput shape
         * reads that -- twice the points per LDS byte, half the bytes per streamed point.  Not in team mode (a member's share
         * does not start on a 64-point chunk). */
        /* ... and only where part of the list would be streamed as 8-byte points: the 4-byte decode costs ~17 % more vector
         * instructions per point.  Rounds 3-4 (issue stage the nearer ceiling): 640x480 level 0 (14.8 k points, 8.2 k fit as 8-byte
         * points) a wash, 1920x1080 +3 % -- taken from three times the LDS capacity.  Round 5 (the kernel draws the HBM's whole
         * achievable rate): from ONE times the capacity, 789 k -> 802 k aligns/s at 640x480x4 (sc.pt4_factor, dvo_capi.cpp). */
        const bool pt4 = !TEAM && mode == TEX_P4 && !r16 && !sc.no_pt4 && N >= sc.pt4_factor * (((lds_words - pal_words - img_words) >> 1) & ~1) && L.pt4_ok &&
                         __builtin_amdgcn_readfirstlane(L.pt4_ok[dpair]) != 0;
        float *const lds_pts = lds_dyn + pal_words + img_words;
        const int cap = (mode == TEX_L16) ? n_pad
                        : (pt4 ? ((lds_words - pal_words - img_words) & ~1) : (((lds_words - pal_words - img_words) >> 1) & ~1));      /* points the LDS holds */
        const int n_lds = (N <= cap) ? N : (cap / (2 * BLOCK)) * (2 * BLOCK);      /* whole rounds only */
        float *const lds_tex = lds_pts + 2 * cap;
        if (tid == 0) { st.exact_ran = 0; st.e2_open = 0; st.e2_ran = 0; }
        if (mode == TEX_P4 && works) {
            const float2 *__restrict__ pg = L.pal + (size_t)dpair * DVO_PAL_MAX;
            float2 *pl = reinterpret_cast<float2 *>(lds_dyn);
            for (int i = tid; i < n_pal_lds; i += BLOCK) pl[i] = pg[i];
        }
        const unsigned *__restrict__ g4pts = L.cpt4 + (size_t)dpair * L.pt_cap + pfirst;
        if (pt4) {   /* 16-byte loads of four points, four in flight per lane */
            const uint4 *g4 = reinterpret_cast<const uint4 *>(g4pts);
            uint4 *d4 = reinterpret_cast<uint4 *>(lds_pts);
            const int n4 = n_lds >> 2;
            int i = tid;
            for (; i + 3 * BLOCK < n4; i += 4 * BLOCK) {
                uint4 v[4];
#pragma unroll
                for (int q = 0; q < 4; q++) v[q] = g4[i + q * BLOCK];
#pragma unroll
                for (int q = 0; q < 4; q++) d4[i + q * BLOCK] = v[q];
            }
            for (; i < n4; i += BLOCK) d4[i] = g4[i];
            for (int j = (n4 << 2) + tid; j < n_lds; j += BLOCK) reinterpret_cast<unsigned *>(lds_pts)[j] = g4pts[j];
        } else {   /* 16-byte loads (whole 128-byte lines per request), four in flight per lane */
            const uint4 *g4 = reinterpret_cast<const uint4 *>(gpts);
            uint4 *d4 = reinterpret_cast<uint4 *>(lds_pts);
            const int n2 = n_lds >> 1;
            int i = tid;
            for (; i + 3 * BLOCK < n2; i += 4 * BLOCK) {
                uint4 v[4];
#pragma unroll
                for (int q = 0; q < 4; q++) v[q] = g4[i + q * BLOCK];
#pragma unroll
                for (int q = 0; q < 4; q++) d4[i + q * BLOCK] = v[q];
            }
            for (; i < n2; i += BLOCK) d4[i] = g4[i];
            if ((n_lds & 1) && tid == 0) reinterpret_cast<uint2 *>(lds_pts)[n_lds - 1] = gpts[n_lds - 1];
        }
        if (mode == TEX_L16) {
            const v4f *g = reinterpret_cast<const v4f *>(tex);
            v4f *d = reinterpret_cast<v4f *>(lds_tex);
            const int n16 = (int)L.tex_stride;
            int i = tid;
            for (; i + 3 * BLOCK < n16; i += 4 * BLOCK) {
                v4f v[4];
#pragma unroll
                for (int q = 0; q < 4; q++) v[q] = g[i + q * BLOCK];
#pragma unroll
                for (int q = 0; q < 4; q++) d[i + q * BLOCK] = v[q];
            }
            for (; i < n16; i += BLOCK) d[i] = g[i];
        }
        LdsPoints lp;
        lp.p = reinterpret_cast<const uint2 *>(lds_pts);
        lp.w4 = reinterpret_cast<const unsigned *>(lds_pts);
        lp.g4 = g4pts;
        lp.hdr = L.chdr + (size_t)dpair * (L.pt_cap >> 6);
        TexSrc ts;
        ts.g16 = tex; ts.tile_col_bytes = (unsigned)c.tiles_per_col * 128u;
        ts.l16 = reinterpret_cast<const char *>(lds_tex);
        ts.p4 = reinterpret_cast<const char *>(L.p4 + (size_t)dpair * L.p4_stride);
        ts.p4_col_bytes = (unsigned)p4_tiles_per_col(L.rows) * 128u;
        ts.r16_org = kStatic + 4u * (unsigned)pal_words;
        ts.r16_col_bytes = 2u * (unsigned)r16_col;
        ts.r16_sent = ts.r16_org + 2u * (unsigned)(r16_col * (L.cols + 2) + r16_col);        /* the element above the centre of the sentinel block */
        if (r16) {
            /* one padded column per wave at a time: element (yy, xx) <- the centre rank word of pixel (reflect101(yy), reflect101(xx)),
             * its rank field as it is (rank * 8 = the palette byte offset) */
            unsigned short *img = reinterpret_cast<unsigned short *>(lds_dyn + pal_words);
            const int wv = tid >> 6, ln = tid & 63;
            for (int xc = wv; xc < L.cols + 2; xc += BLOCK / 64) {
                int xx = xc - 1;
                xx = xx < 0 ? -xx : (xx >= L.cols ? 2 * L.cols - 2 - xx : xx);
                for (int yc = ln; yc < r16_col; yc += 64) {
                    int yy = yc - 1;
                    yy = yy < 0 ? -yy : (yy >= L.rows ? 2 * L.rows - 2 - yy : yy);
                    const unsigned w = *reinterpret_cast<const unsigned *>(ts.p4 + p4_byte_offset(yy, xx, ts.p4_col_bytes) + 4u);
                    img[xc * r16_col + yc] = (unsigned short)(w & 0xfff8u);
                }
            }
            for (int i = tid; i < 2 * r16_col + 4; i += BLOCK) img[r16_col * (L.cols + 2) + i] = (unsigned short)(n_pal << 3);      /* the zero entry */
        }
        __syncthreads();
        DVO_STAMP(ts1);
        DVO_STAMP_ADD(5, ts0, ts1);

        const bool exch = TEAM && !solo;                                     /* this level's sums are the team's */
        for (int itr = 0; itr < iters_run; ++itr) {                          /* :658 */
            /* Instruction-issue priority falls with progress.  Two workgroups share a CU; the hardware favours the older
             * one, so of two that start together one ends ~100 us before the other, which then finishes alone on a half-empty
             * CU (launch timeline, tools/exp_timeline.py: the last 15 % of a 1024-pair launch ran with 256 of 512 slots busy).
             * With the one that is BEHIND getting the issue slots they finish together: workgroup durations 719 +- 69 us ->
             * 765 +- 29 us, slots busy until the end; C2 654 k -> 665 k aligns/s, without misses 733 k -> 769 k.
             * (make EXP=noprio EXPDEFS=-DDVO_NO_PROGRESS_PRIO=1 for the A/B.) */
#ifndef DVO_NO_PROGRESS_PRIO
            if (!TEAM) {
                if (l > 0) __builtin_amdgcn_s_setprio(3);
                else if (2 * itr < iters) __builtin_amdgcn_s_setprio(2);
                else if (4 * itr < 3 * iters) __builtin_amdgcn_s_setprio(1);
                else __builtin_amdgcn_s_setprio(0);
            }
#endif
            /* the iterate this iteration evaluates, and the one its update writes (PoseState: the bookkeeping of this iterate
             * runs beside the update) */
            const PoseCur &pc = st.p[itr & 1];
            PoseCur &pn = st.p[(itr + 1) & 1];
            iter_const_pose(c, pc.Rf, pc.tf);                                 /* :673-674 */

            DVO_STAMP(t0);
            Acc7T<WITH_H> a;
            acc7_zero(a);
            bool any_odd = false;
            /* waves take the lanes of a round in reverse order: the tail of the last, partial round goes to the high
             * waves first, so wave 0 -- whose lane 0 still has the regulariser of the new pose to finish -- most often
             * has a round less */
            const int lane_off = BLOCK - 64 - (tid & ~63) + (tid & 63);
            if (mode == TEX_L16) {                 /* staged levels hold every point in LDS */
                accumulate_points2<BLOCK, true, TEX_L16, 2, 0, false, Acc7T<WITH_H>>(c, ts, lp, gpts, 0, N, lane_off, a, any_odd);             /* :369, :433 */
            } else if (r16) {                      /* every look-up from LDS: no deep prefetch needed */
                accumulate_points2<BLOCK, true, TEX_R16, 2, kStatic, false, Acc7T<WITH_H>>(c, ts, lp, gpts, 0, n_lds, lane_off, a, any_odd);
                accumulate_points2<BLOCK, false, TEX_R16, 2, kStatic, false, Acc7T<WITH_H>>(c, ts, lp, gpts, n_lds, N, lane_off, a, any_odd);
            } else if (mode == TEX_P4 && pt4) {
                accumulate_points2<BLOCK, true, TEX_P4, (WITH_H ? 2 : DVO_P4_DEPTH(BLOCK)), kStatic, true, Acc7T<WITH_H>>(c, ts, lp, gpts, 0, n_lds, lane_off, a, any_odd);
                accumulate_points2<BLOCK, false, TEX_P4, (WITH_H ? 2 : DVO_P4_DEPTH(BLOCK)), kStatic, true, Acc7T<WITH_H>>(c, ts, lp, gpts, n_lds, N, lane_off, a, any_odd);
            } else if (mode == TEX_P4) {
                accumulate_points2<BLOCK, true, TEX_P4, (WITH_H ? 2 : DVO_P4_DEPTH(BLOCK)), kStatic, false, Acc7T<WITH_H>>(c, ts, lp, gpts, 0, n_lds, lane_off, a, any_odd);
                accumulate_points2<BLOCK, false, TEX_P4, (WITH_H ? 2 : DVO_P4_DEPTH(BLOCK)), kStatic, false, Acc7T<WITH_H>>(c, ts, lp, gpts, n_lds, N, lane_off, a, any_odd);
            } else {
                accumulate_points2<BLOCK, true, TEX_G16, 2, 0, false, Acc7T<WITH_H>>(c, ts, lp, gpts, 0, n_lds, lane_off, a, any_odd);
                accumulate_points2<BLOCK, false, TEX_G16, 2, 0, false, Acc7T<WITH_H>>(c, ts, lp, gpts, n_lds, N, lane_off, a, any_odd);       /* beyond the LDS budget */
            }
            {   /* A lane without a visible point keeps its coordinates (round2_issue): its gradient, weight and residual are exact
                 * zeros, so it adds exact zeros -- unless one of its coordinate products overflowed (a point a hair off the camera
                 * plane, an absurd pose): 0 * inf = NaN would then sit in the lane's sums.  NaN and inf are sticky in a sum, so one
                 * test per iteration catches it: such a wave redoes its share with the literal scalar code, which skips invisible
                 * points like the reference does (:371).  (seven additions and a compare per iteration, not per round.) */
                const double chk = ((a.g[0] + a.g[1]) + (a.g[2] + a.g[3])) + ((a.g[4] + a.g[5]) + a.e2);
                any_odd |= (__builtin_amdgcn_ballot_w64(!__builtin_isfinite(chk)) != 0ull);
            }
            if (any_odd || sc.force_exact) {      /* wave-uniform; a point with a degenerate z: this wave's share again, literal divisions */
                acc7_zero(a);
                sweep_literal<BLOCK>(mode, p4_partial, pt4, c, tex, ts, reinterpret_cast<const float2 *>(lds_dyn), lp, gpts, n_lds, N, lane_off, a);
                if ((tid & 63) == 0) st.exact_ran = 1;          /* inspection: dvo_get_level_texel_mode reports it (tests) */
            }
            DVO_STAMP(t1);
            DVO_STAMP_T2_DECL();
            /* ---- the serial part of the iteration (round 5) -----------------------------------------------------------------
             * Every wave leaves its eight totals in LDS; after ONE barrier wave 0 adds them (lane k: sum k) and runs the update
             * from those registers -- in team mode after exchanging them with the other members, also in registers -- while wave
             * 1 does the energy and the best-iterate bookkeeping (:689-705) of the SAME iterate: the two read the iterate `pc`,
             * the update writes `pn`.  (Rounds 1-4: a second pass through LDS and a second barrier for the totals, then one
             * lane doing bookkeeping, update and all LDS traffic in a row.) */
            wave_sums7(a, red);
            if constexpr (WITH_H) {       /* the wave's 21 sums of H: three passes of the 8-value reduce-scatter */
#pragma unroll
                for (int q = 0; q < 3; q++) {
                    double d[8];
#pragma unroll
                    for (int k = 0; k < 8; k++) d[k] = (8 * q + k < 21) ? a.H[8 * q + k] : 0.0;
                    wave_reduce_scatter8_dpp(d);
                    const int idx = reduce_scatter8_dpp_index(tid & 63);
                    if ((tid & 63) < 8) redH[tid >> 6][8 * q + idx] = d[0];
                }
            }
            const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
            double neg_step = 0.0;
            if (wave == 0) neg_step = pose_neg_step(uc, itr);        /* :773: needs the iteration index only -- taken while the other waves arrive */
            __syncthreads();
            double sl = 0.0;
            if (wave == 0) {
                sl = block_sum8<BLOCK>(red, lane & 7);
                DVO_STAMP_T2();
                if (exch) {    /* the sums of the other members; identical bits on every member */
                    sl = team_total(sl);
                    if (lane == 6 || lane == 7) tot[lane] = sl;            /* for the bookkeeping wave */
                }
            }
            if (exch) __syncthreads();
            if (wave == 0) {
                double psi[6];
                pose_direction_lanes(st, uc, neg_step, sl, lane, psi);
                if (lane == 0) pose_apply(st, pc, pn, uc, psi);
            } else if (WITH_H && wave == 2) {      /* H of this iterate (dvo_get_level_normal_matrix): waves in order, like the other sums */
                if (lane < 21) {
                    double hs = 0.0;
#pragma unroll
                    for (int w = 0; w < BLOCK / 64; w++) hs += redH[w][lane];
                    out.H[((size_t)pair * sc.e_stride + sc.e_off[l] + itr) * 21 + lane] = hs;
                }
            } else if (wave == 1 && lane == 0) {
                const double e2 = exch ? tot[6] : block_sum8<BLOCK>(red, 6);
                const double nv = exch ? tot[7] : block_sum8<BLOCK>(red, 7);
                float e;
                if (energy_certified(e2, Nall, e) && !sc.force_e2) {          /* no order of additions could have given another float */
                    pose_bookkeep_e(st, pc, itr, Nall, e, (int)nv);
                    if (member == 0) energy[itr] = e;                        /* :690 */
                } else {                                                     /* about Nall 2^-28 of the iterations: settled below */
                    st.e2_fast = e2;
                    st.e2_nvis = (int)nv;
                    st.e2_open = 1;
                }
            }
            if (exch) epoch++;
            DVO_STAMP(t3);
            __syncthreads();
            DVO_STAMP(t4);
            if (st.e2_open) {       /* workgroup- (and team-) uniform: the same sum bits everywhere */
                /* the residuals of this iterate once more (c still holds its pose), added exactly: three limbs per lane -> wave ->
                 * workgroup (-> team), all of them integers below 2^53 in doubles, so the sums are exact whatever their order */
                AccE2 ae;
                e2_limbs_zero(ae.l);
                ae.nvis = 0;
                sweep_literal<BLOCK>(mode, p4_partial, pt4, c, tex, ts, reinterpret_cast<const float2 *>(lds_dyn), lp, gpts, n_lds, N, lane_off, ae);
                wave_sums_e2(ae, red);
                __syncthreads();
                if (wave == 0) {
                    double sl2 = block_sum8<BLOCK>(red, lane & 7);
                    if (exch) sl2 = team_total(sl2);
                    if (lane < 3) tot[lane] = sl2;
                }
                if (exch) epoch++;
                __syncthreads();
                if (wave == 1 && lane == 0) {
                    const float e = (float)sqrt(e2_from_limbs(tot[0], tot[1], tot[2], st.e2_fast));
                    pose_bookkeep_e(st, pc, itr, Nall, e, st.e2_nvis);
                    if (member == 0) energy[itr] = e;                        /* :690 */
                    st.e2_open = 0;
                    st.e2_ran++;
                }
                __syncthreads();
            }
            DVO_STAMP_ADD(0, t0, t1); DVO_STAMP_ADD(1, t1, t2); DVO_STAMP_ADD(2, t2, t3); DVO_STAMP_ADD(3, t3, t4);
            DVO_STAMP_ADD(4, t0, t0 + 1);
            if (st.stop) break;                                              /* :877 */
            /* log(new pose) for the next iteration's regulariser: lane 0 takes it now, while the other waves are
             * already in their point phase */
            if (tid == 0 && itr + 1 < iters_run) pose_regulariser_precompute(st, pn, uc);
        }

        /* finalEpsilons / finalReprojections = those of the best iterate (:703-704, :1002-1003); recomputed once from
         * the same float pose -> same bits */
        if ((sc.flags & 1) && l == sc.last_level) {
            if (st.bestItr >= 0) {
                iter_const_pose(c, st.bRf, st.btf);
                float *fe = out.final_eps + (size_t)pair * out.final_cap;
                float *fr = out.final_reproj + (size_t)pair * out.final_cap * 3;
                /* written in the order of the compact list (this workgroup's share starts at pfirst); the host hands them out in
                 * the reference's order (LevelSlab.cidx) */
                float *fes = fe + pfirst, *frs = fr + 3 * (size_t)pfirst;
#ifndef DVO_NO_FINAL
                if (mode == TEX_P4 && pt4) {
                    final_outputs2<BLOCK, true, TEX_P4, kStatic, true>(c, ts, lp, gpts, 0, n_lds, fes, frs, p4_partial);
                    final_outputs2<BLOCK, false, TEX_P4, kStatic, true>(c, ts, lp, gpts, n_lds, N, fes, frs, p4_partial);
                } else if (mode == TEX_P4) {
                    final_outputs2<BLOCK, true, TEX_P4, kStatic>(c, ts, lp, gpts, 0, n_lds, fes, frs, p4_partial);
                    final_outputs2<BLOCK, false, TEX_P4, kStatic>(c, ts, lp, gpts, n_lds, N, fes, frs, p4_partial);
                } else if (mode == TEX_L16) {
                    final_outputs2<BLOCK, true, TEX_L16, kStatic>(c, ts, lp, gpts, 0, N, fes, frs);
                } else {
                    final_outputs2<BLOCK, true, TEX_G16, kStatic>(c, ts, lp, gpts, 0, n_lds, fes, frs);
                    final_outputs2<BLOCK, false, TEX_G16, kStatic>(c, ts, lp, gpts, n_lds, N, fes, frs);
                }
#endif
            }
            if (tid == 0 && member == 0) out.final_N[pair] = (st.bestItr >= 0) ? Nall : 0;
        }
        __syncthreads();
        if (tid == 0) {                                                      /* :997-1005 */
            pose_state_finish(st);
            if (member == 0) {
                out.best_idx[pair * DVO_LEVELS + l] = st.bestItr;
                out.ratio[pair * DVO_LEVELS + l] = st.bestRatio;
                out.tex_mode[pair * DVO_LEVELS + l] = mode | (st.exact_ran ? DVO_TEXMODE_EXACT_RAN : 0) | (pt4 ? DVO_TEXMODE_PT4 : 0) | (r16 ? DVO_TEXMODE_RANKS_LDS : 0) |
                                                       (min(st.e2_ran, DVO_TEXMODE_E2_MAX) << DVO_TEXMODE_E2_SHIFT);
            }
        }
        __syncthreads();
        if (TEAM && solo) {
            /* the level's result: member 0's {q, t} to every member, through the exchange that adds lane values -- the others put
             * in -0.0, and x + (-0.0) = x for EVERY x (+0.0 would turn a -0.0 component into +0.0), so the bits arrive unchanged
             * whatever the order of the additions.  R = the matrix of q (pose_state_finish), recomputed alike everywhere. */
            if (tid < 64) {
                const int k = tid & 7;
                double mine = -0.0;
                if (member == 0 && k < 7) mine = (k < 4) ? st.p[0].q[k] : st.p[0].t[k - 4];
                const double v = team_total(mine);
                if (tid < 4) st.p[0].q[tid] = v;
                else if (tid < 7) st.p[0].t[tid - 4] = v;
            }
            epoch++;
            __syncthreads();
            if (tid == 0) {
                double q[4], R[9];
#pragma unroll
                for (int k = 0; k < 4; k++) q[k] = st.p[0].q[k];
                quat_to_matrix(q, R);
#pragma unroll
                for (int k = 0; k < 9; k++) st.R[k] = R[k];
            }
            __syncthreads();
        }
    }

#ifdef DVO_STAMPS
    if (tid == 0 && out.dbg) out.dbg[(size_t)pair * 64 + 63] = stamp_real();      /* workgroup end */
#endif
    if (tid == 0 && member == 0) {
        double *p = out.poses + (size_t)pair * 12;
#pragma unroll
        for (int k = 0; k < 9; k++) p[k] = st.R[k];
#pragma unroll
        for (int k = 0; k < 3; k++) p[9 + k] = st.p[0].t[k];
    }
}

template <int BLOCK>
static hipError_t launch_fused2_b(const LevelSet &lv, const Schedule &sc, const Intrinsics &K, const DevParams &prm,
                                  const Outputs &out, int first_pair, int n_pairs, hipStream_t s) {
    const size_t dyn = (size_t)sc.lds_bytes;
    hipError_t e;
    if constexpr (BLOCK == 512) {
        if (sc.flags & 4) {      /* DVO_FLAG_NORMAL_MATRIX: H per iterate: one 512-thread workgroup per pair, no teams */
            if (sc.team > 1 || !out.H) return hipErrorInvalidValue;
            auto kern = align_fused2_kernel<512, false, true>;
            if ((e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn)) != hipSuccess) return e;
            hipLaunchKernelGGL(kern, dim3(n_pairs), dim3(512), dyn, s, lv, sc, K, prm, out, first_pair);
            return hipGetLastError();
        }
    } else if constexpr (BLOCK == 256) {
        if (sc.flags & 4) {
            if (sc.team > 1 || !out.H) return hipErrorInvalidValue;
            auto kern = align_fused2_kernel<256, false, true>;
            if ((e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn)) != hipSuccess) return e;
            hipLaunchKernelGGL(kern, dim3(n_pairs), dim3(256), dyn, s, lv, sc, K, prm, out, first_pair);
            return hipGetLastError();
        }
    } else {
        if (sc.flags & 4) return hipErrorInvalidValue;      /* DVO_FLAG_NORMAL_MATRIX */
    }
    if (sc.team > 1) {
        auto kern = align_fused2_kernel<BLOCK, true>;
        if ((e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn)) != hipSuccess) return e;
        const int grid = (sc.team > DVO_TEAM_MAX) ? sc.team : 8 * ((n_pairs + 7) / 8) * sc.team;
        /* The members of a team wait for each other inside the kernel, so ALL workgroups of the launch must be resident at
         * once.  The engine checks that itself -- the grid against what the occupancy query says the device can hold of THIS
         * kernel with THIS much LDS, which is all hipLaunchCooperativeKernel checks (MI355X_MICROARCH.md, residency and
         * cooperative launch: plain and cooperative launches have identical residency) -- and refuses the launch otherwise
         * (the host then runs the batch without teams).  A member that still finds itself alone (another process or stream
         * holding compute units) ends in the bounded spin's error flag, which every output getter reports.
         * Round 3 made the launch itself cooperative; round 4 went back to a plain launch behind the same check because
         * (1) ROCm 7.2 kills any process that made ONE cooperative launch inside the HIP runtime's own exit handler when it runs
         * under rocprofv3 (reproduced with examples/solve_dvo_demo.cpp against the system runtime, no engine resource alive:
         * tools/experiments/r04_exit_segv*.sh), and (2) it cost 0.03 ms of every 0.3 ms small-batch
         * step.  DVO_TEAM_COOP_LAUNCH=1 brings the cooperative launch back for A/B measurements. */
        {   /* the answer only depends on (device, dynamic LDS): asked once per thread and shape, not per launch */
            struct Seen { int dev = -1; size_t dyn = 0; long long limit = 0; };
            thread_local Seen seen;
            int dev = 0;
            if ((e = hipGetDevice(&dev)) != hipSuccess) return e;
            if (seen.dev != dev || seen.dyn != dyn) {
                int per_cu = 0, n_cu = 0;
                if ((e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)kern, BLOCK, dyn)) != hipSuccess) return e;
                if ((e = hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev)) != hipSuccess) return e;
                seen.dev = dev; seen.dyn = dyn; seen.limit = (long long)per_cu * n_cu;
            }
            if ((long long)grid > seen.limit) return hipErrorCooperativeLaunchTooLarge;
        }
        static const bool coop = std::getenv("DVO_TEAM_COOP_LAUNCH") != nullptr;
        if (!coop) {
            hipLaunchKernelGGL(kern, dim3(grid), dim3(BLOCK), dyn, s, lv, sc, K, prm, out, first_pair);
        } else {
            LevelSet a0 = lv; Schedule a1 = sc; Intrinsics a2 = K; DevParams a3 = prm; Outputs a4 = out; int a5 = first_pair;
            void *args[] = {&a0, &a1, &a2, &a3, &a4, &a5};
            if ((e = hipLaunchCooperativeKernel((const void *)kern, dim3(grid), dim3(BLOCK), args, (unsigned)dyn, s)) != hipSuccess) {
                (void)hipGetLastError();
                return e;
            }
        }
    } else {
        auto kern = align_fused2_kernel<BLOCK, false>;
        if ((e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn)) != hipSuccess) return e;
        hipLaunchKernelGGL(kern, dim3(n_pairs), dim3(BLOCK), dyn, s, lv, sc, K, prm, out, first_pair);
    }
    return hipGetLastError();
}

/* static LDS of align_fused2_kernel<BLOCK> (the host sizes the dynamic part against the CU's 160 KiB) */
size_t fused2_static_lds(int block_threads, bool with_h) {
    return ((sizeof(PoseState) + 15) & ~(size_t)15) + (size_t)(block_threads / 64) * 64 + 64 + (with_h ? (size_t)(block_threads / 64) * 24 * 8 : 0) + 16;
}

hipError_t launch_align_fused2(int block_threads, const LevelSet &lv, const Schedule &sc, const Intrinsics &K,
                               const DevParams &prm, const Outputs &out, int first_pair, int n_pairs, hipStream_t s) {
    if (n_pairs <= 0) return hipSuccess;
    if (sc.team > DVO_TEAM_MAX && !(n_pairs == 1 && (sc.team == 64 || sc.team == 128 || sc.team == 256))) return hipErrorInvalidValue;
    switch (block_threads) {
    case 256: return launch_fused2_b<256>(lv, sc, K, prm, out, first_pair, n_pairs, s);
    case 1024: return launch_fused2_b<1024>(lv, sc, K, prm, out, first_pair, n_pairs, s);
    default: return launch_fused2_b<512>(lv, sc, K, prm, out, first_pair, n_pairs, s);
    }
}

/* ---- the launch of one iteration of the tiled / wide schedule with THIS file's point loop (round 5) -------------------------------
 * tiled_step_kernel (dvo_kernels.hip) runs one point per lane over the reference's 3 x N float list: 7.5 us for the 2.4 k points a
 * workgroup gets of a 4096 x 3072 level 0, issue-bound.  Here the same launch -- head, hand-off and last arriver are shared
 * (dvo_tiled_step.h) -- streams its share of the COMPACT list (8 bytes per point, a round ahead) through the packed
 * two-points-per-lane rounds, looking the texels up in the level's 16-byte image (the tiled schedule replicates that image on every
 * rank; its compact form would need the palette in every workgroup's LDS at every launch).  Lists the engine's own enlist kernels
 * built have the compact twin; without it, or with interpolate_dt, the launch is tiled_step_kernel.  WITH_H: the 21 sums of H = sum w J J^T
 * ride along (DVO_FLAG_NORMAL_MATRIX; BASELINE configs[4] names "6x6 JtJ + 6x1 Jtr" for the all-reduce).
 * Measured and not kept: round 0's points of every lane requested in the shadow of the head's loads (they do not depend on the pose) --
 * 0.541-0.545 ms per 4096 x 3072 x 5 alignment against 0.538. */
template <bool WITH_H>
__global__ void __launch_bounds__(DVO_STEP_THREADS)
tiled_step_pk_kernel(LevelSlab L, int pair, int level, Intrinsics K, const PoseState *st_in, PoseState *st_out,
                     const double *__restrict__ acc_in, int itr, int apply_prev, int n_total, int first, int n,
                     double *partials, unsigned *ticket, double *acc_out, float *energy, double *H_prev) {
    __shared__ double red[DVO_STEP_THREADS / 64][8];
    __shared__ double red2[DVO_STEP_THREADS / 64][8];                        /* the limbs of the exact sum of eps^2 */
    __shared__ double redH[WITH_H ? DVO_STEP_THREADS / 64 : 1][24];
    __shared__ TiledStepLds m;
    tiled_step_body<WITH_H>(m, st_in, st_out, acc_in, itr, apply_prev, n_total, first, n, partials, ticket, acc_out, energy, H_prev,
        [&](const PoseCur &pc, bool run, int b0, int b1, double *tot) {
            const int tid = threadIdx.x;
            typedef Acc7T<WITH_H, true> AccS;
            AccS a;
            acc7_zero(a);
            if (run) {
                const char *__restrict__ tex = reinterpret_cast<const char *>(L.tex + (size_t)pair * L.tex_stride);
                const uint2 *__restrict__ gpts = L.cpts + (size_t)pair * L.pt_cap;
                IterConst c;
                level_consts(c, K, level, L.rows, L.cols);
#pragma unroll
                for (int k = 0; k < 9; k++) c.r[k] = uniform_f(pc.Rf[k]);
#pragma unroll
                for (int k = 0; k < 3; k++) c.t[k] = uniform_f(pc.tf[k]);
                TexSrc ts = {};
                ts.g16 = tex; ts.tile_col_bytes = (unsigned)c.tiles_per_col * 128u;
                LdsPoints lp = {};
                bool any_odd = false;
                accumulate_points2<DVO_STEP_THREADS, false, TEX_G16, 2, 0, false, AccS>(c, ts, lp, gpts, b0, b1, tid, a, any_odd);
                {   /* a degenerate z somewhere in this wave's share: again with the literal divisions (see align_fused2_kernel) */
                    const double chk = ((a.g[0] + a.g[1]) + (a.g[2] + a.g[3])) + ((a.g[4] + a.g[5]) + a.e2);
                    any_odd |= (__builtin_amdgcn_ballot_w64(!__builtin_isfinite(chk)) != 0ull);
                }
                if (any_odd) {
                    acc7_zero(a);
                    accumulate_points_exact<DVO_STEP_THREADS, false, false, false, AccS>(c, tex, ts, nullptr, lp, gpts, b0, b1, tid, a);
                }
            }
            wave_sums7(a, red);
            wave_sums_e2(a, red2);
            if constexpr (WITH_H) {       /* the wave's 21 sums of H: three passes of the 8-value reduce-scatter (align_fused2_kernel) */
#pragma unroll
                for (int q = 0; q < 3; q++) {
                    double d[8];
#pragma unroll
                    for (int k = 0; k < 8; k++) d[k] = (8 * q + k < 21) ? a.H[8 * q + k] : 0.0;
                    wave_reduce_scatter8_dpp(d);
                    const int idx = reduce_scatter8_dpp_index(tid & 63);
                    if ((tid & 63) < 8) redH[tid >> 6][8 * q + idx] = d[0];
                }
            }
            __syncthreads();
            if (tid < DVO_NACC_PAD) {
                double v = 0.0;
                if (tid >= 21 && tid < 29) v = block_sum8<DVO_STEP_THREADS>(red, tid - 21);
                else if (tid >= 29) v = block_sum8<DVO_STEP_THREADS>(red2, tid - 29);
                else if (WITH_H && tid < 21) {
#pragma unroll
                    for (int w = 0; w < DVO_STEP_THREADS / 64; w++) v += redH[w][tid];      /* waves in order, like the other sums */
                }
                tot[tid] = v;
            }
            __syncthreads();
        });
}
hipError_t launch_tiled_step_pk(const LevelSlab &L, int pair, int level, const Intrinsics &K, const void *st_in, void *st_out,
                                const double *acc_in, int itr, int apply_prev, int n_total, int first_point, int n_points,
                                double *partials, unsigned *ticket, double *acc_out, float *energy, int nblocks, double *H_prev, hipStream_t s) {
    if (H_prev)
        hipLaunchKernelGGL(tiled_step_pk_kernel<true>, dim3(nblocks), dim3(DVO_STEP_THREADS), 0, s, L, pair, level, K, (const PoseState *)st_in,
                           (PoseState *)st_out, acc_in, itr, apply_prev, n_total, first_point, n_points, partials, ticket, acc_out, energy, H_prev);
    else
        hipLaunchKernelGGL(tiled_step_pk_kernel<false>, dim3(nblocks), dim3(DVO_STEP_THREADS), 0, s, L, pair, level, K, (const PoseState *)st_in,
                           (PoseState *)st_out, acc_in, itr, apply_prev, n_total, first_point, n_points, partials, ticket, acc_out, energy, H_prev);
    return hipGetLastError();
}

/* ---- a SMALL level of the tiled / wide schedule as one launch (round 5) -------------------------------------------------------------
 * A step launch costs ~6.8 us whatever the level holds; a level of a few thousand points is finished by ONE workgroup in less than that per
 * iteration (the packed rounds over the compact list, 16-byte texels; the serial part of align_fused2_kernel: wave 0 direction + step,
 * wave 1 bookkeeping).  So such a level runs all its iterations here, then does what tiled_finish_kernel does (the pose <- the best
 * iterate, outputs, the next level's begin).  In a tiled run EVERY rank runs it over the whole list -- same bits everywhere, and no
 * collective for the level's iterations. */
__global__ void __launch_bounds__(DVO_STEP_THREADS)
tiled_level_solo_kernel(LevelSlab L, int pair, int level, Intrinsics K, const PoseState *st_in, PoseState *st_out, int iters, int N,
                        float *energy, double *Rt12, int *best_idx, float *ratio, float *next_energy, int next_iters) {
    constexpr int BLOCK = DVO_STEP_THREADS;
    __shared__ double red[BLOCK / 64][8];
    __shared__ double red2[BLOCK / 64][8];                                   /* the limbs of the exact sum of eps^2 */
    __shared__ PoseState st;
    const int tid = threadIdx.x;
    {
        const unsigned long long *src = reinterpret_cast<const unsigned long long *>(st_in);
        unsigned long long *dst = reinterpret_cast<unsigned long long *>(&st);
        for (int i = tid; i < (int)(sizeof(PoseState) / 8); i += BLOCK) dst[i] = src[i];
    }
    __syncthreads();
    const UpdConst &uc = st.u;
    const char *__restrict__ tex = reinterpret_cast<const char *>(L.tex + (size_t)pair * L.tex_stride);
    const uint2 *__restrict__ gpts = L.cpts + (size_t)pair * L.pt_cap;
    IterConst c;
    level_consts(c, K, level, L.rows, L.cols);
    TexSrc ts = {};
    ts.g16 = tex; ts.tile_col_bytes = (unsigned)c.tiles_per_col * 128u;
    LdsPoints lp = {};
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int lane_off = BLOCK - 64 - (tid & ~63) + (tid & 63);      /* the tail of the last round goes to the high waves first (align_fused2_kernel) */
    for (int itr = 0; itr < iters; ++itr) {                          /* :658 */
        const PoseCur &pc = st.p[itr & 1];
        PoseCur &pn = st.p[(itr + 1) & 1];
#pragma unroll
        for (int k = 0; k < 9; k++) c.r[k] = uniform_f(pc.Rf[k]);    /* :673 */
#pragma unroll
        for (int k = 0; k < 3; k++) c.t[k] = uniform_f(pc.tf[k]);    /* :674 */
        typedef Acc7T<false, true> AccS;      /* one workgroup, all iterations: nothing to gain from the certificate, the limbs ride along */
        AccS a;
        acc7_zero(a);
        bool any_odd = false;
        accumulate_points2<BLOCK, false, TEX_G16, 2, 0, false, AccS>(c, ts, lp, gpts, 0, N, lane_off, a, any_odd);
        {
            const double chk = ((a.g[0] + a.g[1]) + (a.g[2] + a.g[3])) + ((a.g[4] + a.g[5]) + a.e2);
            any_odd |= (__builtin_amdgcn_ballot_w64(!__builtin_isfinite(chk)) != 0ull);
        }
        if (any_odd) {
            acc7_zero(a);
            accumulate_points_exact<BLOCK, false, false, false, AccS>(c, tex, ts, nullptr, lp, gpts, 0, N, lane_off, a);
        }
        wave_sums7(a, red);
        wave_sums_e2(a, red2);
        double neg_step = 0.0;
        if (wave == 0) neg_step = pose_neg_step(uc, itr);
        __syncthreads();
        if (wave == 0) {
            const double sl = block_sum8<BLOCK>(red, lane & 7);
            double psi[6];
            pose_direction_lanes(st, uc, neg_step, sl, lane, psi);
            if (lane == 0) pose_apply(st, pc, pn, uc, psi);
        } else if (wave == 1 && lane == 0) {
            const double e2 = e2_from_limbs(block_sum8<BLOCK>(red2, 0), block_sum8<BLOCK>(red2, 1), block_sum8<BLOCK>(red2, 2), block_sum8<BLOCK>(red, 6));
            energy[itr] = pose_bookkeep(st, pc, itr, N, e2, (int)block_sum8<BLOCK>(red, 7));      /* :690 */
        }
        __syncthreads();
        if (st.stop) break;                                          /* :877 */
        if (tid == 0 && itr + 1 < iters) pose_regulariser_precompute(st, pn, uc);
    }
    __syncthreads();
    if (tid == 0) {                                                  /* :997-1005, then the next level's :642-657 */
        pose_state_finish(st);
        for (int k = 0; k < 9; k++) Rt12[k] = st.R[k];
        for (int k = 0; k < 3; k++) Rt12[9 + k] = st.p[0].t[k];
        *best_idx = st.bestItr;
        *ratio = st.bestRatio;
        if (next_iters > 0) { pose_state_begin(st); pose_regulariser_precompute(st, st.p[0], st.u); }
    }
    for (int i = tid; i < next_iters; i += BLOCK) next_energy[i] = 0.0f;      /* :634 */
    __syncthreads();
    {
        const unsigned long long *src = reinterpret_cast<const unsigned long long *>(&st);
        unsigned long long *dst = reinterpret_cast<unsigned long long *>(st_out);
        for (int i = tid; i < (int)(sizeof(PoseState) / 8); i += BLOCK) dst[i] = src[i];
    }
}
hipError_t launch_tiled_level_solo(const LevelSlab &L, int pair, int level, const Intrinsics &K, const void *st_in, void *st_out, int iters,
                                   int n_points, float *energy, double *Rt12, int *best_idx, float *ratio, float *next_energy,
                                   int next_iters, hipStream_t s) {
    hipLaunchKernelGGL(tiled_level_solo_kernel, dim3(1), dim3(DVO_STEP_THREADS), 0, s, L, pair, level, K, (const PoseState *)st_in,
                       (PoseState *)st_out, iters, n_points, energy, Rt12, best_idx, ratio, next_energy, next_iters);
    return hipGetLastError();
}

}  // namespace dvo
