/*
 * dvo_edt_band.h -- round 6: edge mask -> COMPACT now level in one pass over the pixels (row f1 of SURVEY.md section 8;
 * computeDistTransfrmOfNow, src/SolveDVO.cpp:1768-1795, + imageGradient :1063-1098).  Included by dvo_frames.hip, inside
 * namespace dvo, after the kernels of the three-pass stage (columns -> rows -> rank pack), which remains the path of images this
 * one cannot hold (below) and of everything wider than EB_MAX_COLS / taller than EB_MAX_ROWS.
 *
 * What changed against the three-pass stage, and why it may:
 *
 *   (1) The palette is no longer "the squared distances PRESENT in the image" but "every squared distance an image CAN have":
 *       the sums of two squares below EB_D2_END, in order (EDT_SOS_LUT: a bitmap + prefix popcounts, built at compile time).  A
 *       pixel's rank is then a property of its own d2 -- no presence bitmap, no image-wide dependency before the first rank word
 *       can be written -- and the rank words leave the kernel that computed the distances: g (2 B/pixel written + read), d2
 *       (4 + 4 B/pixel) and the bitmap never exist.  The palette VALUES still need the image's maximum (cv::normalize, :1774):
 *       a tiny launch writes them afterwards (edt_palette_levels_kernel), with the very expressions the three-pass stage uses
 *       (edt_value, weight_of), so every texel decodes to the same bits; entries the image does not use are simply never read.
 *       Measured on the bench scenes the universal palette is 10-22 % longer than the present one (1698 against 1392 entries at
 *       640x480).  Layout: [0] the zero sentinel, [1] the NaN entry of partial forms (both at FIXED indices now: a rank word
 *       must not depend on the palette's length), [2 + r] the r-th sum of two squares.
 *   (2) The column pass is not a pass over pixels any more: edt_colmask writes, per column and 32-row word, the edge bits and
 *       the distances from the word's first / last row to the nearest edge outside it (0.25 B/pixel, transposed so that a band
 *       reads whole lines).  A band -- two tile rows of the compact image = 12 interior rows + the apron row above and below,
 *       every column -- rebuilds g for its rows from two words per column in registers.
 *   (3) The row scan is the packed 16-bit scan of edt_rows_pk_body (wave-uniform step counter in scalar registers), EIGHT rows
 *       per lane, with the trip's two base addresses CLAMPED into 8 columns of "infinity" either side: it runs past the image
 *       border to any radius, so the exact per-lane finish and its second tile are gone.  It stops at radius 184: a pixel
 *       whose squared distance is EB_D2_END or more gets the NaN rank, its image a partial form (dvo_palette.h) and -- from a
 *       list the last band of every image appends to -- the three-pass stage's columns + rows, which leave the exact d2 for
 *       the 16-byte texels such an image also gets (and the exact maximum for its palette).
 *
 * HBM traffic per pixel: 1 B (edge) + 0.25 + 0.5 (masks, written and read) + 5.33 (rank words incl. aprons) = 7.1 B against
 * 19.8 measured for the three-pass stage (profiles/r05_frames/pmc_summary.txt).
 */
#ifndef DVO_EDT_BAND_H_
#define DVO_EDT_BAND_H_

constexpr int EB_PAD = 8;                               /* columns of "infinity" either side of the scanned tile: one trip */
constexpr unsigned EB_D2_END = 32000;                   /* squared distances below this have a rank: 8173 sums of two squares */
constexpr int EB_LUT_WORDS = (int)(EB_D2_END / 32);
constexpr int EB_TRIPS = 23;                            /* 184 steps: 185^2 > EB_D2_END, and (2 * 185 + 1, 185^2) fit 16 bits */
constexpr int EB_NANR = 1;                              /* palette index of the NaN entry; 0: the zero sentinel; 2 + r: real */
constexpr int EB_MAX_ROWS = 512, EB_MAX_COLS = 1024;    /* one 512-row chunk per column; tiles of both kinds within 64 KB of LDS */
static_assert(EB_D2_END % 32 == 0 && (8 * EB_TRIPS + 1) * (8 * EB_TRIPS + 1) >= (int)EB_D2_END && (8 * EB_TRIPS + 1) * (8 * EB_TRIPS + 1) < 65536, "scan radius");

struct EdtSosLut { unsigned bm[EB_LUT_WORDS]; unsigned short pre[EB_LUT_WORDS]; int total; };
constexpr EdtSosLut edt_sos_lut_make() {
    EdtSosLut t{};
    for (unsigned a = 0; a * a < EB_D2_END; a++)
        for (unsigned b = a; a * a + b * b < EB_D2_END; b++) { const unsigned v = a * a + b * b; t.bm[v >> 5] |= 1u << (v & 31u); }
    int run = 0;
    for (int w = 0; w < EB_LUT_WORDS; w++) {
        t.pre[w] = (unsigned short)run;
        unsigned v = t.bm[w];
        while (v) { v &= v - 1u; run++; }
    }
    t.total = run;
    return t;
}
__device__ const EdtSosLut EDT_SOS_LUT = edt_sos_lut_make();
static_assert(edt_sos_lut_make().total + 4 <= DVO_PAL_MAX, "every rank + the two fixed entries + the two entries a consumer copies beyond pal_n fit the 13-bit field");

/* level table of the fused stage (kernel argument) */
struct EdtBandLevels {
    int n, first_pair;
    int rows[DVO_LEVELS], cols[DVO_LEVELS], nwords[DVO_LEVELS], nbands[DVO_LEVELS], tpc[DVO_LEVELS], n_partial[DVO_LEVELS];
    unsigned firstA[DVO_LEVELS + 1], firstB[DVO_LEVELS + 1];
    const unsigned char *edge[DVO_LEVELS]; size_t edge_stride[DVO_LEVELS];
    unsigned *maskT[DVO_LEVELS], *carryT[DVO_LEVELS];      /* [image][word][column] */
    unsigned *imax[DVO_LEVELS]; int *done[DVO_LEVELS]; int *flags[DVO_LEVELS];
    const int *partial[DVO_LEVELS];                         /* the three-pass rows kernel's block maxima (images on the list) */
    unsigned *p4[DVO_LEVELS]; size_t p4_stride[DVO_LEVELS]; float2 *pal[DVO_LEVELS]; int *pal_n[DVO_LEVELS];
    int *list;                                              /* [0] entries, then (level << 24 | image) of every image with a partial form */
};

/* bit j = edge at row y0 + j of the column, j = 0..7 (rows past the image: 0) */
DVO_DEV unsigned eb_load_mask8(const unsigned char *col, bool vec, int y0, int rows) {
    if (y0 >= rows) return 0u;
    if (vec) {
        const uint2 v = *reinterpret_cast<const uint2 *>(col + y0);
        return edt_nonzero_bytes(v.x) | (edt_nonzero_bytes(v.y) << 4);
    }
    unsigned m = 0;
    for (int j = 0; j < 8; j++) if (y0 + j < rows && col[y0 + j] != 0) m |= 1u << j;
    return m;
}

/* (2) above.  512 threads: a workgroup takes 32 adjacent columns, a wave four of them (8 apart, all four loads in flight), a lane
 * eight rows; the 32-row words of the 32 columns meet in LDS and leave as whole 128-byte lines. */
__global__ void __launch_bounds__(512) edt_colmask_levels_kernel(const EdtBandLevels t) {
    __shared__ unsigned sm[16][33], sc[16][33];
    const int l = level_of_block(t.firstA, t.n, blockIdx.x);
    const int bx = (int)(blockIdx.x - t.firstA[l]), by = blockIdx.y;
    const int rows = t.rows[l], cols = t.cols[l], nwords = t.nwords[l];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (bx == 0 && tid == 0) { t.imax[l][by] = 0u; t.done[l][by] = 0; t.flags[l][by] = 0; }
    if (blockIdx.x == 0 && by == 0 && tid == 0) t.list[0] = 0;
    const unsigned char *img = t.edge[l] + (size_t)by * t.edge_stride[l];
    const int INF = DVO_EDT_INF(rows, cols);
    const int x0 = bx * 32;
    unsigned m8[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int xx = x0 + wave + 8 * k;
        const unsigned char *col = img + (size_t)(xx < cols ? xx : 0) * rows;
        const unsigned m = eb_load_mask8(col, ((rows & 7) == 0) && ((reinterpret_cast<size_t>(col) & 7) == 0), lane * 8, rows);
        m8[k] = xx < cols ? m : 0u;
    }
    const int w = lane >> 2;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const unsigned long long bal = __ballot(m8[k] != 0u);
        unsigned v = m8[k] | ((unsigned)__shfl_down((int)m8[k], 1) << 8);
        v |= (unsigned)__shfl_down((int)v, 2) << 16;                      /* lanes 4w: the word's 32 bits */
        const unsigned long long lower = bal & ((1ull << (4 * w)) - 1ull);
        const unsigned long long upper = (w == 15) ? 0ull : (bal & (~0ull << (4 * w + 4)));
        const int lpu = lower ? 63 - __clzll((long long)lower) : 0;
        const int lpd = upper ? __ffsll((long long)upper) - 1 : 0;
        const unsigned mlu = (unsigned)__shfl((int)m8[k], lpu), mld = (unsigned)__shfl((int)m8[k], lpd);
        /* U: from row 32w - 1 up to the nearest edge at or above it; D: from row 32(w + 1) down to the nearest at or below it */
        const int U = lower ? (32 * w - 1) - (lpu * 8 + (31 - __clz((int)mlu))) : INF;
        const int D = upper ? (lpd * 8 + (__ffs((int)mld) - 1)) - 32 * (w + 1) : INF;
        if ((lane & 3) == 0) { sm[w][wave + 8 * k] = v; sc[w][wave + 8 * k] = (unsigned)U | ((unsigned)D << 16); }
    }
    __syncthreads();
    unsigned *mk = t.maskT[l] + (size_t)by * nwords * cols, *cr = t.carryT[l] + (size_t)by * nwords * cols;
    for (int idx = tid; idx < nwords * 32; idx += 512) {
        const int ww = idx >> 5, c = idx & 31, xx = x0 + c;
        if (xx < cols) { mk[(size_t)ww * cols + xx] = sm[ww][c]; cr[(size_t)ww * cols + xx] = sc[ww][c]; }
    }
}

/* (3) above: one band of one image -- T tile rows of the compact image = 6T interior rows + the apron row above and below = NR
 * rows in NS row slots (a multiple of eight), every column.  Slot s stands for image row ya + s with ya = 6T * band - 1; row -1
 * is row 1 and row `rows` is row rows - 2 (reflect-101, cv::filter2D's border): those two slots are COPIES made while the tile is
 * built, so that nothing after it knows about borders in y, and two pad columns get the ranks of columns 1 and cols - 2 for the
 * same in x.  LDS: ONE tile [PAD + cols + PAD][NS / 2 dwords] -- min(g, 255)^2 pairs while the scan runs, the pixels' rank pairs
 * afterwards (a thread keeps the ranks of its <= NI items in registers across the barrier between the two uses) -- and the ranks
 * of the squared distances below EB_DIRECT as a plain table (larger ones go through the bitmap in memory).
 * A lane scans EIGHT rows of one column: one ds_read_b128 per neighbour column and side, then per dword a packed minimum, a
 * saturating packed addition of the step's i^2 and a packed minimum -- 1.5 vector instructions per pixel and step.
 * Two shapes: T = 2, 256 threads (14 of 16 slots used, 29 KB of LDS at 640 columns: small batches, many short workgroups) and
 * T = 5, 512 threads (32 of 32 slots, 50 KB: 30 of every 32 scanned rows are interior rows, against 12 of 16). */
constexpr int EB_DIRECT = 4096;
typedef unsigned eb_v4 __attribute__((ext_vector_type(4)));
struct EdtRank16 { unsigned short r[EB_DIRECT]; };
constexpr EdtRank16 edt_rank16_make() {
    const EdtSosLut L = edt_sos_lut_make();
    EdtRank16 t{};
    int run = 0;
    for (int v = 0; v < EB_DIRECT; v++) { t.r[v] = (unsigned short)(2 + run); if ((L.bm[v >> 5] >> (v & 31)) & 1u) run++; }
    return t;
}
__device__ const EdtRank16 EDT_RANK16 = edt_rank16_make();
DVO_DEV eb_v4 eb_mk4(unsigned x, unsigned y, unsigned z, unsigned w) { eb_v4 v; v.x = x; v.y = y; v.z = z; v.w = w; return v; }
DVO_DEV unsigned eb_rank_global(unsigned v) {
    return 2u + EDT_SOS_LUT.pre[v >> 5] + (unsigned)__popc(EDT_SOS_LUT.bm[v >> 5] & ((1u << (v & 31u)) - 1u));
}
DVO_DEV unsigned eb_pk_lo(unsigned hi_src, unsigned lo_src) { return __builtin_amdgcn_perm(hi_src, lo_src, 0x05040100u); }   /* (hi_src.lo16 << 16) | lo_src.lo16 */
DVO_DEV unsigned eb_pk_hi(unsigned hi_src, unsigned lo_src) { return __builtin_amdgcn_perm(hi_src, lo_src, 0x07060302u); }   /* (hi_src.hi16 << 16) | lo_src.hi16 */
template <int T, int THREADS, int NI>                          /* NI: items (8 rows of a column) per thread, cols * NS / 8 <= THREADS * NI */
__global__ void __launch_bounds__(THREADS, (THREADS == 256) ? ((NI <= 5) ? 5 : 4) : 6) edt_band_levels_kernel(const EdtBandLevels t) {
    constexpr int NR = 6 * T + 2, NS = (NR + 7) & ~7, CD = NS / 2, LPC = NS / 8, CB = CD * 4, C4 = CD / 4;
    constexpr int NPI = (NI * THREADS / LPC / 2 + THREADS - 1) / THREADS;      /* column pairs per thread while the tile is built */
    static_assert(NR <= 32 && (LPC & (LPC - 1)) == 0 && EB_PAD >= 4, "a band's rows lie in one 64-bit window of the column; a trip is four steps");
    extern __shared__ eb_v4 eb_lds4[];
    const int l = level_of_block(t.firstB, t.n, blockIdx.x);
    const int band = (int)(blockIdx.x - t.firstB[l]), by = blockIdx.y;
    const int rows = t.rows[l], cols = t.cols[l];
    if (rows < 2 || cols < 2) return;                            /* the palette launch reports PAL_SHAPE */
    const int tid = threadIdx.x;
#ifdef DVO_EDT_STAMPS
    unsigned long long acc_t[8] = {0, 0, 0, 0, 0, 0, 0, 0};      /* [0] table + g + barrier, [1] scan trips, [2] ranks, [3] rank words, [4] tail, [5] waves, [6] trips, [7] all */
#endif
    EDT_T(t_begin);
    unsigned short *lrank = reinterpret_cast<unsigned short *>(eb_lds4 + (size_t)(cols + 2 * EB_PAD) * C4);

    /* ---- the band's rows of g from the column words, TWO columns per thread in packed 16-bit arithmetic: every load first ---- */
    const int ya = 6 * T * band - 1, yb = ya + NR - 1;
    const int ra = ya < 0 ? 0 : ya, rb = yb > rows - 1 ? rows - 1 : yb;         /* the real rows among them */
    const int off = ra - ya, nreal = rb - ra + 1;                 /* slot of row ra (0, or 1 in band 0); 1 .. NR rows */
    const int s_bot = (yb >= rows) ? rows - ya : -1;              /* slot of row `rows` (a copy of row rows - 2), if the band has it */
    const int w0 = ra >> 5, w1 = rb >> 5, p0 = ra - 32 * w0, nbits = (w1 - w0 + 1) * 32, pend = p0 + nreal - 1;
    {
        const int nwords = t.nwords[l];
        const unsigned *mk = t.maskT[l] + (size_t)by * nwords * cols, *cr = t.carryT[l] + (size_t)by * nwords * cols;
        unsigned ld[NPI][8];
#pragma unroll
        for (int ip = 0; ip < NPI; ip++) {
            const int cA = 2 * (tid + THREADS * ip);
            const int a = cA < cols ? cA : cols - 1, b = cA + 1 < cols ? cA + 1 : cols - 1;
            ld[ip][0] = mk[(size_t)w0 * cols + a]; ld[ip][1] = mk[(size_t)w1 * cols + a]; ld[ip][2] = cr[(size_t)w0 * cols + a]; ld[ip][3] = cr[(size_t)w1 * cols + a];
            ld[ip][4] = mk[(size_t)w0 * cols + b]; ld[ip][5] = mk[(size_t)w1 * cols + b]; ld[ip][6] = cr[(size_t)w0 * cols + b]; ld[ip][7] = cr[(size_t)w1 * cols + b];
        }
        for (int i = tid; i < EB_DIRECT / 2; i += THREADS) reinterpret_cast<unsigned *>(lrank)[i] = reinterpret_cast<const unsigned *>(EDT_RANK16.r)[i];
        if (tid < EB_PAD * C4) {
            eb_lds4[tid] = eb_mk4(~0u, ~0u, ~0u, ~0u);
            eb_lds4[(size_t)(EB_PAD + cols) * C4 + tid] = eb_mk4(~0u, ~0u, ~0u, ~0u);
        }
        auto prep = [&](unsigned m0, unsigned m1, unsigned c0, unsigned c1, unsigned &W, unsigned &u0, unsigned &d0) {
            const unsigned long long M = (w1 != w0) ? ((unsigned long long)m0 | ((unsigned long long)m1 << 32)) : (unsigned long long)m0;
            W = (unsigned)(M >> p0);                               /* bit k: an edge at row ra + k */
            const unsigned long long Mlo = M & ((1ull << p0) - 1ull);
            u0 = (unsigned)(Mlo ? (p0 - 1) - (63 - __clzll((long long)Mlo)) : (int)(c0 & 0xffffu) + p0);       /* row ra - 1 to the nearest edge at or above it */
            const unsigned long long Mhi = M >> (pend + 1);         /* pend <= 62 */
            d0 = (unsigned)(Mhi ? __ffsll((long long)Mhi) - 1 : (int)(c1 >> 16) + (nbits - 1 - pend));        /* row rb + 1 to the nearest edge at or below it */
        };
#pragma unroll
        for (int ip = 0; ip < NPI; ip++) {
            const int cA = 2 * (tid + THREADS * ip);
            if (cA >= cols) break;
            unsigned WA, WB, uA, uB, dA, dB;
            prep(ld[ip][0], ld[ip][1], ld[ip][2], ld[ip][3], WA, uA, dA);
            prep(ld[ip][4], ld[ip][5], ld[ip][6], ld[ip][7], WB, uB, dB);
            const int NWA = (int)~WA, NWB = (int)~WB;
            edt_us2 up = edt_as_us2(eb_pk_lo(uB, uA)), dn = edt_as_us2(eb_pk_lo(dB, dA));
            const edt_us2 one = edt_as_us2(0x00010001u), c255 = edt_as_us2(0x00ff00ffu);
            unsigned nm[NR];                                       /* per row: 0xffff in the half of a column WITHOUT an edge there */
            edt_us2 U[NR];
#pragma unroll
            for (int k = 0; k < NR; k++) {
                nm[k] = eb_pk_lo((unsigned)__builtin_amdgcn_sbfe(NWB, k, 1), (unsigned)__builtin_amdgcn_sbfe(NWA, k, 1));
                up = edt_as_us2(edt_as_u32(up + one) & nm[k]);
                U[k] = up;
            }
            unsigned q[NR];
#pragma unroll
            for (int k = NR - 1; k >= 0; k--) {
                if (k < nreal) {
                    dn = edt_as_us2(edt_as_u32(dn + one) & nm[k]);
                    const edt_us2 gc = __builtin_elementwise_min(__builtin_elementwise_min(U[k], dn), c255);     /* g > 255: its square is beyond every rank */
                    q[k] = edt_as_u32(gc * gc);
                } else q[k] = 0u;
            }
            unsigned qs[NS];                                       /* by slot: row ra + k sits in slot k + off */
#pragma unroll
            for (int sl = 0; sl < NS; sl++) {
                const unsigned v0 = sl < NR ? q[sl] : 0u, v1 = (sl >= 1 && sl - 1 < NR) ? q[sl - 1] : 0u;
                qs[sl] = off ? v1 : v0;
            }
            if (off) qs[0] = qs[2];                                /* row -1 is row 1 */
#pragma unroll
            for (int sl = 2; sl < NR; sl++) if (sl == s_bot) qs[sl] = qs[sl - 2];      /* row `rows` is row rows - 2; the slots after it stay 0 */
            eb_v4 *dA4 = eb_lds4 + (size_t)(EB_PAD + cA) * C4;
#pragma unroll
            for (int c4 = 0; c4 < C4; c4++) {
                dA4[c4] = eb_mk4(eb_pk_lo(qs[8 * c4 + 1], qs[8 * c4]), eb_pk_lo(qs[8 * c4 + 3], qs[8 * c4 + 2]), eb_pk_lo(qs[8 * c4 + 5], qs[8 * c4 + 4]), eb_pk_lo(qs[8 * c4 + 7], qs[8 * c4 + 6]));
                if (cA + 1 < cols)
                    dA4[C4 + c4] = eb_mk4(eb_pk_hi(qs[8 * c4 + 1], qs[8 * c4]), eb_pk_hi(qs[8 * c4 + 3], qs[8 * c4 + 2]), eb_pk_hi(qs[8 * c4 + 5], qs[8 * c4 + 4]), eb_pk_hi(qs[8 * c4 + 7], qs[8 * c4 + 6]));
            }
        }
    }
    __syncthreads();
    EDT_T(t_staged);
    EDT_ACC(0, t_begin, t_staged);

    /* ---- the row scan, eight rows per lane; the ranks of a thread's items stay in registers ---- */
    typedef __attribute__((address_space(3))) const eb_v4 lds_c4;
    const unsigned tq_lds = (unsigned)(size_t)(lds_c4 *)eb_lds4;
    const int total = cols * LPC;
    unsigned mx = 0;
    bool far = false;
    eb_v4 res[NI];
    auto rank_of = [&](unsigned v) -> unsigned {
        if (v < (unsigned)EB_DIRECT) return lrank[v];
        if (v < EB_D2_END) return eb_rank_global(v);
        return (unsigned)EB_NANR;
    };
    /* the ranks of a dword's two squared distances; fast: both below EB_DIRECT (one table look-up each, no branch) */
    auto ranks2_fast = [&](unsigned w) -> unsigned {
        const edt_us2 c = __builtin_elementwise_min(edt_as_us2(w), edt_as_us2((unsigned)(EB_DIRECT - 1) * 0x00010001u));
        return (unsigned)lrank[c.x] | ((unsigned)lrank[c.y] << 16);
    };
    auto ranks2 = [&](unsigned w) -> unsigned {
        const unsigned v0 = w & 0xffffu, v1 = w >> 16;
        return rank_of(v0) | (rank_of(v1) << 16);
    };
#pragma unroll
    for (int it = 0; it < NI; it++) {
        const int p = it * THREADS + tid;
        if (it * THREADS + (tid & ~63) >= total) { res[it] = eb_mk4(0u, 0u, 0u, 0u); continue; }      /* wave-uniform */
        const int cp = p < total ? p : total - 1;
        const int h = cp & (LPC - 1), col = cp / LPC;
        const unsigned ctr = tq_lds + (unsigned)((EB_PAD + col) * CB + h * 16);      /* LDS address of the item's four dwords */
        eb_v4 best = *(lds_c4 *)(size_t)ctr;
        if (p >= total) best = eb_mk4(0u, 0u, 0u, 0u);
        unsigned S = 0x00010001u, Dd = 0x00030003u;
        unsigned la = ctr - 4u * CB, ra_ = ctr;
        const unsigned la_min = tq_lds + (unsigned)(h * 16), ra_max = tq_lds + (unsigned)((EB_PAD + cols + EB_PAD - 5) * CB + h * 16);
        EDT_T(t_s0);
        int trip = 0;
        for (; trip < 2 * EB_TRIPS; trip++) {                    /* four steps per trip */
            const edt_us2 m4 = __builtin_elementwise_max(__builtin_elementwise_max(edt_as_us2(best.x), edt_as_us2(best.y)), __builtin_elementwise_max(edt_as_us2(best.z), edt_as_us2(best.w)));
            const unsigned open = edt_as_u32(__builtin_elementwise_sub_sat(m4, edt_as_us2(S)));
            if (__builtin_amdgcn_ballot_w64(open != 0u) == 0ull) break;               /* i^2 >= best everywhere */
            unsigned lac = (int)la > (int)la_min ? la : la_min, rac = ra_ < ra_max ? ra_ : ra_max;
            asm volatile("" : "+v"(lac), "+v"(rac));               /* the four offsets of either side go into the instructions */
            lds_c4 *ql = (lds_c4 *)(size_t)lac, *qr = (lds_c4 *)(size_t)rac;
            eb_v4 a[4], b[4];
#pragma unroll
            for (int j = 1; j <= 4; j++) { a[j - 1] = ql[(4 - j) * C4]; b[j - 1] = qr[j * C4]; }
#pragma unroll
            for (int j = 1; j <= 4; j++) {
                const edt_us2 sj = edt_as_us2(S);
                auto step = [&](unsigned acc, unsigned x, unsigned y) -> unsigned {
                    return edt_as_u32(__builtin_elementwise_min(edt_as_us2(acc), __builtin_elementwise_add_sat(__builtin_elementwise_min(edt_as_us2(x), edt_as_us2(y)), sj)));
                };
                best.x = step(best.x, a[j - 1].x, b[j - 1].x); best.y = step(best.y, a[j - 1].y, b[j - 1].y);
                best.z = step(best.z, a[j - 1].z, b[j - 1].z); best.w = step(best.w, a[j - 1].w, b[j - 1].w);
                S += Dd; Dd += 0x00020002u;
            }
            la -= 4u * CB; ra_ += 4u * CB;
        }
        EDT_T(t_s1);
        EDT_ACC(1, t_s0, t_s1);
#ifdef DVO_EDT_STAMPS
        acc_t[6] += (unsigned long long)trip;
#endif
        unsigned mi;                                                /* the largest of the item's eight squared distances */
        {
            const edt_us2 m4 = __builtin_elementwise_max(__builtin_elementwise_max(edt_as_us2(best.x), edt_as_us2(best.y)), __builtin_elementwise_max(edt_as_us2(best.z), edt_as_us2(best.w)));
            mi = (unsigned)m4.x > (unsigned)m4.y ? (unsigned)m4.x : (unsigned)m4.y;
            mx = mi > mx ? mi : mx;
            far = far || mi >= EB_D2_END;
        }
        res[it] = eb_mk4(ranks2_fast(best.x), ranks2_fast(best.y), ranks2_fast(best.z), ranks2_fast(best.w));
        if (mi >= (unsigned)EB_DIRECT) res[it] = eb_mk4(ranks2(best.x), ranks2(best.y), ranks2(best.z), ranks2(best.w));      /* rare: far from every edge */
        EDT_T(t_s2);
        EDT_ACC(2, t_s1, t_s2);
    }
    __syncthreads();                                             /* every scan is done: the tile becomes the ranks */
    EDT_T(t_scanned);
#pragma unroll
    for (int it = 0; it < NI; it++) {
        const int p = it * THREADS + tid;
        if (p < total) {
            const int h = p & (LPC - 1), col = p / LPC;
            eb_lds4[(size_t)(EB_PAD + col) * C4 + h] = res[it];
            if (col == 1) eb_lds4[(size_t)(EB_PAD - 1) * C4 + h] = res[it];              /* reflect-101: column -1 is column 1 ... */
            if (col == cols - 2) eb_lds4[(size_t)(EB_PAD + cols) * C4 + h] = res[it];     /* ... and column `cols` is column cols - 2 */
        }
    }
    __syncthreads();

    /* ---- rank words, two at a time (stored rows 2 sp, 2 sp + 1 of a tile = slots 6 tyl + 2 sp and the next: one dword of the
     *      tile): own rank, signed rank steps to the horizontal neighbours; the band's T lines of a tile column are contiguous ---- */
    const int tpc = t.tpc[l], ntc = (cols + 3) >> 2;
    const int ty0 = band * T, nty = (tpc - ty0 < T) ? tpc - ty0 : T;
    unsigned *p4 = t.p4[l] + (size_t)(t.first_pair + by) * t.p4_stride[l];
    if (band == 0 && tid < 32) p4[tid] = 0u;                       /* the sentinel line: palette entry 0 */
    bool part = false;
    {
        typedef __attribute__((address_space(3))) const unsigned lds_cu;
        /* a thread's place among the 16 T word pairs of a tile column is fixed; the THREADS / (16 T) groups walk the tile columns */
        constexpr int PL = 16 * T, G = THREADS / PL;
        const int grp = tid / PL, rem = tid - grp * PL;
        const int tyl = rem >> 4, xl = (rem >> 2) & 3, sp = rem & 3;
        const int ty = ty0 + tyl;
        const int ys0 = ty * DVO_P4_ROWS + 2 * sp - 1;             /* image rows of the two words (-1 / rows: the reflected copies) */
        const bool in0 = sp >= 1 && ys0 < rows;                    /* stored row 2 sp: interior unless it is the apron above (sp = 0) */
        const bool in1 = sp <= 2 && ys0 + 1 < rows;                /* stored row 2 sp + 1: interior unless it is the apron below (sp = 3) */
        const unsigned keep = (in0 ? 0x0000ffffu : 0u) | (in1 ? 0xffff0000u : 0u);      /* the steps of interior words */
        const unsigned cmask = (ys0 + 1 <= rows) ? 0xffffffffu : 0x0000ffffu;         /* no stored row beyond `rows`: that word stays 0 */
        if (grp < G && tyl < nty) {
            unsigned lds_a = tq_lds + (unsigned)((EB_PAD + xl) * CB + (tyl * 3 + sp) * 4) + (unsigned)(grp * 4 * CB);
            unsigned *dst = p4 + 32u + ((size_t)grp * tpc + ty) * 32u + (xl * 8 + 2 * sp);
            for (int tc = grp; tc < ntc; tc += G, lds_a += (unsigned)(G * 4 * CB), dst += (size_t)G * tpc * 32u) {
                unsigned word0 = 0u, word1 = 0u;
                if (ys0 <= rows && tc * 4 + xl < cols) {       /* else: no such stored rows / columns, the words stay 0 */
                    lds_cu *pc = (lds_cu *)(size_t)lds_a;
                    const unsigned C = pc[0] & cmask, R = pc[CD], L = pc[-CD];
                    const edt_us2 dR = edt_as_us2(R) - edt_as_us2(C), dL = edt_as_us2(L) - edt_as_us2(C);
                    /* a step fits its signed byte iff step + 128 < 256 */
                    const unsigned bad = (edt_as_u32(dR + edt_as_us2(0x00800080u)) | edt_as_u32(dL + edt_as_us2(0x00800080u))) & 0xff00ff00u & keep;
                    unsigned steps = __builtin_amdgcn_perm(edt_as_u32(dL), edt_as_u32(dR), 0x06020400u) & keep;      /* bytes: dL.hi, dR.hi, dL.lo, dR.lo */
                    unsigned c0 = C & 0xffffu, c1 = C >> 16;
                    if (bad) {                                   /* rare: this pixel is looked up in the 16-byte texels */
                        if (bad & 0x0000ff00u) { c0 = EB_NANR; steps &= 0xffff0000u; }
                        if (bad & 0xff000000u) { c1 = EB_NANR; steps &= 0x0000ffffu; }
                        part = true;
                    }
                    word0 = (c0 << 3) | (steps << 16);
                    word1 = (c1 << 3) | (steps & 0xffff0000u);
                }
                *reinterpret_cast<uint2 *>(dst) = make_uint2(word0, word1);
            }
        }
    }
    EDT_T(t_words);
    EDT_ACC(3, t_scanned, t_words);

    /* ---- the image's maximum, its flags; the last band of an image puts it on the list if its form is partial.  The atomics are
     *      agent-scope read-modify-writes: one that has RETURNED has been performed, no fence (and none of its cache write-backs) ---- */
    __shared__ int s_red[THREADS / 64];
    for (int o = 32; o > 0; o >>= 1) { const unsigned v = (unsigned)__shfl_down((int)mx, o, 64); mx = v > mx ? v : mx; }
    if ((tid & 63) == 0) s_red[tid >> 6] = (int)mx;
    const int any_far = __syncthreads_or(far ? 1 : 0), any_part = __syncthreads_or(part ? 1 : 0);
    if (tid == 0) {
        int mred = 0;
        for (int k = 0; k < THREADS / 64; k++) mred = s_red[k] > mred ? s_red[k] : mred;
        unsigned seen = atomicMax(t.imax[l] + by, (unsigned)mred);
        if (any_far || any_part) seen += (unsigned)atomicOr(t.flags[l] + by, (int)EDT_FLAG_PARTIAL | (any_far ? (int)EDT_FLAG_FAR : 0));
        asm volatile("s_waitcnt vmcnt(0)" :: "v"(seen) : "memory");
        const bool last = atomicAdd(t.done[l] + by, 1) == t.nbands[l] - 1;
        if (last && (atomicOr(t.flags[l] + by, 0) & (int)EDT_FLAG_PARTIAL)) {
            const int k = atomicAdd(t.list, 1);
            t.list[1 + k] = (l << 24) | by;
        }
    }
#ifdef DVO_EDT_STAMPS
    {
        EDT_T(t_end);
        EDT_ACC(4, t_words, t_end);
        EDT_ACC(7, t_begin, t_end);
        acc_t[5] = 1;
        const unsigned slot = ((blockIdx.y * gridDim.x + blockIdx.x) * (unsigned)(THREADS / 64) + (threadIdx.x >> 6)) & (unsigned)(EDT_STAMP_SLOTS - 1);
        if ((threadIdx.x & 63) == 0)
            for (int k = 0; k < 8; k++) g_edt_stamp[slot][k] = acc_t[k];
    }
#endif
}

/* palette values, once the image's maximum is known (see (1) above): one workgroup per image and level */
__global__ void __launch_bounds__(256) edt_palette_levels_kernel(const EdtBandLevels t) {
    const int l = blockIdx.x, by = blockIdx.y, tid = threadIdx.x;
    const int rows = t.rows[l], cols = t.cols[l], pair = t.first_pair + by;
    if (rows < 2 || cols < 2) { if (tid == 0) t.pal_n[l][pair] = -(int)PAL_SHAPE; return; }
    __shared__ int s_max;
    const int f = t.flags[l][by];
    int m = 0;
    if (f & (int)EDT_FLAG_FAR) {                                   /* the exact maximum: the three-pass rows kernel ran for this image */
        const int *partial = t.partial[l] + (size_t)by * t.n_partial[l];
        for (int k = tid; k < t.n_partial[l]; k += 256) { const int v = partial[k]; m = v > m ? v : m; }
    } else if (tid == 0) m = (int)t.imax[l][by];
    m = block_reduce_256<true>(m);
    if (tid == 0) s_max = m;
    __syncthreads();
    const unsigned m2 = (unsigned)s_max;
    const EdtScale sc = edt_scale(m2, rows, cols);
    const unsigned m2c = m2 < EB_D2_END ? m2 : EB_D2_END - 1u;
    float2 *pal = t.pal[l] + (size_t)pair * DVO_PAL_MAX;
    const int nw = (int)(m2c >> 5) + 1;
    const float qnan = __uint_as_float(0x7fc00000u);
    if (tid == 0) { pal[0] = make_float2(0.0f, 0.0f); pal[1] = make_float2(qnan, qnan); }
    /* per-thread (vector) loads of the table on purpose: for a wave-uniform index the compiler (ROCm 7.2) splits 4 * w between the
     * base and the offset of an s_load_dword, whose base must be dword-aligned -- odd words then read their predecessor */
    for (int w = tid; w < nw; w += 256) {
        unsigned v = EDT_SOS_LUT.bm[w];
        int r = 2 + (int)EDT_SOS_LUT.pre[w];
        while (v) {
            const int b = __ffs((int)v) - 1;
            v &= v - 1u;
            const unsigned val = (unsigned)(w * 32 + b);
            if (val > m2c) break;
            const float P = edt_value(val, sc);
            pal[r] = make_float2(P, weight_of(P));                 /* getWeightOf, SolveDVO.cpp:1047-1053 */
            r++;
        }
        if (w == nw - 1) {                                       /* the owner of the last word knows the palette's length */
            pal[r] = make_float2(0.0f, 0.0f); pal[r + 1] = make_float2(0.0f, 0.0f);      /* copied by consumers, never referenced */
            t.pal_n[l][pair] = r | ((f & (int)EDT_FLAG_PARTIAL) ? DVO_PAL_PARTIAL : 0);
        }
    }
}

/* the three-pass stage's columns and rows for the images on the list (partial forms): their exact d2 and block maxima.
 * A few workgroups walk (entry, block) pairs; with an empty list -- nearly always -- they leave at once. */
struct EdtListShape { int gx_cols[DVO_LEVELS], gxmax_cols, gxmax_rows; };
template <int WAVES>
__global__ void __launch_bounds__(WAVES * 64) edt_columns8_list_kernel(const EdtLevels t, const EdtListShape sh, const int *__restrict__ list) {
    const int n = list[0];
    for (int k = blockIdx.x; k < n * sh.gxmax_cols; k += gridDim.x) {
        const int e = list[1 + k / sh.gxmax_cols], bx = k % sh.gxmax_cols, l = e >> 24, by = e & 0xffffff;
        if (bx < sh.gx_cols[l])
            edt_columns8_body<WAVES>(bx, sh.gx_cols[l], by, t.edge[l], t.edge_stride[l], t.rows[l], t.cols[l], t.R, t.g[l], t.bitmap[l], t.bm_words[l], t.flags[l], true);
        __syncthreads();
    }
}
template <int R>
__global__ void __launch_bounds__(256) edt_rows_pk_list_kernel(const EdtLevels t, const EdtListShape sh, const int *__restrict__ list) {
    const int n = list[0];
    for (int k = blockIdx.x; k < n * sh.gxmax_rows; k += gridDim.x) {
        const int e = list[1 + k / sh.gxmax_rows], bx = k % sh.gxmax_rows, l = e >> 24, by = e & 0xffffff;
        if (bx < t.n_partial[l])
            edt_rows_pk_body<R>(bx, t.n_partial[l], by, t.g[l], t.rows[l], t.cols[l], t.d2[l], t.partial[l], t.bitmap[l], t.bm_words[l], t.flags[l]);
        __syncthreads();
    }
}

#endif
