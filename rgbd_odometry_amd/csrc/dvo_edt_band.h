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

/* (3) above: one band of one image -- two tile rows of the compact image = 12 interior rows + the apron row above and below = 14 rows,
 * every column.  LDS: ONE tile [PAD + cols + PAD][16 row slots as 8 dwords] -- min(g^2, 65535) pairs while the scan runs, the
 * pixels' rank pairs afterwards (a thread keeps the ranks of its <= NI items in registers across the barrier between the two
 * uses) -- and the first 256 words of the rank table (squared distances below 8192; the rest is read from memory): 22.5 KB at 640
 * columns, seven workgroups per CU.  A lane scans EIGHT rows of one column (one ds_read_b128 per neighbour column and side, four
 * packed minima / saturating additions per dword): 2.4 instructions per pixel and step against 4.1 for two rows per lane. */
constexpr int EB_T = 2, EB_NR = 6 * EB_T + 2, EB_CD = 8;       /* tile rows per band, its rows, dwords per tile column (16 row slots) */
constexpr int EB_LDS_LUT_WORDS = 256;
typedef unsigned eb_v4 __attribute__((ext_vector_type(4)));
static_assert(EB_NR <= 2 * EB_CD && EB_NR <= 26, "a band's rows fit the column's slots and, with the first at bit 31 at worst, one 64-bit window");
DVO_DEV eb_v4 eb_mk4(unsigned x, unsigned y, unsigned z, unsigned w) { eb_v4 v; v.x = x; v.y = y; v.z = z; v.w = w; return v; }
DVO_DEV unsigned eb_rank_global(unsigned v) {
    return 2u + EDT_SOS_LUT.pre[v >> 5] + (unsigned)__popc(EDT_SOS_LUT.bm[v >> 5] & ((1u << (v & 31u)) - 1u));
}
template <int NI>                                              /* items (column halves) per thread: 2 * cols <= 256 * NI */
__global__ void __launch_bounds__(256, (NI <= 5) ? 7 : 5) edt_band_levels_kernel(const EdtBandLevels t) {
    constexpr int NC = (NI + 1) / 2;                            /* columns per thread while the tile is built */
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
    unsigned *tq = reinterpret_cast<unsigned *>(eb_lds4);
    unsigned *lbm = tq + (size_t)(cols + 2 * EB_PAD) * EB_CD;
    unsigned short *lpre = reinterpret_cast<unsigned short *>(lbm + EB_LDS_LUT_WORDS);

    /* ---- the band's rows of g from the column words: every load first, then the arithmetic ---- */
    const int ya = band == 0 ? 0 : 6 * EB_T * band - 1;
    const int yb = (rows - 1 < 6 * EB_T * (band + 1)) ? rows - 1 : 6 * EB_T * (band + 1);
    const int n = yb - ya + 1;                                   /* 2 .. 14 rows */
    const int w0 = ya >> 5, w1 = yb >> 5, p0 = ya - 32 * w0, nbits = (w1 - w0 + 1) * 32;
    {
        const int nwords = t.nwords[l];
        const unsigned *mk = t.maskT[l] + (size_t)by * nwords * cols, *cr = t.carryT[l] + (size_t)by * nwords * cols;
        unsigned ma[NC], mb[NC], ca[NC], cb[NC];
#pragma unroll
        for (int k = 0; k < NC; k++) {
            const int col = tid + 256 * k, cc = col < cols ? col : cols - 1;
            ma[k] = mk[(size_t)w0 * cols + cc]; ca[k] = cr[(size_t)w0 * cols + cc];
            mb[k] = mk[(size_t)w1 * cols + cc]; cb[k] = cr[(size_t)w1 * cols + cc];
        }
        lbm[tid] = EDT_SOS_LUT.bm[tid];
        if (tid < EB_LDS_LUT_WORDS / 2) reinterpret_cast<unsigned *>(lpre)[tid] = reinterpret_cast<const unsigned *>(EDT_SOS_LUT.pre)[tid];
        if (tid < EB_PAD * EB_CD / 4) {
            eb_lds4[tid] = eb_mk4(~0u, ~0u, ~0u, ~0u);
            eb_lds4[(size_t)(EB_PAD + cols) * (EB_CD / 4) + tid] = eb_mk4(~0u, ~0u, ~0u, ~0u);
        }
#pragma unroll
        for (int k = 0; k < NC; k++) {
            const int col = tid + 256 * k;
            if (col >= cols) break;
            const unsigned long long M = (w1 != w0) ? ((unsigned long long)ma[k] | ((unsigned long long)mb[k] << 32)) : (unsigned long long)ma[k];
            const int U = (int)(ca[k] & 0xffffu), D = (int)(cb[k] >> 16);
            const unsigned W = (unsigned)(M >> p0);                /* bit r: an edge at the band's row r (p0 + 13 < 64) */
            const unsigned long long Mlo = M & ((1ull << p0) - 1ull);
            int d = Mlo ? (p0 - 1) - (63 - __clzll((long long)Mlo)) : U + p0;      /* row ya - 1 to the nearest edge at or above it */
            int up[EB_NR];
#pragma unroll
            for (int r = 0; r < EB_NR; r++) { d = (W & (1u << r)) ? 0 : d + 1; up[r] = d; }
            const int pend = p0 + n - 1;
            const unsigned long long Mhi = M >> (pend + 1);
            d = Mhi ? __ffsll((long long)Mhi) - 1 : D + (nbits - 1 - pend);        /* row yb + 1 to the nearest edge at or below it */
            unsigned q[2 * EB_CD];
#pragma unroll
            for (int r = 2 * EB_CD - 1; r >= 0; r--) {
                if (r < EB_NR && r < n) {
                    d = (W & (1u << r)) ? 0 : d + 1;
                    const unsigned gd = (unsigned)(up[r] < d ? up[r] : d);
                    q[r] = gd > 255u ? 65535u : gd * gd;
                } else q[r] = 0u;                                /* row slots past the band: finished before they start */
            }
            eb_v4 *dst = eb_lds4 + (size_t)(EB_PAD + col) * (EB_CD / 4);
            dst[0] = eb_mk4(q[0] | (q[1] << 16), q[2] | (q[3] << 16), q[4] | (q[5] << 16), q[6] | (q[7] << 16));
            dst[1] = eb_mk4(q[8] | (q[9] << 16), q[10] | (q[11] << 16), q[12] | (q[13] << 16), q[14] | (q[15] << 16));
        }
    }
    __syncthreads();
    EDT_T(t_staged);
    EDT_ACC(0, t_begin, t_staged);

    /* ---- the row scan, eight rows per lane; the ranks of a thread's items stay in registers ---- */
    typedef __attribute__((address_space(3))) const eb_v4 lds_c4;
    const unsigned tq_lds = (unsigned)(size_t)(lds_c4 *)eb_lds4;
    const int total = cols * 2;
    unsigned mx = 0;
    bool far = false;
    eb_v4 res[NI];
    auto rank_of = [&](unsigned v) -> unsigned {
        if (v < (unsigned)EB_LDS_LUT_WORDS * 32u) return 2u + lpre[v >> 5] + (unsigned)__popc(lbm[v >> 5] & ((1u << (v & 31u)) - 1u));
        if (v < EB_D2_END) return eb_rank_global(v);
        return (unsigned)EB_NANR;
    };
    auto ranks2 = [&](unsigned w) -> unsigned {
        const unsigned v0 = w & 0xffffu, v1 = w >> 16;
        far = far || v0 >= EB_D2_END || v1 >= EB_D2_END;
        mx = v0 > mx ? v0 : mx; mx = v1 > mx ? v1 : mx;
        return rank_of(v0) | (rank_of(v1) << 16);
    };
#pragma unroll
    for (int it = 0; it < NI; it++) {
        const int p = it * 256 + tid;
        if (it * 256 + (tid & ~63) >= total) { res[it] = eb_mk4(0u, 0u, 0u, 0u); continue; }      /* wave-uniform */
        const int cp = p < total ? p : total - 1;
        const int h = cp & 1;
        const int ctr = (EB_PAD * 2 + cp) * 16;                  /* byte offset of the item's four dwords */
        eb_v4 best = eb_lds4[EB_PAD * 2 + cp];
        if (p >= total) best = eb_mk4(0u, 0u, 0u, 0u);
        unsigned S = 0x00010001u, Dd = 0x00030003u;
        int la = ctr - 4 * 32, ra = ctr;
        const int la_min = h * 16, ra_max = (EB_PAD + cols + EB_PAD - 5) * 32 + h * 16;
        EDT_T(t_s0);
        int trip = 0;
        for (; trip < 2 * EB_TRIPS; trip++) {                    /* four steps per trip */
            const edt_us2 s2 = edt_as_us2(S);
            const unsigned open = edt_as_u32(__builtin_elementwise_sub_sat(edt_as_us2(best.x), s2)) | edt_as_u32(__builtin_elementwise_sub_sat(edt_as_us2(best.y), s2)) |
                                  edt_as_u32(__builtin_elementwise_sub_sat(edt_as_us2(best.z), s2)) | edt_as_u32(__builtin_elementwise_sub_sat(edt_as_us2(best.w), s2));
            if (__builtin_amdgcn_ballot_w64(open != 0u) == 0ull) break;               /* i^2 >= best everywhere */
            unsigned lac = tq_lds + (unsigned)(la > la_min ? la : la_min), rac = tq_lds + (unsigned)(ra < ra_max ? ra : ra_max);
            asm volatile("" : "+v"(lac), "+v"(rac));               /* the four offsets of either side go into the instructions */
            lds_c4 *ql = (lds_c4 *)(size_t)lac, *qr = (lds_c4 *)(size_t)rac;
            eb_v4 a[4], b[4];
#pragma unroll
            for (int j = 1; j <= 4; j++) { a[j - 1] = ql[(4 - j) * 2]; b[j - 1] = qr[j * 2]; }
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
            la -= 4 * 32; ra += 4 * 32;
        }
        EDT_T(t_s1);
        EDT_ACC(1, t_s0, t_s1);
#ifdef DVO_EDT_STAMPS
        acc_t[6] += (unsigned long long)trip;
#endif
        res[it] = eb_mk4(ranks2(best.x), ranks2(best.y), ranks2(best.z), ranks2(best.w));
        EDT_T(t_s2);
        EDT_ACC(2, t_s1, t_s2);
    }
    __syncthreads();                                             /* every scan is done: the tile becomes the ranks */
    EDT_T(t_scanned);
#pragma unroll
    for (int it = 0; it < NI; it++) {
        const int p = it * 256 + tid;
        if (p < total) {
            eb_lds4[EB_PAD * 2 + p] = res[it];
            const int col = p >> 1, h = p & 1;
            if (col == 1) eb_lds4[(EB_PAD - 1) * 2 + h] = res[it];              /* reflect-101: column -1 is column 1 ... */
            if (col == cols - 2) eb_lds4[(EB_PAD + cols) * 2 + h] = res[it];     /* ... and column `cols` is column cols - 2 */
        }
    }
    __syncthreads();

    /* ---- rank words: the band's two lines of every tile column, contiguous in memory.  A thread's place in the 64 words of a
     *      tile column is fixed; it walks the tile columns tid / 64, + 4, ... ---- */
    const int tpc = t.tpc[l], ntc = (cols + 3) >> 2;
    const int ty0 = band * EB_T, nty = (tpc - ty0 < EB_T) ? tpc - ty0 : EB_T;
    unsigned *p4 = t.p4[l] + (size_t)(t.first_pair + by) * t.p4_stride[l];
    if (band == 0 && tid < 32) p4[tid] = 0u;                       /* the sentinel line: palette entry 0 */
    bool part = false;
    {
        const int rem = tid & 63, tyl = rem >> 5, wd = rem & 31, xl = wd >> 3, srow = wd & 7;
        const int ty = ty0 + tyl, ys = ty * DVO_P4_ROWS + srow - 1;
        const bool row_ok = tyl < nty && ys <= rows;
        const bool interior = srow >= 1 && srow <= DVO_P4_ROWS && ys < rows;
        const int r = reflect101(ys <= rows ? ys : rows, rows) - ya;
        typedef __attribute__((address_space(3))) const unsigned short lds_cs;
        for (int tc = tid >> 6; tc < ntc; tc += 4) {
            const int xx = tc * 4 + xl;
            unsigned word = 0u;
            if (row_ok && xx < cols) {
                lds_cs *pc = (lds_cs *)(size_t)(tq_lds + (unsigned)((EB_PAD + xx) * 32 + 2 * r));
                int c = (int)pc[0];
                if (interior) {
                    int dr = (int)pc[16] - c, dl = (int)pc[-16] - c;
                    if ((unsigned)(dr + 127) > 254u || (unsigned)(dl + 127) > 254u) { c = EB_NANR; dr = 0; dl = 0; }    /* this pixel is looked up in the 16-byte texels */
                    word = (((unsigned)dr & 0xffu) << 16) | (((unsigned)dl & 0xffu) << 24);
                }
                if (c == EB_NANR) part = true;
                word |= (unsigned)c << 3;
            }
            if (tyl < nty) p4[32u + ((size_t)tc * tpc + ty) * 32u + wd] = word;
        }
    }
    EDT_T(t_words);
    EDT_ACC(3, t_scanned, t_words);

    /* ---- the image's maximum, its flags; the last band of an image puts it on the list if its form is partial.  The atomics are
     *      agent-scope read-modify-writes: one that has RETURNED has been performed, no fence (and none of its cache write-backs) ---- */
    const int mred = block_reduce_256<true>((int)mx);
    const int any_far = __syncthreads_or(far ? 1 : 0), any_part = __syncthreads_or(part ? 1 : 0);
    if (tid == 0) {
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
        const unsigned slot = ((blockIdx.y * gridDim.x + blockIdx.x) * 4u + (threadIdx.x >> 6)) & (unsigned)(EDT_STAMP_SLOTS - 1);
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
