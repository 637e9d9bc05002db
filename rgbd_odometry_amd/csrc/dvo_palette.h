/*
 * dvo_palette.h -- layout of the compact ("P4") form of a now level: 4 bytes per pixel instead of the 16-byte texel
 * {DT, gx, gy, w} of dvo_device_math.h.  Shared by the builder (dvo_palette.hip), the fused kernel (dvo_fused.hip) and the
 * host code.  Internal.
 *
 * Why: the fused alignment kernel is bound by the number of memory requests that leave the L2: every
 * iteration touches the distinct 128-byte lines under the reprojected contour once, and with 16-byte texels a line holds
 * 8 pixels of which a one-pixel-wide contour uses 2-3.  The four floats of a texel are redundant for every now level the
 * reference can produce: DT is the normalised distance transform (SolveDVO.cpp:1768-1795: few hundred distinct values per
 * image), gx / gy are imageGradient(DT) (:1063-1098: 0.5*(DT[x+1]-DT[x-1]), 0.5*(DT[y+1]-DT[y-1]), reflect-101 borders) and
 * w = getWeightOf(DT) (:1047-1053).  So a pixel is fully described by the RANK of its DT value in the image's sorted list of
 * distinct values (the "palette") and the ranks of its four neighbours.
 *
 *   palette : per (pair, level) up to DVO_PAL_MAX entries {P = DT value, W = getWeightOf(P)} (float2), sorted by P;
 *             copied into LDS at the start of a level
 *   P4 slab : lines of 128 bytes = 4 pixel columns x 8 stored rows of one dword:  6 interior image rows plus the row above
 *             and the row below (an apron, so that the vertical neighbours of every interior pixel are in the SAME line and
 *             one 12-byte load fetches up / centre / down).  Image borders hold the reflect-101 rows.  24 pixels per line
 *             instead of 8: modelled on the bench scenes, 97 k -> 51 k distinct lines per 640x480x4x10 alignment.
 *   dword   : bits 3..15  rank * 8   (byte offset of the palette entry; rank < DVO_PAL_MAX = 8192)
 *             bits 16..23 rank(x+1) - rank, signed     bits 24..31 rank(x-1) - rank, signed
 *             (apron / border entries carry their pixel's rank only)
 *   sentinel: line 0 of every image is not a tile: its 32 words point at palette entry n (one past the real entries), which
 *             is {0, 0}.  A lane without a visible point gathers from offset 0 and so decodes DT = gx = gy = w = 0 -- exact
 *             zeros in every sum -- without a single select.
 *
 * Two builders.  NATIVE (round 3, dvo_frames.hip): the engine's own distance transform knows every pixel's integer squared
 * distance d2, and DT = (float)sqrt(d2) * scale is a non-decreasing function of it -- the palette is the sorted list of the
 * d2 values present (a presence bitmap + prefix popcounts: no hashing, no sorting), a pixel's rank one table look-up; the
 * compact form is then the ONLY form the distance-transform stage writes (16-byte texels are decoded from it on demand for the
 * inspection / host-driven paths).  Exact by construction: palette value, weight and gradients are produced by the very
 * expressions the 16-byte path used.  GENERIC (dvo_palette.hip): for caller-supplied float images (dvo_set_now_level):
 *
 * The form is LOSSLESS BY VERIFICATION: the generic builder re-derives {DT, gx, gy, w} of every pixel from the palette exactly as
 * the kernel will and compares them bit for bit with the 16-byte texel; one mismatch (a caller-supplied gradient that is not
 * imageGradient(DT), more than 4095 distinct values, a rank step beyond +-127, NaN / negative DT) and the level keeps
 * the 16-byte path for that pair.  Results are therefore identical bit for bit whichever form is read.
 */
#ifndef DVO_PALETTE_H_
#define DVO_PALETTE_H_

#include <hip/hip_runtime.h>
#include <stddef.h>

#define DVO_PAL_MAX 8192          /* palette entries per (pair, level) incl. the sentinel: the 13-bit rank field of a word */
#define DVO_PAL_BUILD_MAX 4096    /* limit of the GENERIC builder (dvo_palette.hip: LDS hash sets); the distance-transform path
                                     (dvo_frames.hip) ranks by a bitmap of squared distances and fills the whole format */
/* PARTIAL compact forms (round 5; native builder only).  An image with more distinct distances than the palette holds, with a
 * pixel beyond the range of the presence bitmap (512 px or more from every edge) or with a horizontal rank step beyond +-127 used
 * to be REFUSED as a whole (pal_n = -reason: 16-byte texels for every look-up).  But the points of an alignment land near the now
 * frame's edges, i.e. on SMALL distances = low ranks (the palette is sorted by distance).  A partial form keeps the lowest
 * DVO_PAL_CAP_PARTIAL ranks; every other pixel -- and every pixel whose step does not fit -- carries the rank of a NaN entry
 * {NaN, NaN} that every native palette now has (index count + 1, right after the zero sentinel at index count).  A look-up that meets it poisons
 * its lane's sums; the kernel's once-per-iteration finiteness test then sends that wave through the literal scalar path on the
 * image's 16-byte texels, which such an image also gets (pal_n carries DVO_PAL_PARTIAL: the texels are its complete form, the
 * compact form its fast path).  bench.py `sparse_scenes`, tests/test_gpu_sparse_scenes.py. */
#define DVO_PAL_PARTIAL (1 << 20)
#define DVO_PAL_CAP_PARTIAL 4094
#define DVO_P4_ROWS 6             /* interior image rows per 128-byte line (8 stored rows) */
/* squared distances the native builder can rank: bits of the per-image presence bitmap (distances below 512 pixels) */
#define DVO_EDT_BITMAP_BITS (1 << 18)

namespace dvo {

/* reasons for "no compact form" (pal_n = -reason) */
enum { PAL_BAD_VALUE = 1, PAL_TOO_MANY = 2, PAL_STEP = 3, PAL_GRADIENT = 4, PAL_WEIGHT = 5, PAL_SHAPE = 6, PAL_FAR = 7 };

/* pal_n > 0: entries of the palette (the zero sentinel sits at that index, a partial form's NaN entry one further), with
 * DVO_PAL_PARTIAL or-ed in for a partial form */
__host__ __device__ inline int pal_count(int pal_n) { return pal_n > 0 ? (pal_n & (DVO_PAL_PARTIAL - 1)) : 0; }
__host__ __device__ inline bool pal_partial(int pal_n) { return pal_n > 0 && (pal_n & DVO_PAL_PARTIAL) != 0; }
__host__ __device__ inline int p4_tiles_per_col(int rows) { return (rows + DVO_P4_ROWS - 1) / DVO_P4_ROWS; }
/* dwords of one image */
__host__ __device__ inline size_t p4_count(int rows, int cols) {
    return (size_t)p4_tiles_per_col(rows) * (size_t)((cols + 3) >> 2) * 32u + 32u;      /* + the sentinel line */
}
/* dword index of stored row `srow` (0 = apron above, 1..6 interior, 7 = apron below) of tile row `ty`, pixel column xx */
__host__ __device__ inline size_t p4_slot(int ty, int srow, int xx, int tiles_per_col) {
    return 32u + ((size_t)(xx >> 2) * tiles_per_col + ty) * 32u + (size_t)((xx & 3) * 8 + srow);
}

}  // namespace dvo
#endif
