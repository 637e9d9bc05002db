/*
 * dvo_launch.h -- internal interface between the C-ABI host code (dvo_capi.cpp)
 * and the HIP kernels (dvo_kernels.hip).  Not installed; plain structs only.
 */
#ifndef DVO_LAUNCH_H_
#define DVO_LAUNCH_H_

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "dvo_device_math.h"

#define DVO_LEVELS 8          /* == DVO_MAX_LEVELS of include/dvo_amd.h */
#define DVO_NACC 29           /* == DVO_NUM_ACC */
#define DVO_NACC_PAD 32
/* where the fused kernel reads the now level of a (pair, level) from */
enum { DVO_TEXMODE_GLOBAL16 = 0, DVO_TEXMODE_LDS16 = 1, DVO_TEXMODE_PAL4 = 2,
       DVO_TEXMODE_EXACT_RAN = 0x100 /* flag: a wave of the packed kernel took the literal-division fallback at this level */,
       DVO_TEXMODE_PT4 = 0x200       /* flag: the level's reference points were read in their 4-byte form */,
       DVO_TEXMODE_E2_SHIFT = 12, DVO_TEXMODE_E2_MAX = 0xffff /* bits 12..27: iterations of the level whose energy came from the exact sweep (round 6) */,
       DVO_TEXMODE_RANKS_LDS = 0x400 /* flag (with DVO_TEXMODE_PAL4): the level's ranks were looked up in an LDS copy of the whole level (round 5) */ };

namespace dvo {

/* One pyramid level of every pair of a context ("slab" layout in HBM):
 *   tex : n_pairs x tex_stride texels {DT, gx, gy, 0} (16 B / pixel), pixel (yy,xx) of pair p at
 *         tex[p*tex_stride + texel_index(yy, xx, tiles_per_col)]   (tiled, see dvo_device_math.h)
 *   pts : n_pairs x pt_cap x 3 floats, point i of pair p at pts[(p*pt_cap + i)*3]
 *         (the reference's 3xN column-major SpaceCordList, 12 B / point)
 *   N   : n_pairs ints                                                          */
struct LevelSlab {
    const float4 *tex;
    const float *pts;
    const uint2 *cpts;      /* compact points {xx | yy << 16, Z}, pt_cap per pair; valid where the host says so (Schedule.compact) */
    const unsigned *cidx;   /* index of each compact point in the 3 x N list (block order -> reference order), pt_cap per pair */
    const unsigned *cpt4;   /* 4-byte points (pt4_decode), pt_cap per pair */
    const unsigned *chdr;   /* chunk headers of the 4-byte points, pt_cap / 64 per pair */
    const int *pt4_ok;      /* per pair: the 4-byte list is valid (NULL: never built) */
    const int *N;
    size_t tex_stride;      /* texels per pair */
    int pt_cap;             /* points per pair (capacity) */
    int rows, cols;
    /* compact form of the now level (dvo_palette.h); pal_n[p] > 0: pair p has one (its palette size), <= 0: it has not */
    const unsigned *p4;     /* n_pairs x p4_stride rank words */
    const float2 *pal;      /* n_pairs x DVO_PAL_MAX palette entries {DT value, weight} */
    const int *pal_n;       /* n_pairs (NULL: never built) */
    size_t p4_stride;       /* dwords per pair */
};
struct LevelSet { LevelSlab l[DVO_LEVELS]; };

struct Schedule {
    int n_levels;
    int iters[DVO_LEVELS];
    int e_off[DVO_LEVELS];   /* offset of level l inside a pair's energy block */
    int e_stride;            /* floats per pair = sum iters */
    int last_level;          /* smallest l with iters[l] > 0 (its outputs survive, SolveDVO.cpp:2102) */
    int flags;
    int alias_mod;           /* diagnostics: data of pair p % alias_mod (0 = off) */
    int lds_points;          /* reference points kept resident in LDS per workgroup (3 words each, 2 when compact) */
    int lds_bytes;           /* dynamic LDS of the launch (dvo_fused.hip splits it per level between points and the now level) */
    int no_lds_tex;          /* diagnostics: never stage the now level into LDS */
    int no_p4;               /* diagnostics: never read the compact form of a now level (dvo_palette.h) */
    int team;                /* workgroups per pair of the packed kernel (1 = none; > 1: team mode, see dvo_fused.hip) */
    int n_pairs_launch;      /* pairs of this launch (team mode maps workgroups to pairs itself) */
    int force_e2;            /* tests (engine_variant 5): the energy certificate always fails -> every iteration takes the exact sweep of the residuals */
    int force_exact;         /* tests: every wave takes the literal-division fallback of the packed kernel (accumulate_points_exact) */
    int compact;             /* every pair/level of this launch has a compact point list: read 8 B / point instead of 12 */
    int no_pt4;              /* diagnostics: never read the 4-byte form of a reference list (DVO_POINTS4=off) */
    int pt4_factor;          /* 4-byte points for lists of at least this many times what the LDS holds as 8-byte points (default 3) */
    int final_blk;           /* host bookkeeping: the final outputs of this launch are stored in the compact lists' (block) order */
    int team_no_plain;       /* diagnostics (DVO_TEAM_PLAIN_STORES=off): team records always travel as sc1 stores, even inside one XCD */
    int no_r16;              /* diagnostics (DVO_RANKS_LDS=off): never stage a coarse level's ranks into LDS */
    unsigned team_epoch0;    /* team mode: tags of this launch's records start above every tag an earlier launch left in the buffer (no memset between launches) */
    int team_solo_max;       /* team mode: a level with at most this many points is run by member 0 alone, the others pick its pose up
                                at the level's end (dvo_fused.hip: solo levels; DVO_TEAM_SOLO_MAX, 0 = every level by the whole team) */
};

#define DVO_TILED_SOLO_MAX_DEFAULT 6144     /* tiled / wide schedule: levels of at most this many points run as one launch (dvo_fused.hip) */
#define DVO_TEAM_SOLO_MAX_DEFAULT 0          /* measured and not taken, see dvo_fused.hip: solo levels */

struct Intrinsics { float fx, fy, cx, cy; int interp; /* dvo_params.interpolate_dt, travels with the camera model to every kernel */ };

struct Outputs {
    double *poses;           /* n_pairs x 12 : R[9] col-major, t[3] */
    float *energy;           /* n_pairs x e_stride */
    int *best_idx;           /* n_pairs x DVO_LEVELS */
    float *ratio;            /* n_pairs x DVO_LEVELS */
    float *final_eps;        /* n_pairs x final_cap */
    float *final_reproj;     /* n_pairs x 3*final_cap */
    int *final_N;            /* n_pairs */
    int final_cap;
    unsigned long long *dbg; /* diagnostics (DVO_STAMPS builds): n_pairs x 64 counters, else NULL */
    double *H;               /* DVO_FLAG_NORMAL_MATRIX: n_pairs x e_stride x 21 (upper triangle of sum w J J^T per iterate), else NULL */
    double *team_buf;        /* team mode: n_pairs x 2 x 16 x 8 records of 16 bytes {value, tag} (partial sums of the members, double-buffered; tags of earlier launches are all below Schedule.team_epoch0 + 1) */
    unsigned *team_cnt;      /* (unused by the kernel; the int after n_pairs entries is the error flag) */
    int *team_err;           /* set to 1 by a member that gave up waiting (members not co-resident) */
    int *tex_mode;           /* n_pairs x DVO_LEVELS: where the fused kernel read the now level from (DVO_TEXMODE_*), inspection */
    const int *order;        /* launch order of the pairs (longest first): workgroup b aligns pair first_pair + order[b]; NULL = b */
};

hipError_t launch_pack_texels(const float *dt, const float *gx, const float *gy, float4 *out,
                              int rows, int cols, hipStream_t s);
/* compact form of now levels [first_pair, first_pair+count) of one pyramid level from their 16-byte texels (dvo_palette.hip) */
size_t palette_work_ints(int count);        /* scratch of one launch_palette_build over `count` images */
hipError_t launch_palette_build(const float4 *tex, size_t tex_stride, int rows, int cols, unsigned *p4, size_t p4_stride,
                                float2 *pal, int *pal_n, int first_pair, int count, unsigned *work, hipStream_t s);
bool fused_uses_compact(int points_in_flight, int interp);
hipError_t launch_replicate_level(float4 *tex, const unsigned char *src_has_tex /* device, n_src flags, or NULL = all */, size_t tex_stride, float *pts,
                                  uint2 *cpts, unsigned *cidx, unsigned *cpt4, unsigned *chdr, int *pt4_ok, int pt_cap, int *N, int n_src, int dst_first,
                                  int dst_count, hipStream_t s);
/* 4-byte twins of the compact lists of pairs [first_pair, first_pair + count): encode, decode with pt4_decode, compare with the
 * 8-byte form; pt4_ok[pair] = 1 only if every point survives */
/* a caller's 3xN list -> {xx | yy << 16, Z} where every point verifies bit for bit (dvo_frames.hip); *fail is set otherwise */
hipError_t launch_points_recover_compact(const float *xyz, int N, int level, const Intrinsics &K, uint2 *compact, unsigned *cidx, int *fail, hipStream_t s);
hipError_t launch_points4_build(const uint2 *cpts, const int *N, int pt_cap, int rows, unsigned *cpt4, unsigned *chdr, int *pt4_ok,
                                int first_pair, int count, hipStream_t s);
/* final outputs stored in block order -> the reference's order: out[cidx[i]] = in[i] */
hipError_t launch_final_permute(const unsigned *cidx, const float *fe_blk, const float *fr_blk, int n, float *fe, float *fr, hipStream_t s);
/* the compact now form of slots [dst_first, dst_first+dst_count) <- that of pair (p - dst_first) % n_src */
hipError_t launch_replicate_compact(unsigned *p4, size_t p4_stride, float2 *pal, int *pal_n, int n_src, int dst_first,
                                    int dst_count, hipStream_t s);
hipError_t launch_align_fused(int block_threads, int points_in_flight, const LevelSet &lv,
                              const Schedule &sc, const Intrinsics &K, const DevParams &prm,
                              const Outputs &out, int first_pair, int n_pairs, hipStream_t s);
/* the same schedule on two points per lane in packed float32 (dvo_fused.hip); needs sc.compact */
hipError_t launch_align_fused2(int block_threads, const LevelSet &lv, const Schedule &sc, const Intrinsics &K,
                               const DevParams &prm, const Outputs &out, int first_pair, int n_pairs, hipStream_t s);
size_t fused2_static_lds(int block_threads, bool with_h = false);
/* per-point dump at a float pose (inspection) */
hipError_t launch_eval_points(const LevelSlab &L, int pair, int level, const Intrinsics &K,
                              const float *Rf, const float *tf,
                              float *reproj, float *J, float *eps, float *w, int *vis, hipStream_t s);
/* 29 accumulators at a float pose: partials [nblocks x 32] then acc[32] */
hipError_t launch_accumulate(const LevelSlab &L, int pair, int level, const Intrinsics &K,
                             const float *Rf, const float *tf, int first_point, int n_points,
                             double *partials, int nblocks, double *acc, hipStream_t s);
int accumulate_blocks_for(int n_points);
/* host-driven iteration: optimiser state (opaque, pose_state_bytes() each) in HBM */
size_t pose_state_bytes();
hipError_t launch_iter_begin(void *state, const DevParams &prm, const double *Rt12, float *energy, int max_iters, hipStream_t s);
hipError_t launch_iter_accumulate(const LevelSlab &L, int pair, int level, const Intrinsics &K, const void *state,
                                  int first_point, int n_points, double *partials, int nblocks, double *acc,
                                  hipStream_t s);
hipError_t launch_iter_update(void *state, const DevParams &prm, int itr, int n_total, const double *acc,
                              float *energy, hipStream_t s);
hipError_t launch_iter_step_fused(const LevelSlab &L, int pair, int level, const Intrinsics &K, void *state,
                                  const DevParams &prm, int itr, int n_points, double *partials, int nblocks,
                                  float *energy, hipStream_t s);
hipError_t launch_iter_end(void *state, double *Rt12, int *best_idx, float *ratio, hipStream_t s);
/* one launch per iteration (round 4, dvo_kernels.hip: tiled_step_kernel): pending update of iteration itr - 1 (apply_prev) from the
 * reduced sums acc_in and the state st_in -> st_out, this rank's point range at the new pose, the 32 sums of the launch in acc_out */
int tiled_step_blocks(int n_points, int n_cu);
hipError_t launch_tiled_step(const LevelSlab &L, int pair, int level, const Intrinsics &K, const DevParams &prm, const void *st_in,
                             void *st_out, const double *acc_in, int itr, int apply_prev, int n_total, int first_point, int n_points,
                             double *partials, unsigned *ticket, double *acc_out, float *energy, int nblocks,
                             double *H_prev /* NULL: the 21 H sums are not formed; else: where H of iterate itr - 1 goes (21 doubles) */, hipStream_t s);
/* the same launch with the packed point loop over the compact list (dvo_fused.hip: tiled_step_pk_kernel) */
hipError_t launch_tiled_step_pk(const LevelSlab &L, int pair, int level, const Intrinsics &K, const void *st_in, void *st_out,
                                const double *acc_in, int itr, int apply_prev, int n_total, int first_point, int n_points,
                                double *partials, unsigned *ticket, double *acc_out, float *energy, int nblocks, double *H_prev, hipStream_t s);
/* a small level of the schedule as ONE launch of one workgroup: all its iterations, then what launch_tiled_finish does (dvo_fused.hip) */
hipError_t launch_tiled_level_solo(const LevelSlab &L, int pair, int level, const Intrinsics &K, const void *st_in, void *st_out, int iters,
                                   int n_points, float *energy, double *Rt12, int *best_idx, float *ratio, float *next_energy,
                                   int next_iters, hipStream_t s);
hipError_t launch_tiled_finish(const void *st_in, void *st_out, const DevParams &prm, const double *acc_in, int itr_last, int n_total,
                               float *energy, double *Rt12, int *best_idx, float *ratio, double *H_last, float *next_energy, int next_iters,
                               hipStream_t s);      /* next_iters > 0: also what iter_begin does for the next level (its energies: next_energy) */
/* finalEpsilons / finalReprojections of points [first, first+n) at the best iterate kept in `state` (host-driven / tiled paths) */
hipError_t launch_final_outputs_state(const LevelSlab &L, int pair, int level, const Intrinsics &K, const void *state,
                                      int first_point, int n_points, float *final_eps, float *final_reproj, int *final_N,
                                      hipStream_t s);
hipError_t launch_unpack_texels(const float4 *tex, int rows, int cols, float *dt, float *gx, float *gy, hipStream_t s);
/* one wave asleep for `us` microseconds of real time (dvo_set_keep_warm) */
hipError_t launch_keep_warm(int us, hipStream_t s);
/* SE(3) helpers on one lane (property tests) */
hipError_t launch_se3_exp(const double *psi, double *Rt12, hipStream_t s);
hipError_t launch_se3_log(const double *Rt12, double *psi, hipStream_t s);
hipError_t launch_rotationize(double *R9, hipStream_t s);

/* ---- per-frame preprocessing (dvo_frames.hip).  Every launcher is batched over g.count same-geometry
 * images; image b of a buffer is at base + b*stride (in elements of that buffer). ---- */
struct ImgBatch { int rows, cols, count; };

/* host-format image -> resident column-major grey (u8) / depth (f32 mm).  dtype: 0 u8, 1 u16, 2 f32 */
hipError_t launch_import_grey(const void *src, int dtype, int row_major, size_t src_stride,
                              unsigned char *grey, size_t stride, ImgBatch g, hipStream_t s);
hipError_t launch_import_depth(const void *src, int dtype, int row_major, size_t src_stride,
                               float *depth_mm, size_t stride, ImgBatch g, hipStream_t s);
/* row f2: full-resolution BGR8 (+ depth in metres, may be NULL) row-major -> pyramid level decimated by 2^shift.
 * SrcTab (round 6): DEVICE arrays of image pointers, one per image of the launch; non-NULL = the images are read where they are (camera
 * frames already in HBM) instead of at base + i * stride */
struct SrcTab { const void *const *bgr; const void *const *depth; };
hipError_t launch_gather_images(const void *const *src, int count, void *dst, size_t bytes, size_t stride, hipStream_t s, int max_wgs_per_image = 64);
hipError_t launch_camera_level(const unsigned char *bgr, size_t bgr_stride, const float *depth_m, size_t depth_stride,
                               int src_rows, int src_cols, int shift, const short2 *umap_xy, const unsigned short *umap_frac,
                               int depth_raw, unsigned char *grey, float *depth_mm, size_t stride, ImgBatch g, hipStream_t s, SrcTab tab = {nullptr, nullptr});
hipError_t launch_camera_levels(const unsigned char *bgr, size_t bgr_stride, const float *depth_m, size_t depth_stride, int src_rows, int src_cols,
                                int n, const int *shift, const int *rows, const int *cols, const short2 *umap_xy, const unsigned short *umap_frac,
                                int depth_raw, unsigned char *const *grey, float *const *depth_mm, const size_t *stride, int count, hipStream_t s,
                                SrcTab tab = {nullptr, nullptr});
/* levels 1 .. n of a pyramid as nearest-neighbour decimations of its level 0 (valid when camera_levels_decimate_ok: no clamped index) */
bool camera_levels_decimate_ok(int n_levels, const int *rows, const int *cols);
hipError_t launch_camera_decimate_levels(const unsigned char *grey0, const float *depth0, size_t stride0, int rows0, int cols0, int n,
                                         const int *rows, const int *cols, unsigned char *const *grey, float *const *depth, const size_t *stride,
                                         int count, hipStream_t s);
/* row f1: cv::Canny(grey, low/high as squared integer thresholds, 3, L2).  work: canny_work_ints() ints;
 * edge out: 0/255 u8 */
size_t canny_work_ints(int rows, int cols, int count);
/* Canny of all pyramid levels of the same images in four launches (dvo_frames.hip) */
size_t canny_levels_work_ints(int n, const int *rows, const int *cols, int count);
bool canny_levels_ok(int n, const int *rows, const int *cols, unsigned char *const *edge, const size_t *edge_stride);
hipError_t launch_canny_levels(int n, const int *rows, const int *cols, const unsigned char *const *grey, const size_t *grey_stride,
                               unsigned char *const *edge, const size_t *edge_stride, int count, int low, int high, int *work, hipStream_t s);
hipError_t launch_canny(const unsigned char *grey, size_t stride, ImgBatch g, int low, int high, int *work,
                        unsigned char *edge, size_t edge_stride, hipStream_t s);
hipError_t launch_count_edges(const unsigned char *edge, size_t n, int *out, hipStream_t s);
/* edge mask -> distance transform -> the now level's resident form (SolveDVO.cpp:1768-1795, :1063-1098), for pairs
 * first_pair .. first_pair + g.count - 1.  p4 != NULL: the COMPACT form (dvo_palette.h) is written natively from the integer
 * squared distances -- p4 / pal / pal_n are the level's slabs (indexed by pair), tex_out (already offset to first_pair) only
 * receives the 16-byte texels of images the compact form cannot hold (pal_n < 0 then).  p4 == NULL: 16-byte texels of every
 * image.  work: edt_work_ints() ints */
/* launch_edges_to_now for all pyramid levels of the same images at once (dvo_frames.hip) */
bool edt_levels_ok(int n, const int *rows, const int *cols);
size_t edt_levels_work_ints(int n, const int *rows, const int *cols, int count);
hipError_t launch_edges_to_now_levels(int n, const int *rows, const int *cols, const unsigned char *const *edge, const size_t *edge_stride, int count,
                                      int *work, float4 *const *tex_out, const size_t *tex_stride, unsigned *const *p4, const size_t *p4_stride,
                                      float2 *const *pal, int *const *pal_n, int first_pair, hipStream_t s,
                                      bool only_texels = false /* only the pass that writes 16-byte texels (for images without a compact form); work as left by a full run */);
/* caller-supplied float images -> compact form (see dvo_frames.hip); work: float_level_work_ints() ints */
size_t float_level_work_ints(int rows, int cols);
hipError_t launch_float_level_to_compact(const float *dt, const float *gx, const float *gy, int rows, int cols, int *work,
                                         unsigned *p4, size_t p4_stride, float2 *pal, int *pal_n, int pair, hipStream_t s);
size_t edt_work_ints(int rows, int cols, int count);
hipError_t launch_edges_to_now(const unsigned char *edge, size_t edge_stride, ImgBatch g, int *work,
                               float4 *tex_out, size_t tex_stride, unsigned *p4, size_t p4_stride, float2 *pal, int *pal_n,
                               int first_pair, hipStream_t s, bool only_texels = false);
/* 16-byte texels of pairs [first_pair, first_pair + count) decoded from their compact form (pairs without one are skipped) */
hipError_t launch_p4_decode_texels(const unsigned *p4, size_t p4_stride, const float2 *pal, const int *pal_n, float4 *tex,
                                   size_t tex_stride, int rows, int cols, int first_pair, int count, hipStream_t s);
/* selectedPts + enlistRefEdgePts (SolveDVO.cpp:1230-1264, :224-264).  col_counts: (cols+2) ints per image;
 * after the count pass col_counts[cols] and [cols+1] hold N.  edge: int32 or u8 (>0 = edge).
 * blk_counts (may be NULL): enlist_block_ints(rows, cols) ints per image -- per (pixel column, 16-row segment) counts in
 * block order; with it the compact twin list is written in 16x16-block order (dvo_frames.hip), the 3xN float list and uv
 * always in the reference's order. */
size_t enlist_block_ints(int rows, int cols);
hipError_t launch_enlist_count(const void *edge, int edge_is_u8, size_t edge_stride, const float *depth_mm,
                               size_t depth_stride, ImgBatch g, int *col_counts, int *blk_counts, hipStream_t s);
hipError_t launch_enlist_write(const void *edge, int edge_is_u8, size_t edge_stride, const float *depth_mm,
                               size_t depth_stride, ImgBatch g, int level, const Intrinsics &K, const int *col_counts,
                               const int *blk_counts, float *xyz, size_t xyz_stride,
                               uint2 *compact /* same stride in points / 3, or nullptr */, unsigned *cidx /* likewise */,
                               float *uv, int capacity, int *N_dst, hipStream_t s);

/* ---- photometric Gauss-Newton (dvo_photo.hip): RGBDOdometry's engine ---- */
hipError_t launch_photo_reference(const unsigned char *grey, const float *depth, int rows, int cols, int level,
                                  double fx, double fy, double cx, double cy, int fixed, double grad_threshold, int capacity,
                                  int *col_work /* 2*(cols+1) ints */, double *J, int *sel, double *zref, float *gref, double *A36,
                                  int *n_out, hipStream_t s);
hipError_t launch_photo_gauss_newton(const double *J, const int *sel, const double *zref, const float *gref, const int *n_dev,
                                     const double *A36, const unsigned char *grey_now, int rows, int cols, int level,
                                     double fx, double fy, double cx, double cy, int fixed, int max_iters, double eps_stop,
                                     double *T16, double *eps_norms, int *updates, double *eps_dump, hipStream_t s);

}  // namespace dvo
#endif
