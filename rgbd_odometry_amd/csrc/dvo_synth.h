/*
 * dvo_synth.h -- seeded synthetic RGB-D edge scene generator (see dvo_synth.cpp).
 * Plain C ABI so tests/bench can drive it through ctypes.  Host only.
 */
#ifndef DVO_SYNTH_H_
#define DVO_SYNTH_H_
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct dvo_synth_scene dvo_synth_scene;

/* Level sizes follow cv::resize(..., scale, scale): cvRound(dim * 2^-level). */
int dvo_synth_level_rows(int H, int level);
int dvo_synth_level_cols(int W, int level);

dvo_synth_scene *dvo_synth_create(int W, int H, int n_levels, uint64_t seed);
/* sparse scenes: n_seg segments (<= 0: the default 60 W/320) drawn inside the columns [0, x_frac W) only */
dvo_synth_scene *dvo_synth_create_ex(int W, int H, int n_levels, uint64_t seed, int n_seg, double x_frac);
void dvo_synth_destroy(dvo_synth_scene *sc);

int dvo_synth_rows(const dvo_synth_scene *sc, int level);
int dvo_synth_cols(const dvo_synth_scene *sc, int level);
/* all images column-major rows x cols */
const int32_t *dvo_synth_ref_edge(const dvo_synth_scene *sc, int level);   /* 0 / 255 */
const float   *dvo_synth_ref_depth(const dvo_synth_scene *sc, int level);  /* mm */
const int32_t *dvo_synth_now_edge(const dvo_synth_scene *sc, int level);
const float   *dvo_synth_now_dt(const dvo_synth_scene *sc, int level);
const float   *dvo_synth_now_gx(const dvo_synth_scene *sc, int level);
const float   *dvo_synth_now_gy(const dvo_synth_scene *sc, int level);
void dvo_synth_intrinsics(const dvo_synth_scene *sc, float *fx_fy_cx_cy);  /* level-0 K */
void dvo_synth_true_pose(const dvo_synth_scene *sc, double *R9_colmajor, double *t3);

#ifdef __cplusplus
}
#endif
#endif
