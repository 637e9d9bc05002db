/*
 * dvo_amd.h -- C ABI of the MI355X-native dense RGB-D edge-alignment engine.
 *
 * Drop-in boundary for ONE hot path of mpkuse/rgbd_odometry: the per-level pose
 * iteration loop SolveDVO::runIterations and its coarse-to-fine schedule.  The
 * reference has no plugin / FFI layer; the seam is the private method
 *   SolveDVO::runIterations          include/SolveDVO.h:228-230, src/SolveDVO.cpp:619-1017
 * and its helpers
 *   computeJacobianOfNowFrame        include/SolveDVO.h:312,     src/SolveDVO.cpp:306-414
 *   getReprojectedEpsilons           include/SolveDVO.h:313,     src/SolveDVO.cpp:425-462
 * called from SolveDVO::loop (src/SolveDVO.cpp:2097-2104, :2220-2227) and
 * casualTestFunction (:2426).  INTEGRATION.md shows the patch that makes the
 * reference's SolveDVO call these entry points.
 *
 * Conventions (identical to the reference's Eigen members):
 *   - images are COLUMN-major rows x cols float32, element (yy,xx) at yy + xx*rows
 *     (Eigen::MatrixXf; now_distance_transform / now_DT_gradientX/Y, SolveDVO.h:279-282)
 *   - reference edge points are 3 x N column-major float32 in metres
 *     (SpaceCordList, SolveDVO.h:137,304), in the column-major scan order of
 *     enlistRefEdgePts (SolveDVO.cpp:237-239)
 *   - pose is (R 3x3 column-major double, t 3 double), "now in ref":
 *     P_now = R^T (P_ref - t)  (SolveDVO.cpp:330); in/out like cR,cT
 *   - intrinsics are the LEVEL-0 K; level l uses diag(s,s,1)*K, s = 2^-l (:334-337)
 *
 * Ownership: the caller owns every host buffer (borrowed for the duration of the
 * call); the context owns all device memory.  One context per host thread and
 * HIP stream; calls on one context are not re-entrant.
 * Errors: int status (0 = DVO_OK); dvo_last_error() gives the message.  The
 * reference's behaviour on the same conditions is assert->abort (SolveDVO.h:124).
 * There is NO CPU fallback: without a HIP device every compute entry point
 * returns DVO_ERR_NO_DEVICE.
 */
#ifndef DVO_AMD_H_
#define DVO_AMD_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DVO_MAX_LEVELS 8
#define DVO_NUM_ACC 29      /* 21 H (upper triangle, row-major) + 6 g + sum eps^2 + n_visible */

enum {
    DVO_OK = 0,
    DVO_ERR_INVALID = 1,     /* bad argument */
    DVO_ERR_NO_DEVICE = 2,   /* no HIP device / runtime */
    DVO_ERR_HIP = 3,         /* HIP runtime error, see dvo_last_error */
    DVO_ERR_STATE = 4,       /* data for the requested pair/level not set */
    DVO_ERR_NOMEM = 5
};

/* flags for the align entry points */
enum {
    DVO_FLAG_FINAL_OUTPUTS = 1,   /* also produce finalEpsilons / finalReprojections (SolveDVO.cpp:1002-1003) */
    DVO_FLAG_IDENTITY_START = 2,  /* start from cR=I, cT=0 instead of the stored pose: what the reference does
                                     on a keyframe switch (SolveDVO.cpp:2210-2211); enqueue form only */
    DVO_FLAG_NORMAL_MATRIX = 4    /* also accumulate H = sum_i w_i J_i^T J_i (21 sums, double) in every iteration of the fused launch
                                     and keep it per iterate (dvo_get_level_normal_matrix): the other 21 of the "21+6" normal-equation
                                     accumulators.  The reference's update never forms H (SolveDVO.cpp:777 uses g only; the pattern is
                                     SolvePnP.cpp:168-182), so poses / energies are unchanged; costs throughput */
};

/* Every literal of SolveDVO::runIterations as a runtime parameter; defaults are
 * the reference's values (file:line in the comments). */
typedef struct dvo_params {
    double beta;               /* heavy-ball BETA = 0.5                       SolveDVO.cpp:653 */
    double precond_rot;        /* PFactor = .5, P = diag(1,1,1,.5,.5,.5)      :724-730 */
    double reg_lambda;         /* regularizationLambda = 0.05                 :742 */
    double step_a;             /* 9.0     stepLength = 9.0*1.0E-2/(...)       :773 */
    double step_b;             /* 1.0E-2                                      :773 */
    int    step_decay_after;   /* 5  : (itr>5) ? (itr-4) : 1                  :773 */
    int    step_decay_offset;  /* 4                                           :773 */
    float  trust_radius;       /* trustRegionHyperSphereRadius = 0.003 (float member) :25 */
    float  psi_norm_stop;      /* psiNormTerminationThreshold  = 1.0E-7 (float member) :24 */
    int    enable_rotationize; /* __ENABLE_ROTATIONIZE__       SolveDVO.h:107 */
    int    enable_l2_reg;      /* __ENABLE_L2_REGULARIZATION   SolveDVO.h:112 */
    int    interpolate_dt;     /* __INTERPOLATE_DISTANCE_TRANSFORM (off in the reference, SolveDVO.h:97): eps from
                                  SolveDVO::interpolate (:1285-1308) instead of the nearest lookup (:446) */
    int    block_threads;      /* engine tuning: threads per workgroup of the fused kernel (256/512/1024; 0 = chosen from the
                                  point-list sizes and the batch size) */
    int    points_in_flight;   /* engine tuning: reference points per lane and pipeline stage (1/2/4; 0 = default 1) */
    int    engine_variant;     /* engine tuning / diagnostics: 0 = auto; 1 = always the one-point-per-lane fused kernel;
                                  2 = packed kernel, but never stage a now level into LDS; 3 = packed kernel with every wave
                                  forced through its literal-division fallback (tests); 4 = packed kernel, but never the
                                  compact form of a now level (dvo_now_prepare); 5 = packed kernel with every iteration's energy
                                  taken from the exact sweep (dvo_get_level_energy_sweeps; tests) */
    int    lds_point_bytes;    /* engine tuning: LDS bytes per workgroup for the level's resident point list
                                  (0 = auto from block_threads, < 0 = none) */
    int    debug_alias_mod;    /* diagnostics only: if > 0, pair p reads the inputs of pair p % debug_alias_mod
                                  (shrinks the HBM working set without changing the arithmetic); 0 = off */
    int    canny_threshold1;   /* cv::Canny(img, edge, 150, 100, 3, true): the two thresholds (order-free, the detector */
    int    canny_threshold2;   /* swaps them), SolveDVO.cpp:1704,1764; used by the dvo_frame* entry points; 0,0 = 150,100 */
    int    team_size;          /* engine tuning: workgroups per frame pair of the fused launch when there are fewer pairs than compute
                                  units (each takes a contiguous share of every level's points; sums exchanged through L2 once per
                                  iteration, identical update on every member).  0 = auto (from the point counts and the number of pairs:
                                  at most 32 = one XCD; a single pair with >= 100 k points: 64 / 128 over all XCDs, two-stage exchange),
                                  1 = off, k = force k (<= 32, or 64 / 128 / 256 for a single pair) */
} dvo_params;

typedef struct dvo_ctx dvo_ctx;

/* ---- lifecycle ------------------------------------------------------------ */
int  dvo_params_default(dvo_params *p);
/* One frame pair, current HIP device.  Mirrors constructing a SolveDVO (SolveDVO.cpp:5-70). */
int  dvo_create(const dvo_params *p, dvo_ctx **out);
/* n_pairs independent frame pairs resident at once (batch / throughput mode). */
int  dvo_create_batch(const dvo_params *p, int n_pairs, dvo_ctx **out);
int  dvo_destroy(dvo_ctx *ctx);
const char *dvo_last_error(const dvo_ctx *ctx);      /* ctx may be NULL: last creation error */
int  dvo_num_pairs(const dvo_ctx *ctx);
/* Launch on an existing HIP stream (hipStream_t as void*; NULL = the HIP null stream, which is
 * what torch.cuda.current_stream().cuda_stream reports for torch's default stream).
 * dvo_use_own_stream() switches back to the context's private non-blocking stream. */
int  dvo_set_stream(dvo_ctx *ctx, void *hip_stream);
int  dvo_use_own_stream(dvo_ctx *ctx);
int  dvo_synchronize(dvo_ctx *ctx);
/* Single camera stream (the reference's loop: one frame every ~30 ms, SolveDVO.cpp:1945): between frames the GPU is idle.  A
 * process that only ever submits such sparse work can find the GPU parked at its lowest clocks and never raise them -- every
 * 0.5 ms alignment then takes 15-30 ms (measured: profiles/r03_single_stream).  dvo_set_keep_warm2 starts a host thread of the
 * context that keeps ONE wave busy on a stream of its own: launches of busy_us microseconds of real time, pause_us apart
 * (pause_us = 0: back to back); busy_us = 0 stops it.  dvo_set_keep_warm(period_us) = a 5 us launch every period_us.  Off by
 * default: a batch workload never idles.  DVO_KEEP_WARM="busy_us,pause_us" in the environment switches it on for every new
 * context.  (The alternative is an administrator pinning the performance level with rocm-smi; this needs no privileges.) */
int  dvo_set_keep_warm2(dvo_ctx *ctx, int busy_us, int pause_us);
int  dvo_set_keep_warm(dvo_ctx *ctx, int period_us);

/* ---- inputs ----------------------------------------------------------------
 * setCameraMatrix (SolveDVO.cpp:88-126): level-0 fx, fy, cx, cy as floats. */
int  dvo_set_intrinsics(dvo_ctx *ctx, float fx, float fy, float cx, float cy);

/* _ref_edge_3d[level] (SolveDVO.h:304) of pair `pair`: 3 x N floats. */
int  dvo_set_ref_level(dvo_ctx *ctx, int level, const float *xyz_3xN, int N);
int  dvo_set_ref_level_pair(dvo_ctx *ctx, int pair, int level, const float *xyz_3xN, int N);

/* now_distance_transform / now_DT_gradientX / now_DT_gradientY [level]
 * (SolveDVO.h:279-282), column-major rows x cols.  All pairs of one context
 * must use the same rows x cols per level. */
int  dvo_set_now_level(dvo_ctx *ctx, int level, const float *dt, const float *gx, const float *gy,
                       int rows, int cols);
int  dvo_set_now_level_pair(dvo_ctx *ctx, int pair, int level, const float *dt, const float *gx,
                            const float *gy, int rows, int cols);
/* Same, from DEVICE pointers (e.g. torch tensors); asynchronous on the context stream. */
int  dvo_set_ref_level_device(dvo_ctx *ctx, int pair, int level, const float *d_xyz_3xN, int N);
int  dvo_set_now_level_device(dvo_ctx *ctx, int pair, int level, const float *d_dt, const float *d_gx,
                              const float *d_gy, int rows, int cols);

/* computeDistTransfrmOfNow after the Canny step (SolveDVO.cpp:1768-1795) + imageGradient (:1063-1098) on
 * the GPU: edge (uint8, >0 = edge pixel, column-major rows x cols, host pointer) -> exact Euclidean
 * distance transform -> cv::normalize(0,255,NORM_MINMAX) (:1774) -> [-.5 0 .5] gradients with a
 * reflect-101 border (:1077-1090) -> resident now level.  1 byte per pixel crosses PCIe instead of 12.
 * (the Canny detector itself stays with the caller.) */
int  dvo_set_now_level_from_edges(dvo_ctx *ctx, int pair, int level, const unsigned char *edge, int rows, int cols);
/* The planar DT / gradient images of a resident now level (host outputs, rows*cols floats each, any may
 * be NULL). */
int  dvo_get_now_level(dvo_ctx *ctx, int pair, int level, float *dt, float *gx, float *gy);

/* The resident reference point list of a level (3 x N floats, host output of `capacity` points; *N_out = N). */
int  dvo_get_ref_level(dvo_ctx *ctx, int pair, int level, float *xyz_out, int capacity, int *N_out);

/* Batch set-up helper: pair slot p in [dst_first, dst_first+dst_count) becomes a device-side copy of
 * pair (p - dst_first) % n_src (all levels that are set), one launch per level.  dst_first = 0 leaves the
 * sources in place and fills the rest of the range cyclically. */
int  dvo_replicate_pairs(dvo_ctx *ctx, int n_src, int dst_first, int dst_count);

/* selectedPts + enlistRefEdgePts (SolveDVO.cpp:1230-1264, :224-264) on the GPU:
 * edge (int32, >0 = edge) and depth_mm (f32), column-major rows x cols, host
 * pointers.  Builds the 3xN list in the reference's column-major scan order and
 * installs it as the ref level of `pair`.  xyz_out / uv_out (host, capacity
 * points) may be NULL.  *N_out receives N. */
int  dvo_set_ref_level_from_images(dvo_ctx *ctx, int pair, int level, const int32_t *edge,
                                   const float *depth_mm, int rows, int cols,
                                   float *xyz_out, float *uv_out, int capacity, int *N_out);

/* ---- the hot path -----------------------------------------------------------
 * SolveDVO::runIterations (SolveDVO.cpp:619-1017) for pair 0.
 *   R[9], t[3]        in/out   cR, cT
 *   energy[max_iters] out      energyAtEachIteration (zero after an early exit, :634)
 *   final_eps[N]      out/NULL finalEpsilons
 *   final_reproj[3N]  out/NULL finalReprojections (3 x N column-major, row 2 = z*(1/z))
 *   best_idx          out      bestEnergyIndex (-1 if no iterate was accepted)
 *   visible_ratio     out      finalVisibleRatio */
int  dvo_run_iterations(dvo_ctx *ctx, int level, int max_iters, double *R, double *t,
                        float *energy, float *final_eps, float *final_reproj,
                        int *best_idx, float *visible_ratio);
int  dvo_run_iterations_pair(dvo_ctx *ctx, int pair, int level, int max_iters, double *R, double *t,
                             float *energy, float *final_eps, float *final_reproj,
                             int *best_idx, float *visible_ratio);

/* The level schedule of SolveDVO::loop (SolveDVO.cpp:2097-2104) fused into one
 * launch: for f = n_levels-1 .. 0: if iters[f] > 0: runIterations(f, iters[f], R, t).
 * Synchronous; R (9*n_pairs) and t (3*n_pairs) are in/out for pairs
 * [first_pair, first_pair+n_pairs). */
int  dvo_align_pyramid(dvo_ctx *ctx, int n_levels, const int *iters, int flags, double *R, double *t);
int  dvo_align_batch(dvo_ctx *ctx, int first_pair, int n_pairs, int n_levels, const int *iters,
                     int flags, double *R, double *t);

/* Asynchronous form: poses stay device-resident between calls (the reference
 * carries cR_64/cT_64 from frame to frame, SolveDVO.cpp:2102). */
int  dvo_set_poses(dvo_ctx *ctx, int first_pair, int n_pairs, const double *R, const double *t);
int  dvo_align_batch_enqueue(dvo_ctx *ctx, int first_pair, int n_pairs, int n_levels,
                             const int *iters, int flags);
int  dvo_get_poses(dvo_ctx *ctx, int first_pair, int n_pairs, double *R, double *t);   /* synchronises */

/* Per-level outputs of the last align call for `pair` (synchronises).
 * energy: iters[level] floats; any pointer may be NULL. */
int  dvo_get_level_report(dvo_ctx *ctx, int pair, int level, float *energy, int n_energy,
                          int *best_idx, float *visible_ratio);
/* H = sum_i w_i J_i^T J_i (6x6 symmetric, row-major, tangent order [translation(3), rotation(3)] like psi) at iterate
 * `itr` of `level` of the last align call made with DVO_FLAG_NORMAL_MATRIX; itr < 0 = the best iterate (:696).  With g
 * (not kept) it is the Gauss-Newton system of the reference's residual; H^-1 scaled by the residual variance is the usual
 * covariance estimate of the aligned pose. */
int  dvo_get_level_normal_matrix(dvo_ctx *ctx, int pair, int level, int itr, double *H36);
/* finalEpsilons / finalReprojections of the last level run (needs DVO_FLAG_FINAL_OUTPUTS). */
int  dvo_get_final_outputs(dvo_ctx *ctx, int pair, float *final_eps, float *final_reproj, int capacity,
                           int *N_out);

/* ---- host-driven iteration: one very large frame, or one frame tiled over several GPUs --------
 * runIterations (SolveDVO.cpp:619-1017) opened up at the only point where its per-point work
 * couples: the sum.  Per iteration the caller runs
 *     dvo_iter_accumulate(points [first, first+n) of this GPU's shard) -> 32 doubles on the device
 *     [all-reduce of those doubles over the GPUs that share the frame: RCCL, ncclSum]
 *     dvo_iter_update(the reduced sums)        -- the reference's 6-DoF update, on the device
 * The grid of the accumulate kernel spans all CUs, so this is also the path for frames whose
 * point lists are too long for one workgroup (1920x1080, 4096x3072).  All calls except
 * dvo_iter_begin/_end are asynchronous on the context stream.
 *   d_acc32: DEVICE pointer to 32 doubles: [0..20] H upper triangle, [21..26] g, [27] sum eps^2 as this GPU added it,
 *            [28] visible points, [29..31] the three 32-bit limbs of the EXACT sum of eps^2 (integers below 2^53 on a grid of
 *            2^-68: sums of them over shards are exact in any order; dvo_iter_update rounds the exact total once and takes the
 *            energy from that -- the same float on every rank and for every sharding; [27] is used only if a residual was
 *            outside [2^-11, 2^12), which no normalised distance is).
 *   n_total: number of reference points of the level over ALL shards (visible ratio, :457).
 * After an early termination (:877) later dvo_iter_update calls are no-ops, like the reference's break. */
int  dvo_iter_begin(dvo_ctx *ctx, int pair, int level, int max_iters, const double *R, const double *t);
int  dvo_iter_accumulate(dvo_ctx *ctx, int pair, int level, int first_point, int n_points, double *d_acc32);
int  dvo_iter_update(dvo_ctx *ctx, int pair, int level, int itr, int n_total, const double *d_acc32);
int  dvo_iter_end(dvo_ctx *ctx, int pair, int level, double *R, double *t, float *energy /*[max_iters] or NULL*/,
                  int *best_idx, float *visible_ratio);

/* The same loop for ONE GPU, enqueued from C: the level schedule of SolveDVO::loop with every iteration
 * spread over all CUs (frames whose point lists are too long for one workgroup).  Synchronous; per-level
 * energies / best index / ratio afterwards through dvo_get_level_report.  flags: DVO_FLAG_FINAL_OUTPUTS
 * (finalEpsilons / finalReprojections of the last level, SolveDVO.cpp:703-704, :1002-1003 -> dvo_get_final_outputs) and / or
 * DVO_FLAG_NORMAL_MATRIX (H = sum w J J^T of every iterate -> dvo_get_level_normal_matrix; + 40 % on every launch).
 * One launch per iteration (the update of an iteration rides at the head of the next launch), replayed as a graph. */
int  dvo_align_pyramid_wide(dvo_ctx *ctx, int pair, int n_levels, const int *iters, int flags, double *R, double *t);

/* ---- tiled mode from C: one large frame sharded over the GPUs of a node (BASELINE configs[4]) ----------
 * The same loop with the all-reduce done by RCCL over xGMI and everything enqueued from C on the context stream -- what a
 * C++ node calls (one process or thread per GPU, each with its own context and its rank's ncclComm_t):
 *     dvo_tiled_attach(ctx, comm, rank, world, NULL);           once
 *     dvo_align_pyramid_tiled(ctx, 0, n_levels, iters, flags, R, t);   per frame pair; every rank gets the same pose
 * Each rank must hold the SAME inputs for `pair` (full reference lists and now pyramid); rank r processes the contiguous
 * index range r of every level's list, the 32 sums are all-reduced (ncclDouble, ncclSum) per iteration, and every rank
 * executes the identical update.  nccl_comm: the caller's ncclComm_t.  rccl_library: path of the RCCL library the
 * communicator was created with; NULL = the RCCL already loaded in the process, else librccl.so.1.  libdvo_amd.so has no
 * link-time dependency on RCCL.  Replaces the reference seam include/SolveDVO.h:228-230 for frames one GPU cannot hold
 * or does not finish fast enough (break-even: DESIGN.md section 5). */
int  dvo_tiled_attach(dvo_ctx *ctx, void *nccl_comm, int rank, int world, const char *rccl_library);
int  dvo_tiled_detach(dvo_ctx *ctx);
/* flags: DVO_FLAG_FINAL_OUTPUTS and / or DVO_FLAG_NORMAL_MATRIX (H = sum w J J^T travels in the same 32 all-reduced doubles and is
 * kept per iterate: dvo_get_level_normal_matrix; without it the 21 slots are zeros).  One kernel + one ncclAllReduce per
 * iteration, the whole schedule captured once and replayed as a graph (dvo_tiled_graph_replayed).
 * With DVO_FLAG_FINAL_OUTPUTS every rank computes finalEpsilons / finalReprojections (SolveDVO.cpp:703-704,
 * :1002-1003) of ITS shard of the last level's list, at the points' own indices: dvo_get_final_outputs then returns arrays in
 * which only [first, first + count) of dvo_tiled_shard is filled in on this rank -- the caller concatenates the shards
 * (SURVEY.md 8e).  Thread safety: contexts of different GPUs may be driven from different host threads of one process
 * (each entry point makes the context's device current; the attachment registry is locked); one context is still
 * one-thread-at-a-time. */
int  dvo_align_pyramid_tiled(dvo_ctx *ctx, int pair, int n_levels, const int *iters, int flags, double *R, double *t);
/* the contiguous index range of `level`'s reference list this rank works on */
int  dvo_tiled_shard(dvo_ctx *ctx, int pair, int level, int *first, int *count);
/* inspection: *graph_replayed = 1 if the last dvo_align_pyramid_tiled replayed its captured graph (one kernel + one ncclAllReduce
 * per iteration, no host work in between), 0 if the schedule was submitted launch by launch (the runtime refused to capture the
 * collective, DVO_TILED_NO_GRAPH=1, or the legacy null stream) */
int  dvo_tiled_graph_replayed(dvo_ctx *ctx, int *graph_replayed);
/* inspection: bit l of *levels_mask = 1 if level l of the last dvo_align_pyramid_wide / _tiled schedule that was ENQUEUED (a replayed
 * graph keeps the mask of its capture) ran the packed two-points-per-lane step kernel over the compact list (round 5; lists built by
 * the engine's own reference-point kernels have one), 0 for the one-point-per-lane kernel over the 3 x N list (caller-supplied
 * lists, dvo_params.interpolate_dt);
 * *solo_mask (may be NULL): the levels among them that ran as ONE launch of one workgroup for all their iterations (levels of at most
 * DVO_TILED_SOLO_MAX = 6144 points; over several ranks every rank runs such a level whole, without a collective) */
int  dvo_wide_packed_levels(dvo_ctx *ctx, int *levels_mask, int *solo_mask);
/* round 6: *levels_mask = the levels of the last dvo_align_pyramid_wide that ran inside ONE launch of the fused kernel in team mode (the
 * coarse levels of a large frame: levels of at most DVO_WIDE_TEAM_MAX = 200 000 points from the coarsest down, when finer ones remain for
 * the step launches); 0 when the schedule ran as step launches only */
int  dvo_wide_team_levels(dvo_ctx *ctx, int *levels_mask);

/* ---- inspection (used by the parity tests) ---------------------------------
 * One evaluation of computeJacobianOfNowFrame + getReprojectedEpsilons at the
 * given pose (cast to float exactly as SolveDVO.cpp:673-674).  Host outputs, any
 * may be NULL: reproj 3xN, J Nx6 row-major, eps[N], w[N], visible[N]. */
int  dvo_eval_points(dvo_ctx *ctx, int pair, int level, const double *R, const double *t,
                     float *reproj, float *J, float *eps, float *w, int *visible);
/* The 29 accumulators of one iteration at the given pose:
 * acc[0..20] = upper triangle of sum_i w_i J_i J_i^T, acc[21..26] = g = J^T W eps
 * (SolveDVO.cpp:777), acc[27] = sum eps_i^2 (the correctly rounded exact sum: no order of additions enters), acc[28] = number of
 * visible points. */
int  dvo_accumulate(dvo_ctx *ctx, int pair, int level, const double *R, const double *t,
                    double *acc29);
/* Device SE(3) helpers exposed for property tests (same code the kernels use). */
int  dvo_device_se3_exp(dvo_ctx *ctx, const double *psi6, double *R, double *t);
int  dvo_device_se3_log(dvo_ctx *ctx, const double *R, const double *t, double *psi6);
int  dvo_device_rotationize(dvo_ctx *ctx, double *R);

/* Where the fused kernel read the now level of (pair, level) from in the last batch launch that used the packed kernel:
 * 0 = 16-byte texels gathered from HBM/L2, 1 = the level's texels staged once per level into LDS ("LDS-staged image
 * tiles": the reference re-copies the three images every iteration, SolveDVO.cpp:310,316-317,427) -- taken when the whole
 * level fits beside its point list; 2 = the level's compact form (below); -1 = not run.  Inspection / tests. */
int  dvo_get_level_texel_mode(dvo_ctx *ctx, int pair, int level, int *mode);
/* *ran = 1 if, at that level of that launch, a wave of the packed kernel took its literal-division fallback (a reference point
 * whose reprojected z left the range the fast reciprocal is proven exact on, or dvo_params.engine_variant = 3).  Tests. */
int  dvo_get_level_exact_fallback(dvo_ctx *ctx, int pair, int level, int *ran);
/* *n = the iterations of that level of that launch whose energy came from the exact sweep (round 6).  The energy is defined without an
 * order of summation: E = (float)sqrt(S), S = the correctly rounded double of the exact sum of eps^2 (SolveDVO.cpp:689, :1310-1312 is a
 * float norm whose order is Eigen's).  The packed kernel adds eps^2 in its own order and certifies that no order could have rounded to
 * another float; where it cannot (about N 2^-28 of the iterations of an N-point level) every wave sweeps the residuals once more into
 * exact 32-bit limbs.  dvo_params.engine_variant = 5 sends every iteration that way (tests).  The kernels whose sums travel between
 * launches or ranks (tiled mode) carry the limbs always: they report 0. */
int  dvo_get_level_energy_sweeps(dvo_ctx *ctx, int pair, int level, int *n);
/* *used = 1 if that level's reference points were read in their 4-byte form (engine detail: block-relative pixel + depth in
 * whole millimetres + chunk headers, validated bit for bit against the 8-byte list when the list is built; taken for lists
 * of at least three times what fits in LDS, where the per-iteration stream of the rest dominates the memory requests).  Tests. */
int  dvo_get_level_points4(dvo_ctx *ctx, int pair, int level, int *used);
/* 1 if the last fused launch read that level's ranks from an LDS copy of the whole level (coarse levels of large batches, round 5) */
int  dvo_get_level_ranks_in_lds(dvo_ctx *ctx, int pair, int level, int *used);

/* Shape the engine chose for the last fused (batch) launch: threads per workgroup (256: two workgroups per compute unit,
 * 512 / 1024: one), workgroups per frame pair (team mode, 1 = none), packed = 1: the two-points-per-lane kernel.  Inspection. */
int  dvo_get_last_launch_shape(dvo_ctx *ctx, int *block_threads, int *team_size, int *packed);

/* Compact form of resident now levels (engine detail, results are bit-identical with or without it).  The three images the
 * reference keeps per now level (dist transform :1768-1795, its imageGradient :1063-1098; the weight :1047-1053 is a function
 * of the first) are redundant: a pixel is described by the rank of its distance value among the image's distinct values and
 * the ranks of its four neighbours -- 4 bytes per pixel, 24 pixels per 128-byte memory line instead of 8: half the memory
 * requests of the alignment kernel.
 *   NATIVE (round 3): a now level produced by the engine's own distance transform (dvo_set_now_level_from_edges,
 *     dvo_frames_as_now, the now_first_pair argument of the upload calls) is written in this form and in no other: the integer
 *     squared distances are ranked through a presence bitmap (no hashing, sorting or verification pass), the 16-byte texels are
 *     never written (17 instead of 45 bytes of HBM traffic per pixel) and are decoded on demand for the entry points that read
 *     them (dvo_get_now_level, dvo_eval_points, the host-driven / tiled iteration).  Images the form cannot hold (more than
 *     8191 distinct distances, a pixel further than 511 pixels from every edge, a rank step beyond +-127) get 16-byte texels
 *     from the same launch instead.
 *   GENERIC: for caller-supplied float images (dvo_set_now_level) the engine derives the form from the 16-byte texels and
 *     VERIFIES per pixel that it reproduces {DT, gx, gy, w} bit for bit, keeping the 16-byte form for that pair and level
 *     otherwise (gradients that are not imageGradient(DT), more than 4095 distinct values, ...).  That build reads the level
 *     twice and costs about four and a half alignments of the same pair, so the engine makes it by itself only for a now level
 *     that has already been aligned DVO_COMPACT_NOW_AFTER times; dvo_now_prepare builds it now for the given pairs.
 * dvo_params.engine_variant = 4 (or DVO_COMPACT_NOW=off in the environment) disables both: 16-byte texels everywhere. */
#define DVO_COMPACT_NOW_AFTER 16
int  dvo_now_prepare(dvo_ctx *ctx, int first_pair, int count);
/* DIRECT (round 3): with on != 0, dvo_set_now_level / _pair / _device build the compact form from the three float images at
 * installation: the unit of a normalised exact distance transform is its smallest positive value s; d2 = round((DT / s)^2) per
 * pixel is accepted only if (float)sqrt(d2) * s reproduces DT bit for bit; the native builder's rank pass then writes the words,
 * and the caller's gx / gy are compared bit for bit with what the kernel will decode.  Any positive scale is accepted; a DT that
 * is no exact transform (-1) or gradients that are not imageGradient(DT) (-4) leave the pair on its 16-byte texels (written in
 * any case) and to the policy above.  Four single-image launches, about 40 us per 640x480 level -- against 98 us of PCIe for the
 * three images, or 4 us for the texels alone when they come from device memory: OFF by default, worth it only for a now level
 * that is aligned many times (DVO_DIRECT_COMPACT=on / off in the environment overrides the call). */
int  dvo_set_direct_compact(dvo_ctx *ctx, int on);
/* palette_size: > 0 number of palette entries of the compact form, 0 not built (yet / stale),
 * < 0 no compact form: -1 negative/inf/nan value, -2 too many distinct values (generic builder: 4095), -3 rank step beyond
 * +-127, -4 gradient is not imageGradient(DT), -5 weight is not getWeightOf(DT), -6 image narrower than 2 pixels, -7 a pixel
 * further than 511 pixels from every edge.  Round 5: the engine's own distance transform no longer refuses an image for -2 / -3 /
 * -7 -- it writes a PARTIAL compact form (the lowest 4094 ranks; the other pixels are looked up in the image's 16-byte texels,
 * which such an image also gets) and dvo_get_now_compact_partial says so; float images handed in directly keep those refusals. */
int  dvo_get_now_compact_info(dvo_ctx *ctx, int pair, int level, int *palette_size);
int  dvo_get_now_compact_partial(dvo_ctx *ctx, int pair, int level, int *partial);

/* Diagnostic builds only (make STAMPS=1): per-level phase cycle counters of `pair`,
 * out64[level*8 + {0: per-point loop, 1: reduction, 2: pose update, 3: barrier, 4: iterations}];
 * all zeros in the product library. */
int  dvo_debug_stamps(dvo_ctx *ctx, int pair, unsigned long long *out64);

/* ---- measurement support ------------------------------------------------------
 * Algorithmic (compulsory) bytes of one alignment of `pair` under the given
 * schedule: sum_l [12*rows*cols + 12*N_l] over levels with iters>0, + 16*N_last
 * when DVO_FLAG_FINAL_OUTPUTS. */
int  dvo_algorithmic_bytes(dvo_ctx *ctx, int pair, int n_levels, const int *iters, int flags,
                           uint64_t *bytes);
/* Sum over levels of iters[l]*N_l (point-iterations) for `pair`. */
int  dvo_point_iterations(dvo_ctx *ctx, int pair, int n_levels, const int *iters, uint64_t *count);

/* ---- frames in (SURVEY.md section 8f rows f1 + f2) ----------------------------------------------------
 * Everything between the camera / the RGBDFramePyd message and the hot path, on the GPU:
 *   row f2  camTopic2PublisherPyD.cpp:73-77,322-347   depth m -> mm u16 (0 -> 1), INTER_NEAREST pyramid, BGR2GRAY
 *   row f1  SolveDVO.cpp:1679-1799                     Canny(150,100,3,L2) -> distance transform -> normalise ->
 *                                                      gradients (now side); :1700-1712 + :1230-1264 + :224-264
 *                                                      Canny -> selectedPts -> enlistRefEdgePts (ref side)
 * A context holds a FRAME STORE of `n_slots` frames (grey, depth in mm, Canny edge map per level, all in
 * HBM).  A stored frame can be installed as the now frame and/or as the reference frame of any pair without
 * another upload -- what setRcvdFrameAsNowFrame / setRcvdFrameAsRefFrame / setPrevFrameAsRefFrame
 * (SolveDVO.cpp:535-618) do with host copies.  All slots share one pyramid geometry.
 * Frame-store calls are batched: `count` consecutive slots per call, one kernel launch per stage and level. */
enum { DVO_PIX_U8 = 0, DVO_PIX_U16 = 1, DVO_PIX_F32 = 2 };
enum { DVO_LAYOUT_COL_MAJOR = 0,    /* Eigen (im_n[level].data()): (yy,xx) at yy + xx*rows */
       DVO_LAYOUT_ROW_MAJOR = 1 };  /* cv::Mat / sensor_msgs::Image: (yy,xx) at yy*cols + xx */
enum { DVO_UPLOAD_ASYNC = 1,        /* do not wait for the copies: with DVO_UPLOAD_DIRECT the host buffers stay borrowed until dvo_synchronize() */
       DVO_UPLOAD_DEPTH_RAW = 2,    /* dvo_frames_upload_cameras: the depth images are already in sensor units (what a mono16 depth topic
                                       carries, as float): no x1000, no rounding, no 0 -> 1 -- what the rgbdSubsc node works on */
       DVO_UPLOAD_DIRECT = 4 };     /* DMA straight out of the caller's buffers.  Only for buffers that are pinned, or at least never
                                       unmapped while the context lives (a pool the caller keeps).  Default (round 3): the images are first
                                       copied into the engine's own pinned mirror (~0.1 ms per 640x480 frame) and the caller's memory is
                                       free again when the call returns.  Why: the HIP runtime registers pageable source buffers with the
                                       driver; when the application later frees them (free -> munmap of a large block), the driver stalls
                                       the process's GPU queues for 14-33 ms -- measured on the C++ file replay, whose loader allocates
                                       and frees a pyramid per frame: 24 ms per frame instead of 0.6 (profiles/r03_single_stream) */
enum { DVO_UPLOAD_MAPPED = 16 };    /* dvo_frames_upload_cameras / _pyramids: the images sit in PINNED host memory the GPU can address (hipHostMalloc,
                                       hipHostRegister'ed + mapped, torch pin_memory): a kernel pulls them over PCIe -- one launch per 32
                                       images at the full link rate (56 GB/s measured) instead of one DMA per image (38 GB/s: the
                                       per-copy submission cost) -- in the same double-buffered chunks as the DMA path, so the pull of
                                       chunk k+1 overlaps the preprocessing of chunk k.  Borrowing rules as for DVO_UPLOAD_DIRECT */
/* Pinned host memory the GPU can address, for callers that do not link the HIP runtime themselves (a ROS node's image pool):
 * what DVO_UPLOAD_MAPPED wants.  NULL when the allocation fails.  Free with dvo_host_free_mapped, never with free(). */
void *dvo_host_alloc_mapped(size_t bytes);
void  dvo_host_free_mapped(void *p);
enum { DVO_UPLOAD_DEVICE = 8 };     /* dvo_frames_upload_cameras: the image pointers are DEVICE pointers of this context's GPU (a decoder or
                                       a camera driver that lands frames in HBM): device-to-device copies into the landing buffer, no PCIe.
                                       The buffers stay borrowed until the call returns (until dvo_synchronize() with DVO_UPLOAD_ASYNC) */

typedef struct dvo_image {          /* one single-channel host image */
    const void *data;
    int rows, cols;
    int dtype;                      /* DVO_PIX_* : grey U8 or F32 (values 0..255), depth U16 (mm) or F32 (mm) */
    int layout;                     /* DVO_LAYOUT_* */
} dvo_image;

/* (re)size the frame store; default on first use: min(2*n_pairs + 2, 64) slots */
int  dvo_frames_reserve(dvo_ctx *ctx, int n_slots);
/* the pyramids of `count` frames as the dvo node receives them (RGBDFramePyd: framemono[] mono8 + dframe[] mono16,
 * imageArrivedCallBack SolveDVO.cpp:490-534) or as the class holds them (im_n/dim_n: F32 column-major).
 * grey[f*n_levels + l], depth[f*n_levels + l]; depth may be NULL (frames that will only ever be "now" frames).
 * U16 depth gets the node's 0 -> 1 treatment (:514); F32 depth is taken as is.  Runs Canny per level.
 * now_first_pair >= 0: slot first_slot+i is also installed as the now frame of pair now_first_pair+i (as
 * dvo_frames_as_now would), chunk by chunk in the shadow of the next chunk's host-to-device copies; -1: no. */
int  dvo_frames_upload_pyramids(dvo_ctx *ctx, int first_slot, int count, int n_levels,
                                const dvo_image *grey, const dvo_image *depth, int now_first_pair, int flags);
/* cv::undistort of the pyramid publisher (camTopic2PublisherPyD.cpp:88-107, :306-308), applied to BOTH images of every
 * frame given to dvo_frames_upload_cameras from now on (the depth image after its conversion to 16-bit millimetres, as the
 * publisher does): K4 = fx, fy, cx, cy and D5 = k1, k2, p1, p2, k3 of the sensor_msgs/CameraInfo the publisher listens to
 * (:52-61), rows x cols = the camera's resolution.  OpenCV 2.4 semantics: fixed-point bilinear remap (5 fraction bits),
 * zero outside the source.  K4 = D5 = NULL switches it off again (the publisher's behaviour without a camera-info topic).
 * The map is built once per call on the host. */
int  dvo_frames_set_undistort(dvo_ctx *ctx, int rows, int cols, const double *K4, const double *D5);
/* camera frames: full-resolution BGR8 (rows x cols x 3, row-major) + depth in metres (F32 row-major, may be NULL);
 * level l is decimated by 2^(first_shift + l) (the reference publishes first_shift = 1: 320x240 .. 40x30).
 * Builds the pyramid on the device, then as above. */
int  dvo_frames_upload_cameras(dvo_ctx *ctx, int first_slot, int count, const unsigned char *const *bgr8,
                               const float *const *depth_m, int rows, int cols, int n_levels, int first_shift,
                               int now_first_pair, int flags);
/* computeDistTransfrmOfNow (SolveDVO.cpp:1740-1799): slot first_slot+i becomes the now frame of pair first_pair+i.
 * Asynchronous on the context stream. */
int  dvo_frames_as_now(dvo_ctx *ctx, int first_slot, int first_pair, int count);
/* computeDistTransfrmOfRef's edge map + preProcessRefFrame (:269-303): slot first_slot+i becomes the reference
 * frame of pair first_pair+i.  N_out[i*n_levels + l] (may be NULL) receives the point counts.  Needs intrinsics
 * and depth.  One host synchronisation (the point counts size the slabs). */
int  dvo_frames_as_ref(dvo_ctx *ctx, int first_slot, int first_pair, int count, int *N_out);
/* inspection: geometry and resident images of one stored level (host outputs, column-major, any may be NULL) */
int  dvo_frame_get_level(dvo_ctx *ctx, int slot, int level, int *rows, int *cols, unsigned char *grey,
                         float *depth_mm, unsigned char *edge, int *n_edges);
int  dvo_frames_num_levels(const dvo_ctx *ctx);

/* ---- the legacy photometric Gauss-Newton odometry: the engine behind RGBDOdometry -----------
 * RGBDOdometry (include/RGBDOdometry.h:41-43, src/RGBDOdometry.cpp) aligns intensities instead of edge distances: per
 * reference frame a semi-dense Jacobian J (pixels with x-gradient >= 5) and A = J^T J per pyramid level (:363-508); per new
 * frame and level up to three iterations of eps_i = I_ref(i) - I_now(warp(i, T)), b = -J^T eps, A psi = b by
 * colPivHouseholderQr, T = T exp(psi)^-1 (:514-700), levels 3 then 2 (:162-163).  All in double, like the reference.
 *
 * The reference code has defects.  `fixed` = 0 (default) REPRODUCES its arithmetic as written, so that a user
 * of the rgbdSubsc node gets that node's numbers; `fixed` = 1 corrects: D1 tJ(0) = fx*fx/Z -> fx*gx/Z (:485); D2 tJ(5)'s
 * second term fx*gy*Y/Z -> fx*gx*Y/Z (:490); D4 level-0 intrinsics at every level -> scaled by 2^-level (:475-476);
 * D7 exponentialMap dropping the translation when |w| < 1e-12 (:727-731).  Kept in both modes: D3 the transposed image
 * convention (X from the row index with cx, fx, :475/:661/:683 -- self-consistent), D5 depth and translation in sensor units
 * (mm), D6 the selection rule gx >= 5, D8 the absolute stop threshold |eps| < 200 (:556).
 * Frames live in the frame store: upload them with dvo_frames_upload_cameras(..., n_levels = 4, first_shift = 0,
 * flags | DVO_UPLOAD_DEPTH_RAW) -- the node's own INTER_NEAREST pyramid of the full-resolution frame (:316-318, :347-350). */
typedef struct dvo_photo_params {
    double fx, fy, cx, cy;        /* cameraMatrix of params.xml, level 0                       RGBDOdometry.cpp:59-62 */
    int    gradient_threshold;    /* const_gradientThreshold = 5                               :32 */
    int    max_jacobian_size;     /* const_maxJacobianSize = 50000 (more selected pixels: error, the reference asserts :464) */
    int    min_required_pts;      /* const_minimumRequiredPts = 100 (fewer: error, :500)        :34 */
    int    iterations;            /* 3                                                          :545 */
    double eps_norm_stop;         /* 200.0                                                      :556 */
    int    fixed;                 /* 0 = the reference's arithmetic, 1 = defects D1 D2 D4 D7 corrected */
    int    reserved;
} dvo_photo_params;
int  dvo_photo_params_default(dvo_photo_params *p);
int  dvo_photo_configure(dvo_ctx *ctx, const dvo_photo_params *p);
/* setRefFrame + computeJacobianAllLevels (:296-327, :363-398) for levels first_level .. n_levels-1 of stored frame `slot`
 * (the reference: first_level = 1).  n_selected[n_levels] (may be NULL) receives the rows of J per level. */
int  dvo_photo_set_ref(dvo_ctx *ctx, int slot, int first_level, int *n_selected);
/* gaussNewtonIterations(level, T) (:514-597) for levels[0], levels[1], ... in that order (the reference: {3, 2}, :162-163) on
 * stored frame `now_slot`.  T16: TransformRep::matrix(), 4x4 row-major, in/out.  eps_norms[n_run * iterations]: |eps| of every
 * iteration (-1 where not run); updates[n_run]: iterations that changed T.  Both may be NULL. */
int  dvo_photo_align(dvo_ctx *ctx, int now_slot, const int *levels, int n_run, double *T16, double *eps_norms, int *updates);
/* inspection: J (n x 6 row-major), selected pixels (row, column) and A = J^T J (6x6) of a reference level */
int  dvo_photo_get_jacobian(dvo_ctx *ctx, int level, double *J, int *sel_i, int *sel_j, int capacity, double *A36, int *n_out);

#ifdef __cplusplus
}
#endif
#endif /* DVO_AMD_H_ */
