/*
 * dvo_oracle_batch.cpp -- TEST INFRASTRUCTURE, like the rest of oracle/: the CPU restatement of the reference path over a BATCH of
 * independent frame pairs, one pair per OpenMP thread at a time.  This is BASELINE.md section 4 (ii) / SURVEY.md section 8(d) "CPU
 * baseline (ii)": "one alignment per core over all cores for the batch config (OpenMP over pairs)" -- every thread runs the
 * single-threaded reference algorithm (the reference itself is single-threaded: EIGEN_DONT_PARALLELIZE, include/SolveDVO.h:14) on
 * its own pair, through the level schedule of SolveDVO::loop (src/SolveDVO.cpp:2097-2104).  Only bench.py's cpu_baseline legs and
 * tests/ call it; no product path does.
 */
#include "dvo_oracle.h"

#include <omp.h>
#include <vector>

extern "C" {

/* n_pairs alignments from the identity; pair i aligns scene i % n_scenes.  Per-level inputs of scene s, level l at index
 * s * n_levels + l.  R_out (n_pairs x 9, column-major) / t_out (n_pairs x 3) may be NULL.  thread_seconds / thread_pairs
 * (n_threads entries each, may be NULL): time spent aligning and pairs aligned by each thread.  Returns the threads used;
 * *seconds_out = wall time of the parallel region. */
int dvo_oracle_align_batch_omp(const dvo_oracle_params *prm, int n_pairs, int n_scenes, int n_levels, const int *iters,
                               const float *const *xyz, const int *N, const float *const *dt, const float *const *gx,
                               const float *const *gy, const int *rows, const int *cols, float fx, float fy, float cx, float cy,
                               int n_threads, double *R_out, double *t_out, double *seconds_out, double *thread_seconds,
                               int *thread_pairs) {
    if (n_threads < 1) n_threads = omp_get_max_threads();
    int sum_it = 0;
    for (int l = 0; l < n_levels; l++) sum_it += iters[l] > 0 ? iters[l] : 0;
    int used = 0;
    const double t0 = omp_get_wtime();
#pragma omp parallel num_threads(n_threads)
    {
        const int tid = omp_get_thread_num();
#pragma omp single
        used = omp_get_num_threads();
        std::vector<float> energy((size_t)(sum_it > 0 ? sum_it : 1));
        std::vector<int> best((size_t)n_levels);
        std::vector<float> ratio((size_t)n_levels);
        double mine = 0.0;
        int done = 0;
#pragma omp for schedule(dynamic, 1)
        for (int i = 0; i < n_pairs; i++) {
            const int s = i % n_scenes;
            double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, t[3] = {0, 0, 0};
            const double a = omp_get_wtime();
            dvo_oracle_align_pyramid(prm, n_levels, iters, xyz + (size_t)s * n_levels, N + (size_t)s * n_levels, dt + (size_t)s * n_levels,
                                     gx + (size_t)s * n_levels, gy + (size_t)s * n_levels, rows + (size_t)s * n_levels,
                                     cols + (size_t)s * n_levels, fx, fy, cx, cy, R, t, energy.data(), best.data(), ratio.data(),
                                     nullptr, nullptr);
            mine += omp_get_wtime() - a;
            done++;
            if (R_out) for (int k = 0; k < 9; k++) R_out[(size_t)i * 9 + k] = R[k];
            if (t_out) for (int k = 0; k < 3; k++) t_out[(size_t)i * 3 + k] = t[k];
        }
        if (thread_seconds && tid < n_threads) thread_seconds[tid] = mine;
        if (thread_pairs && tid < n_threads) thread_pairs[tid] = done;
    }
    if (seconds_out) *seconds_out = omp_get_wtime() - t0;
    return used;
}

}  /* extern "C" */
