/* One synthetic frame pair through the C++ class mirror (dvo_amd::SolveDVO): prints, per level, the
 * energies, best index and visible ratio, then the final pose -- the numbers the reference's
 * casualTestFunction (SolveDVO.cpp:2377-2442) would print.  tests/test_cpp_host.py compares them with the oracle.
 *   usage: solve_dvo_demo W H levels iters seed */
#include <cstdio>
#include <cstdlib>

#include "dvo_amd.hpp"
#include "../rgbd_odometry_amd/csrc/dvo_synth.h"

int main(int argc, char **argv) {
    const int W = argc > 1 ? std::atoi(argv[1]) : 320, H = argc > 2 ? std::atoi(argv[2]) : 240;
    const int nl = argc > 3 ? std::atoi(argv[3]) : 4, iters = argc > 4 ? std::atoi(argv[4]) : 10;
    const unsigned long long seed = argc > 5 ? std::strtoull(argv[5], nullptr, 10) : 0;
    try {
        dvo_synth_scene *sc = dvo_synth_create(W, H, nl, seed);
        if (!sc) { std::fprintf(stderr, "bad scene arguments\n"); return 2; }
        dvo_amd::SolveDVO dvo;
        float k[4];
        dvo_synth_intrinsics(sc, k);
        dvo.setCameraMatrix(k[0], k[1], k[2], k[3]);
        std::vector<dvo_amd::ImageI> edge(nl);
        std::vector<dvo_amd::ImageF> depth(nl);
        dvo_amd::PyramidalStorageStruct now;
        for (int l = 0; l < nl; l++) {
            const int r = dvo_synth_rows(sc, l), c = dvo_synth_cols(sc, l);
            edge[l] = dvo_amd::ImageI(r, c); depth[l] = dvo_amd::ImageF(r, c);
            dvo_amd::ImageF dt(r, c), gx(r, c), gy(r, c);
            for (size_t i = 0; i < (size_t)r * c; i++) {
                edge[l].data[i] = dvo_synth_ref_edge(sc, l)[i];
                depth[l].data[i] = dvo_synth_ref_depth(sc, l)[i];
                dt.data[i] = dvo_synth_now_dt(sc, l)[i];
                gx.data[i] = dvo_synth_now_gx(sc, l)[i];
                gy.data[i] = dvo_synth_now_gy(sc, l)[i];
            }
            now.addLevel(l, dt, gx, gy);
        }
        dvo.setRefFrame(edge, depth);
        dvo.setNowFrame(now);
        dvo.iterationsConfig.assign(nl, iters);
        double cR[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, cT[3] = {0, 0, 0};
        /* the reference's loop: for f = size-1 .. 0: runIterations(f, ...) (SolveDVO.cpp:2097-2104) */
        for (int f = nl - 1; f >= 0; f--) {
            std::vector<float> energy, eps, reproj;
            int best; float ratio;
            dvo.runIterations(f, dvo.iterationsConfig[f], cR, cT, energy, eps, reproj, best, ratio);
            std::printf("level %d best %d ratio %.9g energies", f, best, ratio);
            for (float e : energy) std::printf(" %.9g", e);
            std::printf("\n");
        }
        std::printf("pose");
        for (double v : cR) std::printf(" %.17g", v);
        for (double v : cT) std::printf(" %.17g", v);
        std::printf("\n");
        /* the fused schedule must give the same pose */
        double fR[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, fT[3] = {0, 0, 0};
        dvo.alignPyramid(fR, fT);
        std::printf("fused");
        for (double v : fR) std::printf(" %.17g", v);
        for (double v : fT) std::printf(" %.17g", v);
        std::printf("\n");
        dvo_synth_destroy(sc);
    } catch (const std::exception &e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}
