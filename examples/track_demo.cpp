/*
 * track_demo.cpp -- the reference's file-replay mode (SolveDVO::loop with __DATA_FROM_XML_FILES__,
 * src/SolveDVO.cpp:1950-2050) on the MI355X engine: reads <dir>/framemono_%04d.xml (OpenCV FileStorage XML with
 * mono_0..3 / depth_0..3, written by camTopic2PublisherPyD.cpp:306-383), tracks with the key-frame policy of the
 * reference and writes one "qx qy qz qw tx ty tz" line per frame (printPose, :1341-1354).
 *
 *   track_demo <dir> <start> <end> <skip> <n_levels> <fx> <fy> <cx> <cy> <iters_per_level> <poses.txt>
 *              [<laplacian_b_thresh> <visible_ratio_thresh> <min_points>]     the reference's adaptive key-frame exits (:2129-2152)
 */
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fstream>

#include "dvo_amd.hpp"

int main(int argc, char **argv) {
    if (argc != 12 && argc != 15) {
        std::fprintf(stderr, "usage: %s dir start end skip n_levels fx fy cx cy iters poses.txt [laplacian_b_thresh visible_ratio_thresh min_points]\n", argv[0]);
        return 2;
    }
    const char *dir = argv[1];
    const int start = std::atoi(argv[2]), end = std::atoi(argv[3]), skip = std::atoi(argv[4]), nl = std::atoi(argv[5]);
    const int iters = std::atoi(argv[10]);
    try {
        dvo_amd::SolveDVO dvo;
        dvo.setCameraMatrix((float)std::atof(argv[6]), (float)std::atof(argv[7]), (float)std::atof(argv[8]), (float)std::atof(argv[9]));
        dvo.iterationsConfig.assign(nl, iters);
        dvo.syncAfterNowFrame = std::getenv("TRACK_DEMO_SYNC") != nullptr;
        if (argc == 15) {
            dvo.adaptiveKeyFrames = true;
            dvo.laplacianThreshExitCond = (float)std::atof(argv[12]);
            dvo.ratio_of_visible_pts_thresh = (float)std::atof(argv[13]);
            dvo.minReprojectedPoints = std::atoi(argv[14]);
        }
        std::ofstream poses(argv[11]);
        char name[1024];
        double load_ms = 0, track_ms = 0;
        long tracked = 0;
        auto now = [] { return std::chrono::steady_clock::now(); };
        auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
            return std::chrono::duration<double, std::milli>(b - a).count(); };
        for (long n = 0;; n++) {
            const int idx = start + skip * (int)n;                       /* iDataFrameNum, :1953 / :2034 */
            if (idx > end) break;
            std::snprintf(name, sizeof(name), "%s/framemono_%04d.xml", dir, idx);
            const auto t0 = now();
            if (!dvo.loadFromFile(name, nl)) { std::fprintf(stderr, "No More files, Quitting.. (%s)\n", name); break; }
            const auto t1 = now();
            load_ms += ms(t0, t1);
            if (n == 0) {
                dvo.processFirstFrame();                                 /* no pose line for the first frame, like the reference */
                continue;
            }
            const dvo_amd::Pose p = dvo.processFrame();
            dvo_amd::SolveDVO::printPose(p, poses);
            const double dt = ms(t1, now());
            if (dvo.adaptiveKeyFrames && std::getenv("TRACK_DEMO_VERBOSE")) std::printf("frame %ld: b_cap %.9g visible %.9g points %d\n", n, dvo.lastLaplacianB, dvo.lastVisibleRatio, dvo.lastNumPoints);
            if (std::getenv("TRACK_DEMO_VERBOSE")) std::printf("frame %ld: %.3f ms (now-frame %.3f, first alignment %.3f)  t = %.4f %.4f %.4f\n", n, dt, dvo.lastNowFrameMs, dvo.lastAlignMs, p.px, p.py, p.pz);
            track_ms += dt;
            tracked++;
        }
        std::printf("frames %ld keyframes:", dvo.nFrame);
        for (int i = 0; i < dvo.gop.size(); i++) if (dvo.gop.isKeyFrameAt(i)) std::printf(" %d(reason %d)", dvo.gop.getFrameNumAt(i), dvo.gop.getReasonAt(i));
        std::printf("\n");
        if (tracked) std::printf("per frame: load+parse+upload+Canny %.3f ms, now-frame preprocessing + alignment(s) + pose %.3f ms\n",
                                 load_ms / (tracked + 1), track_ms / tracked);
    } catch (const std::exception &e) {
        std::fprintf(stderr, "track_demo: %s\n", e.what());
        return 1;
    }
    return 0;
}
