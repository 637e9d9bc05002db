/*
 * rgbd_odometry_demo.cpp -- the C++ mirrors of the reference's two remaining entry points on the MI355X engine:
 *
 *   rgbd_odometry_demo photo <dir> <n> <rows> <cols> <fx> <fy> <cx> <cy> <fixed>
 *       dvo_amd::RGBDOdometry (rgbdSubscriber.cpp:34-35 -> RGBDOdometry::eventLoop, src/RGBDOdometry.cpp:128-211) on raw frame
 *       files <dir>/bgr_%04d.bin (rows*cols*3 bytes) and <dir>/depth_%04d.bin (rows*cols uint16, sensor units); prints per
 *       frame the pose eventLoop publishes and the 16 entries of T.
 *   rgbd_odometry_demo casual <ref.xml> <now.xml> <n_levels> <fx> <fy> <cx> <cy> <level> <iterations>
 *       dvo_amd::SolveDVO::casualTestFunction (src/SolveDVO.cpp:2377-2442): two OpenCV-XML frame files, runIterations(level,
 *       iterations) from the identity, the energies printed one per line.
 */
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "dvo_amd.hpp"

static bool read_file(const std::string &path, void *dst, size_t bytes) {
    std::FILE *f = std::fopen(path.c_str(), "rb");
    if (!f) return false;
    const size_t n = std::fread(dst, 1, bytes, f);
    std::fclose(f);
    return n == bytes;
}

int main(int argc, char **argv) {
    try {
        if (argc == 11 && !std::strcmp(argv[1], "photo")) {
            const std::string dir = argv[2];
            const int n = std::atoi(argv[3]), rows = std::atoi(argv[4]), cols = std::atoi(argv[5]);
            dvo_amd::RGBDOdometry odo(std::atoi(argv[10]) != 0);
            odo.setCameraMatrix(std::atof(argv[6]), std::atof(argv[7]), std::atof(argv[8]), std::atof(argv[9]));
            int idx = 0;
            char name[64];
            auto next = [&](std::vector<unsigned char> &bgr, std::vector<unsigned short> &depth, int &r, int &c) {
                if (idx >= n) return false;
                bgr.resize((size_t)rows * cols * 3); depth.resize((size_t)rows * cols);
                std::snprintf(name, sizeof(name), "/bgr_%04d.bin", idx);
                if (!read_file(dir + name, bgr.data(), bgr.size())) return false;
                std::snprintf(name, sizeof(name), "/depth_%04d.bin", idx);
                if (!read_file(dir + name, depth.data(), depth.size() * 2)) return false;
                r = rows; c = cols; idx++;
                return true;
            };
            auto publish = [&](const dvo_amd::Pose &p) {
                std::printf("pose %.17g %.17g %.17g %.17g %.17g %.17g %.17g\nT", p.px, p.py, p.pz, p.qx, p.qy, p.qz, p.qw);
                for (int k = 0; k < 16; k++) std::printf(" %.17g", odo.T()[k]);
                std::printf("\n");
            };
            odo.eventLoop(next, publish);
            std::printf("frames %ld, J rows per level: %d %d %d\n", odo.nFrame, odo.nSelected[1], odo.nSelected[2], odo.nSelected[3]);
            return 0;
        }
        if (argc == 11 && !std::strcmp(argv[1], "casual")) {
            dvo_amd::SolveDVO dvo;
            const int nl = std::atoi(argv[4]);
            dvo.setCameraMatrix((float)std::atof(argv[5]), (float)std::atof(argv[6]), (float)std::atof(argv[7]), (float)std::atof(argv[8]));
            dvo.iterationsConfig.assign(nl, 1);
            const std::vector<float> e = dvo.casualTestFunction(argv[2], argv[3], std::atoi(argv[9]), std::atoi(argv[10]), false);
            for (size_t i = 0; i < e.size(); i++) std::printf("%.9g\n", e[i]);
            return 0;
        }
        std::fprintf(stderr, "usage: %s photo dir n rows cols fx fy cx cy fixed | casual ref.xml now.xml n_levels fx fy cx cy level iterations\n", argv[0]);
        return 2;
    } catch (const std::exception &e) {
        std::fprintf(stderr, "rgbd_odometry_demo: %s\n", e.what());
        return 1;
    }
}
