/*
 * dvo_amd.hpp -- C++ host mirror of the reference's class surface for the edge-alignment path,
 * header-only on top of the C ABI (dvo_amd.h).  ROS-, OpenCV- and Eigen-free, so it builds in
 * this image; the reference's node keeps its own ROS plumbing and swaps only the bodies shown in
 * INTEGRATION.md.
 *
 *   dvo_amd::SolveDVO               include/SolveDVO.h:148-360, src/SolveDVO.cpp
 *       setCameraMatrix             :88-126   (fx,fy,cx,cy of the level-0 calibration)
 *       setRefFrame                 setRcvdFrameAsRefFrame :537-557 + preProcessRefFrame :269-303
 *                                   (selectedPts + enlistRefEdgePts run on the GPU)
 *       setNowFrame                 setRcvdFrameAsNowFrame :588-614 (the DT / gradient images of
 *                                   computeDistTransfrmOfNow :1740-1799 are the inputs)
 *       runIterations               :619-1017, same argument order and meaning
 *       alignPyramid                the level loop of SolveDVO::loop :2097-2104 (+ :2220-2227)
 *       iterationsConfig            :30-33, SolveDVO.h:354
 *   dvo_amd::PyramidalStorageStruct include/PyramidalStorage.h:37-78 (addLevel/getLevel/clearPyramid/printSize):
 *                                   here the per-level container of the now-frame pyramid
 *   dvo_amd::RGBDOdometry           include/RGBDOdometry.h:41-43: the legacy photometric node; only the
 *                                   class surface is kept (its arithmetic is out of scope)
 *
 * Error behaviour: the reference asserts (NDEBUG is force-undefined, SolveDVO.h:124); these classes
 * throw std::runtime_error carrying dvo_last_error().
 */
#ifndef DVO_AMD_HPP_
#define DVO_AMD_HPP_

#include <cstdio>
#include <stdexcept>
#include <string>
#include <vector>

#include "dvo_amd.h"

namespace dvo_amd {

/* Column-major float image, the layout of Eigen::MatrixXf (element (yy,xx) at yy + xx*rows). */
struct ImageF {
    int rows = 0, cols = 0;
    std::vector<float> data;
    ImageF() = default;
    ImageF(int r, int c) : rows(r), cols(c), data((size_t)r * c, 0.0f) {}
    float &operator()(int yy, int xx) { return data[(size_t)yy + (size_t)xx * rows]; }
    float operator()(int yy, int xx) const { return data[(size_t)yy + (size_t)xx * rows]; }
};
struct ImageI {
    int rows = 0, cols = 0;
    std::vector<int32_t> data;
    ImageI() = default;
    ImageI(int r, int c) : rows(r), cols(c), data((size_t)r * c, 0) {}
};

/* Per-level container of the now-frame pyramid {DT, dDT/dx, dDT/dy}. */
class PyramidalStorageStruct {
public:
    void addLevel(int /*level: ignored like the reference, PyramidalStorage.cpp:37*/, const ImageF &dt,
                  const ImageF &gx, const ImageF &gy) {
        dt_.push_back(dt); gx_.push_back(gx); gy_.push_back(gy);
    }
    void getLevel(int level, ImageF &dt, ImageF &gx, ImageF &gy) const {
        dt = dt_.at(level); gx = gx_.at(level); gy = gy_.at(level);
    }
    void clearPyramid() { dt_.clear(); gx_.clear(); gy_.clear(); }
    void printSize() const { std::printf("PyramidalStorageStruct: %zu levels\n", dt_.size()); }
    size_t size() const { return dt_.size(); }
private:
    std::vector<ImageF> dt_, gx_, gy_;
};

class SolveDVO {
public:
    std::vector<int> iterationsConfig;      /* SolveDVO.cpp:30-33 */

    explicit SolveDVO(const dvo_params *params = nullptr) {
        iterationsConfig = {50, 50, 50, 50};
        if (dvo_create(params, &ctx_) != DVO_OK) throw std::runtime_error(std::string("dvo_create: ") + dvo_last_error(nullptr));
    }
    ~SolveDVO() { dvo_destroy(ctx_); }
    SolveDVO(const SolveDVO &) = delete;
    SolveDVO &operator=(const SolveDVO &) = delete;

    /* SolveDVO::setCameraMatrix(const char*) reads an OpenCV calibration XML (SolveDVO.cpp:88-126); the
     * numbers it keeps are these four (fx, fy, cx, cy of K, SolveDVO.h:178-179). */
    void setCameraMatrix(float fx, float fy, float cx, float cy) { chk(dvo_set_intrinsics(ctx_, fx, fy, cx, cy)); isCameraIntrinsicsAvailable = true; }
    /* minimal reader for the <cameraMatrix> ... <data> fx 0 cx 0 fy cy 0 0 1 </data> node of such a file */
    void setCameraMatrix(const char *calibFile) {
        std::FILE *f = std::fopen(calibFile, "r");
        if (!f) throw std::runtime_error(std::string("Cannot open calibration file ") + calibFile);   /* ROS_ERROR at :94-99 */
        std::string s; char buf[4096]; size_t n;
        while ((n = std::fread(buf, 1, sizeof(buf), f)) > 0) s.append(buf, n);
        std::fclose(f);
        size_t a = s.find("cameraMatrix");
        a = (a == std::string::npos) ? a : s.find("<data>", a);
        if (a == std::string::npos) throw std::runtime_error("cameraMatrix/data node not found");
        double k[9];
        if (std::sscanf(s.c_str() + a + 6, "%lf %lf %lf %lf %lf %lf %lf %lf %lf", k, k + 1, k + 2, k + 3, k + 4, k + 5, k + 6, k + 7, k + 8) != 9)
            throw std::runtime_error("cameraMatrix/data: expected 9 numbers");
        setCameraMatrix((float)k[0], (float)k[4], (float)k[2], (float)k[5]);
    }

    /* reference frame: per level an edge map (>0 = edge) and a depth image in mm; the 3-D edge point
     * lists _ref_edge_3d/_ref_edge_2d are built on the GPU (preProcessRefFrame :269-303). */
    void setRefFrame(const std::vector<ImageI> &edge, const std::vector<ImageF> &depth_mm) {
        if (edge.size() != depth_mm.size()) throw std::runtime_error("edge/depth pyramids differ in size");
        ref_points_.assign(edge.size(), {});
        for (size_t l = 0; l < edge.size(); l++) {
            const int cap = edge[l].rows * edge[l].cols;
            std::vector<float> xyz((size_t)3 * cap);
            int N = 0;
            chk(dvo_set_ref_level_from_images(ctx_, 0, (int)l, edge[l].data.data(), depth_mm[l].data.data(),
                                              edge[l].rows, edge[l].cols, xyz.data(), nullptr, cap, &N));
            xyz.resize((size_t)3 * N);
            ref_points_[l] = xyz;
        }
        isRefFrameAvailable = true;
    }
    /* or directly the 3 x N lists (SpaceCordList, SolveDVO.h:137,304) */
    void setRefEdgePoints(int level, const float *xyz_3xN, int N) {
        chk(dvo_set_ref_level(ctx_, level, xyz_3xN, N));
        if ((int)ref_points_.size() <= level) ref_points_.resize(level + 1);
        ref_points_[level].assign(xyz_3xN, xyz_3xN + (size_t)3 * N);
        isRefFrameAvailable = true;
    }
    /* now frame: now_distance_transform / now_DT_gradientX / now_DT_gradientY of every level */
    void setNowFrame(const PyramidalStorageStruct &pyr) {
        for (size_t l = 0; l < pyr.size(); l++) {
            ImageF dt, gx, gy;
            pyr.getLevel((int)l, dt, gx, gy);
            chk(dvo_set_now_level(ctx_, (int)l, dt.data.data(), gx.data.data(), gy.data.data(), dt.rows, dt.cols));
        }
        isNowFrameAvailable = true;
    }

    /* SolveDVO::runIterations (SolveDVO.h:228-230): cR 3x3 column-major, cT 3, both in/out. */
    void runIterations(int level, int maxIterations, double *cR, double *cT,
                       std::vector<float> &energyAtEachIteration, std::vector<float> &finalEpsilons,
                       std::vector<float> &finalReprojections, int &bestEnergyIndex, float &finalVisibleRatio) {
        if (!(isRefFrameAvailable && isNowFrameAvailable && isCameraIntrinsicsAvailable))          /* :627 */
            throw std::runtime_error("runIterations: reference frame, now frame and intrinsics must be set");
        const size_t N = ref_points_.at(level).size() / 3;
        energyAtEachIteration.assign(maxIterations, 0.0f);                                          /* :634 */
        finalEpsilons.assign(N, 0.0f);
        finalReprojections.assign(3 * N, 0.0f);
        chk(dvo_run_iterations(ctx_, level, maxIterations, cR, cT, energyAtEachIteration.data(),
                               finalEpsilons.data(), finalReprojections.data(), &bestEnergyIndex, &finalVisibleRatio));
    }

    /* the level loop of SolveDVO::loop (:2097-2104): for f = size-1 .. 0: if cfg[f] > 0: runIterations(f, ...) --
     * fused into one kernel launch. */
    void alignPyramid(double *cR, double *cT) {
        chk(dvo_align_pyramid(ctx_, (int)iterationsConfig.size(), iterationsConfig.data(), 0, cR, cT));
    }
    void levelReport(int level, std::vector<float> &energy, int &bestEnergyIndex, float &visibleRatio) {
        energy.assign(iterationsConfig.at(level), 0.0f);
        chk(dvo_get_level_report(ctx_, 0, level, energy.data(), (int)energy.size(), &bestEnergyIndex, &visibleRatio));
    }

    /* frame loop hooks of the reference; the ROS node keeps its own loop() */
    void loopDry() {}
    void loopFromFile() { throw std::runtime_error("loopFromFile: body is commented out in the reference too (SolveDVO.cpp:2444-2678)"); }

    dvo_ctx *handle() { return ctx_; }

private:
    void chk(int rc) { if (rc != DVO_OK) throw std::runtime_error(dvo_last_error(ctx_)); }
    dvo_ctx *ctx_ = nullptr;
    std::vector<std::vector<float>> ref_points_;
    bool isCameraIntrinsicsAvailable = false, isRefFrameAvailable = false, isNowFrameAvailable = false;
};

/* Legacy photometric Gauss-Newton node (rgbdSubsc).  Only the class surface is kept; see SURVEY.md 2.1. */
class RGBDOdometry {
public:
    RGBDOdometry() {}
    void eventLoop() {
        throw std::runtime_error("RGBDOdometry::eventLoop: the legacy photometric path is outside the MI355X engine's "
                                 "scope (SURVEY.md 2.1); use dvo_amd::SolveDVO");
    }
};

}  // namespace dvo_amd
#endif
