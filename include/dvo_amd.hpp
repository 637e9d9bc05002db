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
 *       setRcvdFrame / loadFromFile imageArrivedCallBack :490-534 / loadFromFile :154-190 (OpenCV-XML mono_%d, depth_%d)
 *       setRcvdFrameAsRefFrame, setPrevFrameAsRefFrame, setRcvdFrameAsNowFrame, preProcessRefFrame
 *                                   :535-618, :269-303 -- on the engine's frame store (Canny, distance transform,
 *                                   point extraction on the GPU; the previous now frame stays resident)
 *       processFirstFrame / processFrame   the body of SolveDVO::loop :1970-2241: level schedule, forced key frame
 *                                   every 5 frames with the n-1 re-reference + re-run (__NEW__REF_UPDATE), GOP push
 *       printPose                   :1341-1354 ("qx qy qz qw tx ty tz" lines, __WRITE_EST_POSE_TO_FILE)
 *   dvo_amd::GOP<T>                 include/GOP.h, src/GOP.cpp:138-196 (key-frame relative -> global pose chain)
 *   dvo_amd::PyramidalStorageStruct include/PyramidalStorage.h:37-78 (addLevel/getLevel/clearPyramid/printSize):
 *                                   here the per-level container of the now-frame pyramid
 *   dvo_amd::RGBDOdometry           include/RGBDOdometry.h:41-43: the legacy photometric Gauss-Newton node, on the engine's
 *                                   dvo_photo_* entry points (rows A14 / f4)
 *
 * Error behaviour: the reference asserts (NDEBUG is force-undefined, SolveDVO.h:124); these classes
 * throw std::runtime_error carrying dvo_last_error().
 */
#ifndef DVO_AMD_HPP_
#define DVO_AMD_HPP_

#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <ostream>
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

/* cv::Mat / Eigen stand-ins of the legacy photometric path (row-major 8/16-bit images like cv::Mat; column-major double
 * arrays like Eigen::ArrayXXd / MatrixXd) */
struct ImageU8 { int rows = 0, cols = 0, channels = 1; std::vector<unsigned char> data; };      /* CV_8UC1 / CV_8UC3 */
struct ImageU16 { int rows = 0, cols = 0; std::vector<unsigned short> data; };                  /* CV_16UC1 */
struct ArrayD { int rows = 0, cols = 0; std::vector<double> data; };                            /* Eigen::ArrayXXd / MatrixXd */

/* include/PyramidalStorage.h:37-78.  Two uses:
 *  - the reference's own 11-field per-level record of the photometric estimators (im_r_color, im_r, dim_r, X, Y, Z, J,
 *    grayVals, redVals, greenVals, blueVals): addLevel pushes deep copies and ignores its level argument
 *    (PyramidalStorage.cpp:37-65), getLevel copies them back out (:71-102), clearPyramid, printSize (:105-127);
 *  - the 3-image form {DT, dDT/dx, dDT/dy} SolveDVO::setNowFrame of this header takes (SolveDVO itself keeps
 *    std::vector<Eigen::MatrixXf> members; the container shape is kept for both). */
class PyramidalStorageStruct {
public:
    void addLevel(int /*level: ignored, PyramidalStorage.cpp:37*/, const ImageU8 &im_r_color, const ImageU8 &im_r, const ImageU16 &dim_r,
                  const ArrayD &X, const ArrayD &Y, const ArrayD &Z, const ArrayD &J,
                  const ArrayD &grayVals, const ArrayD &redVals, const ArrayD &greenVals, const ArrayD &blueVals) {
        im_r_color_.push_back(im_r_color); im_r_.push_back(im_r); dim_r_.push_back(dim_r);
        X_.push_back(X); Y_.push_back(Y); Z_.push_back(Z); J_.push_back(J);
        gray_.push_back(grayVals); red_.push_back(redVals); green_.push_back(greenVals); blue_.push_back(blueVals);
    }
    void getLevel(int level, ImageU8 &im_r_color, ImageU8 &im_r, ImageU16 &dim_r, ArrayD &X, ArrayD &Y, ArrayD &Z, ArrayD &J,
                  ArrayD &grayVals, ArrayD &redVals, ArrayD &greenVals, ArrayD &blueVals) const {
        im_r_color = im_r_color_.at(level); im_r = im_r_.at(level); dim_r = dim_r_.at(level);         /* deep copies, :85-99 */
        X = X_.at(level); Y = Y_.at(level); Z = Z_.at(level); J = J_.at(level);
        grayVals = gray_.at(level); redVals = red_.at(level); greenVals = green_.at(level); blueVals = blue_.at(level);
    }
    void addLevel(int /*level*/, const ImageF &dt, const ImageF &gx, const ImageF &gy) { dt_.push_back(dt); gx_.push_back(gx); gy_.push_back(gy); }
    void getLevel(int level, ImageF &dt, ImageF &gx, ImageF &gy) const { dt = dt_.at(level); gx = gx_.at(level); gy = gy_.at(level); }
    void clearPyramid() {
        dt_.clear(); gx_.clear(); gy_.clear();
        im_r_color_.clear(); im_r_.clear(); dim_r_.clear(); X_.clear(); Y_.clear(); Z_.clear(); J_.clear();
        gray_.clear(); red_.clear(); green_.clear(); blue_.clear();
    }
    void printSize() const {                            /* PyramidalStorage.cpp:125-127 prints the vector sizes */
        std::printf("PyramidalStorageStruct: %zu DT levels, %zu photometric levels\n", dt_.size(), im_r_.size());
    }
    size_t size() const { return dt_.size(); }
    size_t photometricLevels() const { return im_r_.size(); }
private:
    std::vector<ImageF> dt_, gx_, gy_;
    std::vector<ImageU8> im_r_color_, im_r_;
    std::vector<ImageU16> dim_r_;
    std::vector<ArrayD> X_, Y_, Z_, J_, gray_, red_, green_, blue_;
};

/* geometry_msgs::Pose stand-in */
struct Pose { double px = 0, py = 0, pz = 0, qx = 0, qy = 0, qz = 0, qw = 1; };

/* Eigen::Quaternion<T>(Matrix3) as GOPElement::matrixToPose uses it (src/GOP.cpp:103-115); R column-major */
template <typename T>
inline void quaternionFromMatrix(const T *R, T &qx, T &qy, T &qz, T &qw) {
    auto m = [&](int i, int j) { return R[i + 3 * j]; };
    T t = m(0, 0) + m(1, 1) + m(2, 2);
    if (t > T(0)) {
        t = std::sqrt(t + T(1.0));
        qw = T(0.5) * t;
        t = T(0.5) / t;
        qx = (m(2, 1) - m(1, 2)) * t; qy = (m(0, 2) - m(2, 0)) * t; qz = (m(1, 0) - m(0, 1)) * t;
    } else {
        int i = 0;
        if (m(1, 1) > m(0, 0)) i = 1;
        if (m(2, 2) > m(i, i)) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        t = std::sqrt(m(i, i) - m(j, j) - m(k, k) + T(1.0));
        T q[3];
        q[i] = T(0.5) * t;
        t = T(0.5) / t;
        qw = (m(k, j) - m(j, k)) * t;
        q[j] = (m(j, i) + m(i, j)) * t;
        q[k] = (m(k, i) + m(i, k)) * t;
        qx = q[0]; qy = q[1]; qz = q[2];
    }
}

/* Key-frame relative poses -> global pose chain (include/GOP.h, src/GOP.cpp:138-196). */
template <typename T>
class GOP {
public:
    struct Element { int frameNum; bool keyFrame; int reason; T R[9]; T t[3]; };
    GOP() { identity(lastKeyFr_R_); lastKeyFr_T_[0] = lastKeyFr_T_[1] = lastKeyFr_T_[2] = T(0); }
    void pushAsOrdinaryFrame(int frameNum, const T *cR, const T *cT) {            /* GOP.cpp:138-153 */
        Element e; compose(cR, cT, e.R, e.t);
        e.frameNum = frameNum; e.keyFrame = false; e.reason = -1;
        v_.push_back(e);
    }
    void pushAsKeyFrame(int frameNum, int reason, const T *cR, const T *cT) {      /* GOP.cpp:162-186 */
        Element e; compose(cR, cT, e.R, e.t);
        e.frameNum = frameNum; e.keyFrame = true; e.reason = reason;
        v_.push_back(e);
        for (int k = 0; k < 9; k++) lastKeyFr_R_[k] = e.R[k];
        for (int k = 0; k < 3; k++) lastKeyFr_T_[k] = e.t[k];
    }
    void updateMostRecentToKeyFrame(int reason) {                                  /* GOP.cpp:189-196 */
        Element &e = v_.at(v_.size() - 1);
        for (int k = 0; k < 9; k++) lastKeyFr_R_[k] = e.R[k];
        for (int k = 0; k < 3; k++) lastKeyFr_T_[k] = e.t[k];
        e.keyFrame = true; e.reason = reason;
    }
    int size() const { return (int)v_.size(); }
    const T *getGlobalRAt(int i) const { return v_.at(i).R; }
    const T *getGlobalTAt(int i) const { return v_.at(i).t; }
    bool isKeyFrameAt(int i) const { return v_.at(i).keyFrame; }
    int getReasonAt(int i) const { return v_.at(i).reason; }
    int getFrameNumAt(int i) const { return v_.at(i).frameNum; }
    Pose getGlobalPoseAt(int i) const {
        const Element &e = v_.at(i);
        Pose p; T qx, qy, qz, qw;
        quaternionFromMatrix<T>(e.R, qx, qy, qz, qw);
        p.px = e.t[0]; p.py = e.t[1]; p.pz = e.t[2]; p.qx = qx; p.qy = qy; p.qz = qz; p.qw = qw;
        return p;
    }
private:
    static void identity(T *R) { for (int k = 0; k < 9; k++) R[k] = (k % 4 == 0) ? T(1) : T(0); }
    void compose(const T *cR, const T *cT, T *gR, T *gT) const {   /* global_T = key_T + key_R*cT; global_R = key_R*cR */
        for (int i = 0; i < 3; i++) {
            gT[i] = lastKeyFr_T_[i] + (lastKeyFr_R_[i] * cT[0] + lastKeyFr_R_[i + 3] * cT[1] + lastKeyFr_R_[i + 6] * cT[2]);
            for (int j = 0; j < 3; j++)
                gR[i + 3 * j] = lastKeyFr_R_[i] * cR[3 * j] + lastKeyFr_R_[i + 3] * cR[3 * j + 1] + lastKeyFr_R_[i + 6] * cR[3 * j + 2];
        }
    }
    std::vector<Element> v_;
    T lastKeyFr_R_[9], lastKeyFr_T_[3];
};

/* One received frame: the pyramid of the RGBDFramePyd message (msg/RGBDFramePyd.msg: framemono[] mono8, dframe[]
 * mono16), row-major like cv::Mat / sensor_msgs::Image. */
struct RGBDFramePyd {
    struct Level { int rows = 0, cols = 0; std::vector<uint8_t> mono; std::vector<uint16_t> depth; };
    std::vector<Level> levels;
};

/* OpenCV FileStorage XML written by the publisher (camTopic2PublisherPyD.cpp:306-383): matrices mono_%d (dt u) and
 * depth_%d (dt w).  Minimal reader for that layout: <name type_id="opencv-matrix"><rows><cols><dt><data>. */
inline bool readOpenCvXmlMatrix(const std::string &xml, const std::string &name, int &rows, int &cols, std::string &dt,
                                std::vector<double> &values) {
    size_t a = xml.find("<" + name + " ");
    if (a == std::string::npos) a = xml.find("<" + name + ">");
    if (a == std::string::npos) return false;
    const size_t end = xml.find("</" + name + ">", a);
    auto field = [&](const char *tag, std::string &out) {
        const std::string open = std::string("<") + tag + ">", close = std::string("</") + tag + ">";
        const size_t b = xml.find(open, a);
        if (b == std::string::npos || b > end) return false;
        const size_t e = xml.find(close, b);
        if (e == std::string::npos) return false;
        out = xml.substr(b + open.size(), e - b - open.size());
        return true;
    };
    std::string r, c, d;
    if (!field("rows", r) || !field("cols", c) || !field("dt", dt) || !field("data", d)) return false;
    rows = std::atoi(r.c_str()); cols = std::atoi(c.c_str());
    while (!dt.empty() && (dt.back() == ' ' || dt.back() == '\n')) dt.pop_back();
    while (!dt.empty() && (dt.front() == ' ' || dt.front() == '\n')) dt.erase(dt.begin());
    values.clear();
    values.reserve((size_t)rows * cols);
    const char *p = d.c_str();
    char *q = nullptr;
    for (;;) {
        const double v = std::strtod(p, &q);
        if (q == p) break;
        values.push_back(v);
        p = q;
    }
    return (int)values.size() == rows * cols;
}

class SolveDVO {
public:
    std::vector<int> iterationsConfig;      /* SolveDVO.cpp:30-33 */
    GOP<double> gop;                        /* SolveDVO.h: GOP<double> gop */

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
    void alignPyramid(double *cR, double *cT, int flags = 0) {
        chk(dvo_align_pyramid(ctx_, (int)iterationsConfig.size(), iterationsConfig.data(), flags, cR, cT));
    }
    /* processResidueHistogram (:1398-1481) without its display: the MLE of the Laplacian scale of the residues = their mean,
     * accumulated in float in the list's order as the reference does (:1455-1462) */
    static float processResidueHistogram(const std::vector<float> &residi) {
        float b_cap = 0;
        for (size_t i = 0; i < residi.size(); i++) b_cap += residi[i];
        return residi.empty() ? 0.0f : b_cap / (float)residi.size();
    }
    void levelReport(int level, std::vector<float> &energy, int &bestEnergyIndex, float &visibleRatio) {
        energy.assign(iterationsConfig.at(level), 0.0f);
        chk(dvo_get_level_report(ctx_, 0, level, energy.data(), (int)energy.size(), &bestEnergyIndex, &visibleRatio));
    }

    /* ---- frames in: the engine's frame store instead of host copies + OpenCV -------------------------------- */
    /* imageArrivedCallBack (:490-534): hand the received pyramid to the engine (upload + Canny per level) */
    void setRcvdFrame(const RGBDFramePyd &f) {
        std::vector<dvo_image> g(f.levels.size()), d(f.levels.size());
        for (size_t l = 0; l < f.levels.size(); l++) {
            const RGBDFramePyd::Level &L = f.levels[l];
            g[l] = dvo_image{L.mono.data(), L.rows, L.cols, DVO_PIX_U8, DVO_LAYOUT_ROW_MAJOR};
            d[l] = dvo_image{L.depth.data(), L.rows, L.cols, DVO_PIX_U16, DVO_LAYOUT_ROW_MAJOR};
        }
        rcvd_slot_ = freeSlot();
        chk(dvo_frames_upload_pyramids(ctx_, rcvd_slot_, 1, (int)f.levels.size(), g.data(), d.data(), -1, 0));
        isFrameAvailable = true;
    }
    /* loadFromFile (:154-190): mono_0..3 / depth_0..3 of an OpenCV-XML frame file; false if it cannot be read */
    bool loadFromFile(const char *xmlFileName, int nLevels = 4) {
        std::FILE *fp = std::fopen(xmlFileName, "rb");
        if (!fp) return false;                                                     /* ROS_ERROR "Cannot Open File" :160 */
        /* The loader's buffers are members and keep their capacity from frame to frame.  Not a nicety: a host process that
         * frees multi-megabyte blocks between frames (free -> munmap) was measured to stall its own GPU queues for 14-33 ms per
         * frame on this pool (profiles/r03_single_stream: 24 ms instead of 0.6 ms per tracked frame, intermittently per
         * process; gone with glibc told never to return memory, MALLOC_MMAP_THRESHOLD_ / MALLOC_TRIM_THRESHOLD_). */
        std::string &s = loadText_; char buf[1 << 16]; size_t n;
        s.clear();
        while ((n = std::fread(buf, 1, sizeof(buf), fp)) > 0) s.append(buf, n);
        std::fclose(fp);
        RGBDFramePyd &f = loadFrame_;
        f.levels.resize(nLevels);
        std::vector<double> &v = loadV_, &w = loadW_;
        for (int i = 0; i < nLevels; i++) {
            int r = 0, c = 0, r2 = 0, c2 = 0; std::string dt;
            if (!readOpenCvXmlMatrix(s, "mono_" + std::to_string(i), r, c, dt, v)) return false;
            if (!readOpenCvXmlMatrix(s, "depth_" + std::to_string(i), r2, c2, dt, w) || r2 != r || c2 != c) return false;
            RGBDFramePyd::Level &L = f.levels[i];
            L.rows = r; L.cols = c;
            L.mono.assign(v.begin(), v.end());
            L.depth.assign(w.begin(), w.end());
        }
        setRcvdFrame(f);
        return true;
    }
    void setRcvdFrameAsRefFrame() {                 /* :535-556 (+ computeDistTransfrmOfRef: the Canny edge map is already resident) */
        need(isFrameAvailable, "setRcvdFrameAsRefFrame: no frame received");
        ref_slot_ = rcvd_slot_;
        isRefFrameAvailable = false; refPreprocessed_ = false;
    }
    void setPrevFrameAsRefFrame() {                 /* :559-583: the previous now frame is still in the store */
        need(prev_slot_ >= 0, "setPrevFrameAsRefFrame: n-1 frame not available");
        ref_slot_ = prev_slot_;
        isRefFrameAvailable = false; refPreprocessed_ = false;
    }
    void preProcessRefFrame() {                     /* :269-303: selectedPts + enlistRefEdgePts per level */
        need(ref_slot_ >= 0, "preProcessRefFrame: no reference frame");
        const int nl = dvo_frames_num_levels(ctx_);
        std::vector<int> N(nl);
        chk(dvo_frames_as_ref(ctx_, ref_slot_, 0, 1, N.data()));
        ref_points_.resize(nl);                      /* keep the capacity (see loadFromFile) */
        for (int l = 0; l < nl; l++) {
            ref_points_[l].resize((size_t)3 * N[l]);
            int n = 0;
            chk(dvo_get_ref_level(ctx_, 0, l, ref_points_[l].data(), N[l], &n));
        }
        isRefFrameAvailable = true; refPreprocessed_ = true;
    }
    void setRcvdFrameAsNowFrame() {                 /* :587-618 incl. computeDistTransfrmOfNow :1740-1799 */
        need(isFrameAvailable, "setRcvdFrameAsNowFrame: FRAME NOT AVAILABLE");
        if (isNowFrameAvailable && now_slot_ >= 0) prev_slot_ = now_slot_;         /* p_now_framemono / p_now_depth :594-600 */
        now_slot_ = rcvd_slot_;
        chk(dvo_frames_as_now(ctx_, now_slot_, 0, 1));
        isNowFrameAvailable = true;
    }

    /* ---- the body of SolveDVO::loop (:1970-2241), one call per received frame ------------------------------------ */
    /* first frame: reference frame + first key frame (:1972-2021) */
    void processFirstFrame() {
        setRcvdFrameAsRefFrame();
        preProcessRefFrame();
        if (warmUpOnFirstFrame) {
            /* one throw-away alignment of the first frame against itself: the first alignment of a process pays for loading the
             * kernels' code objects and for the engine's lazily allocated buffers (measured 12.5 ms against 0.6 ms for every later
             * frame, profiles/r03_single_stream) -- paid here, where the reference's loop has no pose to deliver yet (:1972-2021),
             * instead of on the first tracked frame.  Nothing of it survives: the now level is rewritten by the next frame. */
            chk(dvo_frames_as_now(ctx_, ref_slot_, 0, 1));
            double wR[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, wT[3] = {0, 0, 0};
            alignPyramid(wR, wT);
        }
        lastRefFrame = 0;
        identityPose();
        gop.pushAsKeyFrame((int)nFrame, 1, cR_64, cT_64);
        isFrameAvailable = false;
        nFrame++;
    }
    /* every other frame (:2059-2241 with __NEW__REF_UPDATE): returns the latest global pose (what publishGOP hands to printPose) */
    Pose processFrame() {
        const auto t0 = std::chrono::steady_clock::now();
        setRcvdFrameAsNowFrame();
        if (syncAfterNowFrame) chk(dvo_synchronize(ctx_));                          /* diagnostics: separates the two stages' times */
        const auto t1 = std::chrono::steady_clock::now();
        alignPyramid(cR_64, cT_64, adaptiveKeyFrames ? DVO_FLAG_FINAL_OUTPUTS : 0);  /* :2097-2104 (warm start from the last estimate) */
        const auto t2 = std::chrono::steady_clock::now();
        lastNowFrameMs = std::chrono::duration<double, std::milli>(t1 - t0).count();
        lastAlignMs = std::chrono::duration<double, std::milli>(t2 - t1).count();   /* jdur of :2092-2109 */
        bool signalGetNewRefImage = false;
        int reasonForChange = 0;
        if (adaptiveKeyFrames) {
            /* the three exits the reference has commented out (:2129-2152), evaluated on the last level that ran -- its best iterate's
             * visible ratio, finalEpsilons and point count, as the loop leaves them (:2102).  As written there the statement that
             * follows them (`signalGetNewRefImage = false`, :2154) would void all three; here they are OR-ed with the live rule. */
            int last = -1;
            for (int f = 0; f < (int)iterationsConfig.size() && last < 0; f++) if (iterationsConfig[f] > 0) last = f;
            need(last >= 0, "processFrame: no level has iterations");
            std::vector<float> energy; int best = -1; float visibleRatio = 1.0f;
            levelReport(last, energy, best, visibleRatio);
            int n = 0;
            epsilonVec_.assign(ref_points_.at(last).size() / 3, 0.0f);              /* one residue per reference point of that level */
            chk(dvo_get_final_outputs(ctx_, 0, epsilonVec_.data(), nullptr, (int)epsilonVec_.size(), &n));
            epsilonVec_.resize((size_t)n);
            lastLaplacianB = processResidueHistogram(epsilonVec_);                  /* b_cap :2116-2120 */
            lastVisibleRatio = visibleRatio;
            lastNumPoints = n;
            if (lastLaplacianB > laplacianThreshExitCond) { signalGetNewRefImage = true; reasonForChange = 2; }      /* :2131-2136 */
            if (visibleRatio < ratio_of_visible_pts_thresh) { signalGetNewRefImage = true; reasonForChange = 3; }    /* :2139-2144 */
            if (n < minReprojectedPoints) { signalGetNewRefImage = true; reasonForChange = 4; }                      /* :2146-2151 */
        }
        if ((nFrame - lastRefFrame) == keyFrameEvery) { signalGetNewRefImage = true; reasonForChange = 5; }   /* :2155-2160 */
        if (signalGetNewRefImage && lastRefFrame != nFrame - 1) {                  /* :2198 */
            lastRefFrame = nFrame - 1;
            setPrevFrameAsRefFrame();
            preProcessRefFrame();
            gop.updateMostRecentToKeyFrame(reasonForChange);                       /* :2207 */
            identityPose();                                                        /* :2210-2211 */
            alignPyramid(cR_64, cT_64);                                            /* re-run :2220-2227 */
            gop.pushAsOrdinaryFrame((int)nFrame, cR_64, cT_64);                    /* :2232 */
        } else {
            gop.pushAsOrdinaryFrame((int)nFrame, cR_64, cT_64);                    /* :2239 */
        }
        isFrameAvailable = false;
        nFrame++;
        return gop.getGlobalPoseAt(gop.size() - 1);                                /* MentisVisualHandle::publishGOP :283-300 */
    }
    /* printPose (:1341-1354): the line format of poses/estPoses.txt */
    static void printPose(const Pose &p, std::ostream &stream) {
        stream << p.qx << " " << p.qy << " " << p.qz << " " << p.qw << " " << p.px << " " << p.py << " " << p.pz << "\n";
        stream.flush();
    }
    double lastNowFrameMs = 0, lastAlignMs = 0;                                    /* iterationsComputeTime of :2107 and its preprocessing */
    double cR_64[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, cT_64[3] = {0, 0, 0};          /* key-frame relative estimate (column-major) */
    long nFrame = 0, lastRefFrame = 0;
    int keyFrameEvery = 5;                                                         /* (nFrame - lastRefFrame) == 5  :2156 */
    /* the reference's adaptive key-frame signals (:2129-2152, commented out there; constants :22-23): off by default */
    bool adaptiveKeyFrames = false;
    float ratio_of_visible_pts_thresh = 0.8f, laplacianThreshExitCond = 3.0f;
    int minReprojectedPoints = 50;
    float lastLaplacianB = 0.0f, lastVisibleRatio = 1.0f;                          /* what the last frame's first alignment produced */
    int lastNumPoints = 0;

    /* casualTestFunction (:2377-2442), the reference's only two-frame regression of the hot path: frame `refFile` as the
     * reference frame, `nowFile` as the now frame (TUM_RGBD/fr1_rpy/framemono_0080.xml and _0085.xml there, not shipped),
     * runIterations(0, 100, ...) from the identity (:2426); returns the 100 energies it prints (:2438-2441) */
    std::vector<float> casualTestFunction(const char *refFile, const char *nowFile, int level = 0, int iterations = 100,
                                          bool print = true) {
        const int nl = (int)iterationsConfig.size();        /* levels per frame file (4 in the reference) */
        need(loadFromFile(refFile, nl), "casualTestFunction: cannot read the reference frame file");
        setRcvdFrameAsRefFrame();
        preProcessRefFrame();
        need(loadFromFile(nowFile, nl), "casualTestFunction: cannot read the now frame file");
        setRcvdFrameAsNowFrame();
        identityPose();
        std::vector<float> energy, eps, reproj;
        int best = -1; float ratio = 0;
        runIterations(level, iterations, cR_64, cT_64, energy, eps, reproj, best, ratio);
        if (print) {
            for (size_t i = 0; i < energy.size(); i++) std::printf("%zu: %f\n", i, energy[i]);
            std::printf("best %d, visible ratio %f\n", best, ratio);
        }
        return energy;
    }

    /* frame loop hooks of the reference; the ROS node keeps its own loop() */
    void loopDry() {}
    void loopFromFile() { throw std::runtime_error("loopFromFile: body is commented out in the reference too (SolveDVO.cpp:2444-2678)"); }

    dvo_ctx *handle() { return ctx_; }

private:
    std::vector<float> epsilonVec_;                                                /* keeps its capacity from frame to frame */
    void chk(int rc) { if (rc != DVO_OK) throw std::runtime_error(dvo_last_error(ctx_)); }
    static void need(bool ok, const char *what) { if (!ok) throw std::runtime_error(what); }
public:
    bool warmUpOnFirstFrame = true;                 /* see processFirstFrame */
    bool syncAfterNowFrame = false;                 /* diagnostics (tools/track_latency.py): wait for the now-frame stage before timing the alignment */
private:
    void identityPose() { for (int k = 0; k < 9; k++) cR_64[k] = (k % 4 == 0) ? 1.0 : 0.0; cT_64[0] = cT_64[1] = cT_64[2] = 0.0; }
    int freeSlot() const {                          /* a store slot that holds neither the reference, the now nor the n-1 frame */
        for (int s = 0; s < 4; s++) if (s != ref_slot_ && s != now_slot_ && s != prev_slot_) return s;
        return 3;
    }
    dvo_ctx *ctx_ = nullptr;
    std::string loadText_;                          /* loadFromFile: reused from frame to frame */
    std::vector<double> loadV_, loadW_;
    RGBDFramePyd loadFrame_;
    std::vector<std::vector<float>> ref_points_;
    bool isCameraIntrinsicsAvailable = false, isRefFrameAvailable = false, isNowFrameAvailable = false;
    bool isFrameAvailable = false, refPreprocessed_ = false;
    int rcvd_slot_ = -1, ref_slot_ = -1, now_slot_ = -1, prev_slot_ = -1;
};

/* The legacy photometric Gauss-Newton node (rgbdSubsc): include/RGBDOdometry.h:41-43, src/RGBDOdometry.cpp.  Same method names
 * and per-frame sequence as the reference's eventLoop (:128-211), on the engine's dvo_photo_* entry points; the transport
 * (ROS topics in, odom / path / pose out) stays with the caller, who feeds frames and receives the pose eventLoop publishes.
 * The reference's arithmetic -- defects included -- is the default; RGBDOdometry(true) selects the corrected estimator
 * (the decision per defect: include/dvo_amd.h at dvo_photo_params). */
class RGBDOdometry {
public:
    typedef double TransformRep[16];                 /* Eigen::Transform<double,3,Affine>::matrix(), row-major (RGBDOdometry.h:33) */

    explicit RGBDOdometry(bool fixedDefects = false) : fixed_(fixedDefects) {
        if (dvo_create(nullptr, &ctx_) != DVO_OK) throw std::runtime_error(std::string("dvo_create: ") + dvo_last_error(nullptr));
        identity(T_); identity(base_);
    }
    ~RGBDOdometry() { dvo_destroy(ctx_); }
    RGBDOdometry(const RGBDOdometry &) = delete;
    RGBDOdometry &operator=(const RGBDOdometry &) = delete;

    /* setCameraMatrix (:42-68) keeps fx, fy, cx, cy of params.xml's cameraMatrix */
    void setCameraMatrix(double fx, double fy, double cx, double cy) {
        dvo_photo_params p;
        dvo_photo_params_default(&p);
        p.fx = fx; p.fy = fy; p.cx = cx; p.cy = cy; p.fixed = fixed_ ? 1 : 0;
        chk(dvo_photo_configure(ctx_, &p));
        cameraIntrinsicsReady = true;
    }
    /* imageArrivedCallBack: the received colour frame (bgr8, rows x cols x 3 row-major) and depth frame (sensor units) */
    void setRcvdFrame(const unsigned char *bgr8, const unsigned short *depth, int rows, int cols) {
        rcvd_bgr_.assign(bgr8, bgr8 + (size_t)rows * cols * 3);
        rcvd_depth_.assign(depth, depth + (size_t)rows * cols);                  /* -> float, taken as is (DVO_UPLOAD_DEPTH_RAW) */
        rows_ = rows; cols_ = cols;
        isFrameAvailable = true;
    }
    void setRefFrame() { upload(0); isRefFrameAvailable = isPyramidalRefFrameAvailable = true; isJacobiansAvailable = false; }   /* :296-327 */
    void setNowFrame() { upload(1); isNowFrameAvailable = isPyramidalNowFrameAvailable = true; }                                /* :329-357 */
    void computeJacobianAllLevels() {                                            /* :363-398: levels 1..3 */
        need(isPyramidalRefFrameAvailable && cameraIntrinsicsReady, "computeJacobianAllLevels: reference frame / intrinsics missing");
        chk(dvo_photo_set_ref(ctx_, 0, 1, nSelected));
        isJacobiansAvailable = true;
    }
    void gaussNewtonIterations(int level, TransformRep &T) {                     /* :514-597 */
        need(isPyramidalRefFrameAvailable && isPyramidalNowFrameAvailable && isJacobiansAvailable, "gaussNewtonIterations: frames / Jacobians missing");
        need(level != 0, "Critical error, jacobians at level-0 (base) are not computed for complexity reasons");   /* :518 */
        chk(dvo_photo_align(ctx_, 1, &level, 1, T, lastEpsNorms, &lastUpdates));
    }
    /* the body of eventLoop's while (:138-207) for one received frame; returns what it publishes: position = 1000 * translation
     * of base*T (:185-187), orientation = the quaternion of its rotation (:182, :188-191) */
    Pose processFrame() {
        need(isFrameAvailable, "processFrame: no frame received");
        if ((nFrame % refEvery) == 0) {                                          /* :146 ("renew ref-frame every 30 frames": % 10000) */
            mul(base_, T_, base_);                                               /* base = base * T */
            setRefFrame();
            identity(T_);
            computeJacobianAllLevels();
        }
        setNowFrame();
        gaussNewtonIterations(3, T_);                                            /* :162 */
        gaussNewtonIterations(2, T_);                                            /* :163 */
        double S[16];
        mul(base_, T_, S);                                                       /* toSend = base * T :178 */
        double Rc[9];
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Rc[i + 3 * j] = S[i * 4 + j];
        Pose p;
        quaternionFromMatrix<double>(Rc, p.qx, p.qy, p.qz, p.qw);
        p.px = 1000 * S[3]; p.py = 1000 * S[7]; p.pz = 1000 * S[11];
        nFrame++;
        isFrameAvailable = false;
        return p;
    }
    /* eventLoop (:128-211) without ROS: pull frames from `next` until it returns false, hand every pose to `publish` */
    template <typename NextFrame, typename Publish>
    void eventLoop(NextFrame next, Publish publish) {
        std::vector<unsigned char> bgr; std::vector<unsigned short> depth; int rows = 0, cols = 0;
        while (next(bgr, depth, rows, cols)) {
            setRcvdFrame(bgr.data(), depth.data(), rows, cols);
            publish(processFrame());
        }
    }
    const double *T() const { return T_; }
    long nFrame = 0;
    int refEvery = 10000;                            /* :146 */
    int nSelected[DVO_MAX_LEVELS] = {0};             /* rows of J per level */
    double lastEpsNorms[64] = {0};
    int lastUpdates = 0;
    dvo_ctx *handle() { return ctx_; }

private:
    void chk(int rc) { if (rc != DVO_OK) throw std::runtime_error(dvo_last_error(ctx_)); }
    static void need(bool ok, const char *what) { if (!ok) throw std::runtime_error(what); }
    static void identity(double *T) { for (int k = 0; k < 16; k++) T[k] = (k % 5 == 0) ? 1.0 : 0.0; }
    static void mul(const double *A, const double *B, double *C) {
        double t[16];
        for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) { double s = 0; for (int k = 0; k < 4; k++) s += A[i * 4 + k] * B[k * 4 + j]; t[i * 4 + j] = s; }
        for (int k = 0; k < 16; k++) C[k] = t[k];
    }
    void upload(int slot) {                          /* 4 levels, INTER_NEAREST at 1, 1/2, 1/4, 1/8 (:313-324, :346-355) */
        need(isFrameAvailable, "Frame not retrived");
        std::vector<float> d(rcvd_depth_.begin(), rcvd_depth_.end());
        const unsigned char *b = rcvd_bgr_.data();
        const float *dp = d.data();
        chk(dvo_frames_upload_cameras(ctx_, slot, 1, &b, &dp, rows_, cols_, 4, 0, -1, DVO_UPLOAD_DEPTH_RAW));
    }
    dvo_ctx *ctx_ = nullptr;
    bool fixed_ = false;
    double T_[16], base_[16];
    std::vector<unsigned char> rcvd_bgr_;
    std::vector<unsigned short> rcvd_depth_;
    int rows_ = 0, cols_ = 0;
    bool cameraIntrinsicsReady = false, isFrameAvailable = false, isRefFrameAvailable = false, isNowFrameAvailable = false;
    bool isPyramidalRefFrameAvailable = false, isPyramidalNowFrameAvailable = false, isJacobiansAvailable = false;
};

}  // namespace dvo_amd
#endif
