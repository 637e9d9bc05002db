/*
 * ref_dump.cpp -- NOT built in this repository.  A driver for the maintainer's own build of mpkuse/rgbd_odometry: it feeds
 * SolveDVO::runIterations (include/SolveDVO.h:228-230, src/SolveDVO.cpp:619-1017) the inputs exported by export_inputs.py through the
 * members that function reads, runs the level schedule of SolveDVO::loop (src/SolveDVO.cpp:2097-2104) from the identity, and writes
 * what it returns as text (floats as C99 hex: bit-exact).  See README.md in this directory.
 *
 *   rosrun rgbd_odometry dvo_ref_dump <inputs dir> <outputs.txt>        (roscore running: the constructor subscribes, :41)
 */
#define private public          /* a test driver's liberty: runIterations and the per-level stores are private */
#define protected public
#include <SolveDVO.h>
#undef private
#undef protected

#include <cstdio>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

static bool read_f32(const std::string &path, std::vector<float> &v, size_t n) {
    v.resize(n);
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) return false;
    const size_t got = fread(v.data(), sizeof(float), n, f);
    fclose(f);
    return got == n;
}

int main(int argc, char **argv) {
    ros::init(argc, argv, "dvo_ref_dump");
    if (argc < 3) { fprintf(stderr, "usage: dvo_ref_dump <inputs dir> <outputs.txt>\n"); return 2; }
    const std::string in = argv[1];
    std::ifstream cases((in + "/cases.txt").c_str());
    FILE *out = fopen(argv[2], "w");
    if (!cases || !out) { fprintf(stderr, "cannot open %s/cases.txt or %s\n", in.c_str(), argv[2]); return 2; }
    std::string name;
    while (cases >> name) {
        const std::string dir = in + "/" + name;
        std::ifstream meta((dir + "/meta.txt").c_str());
        int n_levels = 0;
        float fx, fy, cx, cy;
        meta >> n_levels >> fx >> fy >> cx >> cy;
        std::vector<int> iters(n_levels), rows(n_levels), cols(n_levels), N(n_levels);
        for (int l = 0; l < n_levels; l++) meta >> iters[l] >> rows[l] >> cols[l] >> N[l];

        SolveDVO dvo;                                           /* one object per case: every per-level store starts empty */
        dvo.fx = fx; dvo.fy = fy; dvo.cx = cx; dvo.cy = cy;     /* what setCameraMatrix fills (:88-119) */
        dvo.K = Eigen::Matrix3f::Zero();
        dvo.K(0, 0) = fx; dvo.K(1, 1) = fy; dvo.K(0, 2) = cx; dvo.K(1, 2) = cy; dvo.K(2, 2) = 1.0f;
        dvo.K_inv = dvo.K.inverse();
        dvo.isCameraIntrinsicsAvailable = true;
        dvo._ref_edge_3d.clear(); dvo._ref_edge_2d.clear();
        dvo.now_distance_transform.clear(); dvo.now_DT_gradientX.clear(); dvo.now_DT_gradientY.clear();
        for (int l = 0; l < n_levels; l++) {
            std::ostringstream p;
            p << dir << "/level_" << l << "_";
            std::vector<float> xyz, uv, dt, gx, gy;
            const size_t px = (size_t)rows[l] * cols[l];
            if (!read_f32(p.str() + "xyz.f32", xyz, 3 * (size_t)N[l]) || !read_f32(p.str() + "uv.f32", uv, 2 * (size_t)N[l]) ||
                !read_f32(p.str() + "dt.f32", dt, px) || !read_f32(p.str() + "gx.f32", gx, px) || !read_f32(p.str() + "gy.f32", gy, px)) {
                fprintf(stderr, "short read in %s level %d\n", dir.c_str(), l);
                return 3;
            }
            /* all matrices column-major, as Eigen keeps them: 3 x N / 2 x N lists (SolveDVO.h:135-144), H x W images */
            dvo._ref_edge_3d.push_back(Eigen::Map<Eigen::MatrixXf>(xyz.data(), 3, N[l]));
            dvo._ref_edge_2d.push_back(Eigen::Map<Eigen::MatrixXf>(uv.data(), 2, N[l]));
            dvo.now_distance_transform.push_back(Eigen::Map<Eigen::MatrixXf>(dt.data(), rows[l], cols[l]));
            dvo.now_DT_gradientX.push_back(Eigen::Map<Eigen::MatrixXf>(gx.data(), rows[l], cols[l]));
            dvo.now_DT_gradientY.push_back(Eigen::Map<Eigen::MatrixXf>(gy.data(), rows[l], cols[l]));
        }
        dvo.isRefFrameAvailable = true; dvo.isNowFrameAvailable = true;     /* asserted at :627 */

        Eigen::Matrix3d cR = Eigen::Matrix3d::Identity();
        Eigen::Vector3d cT = Eigen::Vector3d::Zero();
        fprintf(out, "case %s %d\n", name.c_str(), n_levels);
        for (int l = n_levels - 1; l >= 0; l--) {              /* SolveDVO.cpp:2097-2104 */
            if (iters[l] <= 0) continue;
            Eigen::VectorXf energy, eps;
            Eigen::MatrixXf reproj;
            int best = -1;
            float ratio = 0.0f;
            dvo.runIterations(l, iters[l], cR, cT, energy, eps, reproj, best, ratio);
            fprintf(out, "level %d %d %d %a\n", l, (int)energy.rows(), best, (double)ratio);
            for (int i = 0; i < energy.rows(); i++) fprintf(out, "%a ", (double)energy[i]);
            fprintf(out, "\n");
            const int head = eps.rows() < 256 ? (int)eps.rows() : 256;
            fprintf(out, "final %d\n", head);
            for (int i = 0; i < head; i++) fprintf(out, "%a ", (double)eps[i]);
            fprintf(out, "\n");
            for (int i = 0; i < head; i++) fprintf(out, "%a %a %a ", (double)reproj(0, i), (double)reproj(1, i), (double)reproj(2, i));
            fprintf(out, "\n");
        }
        fprintf(out, "pose");
        for (int j = 0; j < 3; j++) for (int i = 0; i < 3; i++) fprintf(out, " %a", cR(i, j));     /* column-major */
        for (int i = 0; i < 3; i++) fprintf(out, " %a", cT(i));
        fprintf(out, "\n");
        fflush(out);
    }
    fclose(out);
    return 0;
}
