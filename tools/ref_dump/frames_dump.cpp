/*
 * frames_dump.cpp -- NOT built in this repository.  The second driver of tools/ref_dump (round 6): the boundary BEFORE the hot path,
 * row f1 of SURVEY.md section 8 -- grey level -> edge map, normalised exact distance transform, its gradients -- through the
 * maintainer's own build of mpkuse/rgbd_odometry, i.e. through the OpenCV 2.4 calls this repository could only restate
 * (cv::Canny(150, 100, 3, true), cv::distanceTransform(CV_DIST_L2, CV_DIST_MASK_PRECISE), cv::normalize(NORM_MINMAX), imageGradient's
 * cv::filter2D: src/SolveDVO.cpp:1740-1799, :1063-1098).  It fills SolveDVO::im_n (the per-level grey images the node keeps,
 * include/SolveDVO.h:272-282) with the levels exported by export_frames.py, calls SolveDVO::computeDistTransfrmOfNow() and writes what
 * that leaves in now_edge_map / now_distance_transform / now_DT_gradientX / now_DT_gradientY as text (floats as C99 hex: bit-exact).
 *
 *   rosrun rgbd_odometry dvo_frames_dump <inputs dir> <outputs.txt>        (roscore running: the constructor subscribes, :41)
 */
#define private public          /* a test driver's liberty: the per-level stores and computeDistTransfrmOfNow are private */
#define protected public
#include <SolveDVO.h>
#undef private
#undef protected

#include <cstdio>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

int main(int argc, char **argv) {
    ros::init(argc, argv, "dvo_frames_dump");
    if (argc < 3) { fprintf(stderr, "usage: dvo_frames_dump <inputs dir> <outputs.txt>\n"); return 2; }
    const std::string in = argv[1];
    std::ifstream cases((in + "/frames.txt").c_str());
    FILE *out = fopen(argv[2], "w");
    if (!cases || !out) { fprintf(stderr, "cannot open %s/frames.txt or %s\n", in.c_str(), argv[2]); return 2; }
    std::string name;
    while (cases >> name) {
        const std::string dir = in + "/" + name;
        std::ifstream meta((dir + "/meta.txt").c_str());
        int n_levels = 0;
        meta >> n_levels;
        std::vector<int> rows(n_levels), cols(n_levels);
        for (int l = 0; l < n_levels; l++) meta >> rows[l] >> cols[l];
        SolveDVO dvo;
        dvo.im_n.clear();
        for (int l = 0; l < n_levels; l++) {
            std::ostringstream p;
            p << dir << "/grey_" << l << ".u8";             /* column-major like every Eigen image of the node: (yy, xx) at yy + xx * rows */
            std::vector<unsigned char> g((size_t)rows[l] * cols[l]);
            FILE *f = fopen(p.str().c_str(), "rb");
            if (!f || fread(g.data(), 1, g.size(), f) != g.size()) { fprintf(stderr, "short read: %s\n", p.str().c_str()); return 3; }
            fclose(f);
            Eigen::MatrixXf im(rows[l], cols[l]);
            for (int xx = 0; xx < cols[l]; xx++) for (int yy = 0; yy < rows[l]; yy++) im(yy, xx) = (float)g[(size_t)xx * rows[l] + yy];
            dvo.im_n.push_back(im);                         /* what imageArrivedCallBack leaves there (:508-519): the mono8 level as floats */
        }
        dvo.isNowFrameAvailable = true;                     /* asserted at :1742 */
        dvo.computeDistTransfrmOfNow();                     /* :1740-1799 */
        fprintf(out, "frame %s %d\n", name.c_str(), n_levels);
        for (int l = 0; l < n_levels; l++) {
            const Eigen::MatrixXi &e = dvo.now_edge_map[l];
            const Eigen::MatrixXf &dt = dvo.now_distance_transform[l], &gx = dvo.now_DT_gradientX[l], &gy = dvo.now_DT_gradientY[l];
            fprintf(out, "level %d %d %d\n", l, (int)dt.rows(), (int)dt.cols());
            fprintf(out, "edge");                            /* column-major, run-length: value count value count ... */
            {
                const int n = (int)(e.rows() * e.cols());
                int i = 0;
                while (i < n) { int j = i; while (j < n && e.data()[j] == e.data()[i]) j++; fprintf(out, " %d %d", e.data()[i], j - i); i = j; }
            }
            fprintf(out, "\n");
            const Eigen::MatrixXf *img[3] = {&dt, &gx, &gy};
            const char *tag[3] = {"dt", "gx", "gy"};
            for (int k = 0; k < 3; k++) {
                fprintf(out, "%s", tag[k]);
                for (int i = 0; i < (int)(img[k]->rows() * img[k]->cols()); i++) fprintf(out, " %a", (double)img[k]->data()[i]);
                fprintf(out, "\n");
            }
        }
        fflush(out);
    }
    fclose(out);
    return 0;
}
