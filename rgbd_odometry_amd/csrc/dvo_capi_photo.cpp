/*
 * dvo_capi_photo.cpp -- C ABI of the photometric Gauss-Newton odometry: what the reference's
 * RGBDOdometry does between two frames (src/RGBDOdometry.cpp:146-163):
 *
 *     setRefFrame + computeJacobianAllLevels  (:296-327, :363-398)   dvo_photo_set_ref(ctx, slot)
 *     setNowFrame + gaussNewtonIterations(3,T); gaussNewtonIterations(2,T)  (:329-357, :162-163)   dvo_photo_align(ctx, slot, ...)
 *
 * on frames of the context's frame store (dvo_frames_upload_cameras with first_shift = 0: the node's own 4-level
 * INTER_NEAREST pyramid of the full-resolution frame, :316-318, and DVO_UPLOAD_DEPTH_RAW so that the depth stays in
 * sensor units as the node keeps it).
 */
#include "dvo_ctx.h"

using namespace dvo;
using namespace dvo_host;

struct dvo_photo_state {
    dvo_photo_params prm;
    struct Lvl {
        double *J = nullptr, *zref = nullptr, *A = nullptr;
        int *sel = nullptr, *n_dev = nullptr;
        float *gref = nullptr;
        int n = 0, rows = 0, cols = 0;
        bool ready = false;
    } lv[DVO_LEVELS];
    int *col_work = nullptr;
    size_t col_work_ints = 0;
    double *d_T = nullptr, *d_norms = nullptr, *d_eps = nullptr;     /* T16 | per-call norms | eps dump */
    int *d_updates = nullptr;
    int ref_slot = -1;
};

namespace dvo_host {
void photo_forget(dvo_ctx *c) {
    dvo_photo_state *p = c->photo;
    if (!p) return;
    for (auto &L : p->lv) {
        void *ptrs[] = {L.J, L.zref, L.A, L.sel, L.n_dev, L.gref};
        for (void *q : ptrs) if (q) (void)hipFree(q);
    }
    void *ptrs[] = {p->col_work, p->d_T, p->d_norms, p->d_eps, p->d_updates};
    for (void *q : ptrs) if (q) (void)hipFree(q);
    delete p;
    c->photo = nullptr;
}
}  // namespace dvo_host

namespace {
int photo_state(dvo_ctx *c, dvo_photo_state **out) {
    if (!c->photo) {
        c->photo = new dvo_photo_state();
        dvo_photo_params_default(&c->photo->prm);
    }
    *out = c->photo;
    return DVO_OK;
}
}  // namespace

extern "C" {

int dvo_photo_params_default(dvo_photo_params *p) {
    if (!p) return DVO_ERR_INVALID;
    std::memset(p, 0, sizeof(*p));
    p->gradient_threshold = 5;          /* const_gradientThreshold   RGBDOdometry.cpp:32 */
    p->max_jacobian_size = 50000;       /* const_maxJacobianSize     :33 */
    p->min_required_pts = 100;          /* const_minimumRequiredPts  :34 */
    p->iterations = 3;                  /* :545 */
    p->eps_norm_stop = 200.0;           /* :556 */
    p->fixed = 0;
    return DVO_OK;
}

int dvo_photo_configure(dvo_ctx *c, const dvo_photo_params *prm) {
    DVO_ENTER(c);
    if (!prm || !(prm->fx != 0.0) || !(prm->fy != 0.0) || prm->max_jacobian_size < 1 || prm->iterations < 1 || prm->iterations > 64)
        return fail(c, DVO_ERR_INVALID, "bad photometric parameters (setCameraMatrix: fx, fy, cx, cy of the level-0 camera matrix)");
    dvo_photo_state *p;
    photo_state(c, &p);
    HIPCHK(c, stream_wait(c->stream));
    if (prm->max_jacobian_size != p->prm.max_jacobian_size)       /* capacity changed: drop the per-level buffers */
        for (auto &L : p->lv) {
            void *ptrs[] = {L.J, L.zref, L.sel, L.gref};
            for (void *q : ptrs) if (q) (void)hipFree(q);
            L.J = L.zref = nullptr; L.sel = nullptr; L.gref = nullptr;
        }
    p->prm = *prm;
    for (auto &L : p->lv) L.ready = false;
    p->ref_slot = -1;
    return DVO_OK;
}

/* setRefFrame (:296-327) + computeJacobianAllLevels (:363-398): J, the selected pixels and A = J^T J of levels
 * first_level .. n_levels-1 of the stored frame (the reference computes levels 1..3, :373) */
int dvo_photo_set_ref(dvo_ctx *c, int slot, int first_level, int *n_selected /* [n_levels] or NULL */) {
    DVO_ENTER(c);
    dvo_photo_state *p;
    photo_state(c, &p);
    if (!(p->prm.fx != 0.0)) return fail(c, DVO_ERR_STATE, "camera matrix not set (dvo_photo_configure)");
    const int nl = c->fs.n_levels;
    if (nl < 1) return fail(c, DVO_ERR_STATE, "frame store is empty (dvo_frames_upload_cameras)");
    if (slot < 0 || slot >= c->fs.n_slots || !c->fs.valid[slot]) return fail(c, DVO_ERR_INVALID, "no frame in this slot");
    if (!c->fs.has_depth[slot]) return fail(c, DVO_ERR_STATE, "the reference frame needs depth");
    if (first_level < 0 || first_level >= nl) return fail(c, DVO_ERR_INVALID, "first_level out of range");
    const int cap = p->prm.max_jacobian_size;
    if (!p->d_T) {
        HIPCHK(c, hipMalloc((void **)&p->d_T, sizeof(double) * 16));
        HIPCHK(c, hipMalloc((void **)&p->d_norms, sizeof(double) * 64));
        HIPCHK(c, hipMalloc((void **)&p->d_updates, sizeof(int)));
    }
    for (int l = 0; l < DVO_LEVELS; l++) p->lv[l].ready = false;
    for (int l = first_level; l < nl; l++) {
        FrameLevel &F = c->fs.lv[l];
        dvo_photo_state::Lvl &L = p->lv[l];
        if (F.rows > 65535 || F.cols > 32767) return fail(c, DVO_ERR_INVALID, "image too large for the photometric engine");
        if (!L.J) {
            HIPCHK(c, hipMalloc((void **)&L.J, sizeof(double) * 6 * (size_t)cap));
            HIPCHK(c, hipMalloc((void **)&L.zref, sizeof(double) * (size_t)cap));
            HIPCHK(c, hipMalloc((void **)&L.sel, sizeof(int) * (size_t)cap));
            HIPCHK(c, hipMalloc((void **)&L.gref, sizeof(float) * (size_t)cap));
        }
        if (!L.A) {
            HIPCHK(c, hipMalloc((void **)&L.A, sizeof(double) * 36));
            HIPCHK(c, hipMalloc((void **)&L.n_dev, sizeof(int)));
        }
        const size_t need = 2 * ((size_t)F.cols + 1);
        if (need > p->col_work_ints) {
            if (p->col_work) { HIPCHK(c, stream_wait(c->stream)); HIPCHK(c, hipFree(p->col_work)); }
            HIPCHK(c, hipMalloc((void **)&p->col_work, sizeof(int) * need));
            p->col_work_ints = need;
        }
        HIPCHK(c, launch_photo_reference(F.grey + (size_t)slot * F.npx, F.depth + (size_t)slot * F.npx, F.rows, F.cols, l,
                                         p->prm.fx, p->prm.fy, p->prm.cx, p->prm.cy, p->prm.fixed, (double)p->prm.gradient_threshold,
                                         cap, p->col_work, L.J, L.sel, L.zref, L.gref, L.A, L.n_dev, c->stream));
        HIPCHK(c, hipMemcpyAsync(&L.n, L.n_dev, sizeof(int), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, stream_wait(c->stream));       /* col_work is reused by the next level */
        L.rows = F.rows; L.cols = F.cols;
        if (n_selected) n_selected[l] = L.n;
        /* the reference asserts on both (NDEBUG is undefined): :464 xc < const_maxJacobianSize, :500 xc > const_minimumRequiredPts */
        if (L.n >= cap)
            return fail(c, DVO_ERR_INVALID, "level " + std::to_string(l) + ": " + std::to_string(L.n) + " selected pixels reach max_jacobian_size (RGBDOdometry.cpp:464 asserts)");
        if (L.n <= p->prm.min_required_pts)
            return fail(c, DVO_ERR_INVALID, "level " + std::to_string(l) + ": too few points with good texture (RGBDOdometry.cpp:500 asserts)");
        L.ready = true;
    }
    p->ref_slot = slot;
    return DVO_OK;
}

/* gaussNewtonIterations(level, T) (:514-597) for each listed level in order, on the now frame in `now_slot`.
 * T16: 4x4 row-major (TransformRep::matrix()), in/out.  eps_norms: n_run x iterations doubles (|eps| of every iteration,
 * -1 where not run); updates: n_run ints (iterations that changed T).  Both may be NULL. */
int dvo_photo_align(dvo_ctx *c, int now_slot, const int *levels, int n_run, double *T16, double *eps_norms, int *updates) {
    DVO_ENTER(c);
    dvo_photo_state *p = c->photo;
    if (!p || p->ref_slot < 0) return fail(c, DVO_ERR_STATE, "no reference frame (dvo_photo_set_ref)");
    if (!levels || n_run < 1 || !T16) return fail(c, DVO_ERR_INVALID, "bad arguments");
    if (now_slot < 0 || now_slot >= c->fs.n_slots || !c->fs.valid[now_slot]) return fail(c, DVO_ERR_INVALID, "no frame in this slot");
    const int it = p->prm.iterations;
    if (it * n_run > 64) return fail(c, DVO_ERR_INVALID, "too many level runs");
    for (int r = 0; r < n_run; r++) {
        const int l = levels[r];
        if (l < 0 || l >= c->fs.n_levels || !p->lv[l].ready)
            return fail(c, DVO_ERR_STATE, "no Jacobian for level " + std::to_string(l) + " (the reference asserts level != 0, :518)");
    }
    HIPCHK(c, hipMemcpyAsync(p->d_T, T16, sizeof(double) * 16, hipMemcpyHostToDevice, c->stream));
    std::vector<int> upd(n_run, 0);
    for (int r = 0; r < n_run; r++) {
        const int l = levels[r];
        FrameLevel &F = c->fs.lv[l];
        dvo_photo_state::Lvl &L = p->lv[l];
        HIPCHK(c, launch_photo_gauss_newton(L.J, L.sel, L.zref, L.gref, L.n_dev, L.A, F.grey + (size_t)now_slot * F.npx, F.rows, F.cols, l,
                                            p->prm.fx, p->prm.fy, p->prm.cx, p->prm.cy, p->prm.fixed, it, p->prm.eps_norm_stop,
                                            p->d_T, p->d_norms + (size_t)r * it, p->d_updates, nullptr, c->stream));
        HIPCHK(c, hipMemcpyAsync(&upd[r], p->d_updates, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    }
    HIPCHK(c, hipMemcpyAsync(T16, p->d_T, sizeof(double) * 16, hipMemcpyDeviceToHost, c->stream));
    if (eps_norms) HIPCHK(c, hipMemcpyAsync(eps_norms, p->d_norms, sizeof(double) * it * n_run, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, stream_wait(c->stream));
    if (updates) for (int r = 0; r < n_run; r++) updates[r] = upd[r];
    return DVO_OK;
}

/* inspection: J (n x 6 row-major), the selected pixels (row i, column j), A (6x6) of a reference level */
int dvo_photo_get_jacobian(dvo_ctx *c, int level, double *J, int *sel_i, int *sel_j, int capacity, double *A36, int *n_out) {
    DVO_ENTER(c);
    dvo_photo_state *p = c->photo;
    if (!p || !level_ok(level) || !p->lv[level].ready) return fail(c, DVO_ERR_STATE, "no Jacobian for this level");
    dvo_photo_state::Lvl &L = p->lv[level];
    HIPCHK(c, stream_wait(c->stream));
    const int n = std::min(L.n, capacity);
    if (n_out) *n_out = L.n;
    if (J && n > 0) HIPCHK(c, hipMemcpy(J, L.J, sizeof(double) * 6 * (size_t)n, hipMemcpyDeviceToHost));
    if ((sel_i || sel_j) && n > 0) {
        std::vector<int> s(n);
        HIPCHK(c, hipMemcpy(s.data(), L.sel, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost));
        for (int k = 0; k < n; k++) { if (sel_i) sel_i[k] = s[k] & 0xffff; if (sel_j) sel_j[k] = s[k] >> 16; }
    }
    if (A36) HIPCHK(c, hipMemcpy(A36, L.A, sizeof(double) * 36, hipMemcpyDeviceToHost));
    return DVO_OK;
}

}  // extern "C"
