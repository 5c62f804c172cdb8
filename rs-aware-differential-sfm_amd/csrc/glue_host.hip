// glue_host.hip -- C ABI of the caller-side glue: flatten (main.cc:398-444 + getAlpha/getAlphaK), depth map
// (main.cc:466-509), per-scanline pose table (rsframe.cc:771-800), host- and device-pointer variants.
#include <string.h>

#include "rsdsfm_internal.hpp"

using namespace rsdsfm;

// flatten of a column slab: d_img is the row-major [rows][cols] slab whose first column is image column col0
namespace rsdsfm {
int flatten_device(Ctx* c, const double* d_img, int32_t rows, int32_t cols, int32_t col0, double fx, double fy, double cx,
                          double cy, double gamma, double thr, double* d_q, double* d_u, double* d_alpha, double* d_alpha_k,
                          int64_t* n_out) {
    if (rows < 0 || cols < 0 || col0 < 0 || !n_out) return fail(c, RSDSFM_ERR_INVALID, "bad arguments");
    const int64_t n = (int64_t)rows * cols;
    if (n == 0) {
        *n_out = 0;
        return RSDSFM_OK;
    }
    if (!d_img || !d_q || !d_u || !d_alpha || !d_alpha_k) return fail(c, RSDSFM_ERR_INVALID, "null device pointer");
    const size_t ncells = (size_t)flatten_cells(rows, cols);
    int rc = ensure_ws(c, 2 * Arena::need(sizeof(int64_t) * ncells) + Arena::need(64) + 1024);
    if (rc != RSDSFM_OK) return rc;
    rc = ensure_pinned(c, 64);
    if (rc != RSDSFM_OK) return rc;
    Arena ws(c->d_ws);
    int64_t* d_counts = ws.take<int64_t>(ncells);
    int64_t* d_offsets = ws.take<int64_t>(ncells);
    // The count is final after the scan, BEFORE the scatter pass: the scan kernel stores it straight into host-mapped pinned
    // memory and the host waits for an event recorded behind that kernel -- no copy kernel, and the host's next steps (the RANSAC
    // sampler, the next launches) overlap the scatter pass, which subsequent work follows in stream order anyway.
    int64_t* h_total = static_cast<int64_t*>(c->h_pinned);
    if (!c->ev_ready) RSDSFM_HIP_CHECK(c, hipEventCreateWithFlags(&c->ev_ready, hipEventDisableTiming));
    rc = flatten_launch(c, d_img, rows, cols, col0, fx, fy, cx, cy, gamma, thr, d_q, d_u, d_alpha, d_alpha_k, d_counts, d_offsets, h_total,
                        c->ev_ready);
    if (rc != RSDSFM_OK) return rc;
    RSDSFM_HIP_CHECK(c, hipEventSynchronize(c->ev_ready));
    *n_out = *h_total;
    return RSDSFM_OK;
}
}  // namespace rsdsfm

namespace rsdsfm {
// The flatten of a whole image WITHOUT the host wait: the point count lands in *h_total (host-mapped pinned memory, written by the scan
// kernel) once the stream gets there.  The frame solve uses it to enqueue the RANSAC behind the flatten on the assumption that every
// pixel carries flow (a dense optical flow: n = rows * cols) and checks the count at its next wait.
int flatten_enqueue(Ctx* c, const double* d_img, int32_t rows, int32_t cols, double fx, double fy, double cx, double cy, double gamma,
                    double thr, double* d_q, double* d_u, double* d_alpha, double* d_alpha_k, int64_t* h_total) {
    if (rows <= 0 || cols <= 0 || !h_total) return fail(c, RSDSFM_ERR_INVALID, "bad arguments");
    if (!d_img || !d_q || !d_u || !d_alpha || !d_alpha_k) return fail(c, RSDSFM_ERR_INVALID, "null device pointer");
    const size_t ncells = (size_t)flatten_cells(rows, cols);
    int rc = ensure_ws(c, 2 * Arena::need(sizeof(int64_t) * ncells) + Arena::need(64) + 1024);
    if (rc != RSDSFM_OK) return rc;
    Arena ws(c->d_ws);
    int64_t* d_counts = ws.take<int64_t>(ncells);
    int64_t* d_offsets = ws.take<int64_t>(ncells);
    return flatten_launch(c, d_img, rows, cols, 0, fx, fy, cx, cy, gamma, thr, d_q, d_u, d_alpha, d_alpha_k, d_counts, d_offsets, h_total, nullptr);
}
}  // namespace rsdsfm

extern "C" {

int rsdsfm_flatten_dev(rsdsfm_ctx* ctx, const double* d_img, int32_t rows, int32_t cols, double fx, double fy, double cx, double cy,
                       double gamma, double thr, double* d_q, double* d_u, double* d_alpha, double* d_alpha_k, int64_t* n_out) {
    if (!ctx) return RSDSFM_ERR_INVALID;
    DeviceGuard device_guard_(&ctx->c);
    return flatten_device(&ctx->c, d_img, rows, cols, 0, fx, fy, cx, cy, gamma, thr, d_q, d_u, d_alpha, d_alpha_k, n_out);
}

int rsdsfm_flatten_slab_dev(rsdsfm_ctx* ctx, const double* d_img_slab, int32_t rows, int32_t slab_cols, int32_t col0, double fx,
                            double fy, double cx, double cy, double gamma, double thr, double* d_q, double* d_u, double* d_alpha,
                            double* d_alpha_k, int64_t* n_out) {
    if (!ctx) return RSDSFM_ERR_INVALID;
    DeviceGuard device_guard_(&ctx->c);
    return flatten_device(&ctx->c, d_img_slab, rows, slab_cols, col0, fx, fy, cx, cy, gamma, thr, d_q, d_u, d_alpha, d_alpha_k, n_out);
}

int rsdsfm_flatten(rsdsfm_ctx* ctx, const double* img, int32_t rows, int32_t cols, double fx, double fy, double cx, double cy,
                   double gamma, double thr, double* q, double* u, double* alpha, double* alpha_k, int64_t* n_out) {
    if (!ctx) return RSDSFM_ERR_INVALID;
    Ctx* c = &ctx->c;
    DeviceGuard device_guard_(c);
    if (rows < 0 || cols < 0 || !n_out) return fail(c, RSDSFM_ERR_INVALID, "bad arguments");
    const size_t N = (size_t)rows * (size_t)cols;
    if (N == 0) {
        *n_out = 0;
        return RSDSFM_OK;
    }
    if (!img || !q || !u || !alpha || !alpha_k) return fail(c, RSDSFM_ERR_INVALID, "null pointer");
    int rc = ensure_stage(c, 3 * Arena::need(16 * N) + 2 * Arena::need(8 * N) + 1024);
    if (rc != RSDSFM_OK) return rc;
    Arena sa(c->d_stage);
    double* d_img = sa.take<double>(2 * N);
    double* d_q = sa.take<double>(2 * N);
    double* d_u = sa.take<double>(2 * N);
    double* d_a = sa.take<double>(N);
    double* d_ak = sa.take<double>(N);
    if ((rc = xfer_h2d(c, d_img, img, 16 * N)) != RSDSFM_OK) return rc;  // (host_xfer.hip)
    int64_t cnt = 0;
    rc = rsdsfm_flatten_dev(ctx, d_img, rows, cols, fx, fy, cx, cy, gamma, thr, d_q, d_u, d_a, d_ak, &cnt);
    if (rc != RSDSFM_OK) return rc;
    const size_t Mk = (size_t)cnt;
    if (Mk) {
        RSDSFM_HIP_CHECK(c, hipMemcpyAsync(q, d_q, 16 * Mk, hipMemcpyDeviceToHost, c->stream));
        RSDSFM_HIP_CHECK(c, hipMemcpyAsync(u, d_u, 16 * Mk, hipMemcpyDeviceToHost, c->stream));
        RSDSFM_HIP_CHECK(c, hipMemcpyAsync(alpha, d_a, 8 * Mk, hipMemcpyDeviceToHost, c->stream));
        RSDSFM_HIP_CHECK(c, hipMemcpyAsync(alpha_k, d_ak, 8 * Mk, hipMemcpyDeviceToHost, c->stream));
    }
    RSDSFM_HIP_CHECK(c, hipStreamSynchronize(c->stream));
    *n_out = cnt;
    return RSDSFM_OK;
}

int rsdsfm_depth_map_dev(rsdsfm_ctx* ctx, double* d_inl, int64_t m, double v_inout[3], double fx, double fy, double cx, double cy,
                         int32_t rows, int32_t cols, double* d_depth_map, int32_t* d_xs, int32_t* d_ys, int* flipped) {
    if (!ctx) return RSDSFM_ERR_INVALID;
    DeviceGuard device_guard_(&ctx->c);
    return depth_map_device(&ctx->c, d_inl, m, v_inout, fx, fy, cx, cy, rows, cols, d_depth_map, d_xs, d_ys, flipped, nullptr, 0.0, 0.0,
                            nullptr, nullptr);
}

}  // extern "C"

namespace rsdsfm {
// w_or_null != null: RsFrame::setRelativePose's table for (v', w, k) is enqueued behind the depth map, reading the possibly
// sign-flipped v' from the stage's device header, and the ONE synchronisation at the end covers both (the frame solve's tail)
int depth_map_device(Ctx* c, double* d_inl, int64_t m, double v_inout[3], double fx, double fy, double cx, double cy, int32_t rows,
                     int32_t cols, double* d_depth_map, int32_t* d_xs, int32_t* d_ys, int* flipped, const double* w_or_null, double k,
                     double gamma, double* d_R_rows9, double* d_t_rows3) {
    if (m < 0 || rows < 0 || cols < 0 || !v_inout) return fail(c, RSDSFM_ERR_INVALID, "bad arguments");
    const size_t npix = (size_t)rows * (size_t)cols;
    if ((m > 0 && !d_inl) || (npix > 0 && !d_depth_map)) return fail(c, RSDSFM_ERR_INVALID, "null device pointer");
    int rc = ensure_ws(c, Arena::need(64) + Arena::need(8 * 1024) + 1024);
    if (rc != RSDSFM_OK) return rc;
    rc = ensure_pinned(c, 64);
    if (rc != RSDSFM_OK) return rc;
    Arena ws(c->d_ws);
    double* d_header = ws.take<double>(4);
    double* d_partials = ws.take<double>(1024);
    double* h_header = static_cast<double*>(c->h_pinned);  // written by zsum_decide_kernel itself (host-mapped)
    rc = depth_map_launch(c, d_inl, m, v_inout, fx, fy, cx, cy, rows, cols, d_depth_map, d_xs, d_ys, d_header, d_partials, h_header);
    if (rc != RSDSFM_OK) return rc;
    if (w_or_null && d_R_rows9 && d_t_rows3) {
        Pose pose;
        for (int i = 0; i < 3; ++i) pose.v[i] = v_inout[i], pose.w[i] = w_or_null[i];
        pose.k = k;
        rc = pose_table_launch(c, pose, gamma, rows, d_R_rows9, d_t_rows3, d_header + 1);
        if (rc != RSDSFM_OK) return rc;
    }
    RSDSFM_HIP_CHECK(c, hipStreamSynchronize(c->stream));
    if (flipped) *flipped = h_header[0] != 0.0;
    v_inout[0] = h_header[1];
    v_inout[1] = h_header[2];
    v_inout[2] = h_header[3];
    return RSDSFM_OK;
}
}  // namespace rsdsfm

extern "C" {

int rsdsfm_depth_map(rsdsfm_ctx* ctx, double* inl, int64_t m, double v_inout[3], double fx, double fy, double cx, double cy,
                     int32_t rows, int32_t cols, double* depth_map, int32_t* xs, int32_t* ys, int* flipped) {
    if (!ctx) return RSDSFM_ERR_INVALID;
    Ctx* c = &ctx->c;
    DeviceGuard device_guard_(c);
    if (m < 0 || rows < 0 || cols < 0 || !v_inout || (m > 0 && !inl)) return fail(c, RSDSFM_ERR_INVALID, "bad arguments");
    const size_t M = (size_t)m, npix = (size_t)rows * (size_t)cols;
    if (npix > 0 && !depth_map) return fail(c, RSDSFM_ERR_INVALID, "null depth_map");
    int rc = ensure_stage(c, Arena::need(24 * M) + Arena::need(8 * npix) + 2 * Arena::need(4 * M) + 1024);
    if (rc != RSDSFM_OK) return rc;
    Arena sa(c->d_stage);
    double* d_inl = sa.take<double>(3 * M);
    double* d_map = sa.take<double>(npix);
    int32_t* d_xs = sa.take<int32_t>(M);
    int32_t* d_ys = sa.take<int32_t>(M);
    if (M) RSDSFM_HIP_CHECK(c, hipMemcpyAsync(d_inl, inl, 24 * M, hipMemcpyHostToDevice, c->stream));
    rc = rsdsfm_depth_map_dev(ctx, d_inl, m, v_inout, fx, fy, cx, cy, rows, cols, d_map, d_xs, d_ys, flipped);
    if (rc != RSDSFM_OK) return rc;
    if (M) RSDSFM_HIP_CHECK(c, hipMemcpyAsync(inl, d_inl, 24 * M, hipMemcpyDeviceToHost, c->stream));
    if (npix) RSDSFM_HIP_CHECK(c, hipMemcpyAsync(depth_map, d_map, 8 * npix, hipMemcpyDeviceToHost, c->stream));
    if (xs && M) RSDSFM_HIP_CHECK(c, hipMemcpyAsync(xs, d_xs, 4 * M, hipMemcpyDeviceToHost, c->stream));
    if (ys && M) RSDSFM_HIP_CHECK(c, hipMemcpyAsync(ys, d_ys, 4 * M, hipMemcpyDeviceToHost, c->stream));
    RSDSFM_HIP_CHECK(c, hipStreamSynchronize(c->stream));
    return RSDSFM_OK;
}

int rsdsfm_pose_table_dev(rsdsfm_ctx* ctx, const double v[3], const double w[3], double k, double gamma, int32_t rows, double* d_R,
                          double* d_t) {
    if (!ctx) return RSDSFM_ERR_INVALID;
    Ctx* c = &ctx->c;
    DeviceGuard device_guard_(c);
    if (rows < 0 || !v || !w || (rows > 0 && (!d_R || !d_t))) return fail(c, RSDSFM_ERR_INVALID, "bad arguments");
    Pose pose;
    memcpy(pose.v, v, sizeof(pose.v));
    memcpy(pose.w, w, sizeof(pose.w));
    pose.k = k;
    return pose_table_launch(c, pose, gamma, rows, d_R, d_t);
}

}  // extern "C"
