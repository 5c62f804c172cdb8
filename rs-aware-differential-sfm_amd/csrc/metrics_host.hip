// metrics_host.hip -- C ABI of the accuracy metrics: rotation / translation errors of
// evaluateVelocities (errorMeasure.cpp:178-186; a dozen scalar operations, host only) and the reprojection error /
// error image of the estimated structure (Camera::meanReprojectionError camera.cc:594-691, Camera::createErrorImage
// camera.cc:503-591) on the GPU.
#include <math.h>
#include <string.h>

#include "rsdsfm_internal.hpp"

namespace rsdsfm {
int metrics_blocks_max();
int reprojection_error_launch(Ctx* c, const float* d_est, const double* d_gt_depth, const double* d_est_depth, const double* d_R,
                              const double* d_t, double fx, double fy, double cx, double cy, int rows, int cols, double max_norm,
                              unsigned char* d_error_image, double* d_scale_partials, double* d_header, double* d_error_partials, int* error_rows);
}

using namespace rsdsfm;

extern "C" {

int rsdsfm_velocity_errors(const double w_est[3], const double v_est[3], const double w_true[3], const double v_true[3], double* w_error,
                           double* v_error) {
    if (!w_est || !v_est || !w_true || !v_true || !w_error || !v_error) return RSDSFM_ERR_INVALID;
    const double* w = w_est;
    const double* wt = w_true;
    const double A[9] = {1, -w[2], w[1], w[2], 1, -w[0], -w[1], w[0], 1};        // results_w_rot (errorMeasure.cpp:179-182)
    const double B[9] = {1, -wt[2], wt[1], wt[2], 1, -wt[0], -wt[1], wt[0], 1};  // true_rot (errorMeasure.cpp:127-129)
    double E[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) E[i * 3 + j] = (A[i * 3] * B[j * 3] + A[i * 3 + 1] * B[j * 3 + 1]) + A[i * 3 + 2] * B[j * 3 + 2];
    *w_error = sqrt((E[7] * E[7] + E[2] * E[2]) + E[3] * E[3]);
    const double dot = (v_est[0] * v_true[0] + v_est[1] * v_true[1]) + v_est[2] * v_true[2];
    const double nv = sqrt((v_est[0] * v_est[0] + v_est[1] * v_est[1]) + v_est[2] * v_est[2]);
    const double nt = sqrt((v_true[0] * v_true[0] + v_true[1] * v_true[1]) + v_true[2] * v_true[2]);
    *v_error = acos(dot / (nv * nt));
    return RSDSFM_OK;
}

int rsdsfm_reprojection_error_dev(rsdsfm_ctx* ctx, const float* d_est_coords, const double* d_gt_depth, const double* d_est_depth,
                                  const double* d_R_abs_rows9, const double* d_t_abs_rows3, double fx, double fy, double cx, double cy,
                                  int32_t rows, int32_t cols, double max_norm, rsdsfm_reprojection_stats* stats,
                                  uint8_t* d_error_image_or_null) {
    if (!ctx) return RSDSFM_ERR_INVALID;
    Ctx* c = &ctx->c;
    DeviceGuard device_guard_(c);
    if (rows < 0 || cols < 0 || !stats) return fail(c, RSDSFM_ERR_INVALID, "bad arguments");
    memset(stats, 0, sizeof(*stats));
    if ((int64_t)rows * cols == 0) {
        stats->scale = stats->mean_error = NAN;  // 0 / 0 in the reference
        return RSDSFM_OK;
    }
    if (!d_est_coords || !d_gt_depth || !d_est_depth || !d_R_abs_rows9 || !d_t_abs_rows3) return fail(c, RSDSFM_ERR_INVALID, "null device pointer");
    const size_t G = (size_t)metrics_blocks_max();
    int rc = ensure_ws(c, Arena::need(8 * 3 * G) + 1024);
    if (rc != RSDSFM_OK) return rc;
    rc = ensure_pinned(c, 8 * (8 + 3 * G));
    if (rc != RSDSFM_OK) return rc;
    Arena ws(c->d_ws);
    double* d_scale_partials = ws.take<double>(3 * G);
    // the header (8 doubles) and the error partials behind it are written by the second pass straight into host-mapped pinned memory
    // (3 doubles per workgroup over the host link): no copy command behind the launches, the host only synchronises
    double* h = static_cast<double*>(c->h_pinned);
    int erows = 0;
    rc = reprojection_error_launch(c, d_est_coords, d_gt_depth, d_est_depth, d_R_abs_rows9, d_t_abs_rows3, fx, fy, cx, cy, rows, cols, max_norm,
                                   d_error_image_or_null, d_scale_partials, h, h + 8, &erows);
    if (rc != RSDSFM_OK) return rc;
    RSDSFM_HIP_CHECK(c, hipStreamSynchronize(c->stream));
    // the error sums: the workgroups' partials in workgroup order (a fixed order: the result does not depend on the dispatch); counts are exact
    double sum_error = 0.0, error_inliers = 0.0;
    for (int b = 0; b < erows; ++b) {
        sum_error += h[8 + 3 * b];
        error_inliers += h[8 + 3 * b + 1];
    }
    stats->scale = h[0];
    stats->scale_inliers = (int64_t)h[1];
    stats->number_outliers = (int64_t)h[2];
    stats->sum_error = sum_error;
    stats->error_inliers = (int64_t)error_inliers;
    h[3] = sum_error, h[4] = error_inliers;
    stats->mean_error = h[3] * 1.0 / h[4];
    return RSDSFM_OK;
}

int rsdsfm_reprojection_error(rsdsfm_ctx* ctx, const float* est_coords, const double* gt_depth, const double* est_depth,
                              const double* R_abs_rows9, const double* t_abs_rows3, double fx, double fy, double cx, double cy, int32_t rows,
                              int32_t cols, double max_norm, rsdsfm_reprojection_stats* stats, uint8_t* error_image_or_null) {
    if (!ctx) return RSDSFM_ERR_INVALID;
    Ctx* c = &ctx->c;
    DeviceGuard device_guard_(c);
    if (rows < 0 || cols < 0 || !stats) return fail(c, RSDSFM_ERR_INVALID, "bad arguments");
    const size_t npix = (size_t)rows * (size_t)cols;
    if (npix == 0) return rsdsfm_reprojection_error_dev(ctx, nullptr, nullptr, nullptr, nullptr, nullptr, fx, fy, cx, cy, rows, cols, max_norm, stats, nullptr);
    if (!est_coords || !gt_depth || !est_depth || !R_abs_rows9 || !t_abs_rows3) return fail(c, RSDSFM_ERR_INVALID, "null pointer");
    int rc = ensure_stage(c, Arena::need(12 * npix) + 2 * Arena::need(8 * npix) + Arena::need(72 * (size_t)rows) + Arena::need(24 * (size_t)rows) +
                                 Arena::need(npix) + 2048);
    if (rc != RSDSFM_OK) return rc;
    Arena sa(c->d_stage);
    float* d_est = sa.take<float>(3 * npix);
    double* d_gd = sa.take<double>(npix);
    double* d_ed = sa.take<double>(npix);
    double* d_R = sa.take<double>(9 * (size_t)rows);
    double* d_t = sa.take<double>(3 * (size_t)rows);
    uint8_t* d_img = sa.take<uint8_t>(npix);
    RSDSFM_HIP_CHECK(c, hipMemcpyAsync(d_est, est_coords, 12 * npix, hipMemcpyHostToDevice, c->stream));
    RSDSFM_HIP_CHECK(c, hipMemcpyAsync(d_gd, gt_depth, 8 * npix, hipMemcpyHostToDevice, c->stream));
    RSDSFM_HIP_CHECK(c, hipMemcpyAsync(d_ed, est_depth, 8 * npix, hipMemcpyHostToDevice, c->stream));
    RSDSFM_HIP_CHECK(c, hipMemcpyAsync(d_R, R_abs_rows9, 72 * (size_t)rows, hipMemcpyHostToDevice, c->stream));
    RSDSFM_HIP_CHECK(c, hipMemcpyAsync(d_t, t_abs_rows3, 24 * (size_t)rows, hipMemcpyHostToDevice, c->stream));
    rc = rsdsfm_reprojection_error_dev(ctx, d_est, d_gd, d_ed, d_R, d_t, fx, fy, cx, cy, rows, cols, max_norm, stats, error_image_or_null ? d_img : nullptr);
    if (rc != RSDSFM_OK) return rc;
    if (error_image_or_null) {
        RSDSFM_HIP_CHECK(c, hipMemcpyAsync(error_image_or_null, d_img, npix, hipMemcpyDeviceToHost, c->stream));
        RSDSFM_HIP_CHECK(c, hipStreamSynchronize(c->stream));
    }
    return RSDSFM_OK;
}

}  // extern "C"
