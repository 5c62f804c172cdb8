// gtflow_host.hip -- C ABI of the ground-truth flow search (Camera::calculateTrueFlow,
// camera.cc:209-249; RsFrame::calculateImageCoordinatesRsFrame, rsframe.cc:740-768).
#include "rsdsfm_internal.hpp"

namespace rsdsfm {
int true_flow_launch(Ctx* c, const double* d_wx, const double* d_wy, const double* d_wz, int rows, int cols, const double* d_R2,
                     const double* d_t2, int rows2, double fx, double fy, double cx, double cy, int q5_mode, double* d_flow,
                     int* d_best_row);
}

using namespace rsdsfm;

extern "C" {

int rsdsfm_true_flow_dev(rsdsfm_ctx* ctx, const double* d_world_x, const double* d_world_y, const double* d_world_z, int32_t rows,
                         int32_t cols, const double* d_R2_rows9, const double* d_t2_rows3, int32_t rows2, double fx, double fy, double cx,
                         double cy, int q5_mode, double* d_flow, int32_t* d_best_row_or_null) {
    if (!ctx) return RSDSFM_ERR_INVALID;
    Ctx* c = &ctx->c;
    DeviceGuard device_guard_(c);
    if (rows < 0 || cols < 0 || rows2 < 0) return fail(c, RSDSFM_ERR_INVALID, "bad sizes");
    if (q5_mode != RSDSFM_Q5_COMPAT && q5_mode != RSDSFM_Q5_FIXED) return fail(c, RSDSFM_ERR_INVALID, "unknown q5_mode");
    if ((int64_t)rows * cols == 0) return RSDSFM_OK;
    if (rows2 < 1) return fail(c, RSDSFM_ERR_INVALID, "frame 2 needs at least one scanline");
    if (!d_world_x || !d_world_y || !d_world_z || !d_flow || !d_R2_rows9 || !d_t2_rows3) return fail(c, RSDSFM_ERR_INVALID, "null device pointer");
    return true_flow_launch(c, d_world_x, d_world_y, d_world_z, rows, cols, d_R2_rows9, d_t2_rows3, rows2, fx, fy, cx, cy, q5_mode, d_flow,
                            d_best_row_or_null);
}

int rsdsfm_true_flow(rsdsfm_ctx* ctx, const double* world_x, const double* world_y, const double* world_z, int32_t rows, int32_t cols,
                     const double* R2_rows9, const double* t2_rows3, int32_t rows2, double fx, double fy, double cx, double cy, int q5_mode,
                     double* flow, int32_t* best_row_or_null) {
    if (!ctx) return RSDSFM_ERR_INVALID;
    Ctx* c = &ctx->c;
    DeviceGuard device_guard_(c);
    if (rows < 0 || cols < 0 || rows2 < 0) return fail(c, RSDSFM_ERR_INVALID, "bad sizes");
    const size_t npix = (size_t)rows * (size_t)cols;
    if (npix == 0) return RSDSFM_OK;
    if (rows2 < 1) return fail(c, RSDSFM_ERR_INVALID, "frame 2 needs at least one scanline");
    if (!world_x || !world_y || !world_z || !flow || !R2_rows9 || !t2_rows3) return fail(c, RSDSFM_ERR_INVALID, "null pointer");
    const size_t r2 = (size_t)rows2;
    int rc = ensure_stage(c, 3 * Arena::need(8 * npix) + Arena::need(72 * r2) + Arena::need(24 * r2) + Arena::need(16 * npix) + Arena::need(4 * npix) + 2048);
    if (rc != RSDSFM_OK) return rc;
    Arena sa(c->d_stage);
    double* d_wx = sa.take<double>(npix);
    double* d_wy = sa.take<double>(npix);
    double* d_wz = sa.take<double>(npix);
    double* d_R = sa.take<double>(9 * r2);
    double* d_t = sa.take<double>(3 * r2);
    double* d_flow = sa.take<double>(2 * npix);
    int32_t* d_best = sa.take<int32_t>(npix);
    RSDSFM_HIP_CHECK(c, hipMemcpyAsync(d_wx, world_x, 8 * npix, hipMemcpyHostToDevice, c->stream));
    RSDSFM_HIP_CHECK(c, hipMemcpyAsync(d_wy, world_y, 8 * npix, hipMemcpyHostToDevice, c->stream));
    RSDSFM_HIP_CHECK(c, hipMemcpyAsync(d_wz, world_z, 8 * npix, hipMemcpyHostToDevice, c->stream));
    RSDSFM_HIP_CHECK(c, hipMemcpyAsync(d_R, R2_rows9, 72 * r2, hipMemcpyHostToDevice, c->stream));
    RSDSFM_HIP_CHECK(c, hipMemcpyAsync(d_t, t2_rows3, 24 * r2, hipMemcpyHostToDevice, c->stream));
    rc = rsdsfm_true_flow_dev(ctx, d_wx, d_wy, d_wz, rows, cols, d_R, d_t, rows2, fx, fy, cx, cy, q5_mode, d_flow, best_row_or_null ? d_best : nullptr);
    if (rc != RSDSFM_OK) return rc;
    RSDSFM_HIP_CHECK(c, hipMemcpyAsync(flow, d_flow, 16 * npix, hipMemcpyDeviceToHost, c->stream));
    if (best_row_or_null) RSDSFM_HIP_CHECK(c, hipMemcpyAsync(best_row_or_null, d_best, 4 * npix, hipMemcpyDeviceToHost, c->stream));
    RSDSFM_HIP_CHECK(c, hipStreamSynchronize(c->stream));
    return RSDSFM_OK;
}

}  // extern "C"
