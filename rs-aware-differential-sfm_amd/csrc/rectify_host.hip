// rectify_host.hip -- C ABI of the consumers of the solve's output: RS -> GS back projection
// (RsFrame::backProject / backProjectGs, rsframe.cc:803-878), crack interpolation (Camera::interpolateCrackyImage,
// camera.cc:753-774) and the 8-bit depth image (main.cc:480-509); host- and device-pointer variants.
#include <string.h>

#include "rsdsfm_internal.hpp"

using namespace rsdsfm;

extern "C" {

int rsdsfm_back_project_dev(rsdsfm_ctx* ctx, const uint8_t* d_image_bgr, const double* d_depth_map, const double* d_R_rows9,
                            const double* d_t_rows3, double fx, double fy, double cx, double cy, int32_t rows, int32_t cols, int mode,
                            int q5_mode, uint8_t* d_gs_image_bgr, float* d_coords3d_or_null) {
    if (!ctx) return RSDSFM_ERR_INVALID;
    Ctx* c = &ctx->c;
    DeviceGuard device_guard_(c);
    if (rows < 0 || cols < 0 || (int64_t)rows * cols > (int64_t)INT32_MAX) return fail(c, RSDSFM_ERR_INVALID, "bad image size");
    if (mode != RSDSFM_BACKPROJECT_RS && mode != RSDSFM_BACKPROJECT_GS) return fail(c, RSDSFM_ERR_INVALID, "unknown back-projection mode");
    if (q5_mode != RSDSFM_Q5_COMPAT && q5_mode != RSDSFM_Q5_FIXED) return fail(c, RSDSFM_ERR_INVALID, "unknown q5_mode");
    const size_t npix = (size_t)rows * (size_t)cols;
    if (npix == 0) return RSDSFM_OK;
    if (!d_image_bgr || !d_depth_map || !d_R_rows9 || !d_t_rows3 || !d_gs_image_bgr) return fail(c, RSDSFM_ERR_INVALID, "null device pointer");
    return back_project_launch(c, d_image_bgr, d_depth_map, d_R_rows9, d_t_rows3, fx, fy, cx, cy, rows, cols, mode, q5_mode,
                               d_gs_image_bgr, d_coords3d_or_null);
}

int rsdsfm_back_project(rsdsfm_ctx* ctx, const uint8_t* image_bgr, const double* depth_map, const double* R_rows9,
                        const double* t_rows3, double fx, double fy, double cx, double cy, int32_t rows, int32_t cols, int mode,
                        int q5_mode, uint8_t* gs_image_bgr, float* coords3d_or_null) {
    if (!ctx) return RSDSFM_ERR_INVALID;
    Ctx* c = &ctx->c;
    DeviceGuard device_guard_(c);
    if (rows < 0 || cols < 0) return fail(c, RSDSFM_ERR_INVALID, "bad image size");
    const size_t npix = (size_t)rows * (size_t)cols;
    if (npix == 0) return RSDSFM_OK;
    if (!image_bgr || !depth_map || !R_rows9 || !t_rows3 || !gs_image_bgr) return fail(c, RSDSFM_ERR_INVALID, "null pointer");
    int rc = ensure_stage(c, 2 * Arena::need(3 * npix) + Arena::need(8 * npix) + Arena::need(72 * (size_t)rows) + Arena::need(24 * (size_t)rows) +
                                 Arena::need(12 * npix) + 2048);
    if (rc != RSDSFM_OK) return rc;
    Arena sa(c->d_stage);
    uint8_t* d_img = sa.take<uint8_t>(3 * npix);
    uint8_t* d_gs = sa.take<uint8_t>(3 * npix);
    double* d_dm = sa.take<double>(npix);
    double* d_R = sa.take<double>(9 * (size_t)rows);
    double* d_t = sa.take<double>(3 * (size_t)rows);
    float* d_c3 = sa.take<float>(3 * npix);
    RSDSFM_HIP_CHECK(c, hipMemcpyAsync(d_img, image_bgr, 3 * npix, hipMemcpyHostToDevice, c->stream));
    RSDSFM_HIP_CHECK(c, hipMemcpyAsync(d_dm, depth_map, 8 * npix, hipMemcpyHostToDevice, c->stream));
    RSDSFM_HIP_CHECK(c, hipMemcpyAsync(d_R, R_rows9, 72 * (size_t)rows, hipMemcpyHostToDevice, c->stream));
    RSDSFM_HIP_CHECK(c, hipMemcpyAsync(d_t, t_rows3, 24 * (size_t)rows, hipMemcpyHostToDevice, c->stream));
    rc = rsdsfm_back_project_dev(ctx, d_img, d_dm, d_R, d_t, fx, fy, cx, cy, rows, cols, mode, q5_mode, d_gs, coords3d_or_null ? d_c3 : nullptr);
    if (rc != RSDSFM_OK) return rc;
    RSDSFM_HIP_CHECK(c, hipMemcpyAsync(gs_image_bgr, d_gs, 3 * npix, hipMemcpyDeviceToHost, c->stream));
    if (coords3d_or_null) RSDSFM_HIP_CHECK(c, hipMemcpyAsync(coords3d_or_null, d_c3, 12 * npix, hipMemcpyDeviceToHost, c->stream));
    RSDSFM_HIP_CHECK(c, hipStreamSynchronize(c->stream));
    return RSDSFM_OK;
}

int rsdsfm_interpolate_cracky_dev(rsdsfm_ctx* ctx, const uint8_t* d_image_in_bgr, int32_t rows, int32_t cols, int32_t offset,
                                  uint8_t* d_image_out_bgr) {
    if (!ctx) return RSDSFM_ERR_INVALID;
    Ctx* c = &ctx->c;
    DeviceGuard device_guard_(c);
    if (rows < 0 || cols < 0 || offset < 0) return fail(c, RSDSFM_ERR_INVALID, "bad arguments");
    if ((int64_t)rows * cols == 0) return RSDSFM_OK;
    if (!d_image_in_bgr || !d_image_out_bgr || d_image_in_bgr == d_image_out_bgr) return fail(c, RSDSFM_ERR_INVALID, "null or aliased device pointer");
    return interpolate_cracky_launch(c, d_image_in_bgr, rows, cols, offset, d_image_out_bgr);
}

int rsdsfm_interpolate_cracky(rsdsfm_ctx* ctx, const uint8_t* image_in_bgr, int32_t rows, int32_t cols, int32_t offset,
                              uint8_t* image_out_bgr) {
    if (!ctx) return RSDSFM_ERR_INVALID;
    Ctx* c = &ctx->c;
    DeviceGuard device_guard_(c);
    if (rows < 0 || cols < 0 || offset < 0) return fail(c, RSDSFM_ERR_INVALID, "bad arguments");
    const size_t nb = 3 * (size_t)rows * (size_t)cols;
    if (nb == 0) return RSDSFM_OK;
    if (!image_in_bgr || !image_out_bgr) return fail(c, RSDSFM_ERR_INVALID, "null pointer");
    int rc = ensure_stage(c, 2 * Arena::need(nb) + 1024);
    if (rc != RSDSFM_OK) return rc;
    Arena sa(c->d_stage);
    uint8_t* d_in = sa.take<uint8_t>(nb);
    uint8_t* d_out = sa.take<uint8_t>(nb);
    RSDSFM_HIP_CHECK(c, hipMemcpyAsync(d_in, image_in_bgr, nb, hipMemcpyHostToDevice, c->stream));
    rc = interpolate_cracky_launch(c, d_in, rows, cols, offset, d_out);
    if (rc != RSDSFM_OK) return rc;
    RSDSFM_HIP_CHECK(c, hipMemcpyAsync(image_out_bgr, d_out, nb, hipMemcpyDeviceToHost, c->stream));
    RSDSFM_HIP_CHECK(c, hipStreamSynchronize(c->stream));
    return RSDSFM_OK;
}

int rsdsfm_depth_preview_dev(rsdsfm_ctx* ctx, const double* d_inl, int64_t m, double fx, double fy, double cx, double cy, int32_t rows,
                             int32_t cols, uint8_t* d_depth_est) {
    if (!ctx) return RSDSFM_ERR_INVALID;
    Ctx* c = &ctx->c;
    DeviceGuard device_guard_(c);
    if (m < 0 || m > (int64_t)INT32_MAX || rows < 0 || cols < 0 || (int64_t)rows * cols > (int64_t)INT32_MAX) return fail(c, RSDSFM_ERR_INVALID, "bad arguments");
    const size_t npix = (size_t)rows * (size_t)cols;
    if (npix == 0) return RSDSFM_OK;
    if ((m > 0 && !d_inl) || !d_depth_est) return fail(c, RSDSFM_ERR_INVALID, "null device pointer");
    int rc = ensure_ws(c, Arena::need(8 * 2048) + 1024);
    if (rc != RSDSFM_OK) return rc;
    return depth_preview_launch(c, d_inl, m, fx, fy, cx, cy, rows, cols, d_depth_est, static_cast<double*>(c->d_ws));
}

int rsdsfm_rectify_frame_dev(rsdsfm_ctx* ctx, const double* d_inl, int64_t m, const uint8_t* d_image_bgr, const double* d_depth_map,
                             const double* d_R_rows9, const double* d_t_rows3, double fx, double fy, double cx, double cy, int32_t rows, int32_t cols,
                             int mode, int q5_mode, int32_t offset, uint8_t* d_depth_est, uint8_t* d_gs_image_bgr, float* d_coords3d_or_null,
                             uint8_t* d_fixed_image_bgr) {
    if (!ctx) return RSDSFM_ERR_INVALID;
    Ctx* c = &ctx->c;
    DeviceGuard device_guard_(c);
    if (m < 0 || m > (int64_t)INT32_MAX || rows < 0 || cols < 0 || offset < 0 || (int64_t)rows * cols > (int64_t)INT32_MAX) return fail(c, RSDSFM_ERR_INVALID, "bad arguments");
    if (mode != RSDSFM_BACKPROJECT_RS && mode != RSDSFM_BACKPROJECT_GS) return fail(c, RSDSFM_ERR_INVALID, "unknown back-projection mode");
    if (q5_mode != RSDSFM_Q5_COMPAT && q5_mode != RSDSFM_Q5_FIXED) return fail(c, RSDSFM_ERR_INVALID, "unknown q5_mode");
    if ((int64_t)rows * cols == 0) return RSDSFM_OK;
    if ((m > 0 && !d_inl) || !d_image_bgr || !d_depth_map || !d_R_rows9 || !d_t_rows3 || !d_depth_est || !d_gs_image_bgr || !d_fixed_image_bgr ||
        d_gs_image_bgr == d_fixed_image_bgr)
        return fail(c, RSDSFM_ERR_INVALID, "null or aliased device pointer");
    int rc = ensure_ws(c, Arena::need(8 * 2048) + 1024);
    if (rc != RSDSFM_OK) return rc;
    return rectify_frame_launch(c, d_inl, m, d_image_bgr, d_depth_map, d_R_rows9, d_t_rows3, fx, fy, cx, cy, rows, cols, mode, q5_mode, offset,
                                d_depth_est, d_gs_image_bgr, d_coords3d_or_null, d_fixed_image_bgr, static_cast<double*>(c->d_ws));
}

int rsdsfm_depth_preview(rsdsfm_ctx* ctx, const double* inl, int64_t m, double fx, double fy, double cx, double cy, int32_t rows,
                         int32_t cols, uint8_t* depth_est) {
    if (!ctx) return RSDSFM_ERR_INVALID;
    Ctx* c = &ctx->c;
    DeviceGuard device_guard_(c);
    if (m < 0 || rows < 0 || cols < 0) return fail(c, RSDSFM_ERR_INVALID, "bad arguments");
    const size_t npix = (size_t)rows * (size_t)cols;
    if (npix == 0) return RSDSFM_OK;
    if ((m > 0 && !inl) || !depth_est) return fail(c, RSDSFM_ERR_INVALID, "null pointer");
    int rc = ensure_stage(c, Arena::need(24 * (size_t)m + 8) + Arena::need(npix) + 1024);
    if (rc != RSDSFM_OK) return rc;
    Arena sa(c->d_stage);
    double* d_inl = sa.take<double>(3 * (size_t)m + 1);
    uint8_t* d_out = sa.take<uint8_t>(npix);
    if (m) RSDSFM_HIP_CHECK(c, hipMemcpyAsync(d_inl, inl, 24 * (size_t)m, hipMemcpyHostToDevice, c->stream));
    rc = rsdsfm_depth_preview_dev(ctx, d_inl, m, fx, fy, cx, cy, rows, cols, d_out);
    if (rc != RSDSFM_OK) return rc;
    RSDSFM_HIP_CHECK(c, hipMemcpyAsync(depth_est, d_out, npix, hipMemcpyDeviceToHost, c->stream));
    RSDSFM_HIP_CHECK(c, hipStreamSynchronize(c->stream));
    return RSDSFM_OK;
}

}  // extern "C"
