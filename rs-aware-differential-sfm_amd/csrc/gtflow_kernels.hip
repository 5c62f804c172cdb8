// gtflow_kernels.hip -- SURVEY section 8(f-2): ground-truth optical flow between two rolling-shutter frames (gfx950).
//
// Replaces Camera::calculateTrueFlow (reference camera.cc:209-249) and RsFrame::calculateImageCoordinatesRsFrame
// (rsframe.cc:740-768): for every pixel of frame 1 with a world point W, W is projected with the pose of EVERY scanline
// i of frame 2 and the scanline minimising |y_projected - i| wins (first minimum, strict <).  rows * cols * rows2
// projections (6.6e8 at 1280x720): the only compute-heavy loop outside the solver.  It is VALU-bound fp64 work, not
// GEMM-shaped (one IEEE division per projection, an argmin with an index-dependent target):
//   * one pixel per lane; the scanline loop is uniform across the machine, so the pose rows come through the scalar
//     data path (s_load) and are broadcast operands of the VALU instructions -- no LDS, no vector loads in the loop;
//   * only what decides the argmin is computed in the loop (camera-frame y and z: 6 mul + 6 add, 1 division, the
//     intrinsics and the compare; with -DRSDSFM_FUSED=1: 2 mul + 4 fma + 2 add, the intrinsics as one fma); the x
//     coordinate is computed once for the winner.  Each value that is computed is computed with the reference's operation
//     order (rso_project_scanline of the oracle build with the same arithmetic mode), so flows and winners are
//     bit-identical to the oracle's;
//   * the world maps are column-major (Eigen) and the flow row-major (cv::Mat): 24 B read + 16 B written per pixel,
//     negligible against rows2 x ~30 fp64 instructions per pixel.
#include <math.h>

#include "rsdsfm_internal.hpp"

#ifndef RSDSFM_FUSED
#define RSDSFM_FUSED 0
#endif

namespace rsdsfm {

namespace {
constexpr int kGF = 256;

// one row of worldToCameraFrame: [R t; 0 1] * (W, 1), Eigen's 4x4 product evaluated left to right (rsframe.cc:740-768).
// Default build: the reference's unfused arithmetic.  -DRSDSFM_FUSED=1 (librsdsfm_hip_fused.so): contracted at the places
// the oracle's -DRSO_FUSED build calls fma() (rso_project_scanline) -- the search is bound by fp64 issue (0.78 -> 0.61 ms).
__device__ __forceinline__ double cam_row(double r0, double r1, double r2, double t, double X, double Y, double Z) {
#if RSDSFM_FUSED
    return __builtin_fma(r2, Z, __builtin_fma(r1, Y, r0 * X)) + t * 1.0;
#else
    return ((r0 * X + r1 * Y) + r2 * Z) + t * 1.0;
#endif
}
// spaceToPlane: c / z * f + c0
__device__ __forceinline__ double to_plane(double c, double z, double f, double c0) {
#if RSDSFM_FUSED
    return __builtin_fma(c / z, f, c0);
#else
    return c / z * f + c0;
#endif
}
}

// a workgroup owns a 16 x 16 pixel tile; lanes run along v (the contiguous direction of the column-major world maps:
// 16 x 8 B = one 128-byte line per column), so the maps are fetched once (the row-major pixel order this replaced
// re-fetched every line ~8 times: 182 MB instead of 22 MB at 1280x720)
__global__ __launch_bounds__(kGF) void true_flow_kernel(const double* __restrict__ wx, const double* __restrict__ wy,
                                                       const double* __restrict__ wz, int rows, int cols,
                                                       const double* __restrict__ R2, const double* __restrict__ t2, int rows2,
                                                       double fx, double fyp, double cx, double cy, double2* __restrict__ flow,
                                                       int* __restrict__ best_row_out) {
    const int tiles_u = (cols + 15) / 16;
    const int64_t ntiles = (int64_t)tiles_u * ((rows + 15) / 16);
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int v = (int)(tile / tiles_u) * 16 + (threadIdx.x & 15), u = (int)(tile % tiles_u) * 16 + (threadIdx.x >> 4);
        if (v >= rows || u >= cols) continue;
        const int64_t p = (int64_t)v * cols + u;
        const int64_t cm = (int64_t)u * rows + v;
        const double X = wx[cm], Y = wy[cm], Z = wz[cm];
        double f2x = (double)u, f2y = (double)v;
        int best_row = -1;
        if (sqrt(X * X + Y * Y + Z * Z) != 0) {
            double min_diff = INFINITY;
            best_row = 0;
#pragma unroll 4
            for (int i = 0; i < rows2; ++i) {
                const double* Ri = R2 + (int64_t)i * 9;  // uniform address: scalar loads
                const double* ti = t2 + (int64_t)i * 3;
                const double yc = cam_row(Ri[3], Ri[4], Ri[5], ti[1], X, Y, Z);
                const double zc = cam_row(Ri[6], Ri[7], Ri[8], ti[2], X, Y, Z);
                const double py = to_plane(yc, zc, fyp, cy);
                const double diff = fabs(py - (double)i);
                if (diff < min_diff) {
                    min_diff = diff;
                    best_row = i;
                }
            }
            const double* Ri = R2 + (int64_t)best_row * 9;
            const double* ti = t2 + (int64_t)best_row * 3;
            const double xc = cam_row(Ri[0], Ri[1], Ri[2], ti[0], X, Y, Z);
            const double yc = cam_row(Ri[3], Ri[4], Ri[5], ti[1], X, Y, Z);
            const double zc = cam_row(Ri[6], Ri[7], Ri[8], ti[2], X, Y, Z);
            const double px = to_plane(xc, zc, fx, cx);
            const double py = to_plane(yc, zc, fyp, cy);
            if (sqrt(px * px + py * py) != 0) {
                f2x = px;
                f2y = py;
            }
        }
        flow[p] = make_double2(f2x - (double)u, f2y - (double)v);
        if (best_row_out) best_row_out[p] = best_row;
    }
}

int true_flow_launch(Ctx* c, const double* d_wx, const double* d_wy, const double* d_wz, int rows, int cols, const double* d_R2,
                     const double* d_t2, int rows2, double fx, double fy, double cx, double cy, int q5_mode, double* d_flow,
                     int* d_best_row) {
    int64_t blocks = (int64_t)((cols + 15) / 16) * ((rows + 15) / 16);
    if (blocks < 1) blocks = 1;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(true_flow_kernel, dim3((int)blocks), dim3(kGF), 0, c->stream, d_wx, d_wy, d_wz, rows, cols, d_R2, d_t2, rows2, fx,
                       q5_mode == 0 ? fx : fy, cx, cy, reinterpret_cast<double2*>(d_flow), d_best_row);
    RSDSFM_HIP_CHECK(c, hipGetLastError());
    return RSDSFM_OK;
}

}  // namespace rsdsfm
