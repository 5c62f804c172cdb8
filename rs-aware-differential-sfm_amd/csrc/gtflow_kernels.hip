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
#include <float.h>
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

// ---------------------------------------------------------------------------------------------------
// The same argmin with interval pruning (the default): the exhaustive loop evaluates rows2 projections per pixel although the
// projected row y(i) moves by a fraction of a pixel per scanline, so |y(i) - i| is far above the minimum for all but a few dozen
// candidates.  Scanlines are grouped into blocks of kPB; pose_bounds_kernel stores, per block, centre and radius of the eight pose
// entries the row coordinate depends on (R[3..8], t[1], t[2]).  For a pixel, interval arithmetic over a block bounds camera-frame
// y and z, their quotient and the projected row, hence a LOWER bound of |y(i) - i| over the block; blocks whose bound exceeds
// the best value found so far are skipped, everything else is evaluated with the exhaustive kernel's own code.  Every bound is
// widened by 1e-13-relative slack (hundreds of ulps: it covers the rounding of the interval arithmetic itself and of the exact
// evaluation it stands for), blocks with a z interval containing 0 or any non-finite number are always evaluated, and the winner is the
// lexicographic minimum of (difference, scanline) over the evaluated candidates -- the exhaustive loop's "first strict minimum".
// Winners and flows are bit-identical to the exhaustive search (tests/test_gpu_rectify.py compares the two on adversarial
// pose tables); 1280x720 against 720 scanlines: 0.71 -> 0.14 ms.
// ---------------------------------------------------------------------------------------------------
constexpr int kPB = 32;      // scanlines per block
constexpr int kPBW = 16;     // doubles per block: centre[8], radius[8] of (R3, R4, R5, t1, R6, R7, R8, t2)
constexpr int kPBHead = 24;  // header doubles: max |R3..5|, max |t1|, max |R6..8|, max |t2|, 4 unused, then centre[8] / radius[8] over ALL scanlines

__global__ __launch_bounds__(256) void pose_bounds_kernel(const double* __restrict__ R2, const double* __restrict__ t2, int rows2,
                                                         double* __restrict__ out) {
    const int nb = (rows2 + kPB - 1) / kPB;
    double mx[4] = {0.0, 0.0, 0.0, 0.0};
    double glo[8], ghi[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) glo[e] = INFINITY, ghi[e] = -INFINITY;
    bool gbad = false;
    for (int b = threadIdx.x; b < nb; b += blockDim.x) {
        double lo[8], hi[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) lo[e] = INFINITY, hi[e] = -INFINITY;
        bool bad = false;
        const int i1 = (b * kPB + kPB < rows2) ? b * kPB + kPB : rows2;
        for (int i = b * kPB; i < i1; ++i) {
            const double* Ri = R2 + (int64_t)i * 9;
            const double* ti = t2 + (int64_t)i * 3;
            const double v[8] = {Ri[3], Ri[4], Ri[5], ti[1], Ri[6], Ri[7], Ri[8], ti[2]};
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                if (!(fabs(v[e]) <= DBL_MAX)) bad = true;  // NaN or infinity: the block is always evaluated
                lo[e] = fmin(lo[e], v[e]);
                hi[e] = fmax(hi[e], v[e]);
            }
        }
        double* o = out + kPBHead + (int64_t)b * kPBW;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const double m = 0.5 * lo[e] + 0.5 * hi[e];
            double rad = fmax(hi[e] - m, m - lo[e]);
            rad = rad * (1.0 + 1e-15) + DBL_MIN;  // [m - rad, m + rad] contains [lo, hi] whatever the rounding of m and of the differences
            o[e] = bad ? 0.0 : m;
            o[8 + e] = bad ? INFINITY : rad;
            const double a = fmax(fabs(lo[e]), fabs(hi[e]));
            const int g = e < 3 ? 0 : e == 3 ? 1 : e < 7 ? 2 : 3;
            if (!bad) mx[g] = fmax(mx[g], a);
            glo[e] = fmin(glo[e], lo[e]);
            ghi[e] = fmax(ghi[e], hi[e]);
        }
        gbad = gbad || bad;
    }
    __shared__ double s_mx[4][256];
    __shared__ double s_lo[8][256], s_hi[8][256];
    __shared__ int s_bad[256];
#pragma unroll
    for (int g = 0; g < 4; ++g) s_mx[g][threadIdx.x] = mx[g];
#pragma unroll
    for (int e = 0; e < 8; ++e) s_lo[e][threadIdx.x] = glo[e], s_hi[e][threadIdx.x] = ghi[e];
    s_bad[threadIdx.x] = gbad ? 1 : 0;
    __syncthreads();
    if (threadIdx.x < 4) {
        double m = 0.0;
        for (int j = 0; j < (int)blockDim.x; ++j) m = fmax(m, s_mx[threadIdx.x][j]);
        out[threadIdx.x] = m;
    }
    if (threadIdx.x >= 8 && threadIdx.x < 16) {  // the same centre / radius form over ALL scanlines: the cheap first test of every block
        const int e = threadIdx.x - 8;
        double lo = INFINITY, hi = -INFINITY;
        int bad = 0;
        for (int j = 0; j < (int)blockDim.x; ++j) lo = fmin(lo, s_lo[e][j]), hi = fmax(hi, s_hi[e][j]), bad |= s_bad[j];
        const double m = 0.5 * lo + 0.5 * hi;
        double rad = fmax(hi - m, m - lo);
        rad = rad * (1.0 + 1e-15) + DBL_MIN;
        out[8 + e] = bad ? 0.0 : m;
        out[16 + e] = bad ? INFINITY : rad;
    }
}

// evaluates the scanlines [i0, i1) exactly as the exhaustive loop does and keeps the lexicographic minimum of (difference, scanline)
__device__ __forceinline__ void eval_scanlines(const double* __restrict__ R2, const double* __restrict__ t2, int i0, int i1, double X,
                                               double Y, double Z, double fyp, double cy, double& min_diff, int& best_row) {
#pragma unroll 4
    for (int i = i0; i < i1; ++i) {
        const double* Ri = R2 + (int64_t)i * 9;  // uniform address: scalar loads
        const double* ti = t2 + (int64_t)i * 3;
        const double yc = cam_row(Ri[3], Ri[4], Ri[5], ti[1], X, Y, Z);
        const double zc = cam_row(Ri[6], Ri[7], Ri[8], ti[2], X, Y, Z);
        const double py = to_plane(yc, zc, fyp, cy);
        const double diff = fabs(py - (double)i);
        if (diff < min_diff || (diff == min_diff && i < best_row)) {
            min_diff = diff;
            best_row = i;
        }
    }
}

// [plo, phi] contains the projected row y(i) of the world point for every scanline whose eight pose entries lie in centre +- radius
// (o[0..7] / o[8..15]); left at (-inf, +inf) when the camera-frame z interval contains 0 or anything is not finite
__device__ __forceinline__ void row_interval(const double* __restrict__ o, double X, double Y, double Z, double aX, double aY, double aZ,
                                             double slack_y, double slack_z, double fyp, double cy, double& plo, double& phi) {
    const double yc = ((o[0] * X + o[1] * Y) + o[2] * Z) + o[3];
    const double yr = ((o[8] * aX + o[9] * aY) + o[10] * aZ) + o[11] + slack_y;
    const double zc = ((o[4] * X + o[5] * Y) + o[6] * Z) + o[7];
    const double zr = ((o[12] * aX + o[13] * aY) + o[14] * aZ) + o[15] + slack_z;
    double ylo = yc - yr, yhi = yc + yr, zlo = zc - zr, zhi = zc + zr;
    if (!(zlo > 0.0 || zhi < 0.0)) return;  // (also taken for NaN bounds)
    if (zhi < 0.0) {                         // y / z = (-y) / (-z)
        const double t0 = -yhi, t1 = -ylo, t2z = -zhi, t3 = -zlo;
        ylo = t0, yhi = t1, zlo = t2z, zhi = t3;
    }
    const double qlo = ylo / (ylo >= 0.0 ? zhi : zlo);
    const double qhi = yhi / (yhi >= 0.0 ? zlo : zhi);
    const double pa = qlo * fyp + cy, pb = qhi * fyp + cy;
    const double lo = fmin(pa, pb), hi = fmax(pa, pb);
    const double e = 1e-13 * (fabs(lo) + fabs(hi) + 2.0 * fabs(cy) + 1.0);
    if (lo - e <= hi + e) {  // (false for NaN: the interval stays unbounded)
        plo = lo - e;
        phi = hi + e;
    }
}

__global__ __launch_bounds__(kGF) void true_flow_pruned_kernel(const double* __restrict__ wx, const double* __restrict__ wy,
                                                              const double* __restrict__ wz, int rows, int cols,
                                                              const double* __restrict__ R2, const double* __restrict__ t2, int rows2,
                                                              const double* __restrict__ bounds, double fx, double fyp, double cx,
                                                              double cy, double2* __restrict__ flow, int* __restrict__ best_row_out) {
    const int tiles_u = (cols + 15) / 16;
    const int64_t ntiles = (int64_t)tiles_u * ((rows + 15) / 16);
    const int nb = (rows2 + kPB - 1) / kPB;
    const double rmax_y = bounds[0], tmax_y = bounds[1], rmax_z = bounds[2], tmax_z = bounds[3];
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int v0 = (int)(tile / tiles_u) * 16;
        const int v = v0 + (threadIdx.x & 15), u = (int)(tile % tiles_u) * 16 + (threadIdx.x >> 4);
        const bool inside = v < rows && u < cols;
        const int64_t p = (int64_t)v * cols + u;
        const int64_t cm = (int64_t)u * rows + v;
        double X = 0.0, Y = 0.0, Z = 0.0;
        if (inside) X = wx[cm], Y = wy[cm], Z = wz[cm];
        const bool valid = inside && sqrt(X * X + Y * Y + Z * Z) != 0;
        double min_diff = INFINITY;
        int best_row = 0;
        // the block the tile's own rows fall into first: on real data it holds the winner, and its minimum prunes nearly all the others
        int b0 = (v0 + 8 < rows2 ? v0 + 8 : rows2 - 1) / kPB;
        if (valid) eval_scanlines(R2, t2, b0 * kPB, (b0 * kPB + kPB < rows2) ? b0 * kPB + kPB : rows2, X, Y, Z, fyp, cy, min_diff, best_row);
        const double aX = fabs(X), aY = fabs(Y), aZ = fabs(Z);
        const double slack_y = 1e-13 * (rmax_y * (aX + aY + aZ) + tmax_y) + DBL_MIN;
        const double slack_z = 1e-13 * (rmax_z * (aX + aY + aZ) + tmax_z) + DBL_MIN;
        // the projected row over ALL scanlines: on a smooth table it moves by a few pixels, and every block farther away than that from it
        // is discarded with two subtractions, without looking at its own bounds
        double gplo = -INFINITY, gphi = INFINITY;
        row_interval(bounds + 8, X, Y, Z, aX, aY, aZ, slack_y, slack_z, fyp, cy, gplo, gphi);
        for (int b = 0; b < nb; ++b) {
            if (b == b0) continue;
            const double i0 = (double)(b * kPB), i1 = (double)((b * kPB + kPB < rows2 ? b * kPB + kPB : rows2) - 1);
            const double bar = min_diff + 1e-13 * min_diff;
            bool need = !(fmax(i0 - gphi, gplo - i1) > bar);  // NaN-safe: anything unordered goes on
            if (need) {
                double plo = -INFINITY, phi = INFINITY;
                row_interval(bounds + kPBHead + (int64_t)b * kPBW, X, Y, Z, aX, aY, aZ, slack_y, slack_z, fyp, cy, plo, phi);  // uniform address: scalar loads
                need = !(fmax(i0 - phi, plo - i1) > bar);  // fmax(...) <= |y(i) - i| for every scanline of the block
            }
            if (valid && need)
                eval_scanlines(R2, t2, b * kPB, (b * kPB + kPB < rows2) ? b * kPB + kPB : rows2, X, Y, Z, fyp, cy, min_diff, best_row);
        }
        if (!inside) continue;
        double f2x = (double)u, f2y = (double)v;
        int out_row = -1;
        if (valid) {
            out_row = best_row;
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
        if (best_row_out) best_row_out[p] = out_row;
    }
}

int true_flow_launch(Ctx* c, const double* d_wx, const double* d_wy, const double* d_wz, int rows, int cols, const double* d_R2,
                     const double* d_t2, int rows2, double fx, double fy, double cx, double cy, int q5_mode, double* d_flow,
                     int* d_best_row) {
    int64_t blocks = (int64_t)((cols + 15) / 16) * ((rows + 15) / 16);
    if (blocks < 1) blocks = 1;
    if (blocks > 65536) blocks = 65536;
    const double fyp = q5_mode == 0 ? fx : fy;
    if (c->true_flow_exhaustive == 1 || (c->true_flow_exhaustive == 0 && rows2 < 3 * kPB)) {  // few scanlines: the bounds would cost more than they save
        hipLaunchKernelGGL(true_flow_kernel, dim3((int)blocks), dim3(kGF), 0, c->stream, d_wx, d_wy, d_wz, rows, cols, d_R2, d_t2, rows2, fx,
                           fyp, cx, cy, reinterpret_cast<double2*>(d_flow), d_best_row);
        RSDSFM_HIP_CHECK(c, hipGetLastError());
        return RSDSFM_OK;
    }
    const int nb = (rows2 + kPB - 1) / kPB;
    int rc = ensure_ws(c, sizeof(double) * ((size_t)nb * kPBW + kPBHead) + 256);
    if (rc != RSDSFM_OK) return rc;
    double* d_bounds = static_cast<double*>(c->d_ws);
    hipLaunchKernelGGL(pose_bounds_kernel, dim3(1), dim3(256), 0, c->stream, d_R2, d_t2, rows2, d_bounds);
    RSDSFM_HIP_CHECK(c, hipGetLastError());
    hipLaunchKernelGGL(true_flow_pruned_kernel, dim3((int)blocks), dim3(kGF), 0, c->stream, d_wx, d_wy, d_wz, rows, cols, d_R2, d_t2, rows2,
                       d_bounds, fx, fyp, cx, cy, reinterpret_cast<double2*>(d_flow), d_best_row);
    RSDSFM_HIP_CHECK(c, hipGetLastError());
    return RSDSFM_OK;
}

}  // namespace rsdsfm
