// metrics_kernels.hip -- SURVEY section 8(f-4): accuracy metrics of the estimated structure on MI355X (gfx950).
//
// Replaces the shared passes of Camera::meanReprojectionError (reference camera.cc:594-691) and
// Camera::createErrorImage (camera.cc:503-591):
//   reproj_scale_kernel   per pixel: ground-truth world point (planeToSpace + cameraToWorldFrame with the absolute
//                         pose of the pixel's scanline, rounded to float like the reference's cv::Vec3f), the three
//                         float ratios estimate / truth, outlier rule |ratio| > 10 -> partials {sum, inliers, outliers}
//   reproj_error_kernel   every workgroup first reduces those partials itself (fixed order: all obtain the same scale = sum / inliers,
//                         no single-workgroup launch in between); per pixel: || estimate / scale - truth ||, summed where finite
//                         and < 50, optional 8-bit image -> partials {sum_error, error_inliers}
//   the host adds the (<= 512) error partials in workgroup order after ONE device-to-host copy (round 3 ran four launches -- two
//   streaming passes over 2048 workgroups, each followed by a single-workgroup reduction -- and then copied a 40-byte header)
// Pixels are walked in 16 x 16 tiles with the lanes along y (depth maps are column-major); the estimated points are
// row-major float3.  HBM-bound streaming passes: 28 B/pixel read per pass (12 B point + 2 x 8 B depth), 1 B written.
// Per-pixel arithmetic mirrors oracle/rsdsfm_oracle.c (rso_reprojection_error) exactly; the two global sums differ
// from the reference's sequential order only in summation order (counts are exact).
#include <math.h>

#include <algorithm>

#include "device_math.hpp"
#include "rsdsfm_internal.hpp"

namespace rsdsfm {

namespace {
constexpr int kMB = 256;
constexpr int kMetricsMaxBlocks = 1024;  // four workgroups per CU (latency-bound passes: 16 waves per CU); every workgroup of the second pass reads all partial rows of the first (24 KB)

__device__ __forceinline__ int trunc_int_m(double x) {
    if (!(x > -2147483649.0 && x < 2147483648.0)) return INT32_MIN;
    return (int)x;
}

// ground-truth world point of pixel (x, y), float like the reference's cv::Vec3f
__device__ __forceinline__ void true_point(const double* __restrict__ gt_depth, const double* __restrict__ est_depth,
                                           const double* __restrict__ R, const double* __restrict__ t, double fx, double fy, double cx,
                                           double cy, int rows, int x, int y, float (&out)[3]) {
    double z = gt_depth[(int64_t)x * rows + y];
    if (z == 0) z = est_depth[(int64_t)x * rows + y];  // planeToSpace's default-argument fallback (rsframe.cc:657)
    const double nx = ((double)x - cx) * 1.0 / fx, ny = ((double)y - cy) * 1.0 / fy;
    const double pc0 = z * nx, pc1 = z * ny, pc2 = z * 1.0;
    const double* Rs = R + (int64_t)y * 9;
    const double* ts = t + (int64_t)y * 3;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const double rt0 = Rs[i], rt1 = Rs[3 + i], rt2 = Rs[6 + i];
        const double ti = ((-rt0) * ts[0] + (-rt1) * ts[1]) + (-rt2) * ts[2];
        out[i] = (float)(((rt0 * pc0 + rt1 * pc1) + rt2 * pc2) + ti * 1.0);
    }
}

__device__ __forceinline__ bool tile_pixel(int64_t tile, int tiles_x, int rows, int cols, int& x, int& y) {
    y = (int)(tile / tiles_x) * 16 + (threadIdx.x & 15);
    x = (int)(tile % tiles_x) * 16 + (threadIdx.x >> 4);
    return y < rows && x < cols;
}

// block reduction of {double sum, two integer counts} in fixed order -> partials[3 * block]
__device__ __forceinline__ void block_reduce3(double s, long long a, long long b, double* __restrict__ partials) {
    __shared__ double s_s[kMB / 64];
    __shared__ long long s_a[kMB / 64], s_b[kMB / 64];
    const double ws = wave_sum(s);
    for (int off = 32; off >= 1; off >>= 1) {
        a += __shfl_xor(a, off, 64);
        b += __shfl_xor(b, off, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        s_s[threadIdx.x >> 6] = ws;
        s_a[threadIdx.x >> 6] = a;
        s_b[threadIdx.x >> 6] = b;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double ts = s_s[0];
        long long ta = s_a[0], tb = s_b[0];
        for (int w2 = 1; w2 < kMB / 64; ++w2) {
            ts += s_s[w2];
            ta += s_a[w2];
            tb += s_b[w2];
        }
        partials[3 * blockIdx.x] = ts;
        partials[3 * blockIdx.x + 1] = (double)ta;  // exact: counts < 2^53
        partials[3 * blockIdx.x + 2] = (double)tb;
    }
}

}  // namespace

__global__ __launch_bounds__(kMB) void reproj_scale_kernel(const float* __restrict__ est, const double* __restrict__ gt_depth,
                                                          const double* __restrict__ est_depth, const double* __restrict__ R,
                                                          const double* __restrict__ t, double fx, double fy, double cx, double cy,
                                                          int rows, int cols, double* __restrict__ partials) {
    const int tiles_x = (cols + 15) / 16;
    const int64_t ntiles = (int64_t)tiles_x * ((rows + 15) / 16);
    double sum = 0.0;
    long long inl = 0, outl = 0;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        int x, y;
        if (!tile_pixel(tile, tiles_x, rows, cols, x, y)) continue;
        float pt[3];
        true_point(gt_depth, est_depth, R, t, fx, fy, cx, cy, rows, x, y, pt);
        const float* pe = est + ((int64_t)y * cols + x) * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float ratio = pe[c] / pt[c];
            double sc = (double)ratio;
            if (fabsf(ratio) > 10) {
                sc = 0;
                outl++;
            }
            if (sc != 0 && sc == sc) {
                inl++;
                sum += sc;
            }
        }
    }
    block_reduce3(sum, inl, outl, partials);
}

// fixed-order reduction of partials[nblocks][3] by one workgroup: thread i adds rows i, i + 256, ...; waves by DPP; the four waves in order
__device__ __forceinline__ void reduce3(const double* __restrict__ partials, int nblocks, double (&r)[3]) {
    __shared__ double s_red[3][kMB / 64];
    double a = 0.0, b = 0.0, c2 = 0.0;
    for (int i = threadIdx.x; i < nblocks; i += kMB) {
        a += partials[3 * i];
        b += partials[3 * i + 1];
        c2 += partials[3 * i + 2];
    }
    a = wave_sum(a), b = wave_sum(b), c2 = wave_sum(c2);
    if ((threadIdx.x & 63) == 0) {
        s_red[0][threadIdx.x >> 6] = a;
        s_red[1][threadIdx.x >> 6] = b;
        s_red[2][threadIdx.x >> 6] = c2;
    }
    __syncthreads();
    for (int k = 0; k < 3; ++k) r[k] = ((s_red[k][0] + s_red[k][1]) + s_red[k][2]) + s_red[k][3];
    __syncthreads();
}

// header (written by workgroup 0): [0] scale, [1] scale inliers, [2] outliers
__global__ __launch_bounds__(kMB) void reproj_error_kernel(const float* __restrict__ est, const double* __restrict__ gt_depth,
                                                          const double* __restrict__ est_depth, const double* __restrict__ R,
                                                          const double* __restrict__ t, double fx, double fy, double cx, double cy,
                                                          int rows, int cols, const double* __restrict__ scale_partials, int nscale,
                                                          double* __restrict__ header, double max_norm, double* __restrict__ partials,
                                                          unsigned char* __restrict__ error_image) {
    double sc3[3];
    reduce3(scale_partials, nscale, sc3);
    const double scale = sc3[0] / sc3[1];  // sum / double(inliers) (camera.cc:640)
    if (blockIdx.x == 0 && threadIdx.x == 0) header[0] = scale, header[1] = sc3[1], header[2] = sc3[2];
    const int tiles_x = (cols + 15) / 16;
    const int64_t ntiles = (int64_t)tiles_x * ((rows + 15) / 16);
    double sum = 0.0;
    long long inl = 0;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        int x, y;
        if (!tile_pixel(tile, tiles_x, rows, cols, x, y)) continue;
        float pt[3];
        true_point(gt_depth, est_depth, R, t, fx, fy, cx, cy, rows, x, y, pt);
        const float* pe = est + ((int64_t)y * cols + x) * 3;
        double e[3], tr[3], d[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            e[c] = pe[c] / scale;
            tr[c] = pt[c];
            d[c] = e[c] - tr[c];
        }
        const double norm = sqrt((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]);
        if (e[0] == e[0] && e[1] == e[1] && e[2] == e[2] && tr[0] == tr[0] && tr[1] == tr[1] && tr[2] == tr[2] && norm < 50) {
            sum += norm;
            inl++;
        }
        if (error_image) {
            int v = trunc_int_m(norm * 255 / max_norm + 0.5);
            if (v == INT32_MIN) v = 0;
            error_image[(int64_t)y * cols + x] = (unsigned char)v;
        }
    }
    block_reduce3(sum, inl, 0, partials);
}

static inline int metrics_grid(int rows, int cols) {
    const int64_t tiles = (int64_t)((cols + 15) / 16) * ((rows + 15) / 16);
    return (int)std::min<int64_t>(kMetricsMaxBlocks, std::max<int64_t>(1, tiles));
}

int metrics_blocks_max() { return kMetricsMaxBlocks; }

// d_scale_partials, d_error_partials: >= 3 * metrics_blocks_max() doubles each; d_header: 8 doubles.  *error_rows = rows of d_error_partials
// the caller adds up (in row order)
int reprojection_error_launch(Ctx* c, const float* d_est, const double* d_gt_depth, const double* d_est_depth, const double* d_R,
                              const double* d_t, double fx, double fy, double cx, double cy, int rows, int cols, double max_norm,
                              unsigned char* d_error_image, double* d_scale_partials, double* d_header, double* d_error_partials, int* error_rows) {
    const int grid = metrics_grid(rows, cols);
    hipLaunchKernelGGL(reproj_scale_kernel, dim3(grid), dim3(kMB), 0, c->stream, d_est, d_gt_depth, d_est_depth, d_R, d_t, fx, fy, cx, cy,
                       rows, cols, d_scale_partials);
    RSDSFM_HIP_CHECK(c, hipGetLastError());
    hipLaunchKernelGGL(reproj_error_kernel, dim3(grid), dim3(kMB), 0, c->stream, d_est, d_gt_depth, d_est_depth, d_R, d_t, fx, fy, cx, cy,
                       rows, cols, d_scale_partials, grid, d_header, max_norm, d_error_partials, d_error_image);
    RSDSFM_HIP_CHECK(c, hipGetLastError());
    *error_rows = grid;
    return RSDSFM_OK;
}

}  // namespace rsdsfm
