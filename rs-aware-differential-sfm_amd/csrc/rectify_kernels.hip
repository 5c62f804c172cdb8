// rectify_kernels.hip -- SURVEY section 8(f-1): consumers of the solve's output on MI355X (gfx950).
//
//   back_project_claim_kernel / back_project_write_kernel
//       RsFrame::backProject / backProjectGs (reference rsframe.cc:803-878 with planeToSpace :644-664,
//       cameraToWorldFrame :712-736, worldToCameraFrame :687-709, spaceToPlane :628-641): forward splat of the BGR
//       rolling-shutter image into the global-shutter image through the depth map and the per-scanline pose table.
//       The reference's sequential double loop lets the LAST writer (largest row-major scan index) win; here every
//       source pixel claims its target with an integer atomicMax on the scan index (deterministic), then one pass
//       gathers the winners.  Byte / index work, HBM-bound: 3 B image + 8 B depth read, 3 B + 12 B (world point as
//       float3) written per pixel = 26 B/pixel algorithmic.
//       The depth map arrives column-major (Eigen MatrixXd, what depth_write_kernel produces) while the image is
//       row-major: each workgroup stages a 32 x 64 (x, y) tile of the depth map through LDS (coalesced along y),
//       then walks the tile along x (coalesced image reads / float3 writes).
//   interpolate_cracky_kernel
//       Camera::interpolateCrackyImage (camera.cc:694-774): 4-neighbour fill of black pixels.
//   preview_minmax_kernel / preview_claim_kernel / preview_write_kernel
//       the 8-bit depth image of evaluateSingleRun (main.cc:480-509).
//
// Arithmetic mirrors oracle/rsdsfm_oracle.c (rso_back_project, rso_interpolate_cracky, rso_depth_preview) operation
// for operation; all outputs are integers / bytes (bit-exact) except the float3 world points (bit-exact as well: the
// per-pixel chain is identical and compiled with -ffp-contract=off).
#include <math.h>

#include <algorithm>

#include "device_math.hpp"
#include "rsdsfm_internal.hpp"

namespace rsdsfm {

namespace {

constexpr int kTX = 32;  // tile width  (image columns)
constexpr int kTY = 64;  // tile height (image rows = scanlines)
constexpr int kBP = 256;

// double -> int like the reference's int(x) on x86-64 (cvttsd2si): non-finite / out of range -> INT_MIN
__device__ __forceinline__ int trunc_int(double x) {
    if (!(x > -2147483649.0 && x < 2147483648.0)) return INT32_MIN;
    return (int)x;
}

__device__ __forceinline__ bool is_black(unsigned b, unsigned g, unsigned r) {  // cv::norm(Vec3b) <= 15
    const double n = sqrt((double)b * (double)b + (double)g * (double)g + (double)r * (double)r);
    return n <= 15.0;
}

__device__ __forceinline__ unsigned char saturate_u8(double v) {  // cvRound (nearest even) + clamp
    const long long r = __double2ll_rn(v);
    return (unsigned char)(r < 0 ? 0 : (r > 255 ? 255 : r));
}

}  // namespace

// grid: (ceil(cols / kTX), ceil(rows / kTY)); owner: rows*cols int32 (pre-set to -1), row-major
__global__ __launch_bounds__(kBP) void back_project_claim_kernel(const unsigned char* __restrict__ img,
                                                                const double* __restrict__ depth_cm,
                                                                const double* __restrict__ R, const double* __restrict__ t,
                                                                double fx, double fy, double cx, double cy, double fyp, int rows,
                                                                int cols, int mode, int* __restrict__ owner,
                                                                float* __restrict__ c3d) {
    __shared__ double s_z[kTX][kTY + 1];
    const int x0 = blockIdx.x * kTX, y0 = blockIdx.y * kTY;
    const int tid = threadIdx.x;
    // stage the depth tile: lanes run along y (contiguous in the column-major map)
    {
        const int ly = tid & (kTY - 1);
        for (int lx = tid / kTY; lx < kTX; lx += kBP / kTY) {
            const int x = x0 + lx, y = y0 + ly;
            s_z[lx][ly] = (x < cols && y < rows) ? depth_cm[(int64_t)x * rows + y] : 0.0;
        }
    }
    __syncthreads();
    const int lx = tid & (kTX - 1);
    const int x = x0 + lx;
    if (x >= cols) return;
    const double nx = ((double)x - cx) * 1.0 / fx;
    for (int ly = tid / kTX; ly < kTY; ly += kBP / kTX) {
        const int y = y0 + ly;
        if (y >= rows) break;
        const int64_t s = (int64_t)y * cols + x;
        const unsigned b = img[3 * s], g = img[3 * s + 1], r = img[3 * s + 2];
        if (b == 1 && g == 1 && r == 1) continue;  // marker colour: pixel not processed (rsframe.cc:816)
        const double* Rs = mode == 0 ? R + (int64_t)y * 9 : R;
        const double* ts = mode == 0 ? t + (int64_t)y * 3 : t;
        const double ny = ((double)y - cy) * 1.0 / fy;
        const double z = s_z[lx][ly];
        const double pc0 = z * nx, pc1 = z * ny, pc2 = z * 1.0;
        double pw[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const double rt0 = Rs[i], rt1 = Rs[3 + i], rt2 = Rs[6 + i];  // row i of R^T
            const double ti = ((-rt0) * ts[0] + (-rt1) * ts[1]) + (-rt2) * ts[2];
            pw[i] = ((rt0 * pc0 + rt1 * pc1) + rt2 * pc2) + ti * 1.0;
        }
        double pg[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) pg[i] = ((R[i * 3] * pw[0] + R[i * 3 + 1] * pw[1]) + R[i * 3 + 2] * pw[2]) + t[i] * 1.0;
        const double gx = pg[0] / pg[2] * fx + cx;
        const double gy = pg[1] / pg[2] * fyp + cy;
        if (c3d) {
            c3d[3 * s] = (float)pw[0];
            c3d[3 * s + 1] = (float)pw[1];
            c3d[3 * s + 2] = (float)pw[2];
        }
        const int ix = trunc_int(gx + 0.5), iy = trunc_int(gy + 0.5);
        if (ix >= 0 && ix < cols && iy >= 0 && iy < rows) atomicMax(&owner[(int64_t)iy * cols + ix], (int)s);
    }
}

// 4 target pixels (12 bytes) per thread
__global__ __launch_bounds__(kBP) void back_project_write_kernel(const unsigned char* __restrict__ img,
                                                                const int* __restrict__ owner, int64_t npix,
                                                                unsigned char* __restrict__ gs) {
    const int64_t stride = (int64_t)gridDim.x * kBP * 4;
    for (int64_t p0 = ((int64_t)blockIdx.x * kBP + threadIdx.x) * 4; p0 < npix; p0 += stride) {
        unsigned char v[12];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t p = p0 + j;
            const int o = p < npix ? owner[p] : -1;
            v[3 * j] = o >= 0 ? img[3 * (int64_t)o] : 0;
            v[3 * j + 1] = o >= 0 ? img[3 * (int64_t)o + 1] : 0;
            v[3 * j + 2] = o >= 0 ? img[3 * (int64_t)o + 2] : 0;
        }
        if (p0 + 4 <= npix) {
            unsigned* dst = reinterpret_cast<unsigned*>(gs + 3 * p0);  // 12 p0 bytes: 4-byte aligned
            dst[0] = v[0] | (v[1] << 8) | (v[2] << 16) | ((unsigned)v[3] << 24);
            dst[1] = v[4] | (v[5] << 8) | (v[6] << 16) | ((unsigned)v[7] << 24);
            dst[2] = v[8] | (v[9] << 8) | (v[10] << 16) | ((unsigned)v[11] << 24);
        } else {
            for (int64_t p = p0; p < npix; ++p)
                for (int c = 0; c < 3; ++c) gs[3 * p + c] = v[3 * (p - p0) + c];
        }
    }
}

__global__ __launch_bounds__(kBP) void interpolate_cracky_kernel(const unsigned char* __restrict__ in, int rows, int cols,
                                                                int offset, unsigned char* __restrict__ out) {
    const int64_t npix = (int64_t)rows * cols;
    const int64_t stride = (int64_t)gridDim.x * kBP;
    for (int64_t p = (int64_t)blockIdx.x * kBP + threadIdx.x; p < npix; p += stride) {
        const int row = (int)(p / cols), col = (int)(p - (int64_t)row * cols);
        unsigned b = in[3 * p], g = in[3 * p + 1], r = in[3 * p + 2];
        if (row >= offset && row < rows - offset && col >= offset && col < cols - offset && is_black(b, g, r)) {
            const int64_t nb[4] = {p - (int64_t)offset * cols, p + (int64_t)offset * cols, p - offset, p + offset};
            double s0 = 0, s1 = 0, s2 = 0;
            unsigned count = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const unsigned nb0 = in[3 * nb[j]], nb1 = in[3 * nb[j] + 1], nb2 = in[3 * nb[j] + 2];
                if (!is_black(nb0, nb1, nb2)) {
                    s0 += (double)nb0;
                    s1 += (double)nb1;
                    s2 += (double)nb2;
                    count++;
                }
            }
            if (count > 0) {
                const double inv = 1 / (double)count;
                b = saturate_u8(inv * s0);
                g = saturate_u8(inv * s1);
                r = saturate_u8(inv * s2);
            }
        }
        out[3 * p] = (unsigned char)b;
        out[3 * p + 1] = (unsigned char)g;
        out[3 * p + 2] = (unsigned char)r;
    }
}

// ---- 8-bit depth preview (main.cc:480-509) ----
// partials[2 * block] = min z, [2 * block + 1] = max z (start values +inf / 0 as in the reference)
__global__ __launch_bounds__(kBP) void preview_minmax_kernel(const double* __restrict__ inl, int64_t m, double* __restrict__ partials) {
    __shared__ double s_min[kBP / 64], s_max[kBP / 64];
    double lo = INFINITY, hi = 0.0;
    const int64_t stride = (int64_t)gridDim.x * kBP;
    for (int64_t i = (int64_t)blockIdx.x * kBP + threadIdx.x; i < m; i += stride) {
        const double z = inl[3 * i + 2];
        if (z < lo) lo = z;
        if (z > hi) hi = z;
    }
    for (int off = 32; off >= 1; off >>= 1) {
        const double ol = __shfl_xor(lo, off, 64), oh = __shfl_xor(hi, off, 64);
        if (ol < lo) lo = ol;
        if (oh > hi) hi = oh;
    }
    if ((threadIdx.x & 63) == 0) {
        s_min[threadIdx.x >> 6] = lo;
        s_max[threadIdx.x >> 6] = hi;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w2 = 1; w2 < kBP / 64; ++w2) {
            if (s_min[w2] < lo) lo = s_min[w2];
            if (s_max[w2] > hi) hi = s_max[w2];
        }
        partials[2 * blockIdx.x] = lo;
        partials[2 * blockIdx.x + 1] = hi;
    }
}

// single workgroup: header[0] = z_min, header[1] = multiplier = 244 / (z_max - z_min)
__global__ __launch_bounds__(kBP) void preview_header_kernel(const double* __restrict__ partials, int nblocks, double* __restrict__ header) {
    __shared__ double s_min[kBP], s_max[kBP];
    double lo = INFINITY, hi = 0.0;
    for (int b = threadIdx.x; b < nblocks; b += kBP) {
        if (partials[2 * b] < lo) lo = partials[2 * b];
        if (partials[2 * b + 1] > hi) hi = partials[2 * b + 1];
    }
    s_min[threadIdx.x] = lo;
    s_max[threadIdx.x] = hi;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < kBP; ++i) {
            if (s_min[i] < lo) lo = s_min[i];
            if (s_max[i] > hi) hi = s_max[i];
        }
        header[0] = lo;
        header[1] = 244.0 / (hi - lo);
    }
}

// owner: rows*cols int32 row-major, pre-set to -1; the highest inlier index wins (the reference's last writer)
__global__ __launch_bounds__(kBP) void preview_claim_kernel(const double* __restrict__ inl, int64_t m, double fx, double fy, double cx,
                                                           double cy, int rows, int cols, int* __restrict__ owner) {
    const int64_t stride = (int64_t)gridDim.x * kBP;
    for (int64_t i = (int64_t)blockIdx.x * kBP + threadIdx.x; i < m; i += stride) {
        const int x = (int)(fx * inl[3 * i] + cx + 0.5);
        const int y = (int)(fy * inl[3 * i + 1] + cy + 0.5);
        if (x >= 0 && x < cols && y >= 0 && y < rows) atomicMax(&owner[(int64_t)y * cols + x], (int)i);
    }
}

__global__ __launch_bounds__(kBP) void preview_write_kernel(const double* __restrict__ inl, const int* __restrict__ owner,
                                                           const double* __restrict__ header, int64_t npix,
                                                           unsigned char* __restrict__ out) {
    const double z_min = header[0], mult = header[1];
    const int64_t stride = (int64_t)gridDim.x * kBP;
    for (int64_t p = (int64_t)blockIdx.x * kBP + threadIdx.x; p < npix; p += stride) {
        const int o = owner[p];
        unsigned char v = 0;
        if (o >= 0) {
            int zi = trunc_int((inl[3 * (int64_t)o + 2] - z_min) * mult);
            if (zi == INT32_MIN) zi = 0;
            v = (unsigned char)(10 + zi);  // int -> uchar: modulo 256, like the reference's assignment
        }
        out[p] = v;
    }
}

// ---------------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------------
static inline int stream_grid(int64_t n, int per_thread = 1) {
    int64_t b = (n + (int64_t)kBP * per_thread - 1) / ((int64_t)kBP * per_thread);
    return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

int back_project_launch(Ctx* c, const unsigned char* d_img, const double* d_depth_cm, const double* d_R, const double* d_t, double fx,
                        double fy, double cx, double cy, int rows, int cols, int mode, int q5_mode, unsigned char* d_gs, float* d_c3d,
                        int* d_owner) {
    const int64_t npix = (int64_t)rows * cols;
    RSDSFM_HIP_CHECK(c, hipMemsetAsync(d_owner, 0xFF, sizeof(int) * (size_t)npix, c->stream));
    if (d_c3d) RSDSFM_HIP_CHECK(c, hipMemsetAsync(d_c3d, 0, sizeof(float) * 3 * (size_t)npix, c->stream));
    dim3 grid((cols + kTX - 1) / kTX, (rows + kTY - 1) / kTY);
    hipLaunchKernelGGL(back_project_claim_kernel, grid, dim3(kBP), 0, c->stream, d_img, d_depth_cm, d_R, d_t, fx, fy, cx, cy,
                       q5_mode == 0 ? fx : fy, rows, cols, mode, d_owner, d_c3d);
    RSDSFM_HIP_CHECK(c, hipGetLastError());
    hipLaunchKernelGGL(back_project_write_kernel, dim3(stream_grid(npix, 4)), dim3(kBP), 0, c->stream, d_img, d_owner, npix, d_gs);
    RSDSFM_HIP_CHECK(c, hipGetLastError());
    return RSDSFM_OK;
}

int interpolate_cracky_launch(Ctx* c, const unsigned char* d_in, int rows, int cols, int offset, unsigned char* d_out) {
    hipLaunchKernelGGL(interpolate_cracky_kernel, dim3(stream_grid((int64_t)rows * cols)), dim3(kBP), 0, c->stream, d_in, rows, cols,
                       offset, d_out);
    RSDSFM_HIP_CHECK(c, hipGetLastError());
    return RSDSFM_OK;
}

// d_partials: >= 2 * 1024 doubles; d_header: 2 doubles; d_owner: rows*cols int32
int depth_preview_launch(Ctx* c, const double* d_inl, int64_t m, double fx, double fy, double cx, double cy, int rows, int cols,
                         unsigned char* d_out, double* d_partials, double* d_header, int* d_owner) {
    const int64_t npix = (int64_t)rows * cols;
    const int zb = (int)std::min<int64_t>(1024, std::max<int64_t>(1, (m + kBP - 1) / kBP));
    hipLaunchKernelGGL(preview_minmax_kernel, dim3(zb), dim3(kBP), 0, c->stream, d_inl, m, d_partials);
    RSDSFM_HIP_CHECK(c, hipGetLastError());
    hipLaunchKernelGGL(preview_header_kernel, dim3(1), dim3(kBP), 0, c->stream, d_partials, zb, d_header);
    RSDSFM_HIP_CHECK(c, hipGetLastError());
    RSDSFM_HIP_CHECK(c, hipMemsetAsync(d_owner, 0xFF, sizeof(int) * (size_t)npix, c->stream));
    if (m > 0) {
        hipLaunchKernelGGL(preview_claim_kernel, dim3(stream_grid(m)), dim3(kBP), 0, c->stream, d_inl, m, fx, fy, cx, cy, rows, cols,
                           d_owner);
        RSDSFM_HIP_CHECK(c, hipGetLastError());
    }
    hipLaunchKernelGGL(preview_write_kernel, dim3(stream_grid(npix)), dim3(kBP), 0, c->stream, d_inl, d_owner, d_header, npix, d_out);
    RSDSFM_HIP_CHECK(c, hipGetLastError());
    return RSDSFM_OK;
}

}  // namespace rsdsfm
