// glue_kernels.hip -- RS scale factors, per-scanline pose table.
//   minimal::getAlpha / getAlphaK   reference minimal.cc:179-197
//   RsFrame::setRelativePose        reference rsframe.cc:771-800
#include "device_math.hpp"
#include "rsdsfm_internal.hpp"

namespace rsdsfm {

__global__ __launch_bounds__(256) void alpha_kernel(const double2* __restrict__ flow_px, int64_t n, double h,
                                                    double gamma, double* __restrict__ alpha) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        alpha[i] = 1 + gamma * flow_px[i].y / h;
}

__global__ __launch_bounds__(256) void alpha_k_kernel(const double2* __restrict__ q_px, const double2* __restrict__ flow_px,
                                                      int64_t n, double h, double gamma, double* __restrict__ alpha_k) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        double y = q_px[i].y, fy = flow_px[i].y;
        double part1 = gamma * y / h;
        double part2 = 1.0 + gamma * (y + fy) / h;
        alpha_k[i] = 0.5 * (part2 * part2 - part1 * part1);
    }
}

// one lane per scanline: R_i = I + beta_1(i) skew(w), t_i = beta_1(i) v  (scanline 0 = identity)
__global__ __launch_bounds__(256) void pose_table_kernel(Pose pose, double gamma, int rows, double* __restrict__ R,
                                                         double* __restrict__ t) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows) return;
    double beta_1 = 0.0;
    if (i > 0)
        beta_1 = (gamma * i / rows + 0.5 * pose.k * (gamma * gamma * i * i) / ((double)rows * rows)) * (2.0 / (2.0 + pose.k));
    double* Ri = R + (int64_t)i * 9;
    Ri[0] = 1.0 + beta_1 * 0.0;
    Ri[1] = 0.0 + beta_1 * -pose.w[2];
    Ri[2] = 0.0 + beta_1 * pose.w[1];
    Ri[3] = 0.0 + beta_1 * pose.w[2];
    Ri[4] = 1.0 + beta_1 * 0.0;
    Ri[5] = 0.0 + beta_1 * -pose.w[0];
    Ri[6] = 0.0 + beta_1 * -pose.w[1];
    Ri[7] = 0.0 + beta_1 * pose.w[0];
    Ri[8] = 1.0 + beta_1 * 0.0;
    t[(int64_t)i * 3 + 0] = 0.0 + beta_1 * pose.v[0];
    t[(int64_t)i * 3 + 1] = 0.0 + beta_1 * pose.v[1];
    t[(int64_t)i * 3 + 2] = 0.0 + beta_1 * pose.v[2];
}

static inline int stream_grid(int64_t n) {
    int64_t b = (n + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}

int alpha_launch(Ctx* c, const double* flow_px, int64_t n, double h, double gamma, double* alpha) {
    if (n == 0) return RSDSFM_OK;
    hipLaunchKernelGGL(alpha_kernel, dim3(stream_grid(n)), dim3(256), 0, c->stream,
                       reinterpret_cast<const double2*>(flow_px), n, h, gamma, alpha);
    RSDSFM_HIP_CHECK(c, hipGetLastError());
    return RSDSFM_OK;
}

int alpha_k_launch(Ctx* c, const double* q_px, const double* flow_px, int64_t n, double h, double gamma, double* alpha_k) {
    if (n == 0) return RSDSFM_OK;
    hipLaunchKernelGGL(alpha_k_kernel, dim3(stream_grid(n)), dim3(256), 0, c->stream,
                       reinterpret_cast<const double2*>(q_px), reinterpret_cast<const double2*>(flow_px), n, h, gamma, alpha_k);
    RSDSFM_HIP_CHECK(c, hipGetLastError());
    return RSDSFM_OK;
}

int pose_table_launch(Ctx* c, const Pose& pose, double gamma, int rows, double* R, double* t) {
    if (rows <= 0) return RSDSFM_OK;
    hipLaunchKernelGGL(pose_table_kernel, dim3((rows + 255) / 256), dim3(256), 0, c->stream, pose, gamma, rows, R, t);
    RSDSFM_HIP_CHECK(c, hipGetLastError());
    return RSDSFM_OK;
}

}  // namespace rsdsfm
