// valu_rates.hip -- measures the issue rate of the fp64 VALU instructions the depth / RANSAC kernels are made of
// (v_fma_f64, v_mul_f64, v_add_f64, v_rcp_f64, IEEE division, sqrt) on the device it runs on, so that DESIGN.md can
// price those kernels against a MEASURED compute ceiling.  Standalone: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x)                                                                  \
    do {                                                                          \
        hipError_t e = (x);                                                       \
        if (e != hipSuccess) {                                                    \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e)); \
            exit(1);                                                              \
        }                                                                         \
    } while (0)

constexpr int kIter = 2048;
constexpr int kChains = 8;

template <int OP>
__global__ __launch_bounds__(256) void rate_kernel(double* out, double b, double c) {
    double x[kChains];
    for (int i = 0; i < kChains; ++i) x[i] = 1.0 + 1e-3 * (threadIdx.x + i);
    for (int it = 0; it < kIter; ++it) {
#pragma unroll
        for (int i = 0; i < kChains; ++i) {
            if (OP == 0) x[i] = __builtin_fma(x[i], b, c);
            if (OP == 1) x[i] = x[i] * b;
            if (OP == 2) x[i] = x[i] + c;
            if (OP == 3) x[i] = __builtin_amdgcn_rcp(x[i]);
            if (OP == 4) x[i] = c / x[i];
            if (OP == 5) x[i] = sqrt(x[i]);
            if (OP == 6) x[i] = fmax(x[i], c) ;
            if (OP == 7) x[i] = x[i] * b + c;  // mul + add (contract off)
            if (OP == 8) x[i] = __builtin_amdgcn_rsq(x[i]);
            if (OP == 9) x[i] = x[i] < c ? b : x[i] + c;  // v_cmp_f64 + v_add_f64 + 2 x v_cndmask_b32
        }
    }
    double s = 0;
    for (int i = 0; i < kChains; ++i) s += x[i];
    if (s == 12345.678) out[0] = s;
}

template <int OP>
__global__ __launch_bounds__(256) void rate_kernel_f32(float* out, float b, float c) {
    float x[kChains];
    for (int i = 0; i < kChains; ++i) x[i] = 1.0f + 1e-3f * (threadIdx.x + i);
    for (int it = 0; it < kIter; ++it) {
#pragma unroll
        for (int i = 0; i < kChains; ++i) x[i] = __builtin_fmaf(x[i], b, c);
    }
    float s = 0;
    for (int i = 0; i < kChains; ++i) s += x[i];
    if (s == 12345.678f) out[0] = s;
}

template <class F>
static double time_ms(F launch) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    launch();
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < 5; ++r) launch();
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms / 5;
}

int main() {
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    const double clk = p.clockRate * 1e3;  // Hz
    printf("device %s  CUs %d  clock %.0f MHz\n", p.name, cus, clk / 1e6);
    double* d;
    CHECK(hipMalloc(&d, 64));
    const int wg_per_cu[] = {4, 8};  // 256-thread workgroups per CU: 4 -> 4 waves / SIMD, 8 -> 8 waves / SIMD
    const char* names[] = {"v_fma_f64", "v_mul_f64", "v_add_f64", "v_rcp_f64", "IEEE div f64", "sqrt f64", "v_max_f64", "mul+add f64", "v_rsq_f64", "cmp+add+2cndmask"};
    const int per_op[] = {1, 1, 1, 1, 1, 1, 1, 2, 1, 4};
    for (int w = 0; w < 2; ++w) {
        const int grid = cus * wg_per_cu[w];
        printf("-- %d waves per SIMD\n", wg_per_cu[w]);
        for (int op = 0; op < 10; ++op) {
            double ms = 0;
            switch (op) {
                case 0: ms = time_ms([&] { rate_kernel<0><<<grid, 256>>>(d, 0.999, 1e-3); }); break;
                case 1: ms = time_ms([&] { rate_kernel<1><<<grid, 256>>>(d, 0.999999, 1e-3); }); break;
                case 2: ms = time_ms([&] { rate_kernel<2><<<grid, 256>>>(d, 0.999, 1e-3); }); break;
                case 3: ms = time_ms([&] { rate_kernel<3><<<grid, 256>>>(d, 0.999, 1e-3); }); break;
                case 4: ms = time_ms([&] { rate_kernel<4><<<grid, 256>>>(d, 0.999, 1.5); }); break;
                case 5: ms = time_ms([&] { rate_kernel<5><<<grid, 256>>>(d, 0.999, 1e-3); }); break;
                case 6: ms = time_ms([&] { rate_kernel<6><<<grid, 256>>>(d, 0.999, 1e-3); }); break;
                case 7: ms = time_ms([&] { rate_kernel<7><<<grid, 256>>>(d, 0.999, 1e-3); }); break;
                case 8: ms = time_ms([&] { rate_kernel<8><<<grid, 256>>>(d, 0.999, 1e-3); }); break;
                case 9: ms = time_ms([&] { rate_kernel<9><<<grid, 256>>>(d, 0.999, 1e-3); }); break;
            }
            const double waves_per_simd = wg_per_cu[w];  // 4 waves per workgroup over 4 SIMDs
            const double ops_per_simd = waves_per_simd * (double)kIter * kChains * per_op[op];
            const double cycles = ms * 1e-3 * clk;
            const double lane_ops = (double)grid * 256 * kIter * kChains * per_op[op];
            printf("%-14s %8.3f ms  %6.2f cycles / wave-op / SIMD   %7.2f T lane-op/s\n", names[op], ms, cycles / ops_per_simd,
                   lane_ops / (ms * 1e-3) / 1e12);
        }
        double ms = time_ms([&] { rate_kernel_f32<0><<<grid, 256>>>((float*)d, 0.999f, 1e-3f); });
        printf("%-14s %8.3f ms  %6.2f cycles / wave-op / SIMD\n", "v_fma_f32", ms, ms * 1e-3 * clk / (wg_per_cu[w] * (double)kIter * kChains));
    }
    return 0;
}
