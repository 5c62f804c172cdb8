// chain_latency.hip -- latency of DEPENDENT instructions for ONE wave alone on its SIMD (the regime of the minimal solver's Jacobi SVD chain:
// one hypothesis per wave, 50 waves on the chip): cycles per dependent v_fma_f64 / v_mul_f64+v_add_f64 / v_rcp_f64 / v_rsq_f64, the in-range
// division / square-root cores' shape (rcp + 7 fma), and a dependent LDS write -> read round trip.  s_memtime brackets, one workgroup of one wave.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o tools/chain_latency tools/chain_latency.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

constexpr int kN = 4096;

template <int OP>
__global__ __launch_bounds__(64) void lat_kernel(double* out, unsigned long long* clk, double b, double c) {
    __shared__ double lds[128];
    double x = 1.0 + 1e-3 * threadIdx.x;
    lds[threadIdx.x] = x;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < kN; ++it) {
        if (OP == 0) x = __builtin_fma(x, b, c);
        if (OP == 1) x = x * b + c;
        if (OP == 2) x = __builtin_amdgcn_rcp(x) + c;      // rcp + add
        if (OP == 3) x = __builtin_amdgcn_rsq(x) + c;      // rsq + add
        if (OP == 4) {                                     // LDS write -> dependent read (another lane's slot)
            lds[threadIdx.x] = x;
            __builtin_amdgcn_wave_barrier();
            x = lds[(threadIdx.x + 1) & 63] * b;
        }
        if (OP == 5) x = fmax(x, c) + b;                   // max + add
        if (OP == 6) {                                     // readlane broadcast + add
            const int lo = __builtin_amdgcn_readlane(__double2loint(x), 3), hi = __builtin_amdgcn_readlane(__double2hiint(x), 3);
            x = __hiloint2double(hi, lo) + c + 1e-9 * threadIdx.x;
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
    if (x == 12345.678) out[0] = x;
}

int main() {
    double* d;
    unsigned long long* c;
    hipMalloc(&d, 64);
    hipMalloc(&c, 8 * 64);
    const char* names[] = {"v_fma_f64", "v_mul_f64 + v_add_f64", "v_rcp_f64 + v_add_f64", "v_rsq_f64 + v_add_f64", "ds_write_b64 -> ds_read_b64 + v_mul_f64", "v_max_f64 + v_add_f64", "2 x v_readlane + v_add_f64 x 2"};
    for (int op = 0; op < 7; ++op) {
        for (int rep = 0; rep < 2; ++rep) {
            switch (op) {
                case 0: lat_kernel<0><<<1, 64>>>(d, c, 0.999, 1e-3); break;
                case 1: lat_kernel<1><<<1, 64>>>(d, c, 0.999, 1e-3); break;
                case 2: lat_kernel<2><<<1, 64>>>(d, c, 0.999, 1.0); break;
                case 3: lat_kernel<3><<<1, 64>>>(d, c, 0.999, 1.0); break;
                case 4: lat_kernel<4><<<1, 64>>>(d, c, 0.999, 1e-3); break;
                case 5: lat_kernel<5><<<1, 64>>>(d, c, 1e-3, 0.5); break;
                case 6: lat_kernel<6><<<1, 64>>>(d, c, 0.999, 1e-3); break;
            }
            hipDeviceSynchronize();
        }
        unsigned long long h = 0;
        hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
        printf("%-44s %7.1f shader clocks per iteration (s_memtime)\n", names[op], (double)h / kN);
    }
    return 0;
}
