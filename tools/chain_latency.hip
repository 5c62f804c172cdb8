// chain_latency.hip -- latency of DEPENDENT instructions for ONE wave alone on its SIMD (the regime of the minimal solver's Jacobi SVD chain:
// one hypothesis per wave, 50 waves on the chip): cycles per dependent v_fma_f64 / v_mul_f64+v_add_f64 / v_rcp_f64 / v_rsq_f64, the in-range
// division / square-root cores' shape (rcp + 7 fma), and a dependent LDS write -> read round trip.  s_memtime brackets, one workgroup of one wave.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o tools/chain_latency tools/chain_latency.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>

constexpr int kN = 4096;

// the parameter chain of one two-sided Jacobi rotation as minimal9_kernels.hip computes it (real_2x2_jacobi_svd + makeJacobi through the
// in-range cores, with their range tracking), restated here for timing only
__device__ __forceinline__ double sqrt_core(double x) {
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = y * 0.5;
    const double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    double d = __builtin_fma(-g, g, x);
    g = __builtin_fma(d, h, g);
    d = __builtin_fma(-g, g, x);
    return __builtin_fma(d, h, g);
}
__device__ __forceinline__ double div_core(double n, double d) {
    double r = __builtin_amdgcn_rcp(d);
    double e = __builtin_fma(-d, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-d, r, 1.0);
    r = __builtin_fma(r, e, r);
    const double q = n * r;
    e = __builtin_fma(-d, q, n);
    return __builtin_fma(e, r, q);
}
__device__ __forceinline__ double rcp_core(double d) {
    double r = __builtin_amdgcn_rcp(d);
    double e = __builtin_fma(-d, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-d, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-d, r, 1.0);
    return __builtin_fma(e, r, r);
}
__device__ __forceinline__ void trk(uint32_t& w, double x) { w = max(w, ((uint32_t)__double2hiint(x) & 0x7FFFFFFFu) - 0x28000000u); }
struct Rot {
    double c, s;
};
__device__ __forceinline__ void rotation_params(double m0, double m1, double m2, double m3, uint32_t& wd, Rot& jl, Rot& jr) {
    Rot rot1;
    const double t = m0 + m3, d = m2 - m1;
    if (fabs(d) < 2.2250738585072014e-308) {
        rot1.s = 0.0, rot1.c = 1.0;
    } else {
        trk(wd, t), trk(wd, d);
        const double uu = div_core(t, d);
        const double a1p = 1.0 + uu * uu;
        trk(wd, a1p);
        const double tmp = sqrt_core(a1p);
        rot1.s = rcp_core(tmp);
        trk(wd, uu), trk(wd, tmp);
        rot1.c = div_core(uu, tmp);
    }
    if (!(rot1.c == 1.0 && rot1.s == 0.0)) {
        const double a0 = rot1.c * m0 + rot1.s * m2, a2 = -rot1.s * m0 + rot1.c * m2;
        const double a1 = rot1.c * m1 + rot1.s * m3, a3 = -rot1.s * m1 + rot1.c * m3;
        m0 = a0, m1 = a1, m2 = a2, m3 = a3;
    }
    const double deno = 2.0 * fabs(m1);
    if (deno < 2.2250738585072014e-308) {
        jr.c = 1.0, jr.s = 0.0;
    } else {
        const double xz = m0 - m3;
        trk(wd, xz), trk(wd, deno);
        const double tau = div_core(xz, deno);
        const double ww = tau * tau + 1.0;
        trk(wd, ww);
        const double w = sqrt_core(ww);
        const double den = (tau > 0.0) ? tau + w : tau - w;
        const double tt = rcp_core(den);
        const double sign_t = tt > 0.0 ? 1.0 : -1.0;
        const double n = rcp_core(sqrt_core(tt * tt + 1.0));
        jr.s = -sign_t * copysign(1.0, m1) * fabs(tt) * n;
        jr.c = n;
    }
    const Rot jt = {jr.c, -jr.s};
    jl.c = rot1.c * jt.c - rot1.s * jt.s;
    jl.s = rot1.c * jt.s + rot1.s * jt.c;
}

template <int OP>
__global__ __launch_bounds__(64) void lat_kernel(double* out, unsigned long long* clk, double b, double c) {
    __shared__ double lds[128];
    double x = 1.0 + 1e-3 * threadIdx.x;
    lds[threadIdx.x] = x;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < (OP == 7 ? 0 : kN); ++it) {
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
    if (OP == 7) {
        double m0 = x, m1 = 0.3 * x, m2 = 0.2 + 0.1 * x, m3 = 0.7 * x;
        uint32_t wd = 0;
        for (int it = 0; it < kN; ++it) {
            Rot jl, jr;
            rotation_params(m0, m1, m2, m3, wd, jl, jr);
            // the next block depends on this rotation (kept in range: the values wander around 1)
            m0 = 1.0 + 0.25 * jl.c, m1 = 0.3 + 0.1 * jr.s, m2 = 0.2 + 0.1 * jl.s, m3 = 0.7 + 0.2 * jr.c;
        }
        x = m0 + m1 + m2 + m3 + (double)wd;
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
    const char* names[] = {"v_fma_f64", "v_mul_f64 + v_add_f64", "v_rcp_f64 + v_add_f64", "v_rsq_f64 + v_add_f64", "ds_write_b64 -> ds_read_b64 + v_mul_f64", "v_max_f64 + v_add_f64", "2 x v_readlane + v_add_f64 x 2", "parameter chain of one Jacobi rotation (+ 8 glue ops)"};
    for (int op = 0; op < 8; ++op) {
        for (int rep = 0; rep < 2; ++rep) {
            switch (op) {
                case 0: lat_kernel<0><<<1, 64>>>(d, c, 0.999, 1e-3); break;
                case 1: lat_kernel<1><<<1, 64>>>(d, c, 0.999, 1e-3); break;
                case 2: lat_kernel<2><<<1, 64>>>(d, c, 0.999, 1.0); break;
                case 3: lat_kernel<3><<<1, 64>>>(d, c, 0.999, 1.0); break;
                case 4: lat_kernel<4><<<1, 64>>>(d, c, 0.999, 1e-3); break;
                case 5: lat_kernel<5><<<1, 64>>>(d, c, 1e-3, 0.5); break;
                case 6: lat_kernel<6><<<1, 64>>>(d, c, 0.999, 1e-3); break;
                case 7: lat_kernel<7><<<1, 64>>>(d, c, 0.999, 1e-3); break;
            }
            hipDeviceSynchronize();
        }
        unsigned long long h = 0;
        hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
        printf("%-44s %7.1f shader clocks per iteration (s_memtime)\n", names[op], (double)h / kN);
    }
    return 0;
}
