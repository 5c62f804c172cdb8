// Bit-for-bit check of the in-range cores of fp64 sqrt, reciprocal and division (csrc/device_math.hpp: sqrt_core / rcp_core / div_core) against the
// compiler's own expansions of sqrt(x), 1.0 / x and n / d on gfx950: random bit patterns over the whole in-range domain plus the
// boundaries.  Build and run on a GPU box:  hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I rs-aware-differential-sfm_amd/csrc
//   -I include tools/fastmath_check.hip -o /tmp/fastmath_check && /tmp/fastmath_check [millions of samples per launch] [launches]
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#include "device_math.hpp"

using namespace rsdsfm;

__device__ __forceinline__ uint64_t mix(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// mode 0: x uniform over the bit patterns of [2^-767, DBL_MAX]; mode 1: exponent uniform in a window around the bounds; out: mismatch counters
__global__ void check(uint64_t seed, uint64_t n, int mode, unsigned long long* bad, double* example) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t r = mix(seed + i), r2 = mix(r);
    uint64_t bits;
    if (mode == 0) {
        const uint64_t lo = 0x1000000000000000ull, hi = 0x7FEFFFFFFFFFFFFFull;
        bits = lo + r % (hi - lo + 1);
    } else {
        const uint64_t e = (mode == 1) ? (0x100 + (r2 % 4)) : (0x7FE - (r2 % 4));  // biased exponent next to a bound
        bits = (e << 52) | (r & 0xFFFFFFFFFFFFFull);
        if ((r2 >> 8) % 16 == 0) bits &= ~0xFFFFFFFFull;  // sparse mantissas too
    }
    const double x = __longlong_as_double((long long)bits);
    if (!sqrt_in_range(x)) {
        atomicAdd(&bad[3], 1ull);  // (the generator must stay inside the domain)
        return;
    }
    const double a = sqrt(x), b = sqrt_core(x);
    if (__double_as_longlong(a) != __double_as_longlong(b)) {
        if (atomicAdd(&bad[0], 1ull) == 0) example[0] = x;
    }
    // the reciprocal's domain in the solver: d = 1 + sqrt(x), x in range
    const double d = 1.0 + a;
    const double p = 1.0 / d, q = rcp_core(d);
    if (__double_as_longlong(p) != __double_as_longlong(q)) {
        if (atomicAdd(&bad[1], 1ull) == 0) example[1] = d;
    }
    // and over the in-range doubles themselves, exponents where neither scaling nor the special cases of the division apply
    const double y = __longlong_as_double((long long)((bits & 0x000FFFFFFFFFFFFFull) | ((uint64_t)(0x200 + r2 % 0x3FF) << 52) | ((r2 >> 40) << 63)));
    const double p2 = 1.0 / y, q2 = rcp_core(y);
    if (__double_as_longlong(p2) != __double_as_longlong(q2)) {
        if (atomicAdd(&bad[2], 1ull) == 0) example[2] = y;
    }
    // the general division over its window: both magnitudes in [2^-383, 2^385), any signs; exponents uniform over the window, or
    // pinned to its bounds (modes 1 and 2)
    const uint64_t r3 = mix(r2), r4 = mix(r3);
    uint64_t en = 0x280 + r3 % 0x300, ed = 0x280 + (r3 >> 20) % 0x300;
    if (mode == 1) en = 0x280 + r3 % 2, ed = 0x57F - (r3 >> 20) % 2;
    if (mode == 2) en = 0x57F - r3 % 2, ed = 0x280 + (r3 >> 20) % 2;
    const double nn = __longlong_as_double((long long)((r4 & 0x800FFFFFFFFFFFFFull) | (en << 52)));
    const double dd = __longlong_as_double((long long)((mix(r4) & 0x800FFFFFFFFFFFFFull) | (ed << 52)));
    if (div_range_key(nn) >= kDivRangeKeys || div_range_key(dd) >= kDivRangeKeys) {
        atomicAdd(&bad[3], 1ull);
        return;
    }
    const double p3 = nn / dd, q3 = div_core(nn, dd);
    if (__double_as_longlong(p3) != __double_as_longlong(q3)) {
        if (atomicAdd(&bad[4], 1ull) == 0) {
            example[3] = nn;
            example[4] = dd;
        }
    }
}

int main(int argc, char** argv) {
    const uint64_t per = (argc > 1 ? strtoull(argv[1], nullptr, 10) : 256) * 1000000ull;
    const int launches = argc > 2 ? atoi(argv[2]) : 8;
    unsigned long long* bad;
    double* ex;
    if (hipMalloc(&bad, 5 * sizeof(*bad)) != hipSuccess || hipMalloc(&ex, 5 * sizeof(double)) != hipSuccess) return 2;
    if (hipMemset(bad, 0, 5 * sizeof(*bad)) != hipSuccess || hipMemset(ex, 0, 5 * sizeof(double)) != hipSuccess) return 2;
    for (int l = 0; l < launches; ++l)
        for (int mode = 0; mode < 3; ++mode) {
            const uint64_t n = mode == 0 ? per : per / 8;
            hipLaunchKernelGGL(check, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, 0x1234567ull * (l + 1) + mode, n, mode, bad, ex);
        }
    unsigned long long h[5];
    double he[5];
    if (hipMemcpy(h, bad, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return 2;
    if (hipMemcpy(he, ex, sizeof(he), hipMemcpyDeviceToHost) != hipSuccess) return 2;
    const double total = (double)launches * (per + 2 * (per / 8));
    printf("samples %.3e: sqrt mismatches %llu, 1/(1+sqrt) mismatches %llu, 1/y mismatches %llu, n/d mismatches %llu, out-of-domain samples %llu\n", total, h[0], h[1],
           h[2], h[4], h[3]);
    if (h[0]) printf("  first sqrt mismatch at x = %a\n", he[0]);
    if (h[1]) printf("  first reciprocal mismatch at d = %a\n", he[1]);
    if (h[2]) printf("  first reciprocal mismatch at y = %a\n", he[2]);
    if (h[4]) printf("  first division mismatch at n = %a, d = %a\n", he[3], he[4]);
    return (h[0] || h[1] || h[2] || h[3] || h[4]) ? 1 : 0;
}
