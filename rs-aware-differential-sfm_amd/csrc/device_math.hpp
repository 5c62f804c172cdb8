// device_math.hpp -- per-pixel model of the RS differential-SfM solve, shared by all kernels.
//
// Two arithmetic modes, chosen at COMPILE time for the whole library (build.py builds both):
//   RSDSFM_FUSED == 0 (default, librsdsfm_hip.so): the REFERENCE's arithmetic.  The reference is built with plain
//     `-std=c++11` (src/CMakeLists.txt:18: no -mfma / -march), so on x86-64 every a*b+c of nonlinearRefinement.cc:32-52 and
//     minimal.cc:255-275 is an unfused multiply followed by an add, in source order.  The expressions below keep exactly that
//     order (the file is compiled with -ffp-contract=off), and fp64 * + / sqrt are correctly rounded on gfx950, so per-pixel
//     values -- and the integer outputs derived from them (inlier masks / counts, scanline indices) -- are bit-identical to
//     the CPU oracle's default (unfused) build.
//   RSDSFM_FUSED == 1 (opt-in, librsdsfm_hip_fused.so): sums of products contracted into fused multiply-adds where -- and
//     only where -- __builtin_fma is written out below, at exactly the places the oracle's -DRSO_FUSED build calls fma().
//     An fma is one fp64 instruction where the unfused pair is two and the per-pixel kernels are bound by fp64 issue
//     (measured: dense depth solve +11 %, ransac_lm_kernel -65 us).  tests/test_gpu_fused.py quantifies what it changes
//     against the unfused oracle on every BASELINE config (values <= 1e-5 relative, scanline indices identical, the number of
//     inlier-mask flips bounded).
#pragma once

#include <hip/hip_runtime.h>

#include "rsdsfm_internal.hpp"

#ifndef RSDSFM_FUSED
#define RSDSFM_FUSED 0
#endif

namespace rsdsfm {

__device__ __forceinline__ double clampd(double x, double lo, double hi) { return x < lo ? lo : (x > hi ? hi : x); }

// ---- in-range cores of the correctly rounded fp64 square root and reciprocal ---------------------------------------------------------
// The compiler expands sqrt(x) into  [x < 2^-767 ? scale x by 2^256]  v_rsq_f64 + two Newton steps on (g, h) + two residual corrections
// [scale back by 2^-128]  [x is +-0 or +inf ? x]  -- 18 instructions of which 8 are the bracketed wrapper -- and 1.0 / d into
// v_div_scale x 2, v_rcp_f64, two Newton steps, q = n r, one residual correction (v_div_fmas) and v_div_fixup: 11 instructions of which 4
// do nothing for a normal d with a normal reciprocal and n = 1 (tools/fastmath_check.hip lists the ISA's conditions).  For an argument inside
// the range the wrapper is the identity, so the cores below -- the SAME instructions in the same order -- return the same bits:
// tools/fastmath_check.hip compares them with the compiler's expansions over 1e10 random in-range bit patterns and the range bounds.
// ransac_lm_kernel runs them on every pixel of round 0 and ORs the range tests into one flag word of the launch; a RANSAC whose flag is
// raised (a zero Jacobian, a zero or non-finite error: never on real data) is run again with the standard functions (ransac_host.hip):
// 16 of the kernel's 307 instructions per pixel-hypothesis for the price of two 32-bit instructions per test and a branch per hypothesis.
// (An in-kernel fallback -- the wave recomputes the hypothesis -- was built first: whatever its shape, the second pixel loop behind the
// first one took the kernel from 240 to 300+ registers, one wave per SIMD; so did carrying the tests as a bool through the pixel loop.
// The tests are carried as the maximum of an integer key instead: sqrt_range_track.)
//
// x in [2^-767, DBL_MAX]: positive, finite, and not small enough for sqrt()'s expansion to rescale it (a test on the high word: the
// exponent field in [0x100, 0x7FE], sign clear -- NaNs, infinities, zeros, denormals and negative numbers all fail it)
__device__ __forceinline__ uint32_t sqrt_range_key(double x) { return (uint32_t)__double2hiint(x) - 0x10000000u; }  // in range iff < kSqrtRangeKeys
constexpr uint32_t kSqrtRangeKeys = 0x6FF00000u;
__device__ __forceinline__ bool sqrt_in_range(double x) { return sqrt_range_key(x) < kSqrtRangeKeys; }
// the range tests of many arguments as ONE comparison: `worst` = the maximum of their keys (two 32-bit instructions per argument)
__device__ __forceinline__ void sqrt_range_track(uint32_t& worst, double x) { worst = max(worst, sqrt_range_key(x)); }
__device__ __forceinline__ double sqrt_core(double x) {
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y;
    double h = y * 0.5;
    const double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    double d = __builtin_fma(-g, g, x);
    g = __builtin_fma(d, h, g);
    d = __builtin_fma(-g, g, x);
    return __builtin_fma(d, h, g);
}
// sqrt_core for an argument that may be EXACTLY zero (a vanishing Jacobian; the error of a pixel the model explains exactly: ground-truth
// flow is a supported input, main.cc:380-384): sqrt(0) = 0 is a select, not a reason to leave the cores -- `worst` tracks the range test of
// every other argument (sqrt_range_track).  x is a sum of squares: never -0.
__device__ __forceinline__ double sqrt_core_z(double x, uint32_t& worst) {
    const bool z = x == 0.0;
    worst = max(worst, z ? 0u : sqrt_range_key(x));
    const double r = sqrt_core(x);
    return z ? 0.0 : r;
}
// n / d for operands inside the window where v_div_scale does not rescale and v_div_fixup passes the quotient through: both
// magnitudes in [2^-383, 2^385) (then the exponents differ by less than 768, no operand and no quotient is zero, denormal or infinite).
// The core is the compiler's expansion without the two v_div_scale, with v_div_fmas as the plain fma it is when nothing was scaled,
// and without v_div_fixup.
__device__ __forceinline__ uint32_t div_range_key(double x) { return ((uint32_t)__double2hiint(x) & 0x7FFFFFFFu) - 0x28000000u; }  // in range iff < kDivRangeKeys
constexpr uint32_t kDivRangeKeys = 0x30000000u;
__device__ __forceinline__ void div_range_track(uint32_t& worst, double x) { worst = max(worst, div_range_key(x)); }
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
// 1.0 / d for a normal d whose reciprocal is normal (2^-1021 <= |d| <= 2^1021 is more than the callers need: d = 1 + sqrt(x) lies in [1, 2^513))
__device__ __forceinline__ double rcp_core(double d) {
    double r = __builtin_amdgcn_rcp(d);
    double e = __builtin_fma(-d, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-d, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-d, r, 1.0);  // (q = 1.0 * r is r)
    return __builtin_fma(e, r, r);
}

// One point of the caller glue main.cc:398-444 + getAlpha / getAlphaK (minimal.cc:179-197): pixel (column i, row j) with flow f in
// pixels -> normalised position q, normalised flow u, alpha, alpha_k (pixel units, h = rows: quirk Q6).  The ONE statement of these
// expressions: flatten_tile_kernel writes them to the point arrays, minimal9_kernel's direct mode forms its sampled points with them.
struct FlatPoint {
    double qx, qy, ux, uy, alpha, alpha_k;
};
__device__ __forceinline__ FlatPoint flatten_point(double2 f, int i, int j, double fx, double fy, double cx, double cy, double gamma, double h) {
    FlatPoint p;
    p.qx = (i - cx) * 1.0 / fx;
    p.qy = (j - cy) * 1.0 / fy;
    p.ux = f.x * gamma / fx;
    p.uy = f.y * gamma / fy;
    p.alpha = 1 + gamma * f.y / h;  // minimal.cc:183 with pixel flow, h = rows (quirk Q6)
    const double part1 = gamma * (double)j / h;
    const double part2 = 1.0 + gamma * ((double)j + f.y) / h;
    p.alpha_k = 0.5 * (part2 * part2 - part1 * part1);
    return p;
}

// LevenbergMarquardtStrategy::StepAccepted (Ceres 1.14 levenberg_marquardt_strategy.cc)
__host__ __device__ __forceinline__ double radius_accept(double radius, double q) {
    double t = 2.0 * q - 1.0;
    double f = 1.0 - t * t * t;
    if (f < 1.0 / 3.0) f = 1.0 / 3.0;
    radius = radius / f;
    if (radius > kMaxRadius) radius = kMaxRadius;
    return radius;
}

// Per-pixel constants of the dense depth problem for a fixed pose (v, w, k):
//   r(rho) = u - pred(rho),  pred_j = beta * -1 * (rho * a_j + t1_j - t2_j + t3_j)      (nonlinearRefinement.cc:36-49)
//   J = d r / d rho = beta * a
struct PixelModel {
    double nbeta;       // beta * -1.0
    double a0, a1;      // x v_z - v_x ,  y v_z - v_y
    double t01, t02, t03, t11, t12, t13;
    double ux, uy;
    double J0, J1;

    __device__ __forceinline__ void init(double x, double y, double ux_, double uy_, double alpha, double alpha_k,
                                         const Pose& p, double two_over) {
#if RSDSFM_FUSED
        double beta = two_over * __builtin_fma(p.k, alpha_k, alpha);  // (2/(2+k)) * (alpha + k alpha_k)
        nbeta = beta * -1.0;
        a0 = __builtin_fma(x, p.v[2], -p.v[0]);
        a1 = __builtin_fma(y, p.v[2], -p.v[1]);
        t01 = x * y * p.w[0];
        t02 = __builtin_fma(x, x, 1.0) * p.w[1];
        t03 = y * p.w[2];
        t11 = __builtin_fma(y, y, 1.0) * p.w[0];
#else
        double beta = two_over * (alpha + p.k * alpha_k);  // (2/(2+k)) * (alpha + k alpha_k)
        nbeta = beta * -1.0;
        a0 = x * p.v[2] - p.v[0];
        a1 = y * p.v[2] - p.v[1];
        t01 = x * y * p.w[0];
        t02 = (1.0 + x * x) * p.w[1];
        t03 = y * p.w[2];
        t11 = (1.0 + y * y) * p.w[0];
#endif
        t12 = x * y * p.w[1];
        t13 = x * p.w[2];
        ux = ux_;
        uy = uy_;
        J0 = beta * a0;
        J1 = beta * a1;
    }
    __device__ __forceinline__ void residual(double rho, double& r0, double& r1) const {
#if RSDSFM_FUSED
        r0 = __builtin_fma(-nbeta, __builtin_fma(rho, a0, t01) - t02 + t03, ux);
        r1 = __builtin_fma(-nbeta, __builtin_fma(rho, a1, t11) - t12 - t13, uy);
#else
        double p0 = nbeta * (rho * a0 + t01 - t02 + t03);
        double p1 = nbeta * (rho * a1 + t11 - t12 - t13);
        r0 = ux - p0;
        r1 = uy - p1;
#endif
    }
};

// the small sums of products of the LM loops (lm_common.hpp), in the two arithmetic modes: dot2 = a0 b0 + a1 b1,
// acc2 = acc + (a0 a0 + a1 a1) as the oracle's loops accumulate them, mad = a b + c
__device__ __forceinline__ double dot2(double a0, double b0, double a1, double b1) {
#if RSDSFM_FUSED
    return __builtin_fma(a0, b0, a1 * b1);
#else
    return a0 * b0 + a1 * b1;
#endif
}
__device__ __forceinline__ double mad(double a, double b, double c) {
#if RSDSFM_FUSED
    return __builtin_fma(a, b, c);
#else
    return a * b + c;
#endif
}
__device__ __forceinline__ double acc_sq2(double acc, double r0, double r1) {
#if RSDSFM_FUSED
    return __builtin_fma(r0, r0, __builtin_fma(r1, r1, acc));
#else
    return acc + (r0 * r0 + r1 * r1);
#endif
}
__device__ __forceinline__ double acc_sq(double acc, double x) {
#if RSDSFM_FUSED
    return __builtin_fma(x, x, acc);
#else
    return acc + x * x;
#endif
}

// point_error (below) from the terms a PixelModel of the same pixel and pose already holds.  Reference arithmetic only: every
// term of minimal.cc:255-270 is a term of the model up to sign, and IEEE negation / subtraction are exact in the sign --
//   beta = -nbeta;  A v = (v0 - x v2, v1 - y v2) = (-a0, -a1);  B w = ((-t01 + t02) - t03, (-t11 + t12) + t13)
// -- so  e = beta (A v rho + B w) - u  comes out bit for bit as  beta ((B w) - a rho) - u  (12 instead of 25 multiply / adds per
// pixel-hypothesis in ransac_lm_kernel, whose two fused scores share them).
__device__ __forceinline__ double point_error_from_model(const PixelModel& m, double rho) {
#if RSDSFM_FUSED
    return 0.0;  // (not used: the fused build evaluates point_error itself, whose contractions differ from the model's)
#else
    const double beta = -m.nbeta;
    const double bw0 = (m.t02 - m.t01) - m.t03;
    const double bw1 = (m.t12 - m.t11) + m.t13;
    const double e0 = beta * (bw0 - m.a0 * rho) - m.ux;
    const double e1 = beta * (bw1 - m.a1 * rho) - m.uy;
    return sqrt(e0 * e0 + e1 * e1);
#endif
}
// the same with the in-range core of the square root; `worst` tracks the range test of the argument (sqrt_range_track)
__device__ __forceinline__ double point_error_from_model_core(const PixelModel& m, double rho, uint32_t& worst) {
    const double beta = -m.nbeta;
    const double bw0 = (m.t02 - m.t01) - m.t03;
    const double bw1 = (m.t12 - m.t11) + m.t13;
    const double e0 = beta * (bw0 - m.a0 * rho) - m.ux;
    const double e1 = beta * (bw1 - m.a1 * rho) - m.uy;
    const double ss = e0 * e0 + e1 * e1;
    return sqrt_core_z(ss, worst);
}

// minimal.cc:255-270: residual norm of the flow predicted from (v, w, k, rho)
__device__ __forceinline__ double point_error(double x, double y, double ux, double uy, double alpha, double alpha_k,
                                              const Pose& p, double two_over, double rho) {
#if RSDSFM_FUSED
    double beta = __builtin_fma(p.k, alpha_k, alpha) * two_over;
    double av0 = __builtin_fma(-x, p.v[2], p.v[0]);
    double av1 = __builtin_fma(-y, p.v[2], p.v[1]);
    double bw0 = __builtin_fma(-y, p.w[2], __builtin_fma(__builtin_fma(x, x, 1.0), p.w[1], (-x * y) * p.w[0]));
    double bw1 = __builtin_fma(x, p.w[2], __builtin_fma(x * y, p.w[1], (-__builtin_fma(y, y, 1.0)) * p.w[0]));
    double e0 = __builtin_fma(beta, __builtin_fma(av0, rho, bw0), -ux);
    double e1 = __builtin_fma(beta, __builtin_fma(av1, rho, bw1), -uy);
    return sqrt(__builtin_fma(e0, e0, e1 * e1));
#else
    // A*v, B*w as Eigen evaluates the 2x3 * 3x1 products (terms in column order)
    double beta = (alpha + p.k * alpha_k) * two_over;
    double av0 = p.v[0] + (-x) * p.v[2];
    double av1 = p.v[1] + (-y) * p.v[2];
    double bw0 = (-x * y) * p.w[0] + (1 + x * x) * p.w[1] + (-y) * p.w[2];
    double bw1 = (-(1 + y * y)) * p.w[0] + (x * y) * p.w[1] + x * p.w[2];
    double e0 = beta * (av0 * rho + bw0) - ux;
    double e1 = beta * (av1 * rho + bw1) - uy;
    return sqrt(e0 * e0 + e1 * e1);
#endif
}

// ---- wave64 reductions on the VALU (DPP row shifts / broadcasts; no LDS traffic).  Fixed combination
// order -> deterministic.  The result is returned to every lane (read from lane 63).  wave_max assumes
// non-negative inputs (lanes shifted in read 0), which holds for every max slot (they are absolute values).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_move(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, true);
    return __hiloint2double(hi, lo);
}
// the value lane `l` (wave-uniform) holds in v
__device__ __forceinline__ double lane_value(double v, int l) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double lane63(double v) {
    int lo = __builtin_amdgcn_readlane(__double2loint(v), 63);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum(double v) {
    v += dpp_move<0xb1, 0xf>(v);   // quad_perm:[1,0,3,2]
    v += dpp_move<0x4e, 0xf>(v);   // quad_perm:[2,3,0,1]
    v += dpp_move<0x114, 0xf>(v);  // row_shr:4
    v += dpp_move<0x118, 0xf>(v);  // row_shr:8   -> lanes 12..15 of each row hold the row sum
    v += dpp_move<0x142, 0xa>(v);  // row_bcast:15 into rows 1 and 3
    v += dpp_move<0x143, 0xc>(v);  // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave sum
    return lane63(v);
}
__device__ __forceinline__ double wave_max(double v) {
    v = fmax(v, dpp_move<0xb1, 0xf>(v));
    v = fmax(v, dpp_move<0x4e, 0xf>(v));
    v = fmax(v, dpp_move<0x114, 0xf>(v));
    v = fmax(v, dpp_move<0x118, 0xf>(v));
    v = fmax(v, dpp_move<0x142, 0xa>(v));
    v = fmax(v, dpp_move<0x143, 0xc>(v));
    return lane63(v);
}

}  // namespace rsdsfm
