// device_math.hpp -- per-pixel model of the RS differential-SfM solve, shared by all kernels.
//
// Compiled with -ffp-contract=off; sums of products are fused where -- and only where -- __builtin_fma is written out, at
// exactly the places oracle/rsdsfm_oracle.c calls fma() (rso_residual, jac_rho, the LM loops, point_error), so that integer
// outputs (inlier masks / counts, LM decisions) can be compared bit-exactly with the CPU oracle.  The expressions are the
// reference's (nonlinearRefinement.cc:32-52 for the residual, minimal.cc:255-275 for the scoring error); contracting them is
// what gcc's default -ffp-contract=fast does to the reference on FMA hardware.  Here it is a measured 11 % of the dense depth
// solve (the kernels are bound by fp64 instruction issue: an fma is one instruction, the unfused pair two).
#pragma once

#include <hip/hip_runtime.h>

#include "rsdsfm_internal.hpp"

namespace rsdsfm {

__device__ __forceinline__ double clampd(double x, double lo, double hi) { return x < lo ? lo : (x > hi ? hi : x); }

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
        double beta = two_over * __builtin_fma(p.k, alpha_k, alpha);  // (2/(2+k)) * (alpha + k alpha_k)
        nbeta = beta * -1.0;
        a0 = __builtin_fma(x, p.v[2], -p.v[0]);
        a1 = __builtin_fma(y, p.v[2], -p.v[1]);
        t01 = x * y * p.w[0];
        t02 = __builtin_fma(x, x, 1.0) * p.w[1];
        t03 = y * p.w[2];
        t11 = __builtin_fma(y, y, 1.0) * p.w[0];
        t12 = x * y * p.w[1];
        t13 = x * p.w[2];
        ux = ux_;
        uy = uy_;
        J0 = beta * a0;
        J1 = beta * a1;
    }
    __device__ __forceinline__ void residual(double rho, double& r0, double& r1) const {
        r0 = __builtin_fma(-nbeta, __builtin_fma(rho, a0, t01) - t02 + t03, ux);
        r1 = __builtin_fma(-nbeta, __builtin_fma(rho, a1, t11) - t12 - t13, uy);
    }
};

// minimal.cc:255-270: residual norm of the flow predicted from (v, w, k, rho)
__device__ __forceinline__ double point_error(double x, double y, double ux, double uy, double alpha, double alpha_k,
                                              const Pose& p, double two_over, double rho) {
    double beta = __builtin_fma(p.k, alpha_k, alpha) * two_over;
    double av0 = __builtin_fma(-x, p.v[2], p.v[0]);
    double av1 = __builtin_fma(-y, p.v[2], p.v[1]);
    double bw0 = __builtin_fma(-y, p.w[2], __builtin_fma(__builtin_fma(x, x, 1.0), p.w[1], (-x * y) * p.w[0]));
    double bw1 = __builtin_fma(x, p.w[2], __builtin_fma(x * y, p.w[1], (-__builtin_fma(y, y, 1.0)) * p.w[0]));
    double e0 = __builtin_fma(beta, __builtin_fma(av0, rho, bw0), -ux);
    double e1 = __builtin_fma(beta, __builtin_fma(av1, rho, bw1), -uy);
    return sqrt(__builtin_fma(e0, e0, e1 * e1));
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
