// lma_common.hpp -- the ANALYTIC LM TRAJECTORY of the dense depth solves (reference nonlinearRefinement.cc:109-180 under Ceres 1.14's
// trust-region loop; checker: oracle/rsdsfm_oracle.c rso_lma_trial, "mode 2").
//
// For a fixed pose the residual of a pixel is linear in its own rho (nonlinearRefinement.cc:36-49):  r(rho) = c + rho J.  With
//   h = J.J,  g = J.r(1),  e0 = g / h,  rho* = 1 - e0  (the pixel's optimum)
// an LM step at trust-region radius R (diagonal clamp(s^2 h, 1e-6, 1e32) / R; the Jacobi scaling s cancels) multiplies rho - rho* by
// 1 / (1 + R) for every pixel whose clamp is inactive, so after accepted steps with radii R_1 .. R_K
//   rho_K = rho* + e0 phi_K,   phi_K = prod 1 / (1 + R_i),   |r(rho_K)|^2 = |r(rho*)|^2 + g e0 phi_K^2
// and everything the trust-region loop looks at is a closed form of FIVE sums and a maximum per hypothesis:
//   A = sum |r(rho*)|^2   B = sum g e0   C = sum e0^2   D = sum rho*^2   E = sum rho* e0   G = max |g|
//   cost(phi) = (A + B phi^2) / 2;  a step from phi at radius R (psi = R / (1 + R)):  model change = B phi^2 psi (1 - psi / 2),
//   |step|^2 = C phi^2 psi^2,  |x|^2 = D + 2 E phi + C phi^2,  max |gradient| = G phi.
// ~70 operations per pixel-hypothesis instead of 55 per pixel, hypothesis AND iteration.  It is another arithmetic than the reference's
// (fused multiply-adds where they help: this is the library's own), so guards keep every INTEGER output equal to the iterate-by-iterate
// kernels' (lm_common.hpp), which stay in the library as the path a guard falls back to:
//  (a) a pixel whose clamp may bind (h < kLmaHIrr: within ~1 px of the focus of expansion) enters the sums frozen at rho = 1, is put on the
//      hypothesis' list, and the decide stage walks the reference's exact recurrence for it (lmx_walk) beside the closed form;
//  (b) a pixel whose squared error comes within  m = eta tol (2 + |r(1)|^2 + h) / 2  of tol^2 at an iterate whose score is fused is put on
//      the list as well and scored from the exact iterate (eta is > 1000 x the largest difference between the two arithmetics ever
//      observed: tools/lma_cpu_fuzz.py, tests/test_oracle_lma.py);
//  (c) a global decision within a relative kLmaBand of its threshold, an infinite sum, a list that overflows, a trajectory that leaves
//      the tabulated plan while listed pixels walk beside it: the RANSAC is run again on the iterate-by-iterate kernels;
//  (d) a tie in the inlier count whose error sums differ by less than kLmaTie x count (noise-free data: the reference's winner is decided
//      by rounding noise no other arithmetic reproduces): likewise.
#pragma once

#include "device_math.hpp"
#include "lm_common.hpp"
#include "rsdsfm_internal.hpp"

namespace rsdsfm {

constexpr double kLmaHIrr = 1.01e-6;  // s^2 h = h / (1 + sqrt h)^2 reaches the clamp's 1e-6 at h = 1.002003e-6
constexpr double kLmaEta = 1e-11;     // guard (b)
// ... and the floor of its margin's coefficient: 128 u (u = 2^-53).  The derived forward-error bound of the two arithmetics' squared errors is
// 128 u (2 + |r(1)|^2 + h); eta tol / 2 covers it from tol = 2.9e-3 on, the floor covers it for every smaller tolerance.
constexpr double kLmaMarginFloor = 128.0 * 0x1p-53;
constexpr double kLmaBand = 1e-6;     // guard (c)
// (kLmaTie, guard (d): rsdsfm_internal.hpp -- the host files launch the picks with it)
constexpr double kLmaSqrtMin = 0x1p-767;  // squared errors below this (incl. 0) add 0 to the inlier error SUM (the range of sqrt_core)

constexpr int kLmaNC = 3;    // iterates whose score the pixel pass can fuse (at most)
constexpr int kLmaKP = 4;    // planned LM steps the listed pixels' exact walk is tabulated for (3 accepted steps + the step that terminates)
constexpr int kLmaListCap = 1024;  // listed pixels per hypothesis (and rank)
// per-(workgroup, hypothesis) partial row of the pixel pass
constexpr int kLmaSlots = 6 + 2 * kLmaNC;  // A B C D E | G | {count, error sum} x kLmaNC
constexpr int kLmaG = 5;
// per-(rank, hypothesis) row the decide stage consumes (the all-gather payload of the column-tiled solve):
//   [0..4] A' B C D' E (listed clamped pixels removed)  [5] G  [6] clamped pixels  [7] overflow / non-finite flag
//   [8] XC_0  [9] Xg_0   then per planned step k < kLmaKP: XM_k XS_k XC_{k+1} XX_{k+1} Xg_{k+1}
//   then per fused iterate: count, error sum (listed pixels corrected)
constexpr int kLmaRowX0 = 8;
constexpr int kLmaRowXk = 10;
constexpr int kLmaRowScore = kLmaRowXk + 5 * kLmaKP;
constexpr int kLmaRow = kLmaRowScore + 2 * kLmaNC;

// the planned trajectory: radius of step k, and phi after it (every step accepted with quality ~1: radius / (1 / 3))
struct LmaPlan {
    double radius[kLmaKP + 1];
    double psi[kLmaKP + 1];  // psi of step k (radius[k])
    double phi[kLmaKP + 2];  // phi[0] = 1
};
__host__ __device__ inline void lma_phi_step(double radius, double phi, double& psi, double& phic) {
    const double ir = 1.0 / radius;
    psi = 1.0 / (1.0 + ir);
    phic = phi * (ir * psi);
}
__host__ __device__ inline LmaPlan lma_plan() {
    LmaPlan p;
    double r = kInitialRadius, phi = 1.0;
    p.phi[0] = 1.0;
    for (int k = 0; k <= kLmaKP; ++k) {
        p.radius[k] = r;
        double psi, phic;
        lma_phi_step(r, phi, psi, phic);
        p.psi[k] = psi;
        p.phi[k + 1] = phic;
        phi = phic;
        r = radius_accept(r, 1.0);
    }
    return p;
}

// the fused iterates of a pixel pass (kernel argument, by value)
struct LmaCand {
    int nc;               // fused iterates (<= kLmaNC)
    int steps[kLmaNC];    // accepted steps of each (1 ..  kLmaKP - 1)
    double phi2[kLmaNC];  // phi^2 of each
    double tol, tol2, c1, c1x2;  // tolerance, its square, eta tol / 2 and eta tol
    int count_only;              // the pass fuses the inlier COUNTS of its iterates only (ransac_lma_kernel ERR = false): the error sums in the rows stay 0
    LmaPlan plan;                // the planned trajectory (computed once on the host: a chain of divisions)
};

struct LmaPx {
    double a, ge, e0, rhos, h, g;  // |r(rho*)|^2, g e0, e0, rho*, J.J, J.r(1)  (g = e0 = 0 for a clamped pixel)
    bool clamped;
};

// max(x, lo) as the one instruction it is (v_max_f64 returns the other operand for a NaN, like fmax: a NaN h is floored too)
__device__ __forceinline__ double lma_max_pos(double x, double lo) {
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(x), "v"(lo));
    return r;
}
// oracle/rsdsfm_oracle.c lma_pixel(): the SAME operations in the same order
// FULL: the standard division (the rows stage, which sees a handful of pixels: same bits in range)
template <bool FULL = false>
__device__ __forceinline__ LmaPx lma_pixel(double x, double y, double ux, double uy, double al, double ak, const Pose& p, double two_over) {
    LmaPx o;
    const double beta = two_over * __builtin_fma(p.k, ak, al);
    const double a0 = __builtin_fma(x, p.v[2], -p.v[0]), a1 = __builtin_fma(y, p.v[2], -p.v[1]);
    const double J0 = beta * a0, J1 = beta * a1;
    const double xy = x * y, xx1 = __builtin_fma(x, x, 1.0), yy1 = __builtin_fma(y, y, 1.0);
    const double bw0 = __builtin_fma(xx1, p.w[1], __builtin_fma(-xy, p.w[0], -(y * p.w[2])));
    const double bw1 = __builtin_fma(xy, p.w[1], __builtin_fma(-yy1, p.w[0], x * p.w[2]));
    const double c0 = __builtin_fma(-beta, bw0, ux), c1 = __builtin_fma(-beta, bw1, uy);
    const double r0 = c0 + J0, r1 = c1 + J1;
    const double h = __builtin_fma(J0, J0, J1 * J1);
    double g = __builtin_fma(J0, r0, J1 * r1);
    // a clamped pixel enters the sums frozen at rho = 1: g = e0 = 0 (g times a 0 / 1 mask whose high word is one select, h floored at the
    // threshold: four instructions instead of seven selects).  g / h through the in-range core of the division (device_math.hpp: the
    // compiler's own expansion without the operand rescaling; the same bits for |g| in [2^-383, 2^385) or 0 and h in [1e-6, 2^385) -- sums of
    // other magnitudes are infinite and send the run to guard (c))
    const bool clamped = h < kLmaHIrr;
    g = g * __hiloint2double(clamped ? 0 : 0x3FF00000, 0);
    const double e0 = FULL ? g / lma_max_pos(h, kLmaHIrr) : div_core(g, lma_max_pos(h, kLmaHIrr));
    const double rhos = 1.0 - e0;
    const double s0 = __builtin_fma(rhos, J0, c0), s1 = __builtin_fma(rhos, J1, c1);
    o.a = __builtin_fma(s0, s0, s1 * s1);
    o.ge = g * e0;
    o.e0 = e0;
    o.rhos = rhos;
    o.h = h;
    o.g = g;
    o.clamped = clamped;
    return o;
}
// max(m, |x|) as the one instruction it is (v_max_f64 returns the other operand for a NaN, like fmax)
__device__ __forceinline__ double lma_max_abs(double m, double x) {
    double r;
    asm("v_max_f64 %0, %1, |%2|" : "=v"(r) : "v"(m), "v"(x));
    return r;
}
__device__ __forceinline__ double lma_margin(const LmaPx& p, const LmaCand& cd) {
    const double t = (p.a + p.ge) + p.h;
    return __builtin_fma(t, cd.c1, cd.c1x2);
}
// the analytic score of one pixel at phi^2: inlier flag and its term of the error sum
__device__ __forceinline__ void lma_score(const LmaPx& p, double phi2, double tol2, bool& in, double& err) {
    const double e2 = __builtin_fma(p.ge, phi2, p.a);
    in = e2 < tol2;
    err = (in && e2 >= kLmaSqrtMin) ? sqrt_core(e2) : 0.0;  // (in range where it is used: the select drops everything else)
}

// one listed pixel on the reference's exact recurrence along the planned radii (the operations of lm_pixel_t, standard functions):
// base terms and per planned step k the pixel's terms of {model change, |step|^2, candidate cost x 2, candidate rho^2, |J.r(candidate)|},
// and rho after every step
struct LmxWalk {
    double c0, g0;
    double m[kLmaKP], s2[kLmaKP], c[kLmaKP], x2[kLmaKP], g[kLmaKP], rho[kLmaKP + 1];
};
__device__ __forceinline__ void lmx_walk(double x, double y, double ux, double uy, double al, double ak, const Pose& pose, double two_over,
                                         const LmaPlan& plan, int steps, LmxWalk& o) {
    PixelModel m;
    m.init(x, y, ux, uy, al, ak, pose, two_over);
    const double s = 1.0 / (1.0 + sqrt(dot2(m.J0, m.J0, m.J1, m.J1)));
    const double jt0 = m.J0 * s, jt1 = m.J1 * s;
    const double ht = dot2(jt0, jt0, jt1, jt1);
    const double diag = clampd(ht, kMinLmDiag, kMaxLmDiag);
    double rho = 1.0, r0, r1;
    m.residual(rho, r0, r1);
    o.c0 = acc_sq2(0.0, r0, r1);
    o.g0 = fabs(dot2(m.J0, r0, m.J1, r1));
    o.rho[0] = rho;
#pragma unroll
    for (int k = 0; k < kLmaKP; ++k) {
        if (k < steps) {
            const double gt = dot2(jt0, r0, jt1, r1);
            const double step = -(gt / lm_denominator(ht, diag, 1.0 / plan.radius[k]));
            const double m0 = jt0 * step, m1 = jt1 * step;
            double mc = 0.0;
#if RSDSFM_FUSED
            mc = __builtin_fma(-m0, __builtin_fma(m0, 0.5, r0), __builtin_fma(-m1, __builtin_fma(m1, 0.5, r1), mc));
#else
            mc -= m0 * (r0 + m0 / 2.0) + m1 * (r1 + m1 / 2.0);
#endif
            o.m[k] = mc;
            const double cand = mad(step, s, rho);
            const double dx = rho - cand;
            o.s2[k] = acc_sq(0.0, dx);
            m.residual(cand, r0, r1);
            o.c[k] = acc_sq2(0.0, r0, r1);
            o.x2[k] = acc_sq(0.0, cand);
            o.g[k] = fabs(dot2(m.J0, r0, m.J1, r1));
            rho = cand;
        } else {
            o.m[k] = o.s2[k] = o.c[k] = o.x2[k] = o.g[k] = 0.0;
        }
        o.rho[k + 1] = rho;
    }
}

}  // namespace rsdsfm
