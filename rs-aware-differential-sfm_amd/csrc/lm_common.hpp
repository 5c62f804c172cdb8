// lm_common.hpp -- pieces shared by the dense depth kernels and the batched RANSAC kernels:
// the closed-form per-pixel solve, the per-pixel speculative LM trajectory and the Ceres trust-region
// state machine that consumes its sums.  (reference: nonlinearRefinement.cc:109-180 + Ceres 1.14 defaults)
#pragma once

#include "device_math.hpp"
#include "rsdsfm_internal.hpp"

namespace rsdsfm {

__device__ __forceinline__ double closed_form_rho(double x, double y, double ux, double uy, double al, double ak,
                                                  const Pose& pose, double two_over) {
    PixelModel m;
    m.init(x, y, ux, uy, al, ak, pose, two_over);
    double r0, r1;
    m.residual(1.0, r0, r1);
    double h = m.J0 * m.J0 + m.J1 * m.J1;
    double g = m.J0 * r0 + m.J1 * r1;
    return (h > 0.0) ? 1.0 - g / h : 1.0;
}


// ---------------------------------------------------------------------------------------------------
// Ceres trust-region state machine on speculative sums
// ---------------------------------------------------------------------------------------------------
// sums layout: [0] = sum |r|^2 of the current state, [1] = sum rho^2, [2] = max |J.r| ; then for candidate
// j (0-based): base = 3 + 5 j : [sum |r(cand)|^2, model_cost_change, sum step^2, sum cand^2, max |J.r(cand)|]
// `first`: this is the launch of iteration zero (current state = rho == 1, nothing accepted yet).
// `used_K`, `used_write` : the plan the launch that produced `sums` ran with.
// `st` is a register copy of the scalar state; accepted radii are appended to `hist` (device memory).
__device__ __forceinline__ void lm_advance(LmScal& st, double* hist, const double* sums, int64_t n, int first, int used_K,
                                           int used_write, int launch_id) {
    if (first) {
        st.status = 0;
        st.restart = 0;
        st.n_hist = 0;
        st.iteration = 0;
        st.num_successful = 0;
        st.num_unsuccessful = 0;
        st.invalid_run = 0;
        st.termination = -1;
        st.rho_holds = -1;
        st.launches = 0;
        st.radius = kInitialRadius;
        st.decrease_factor = 2.0;
        st.initial_cost = 0.5 * sums[0];
        double r = kInitialRadius;
#pragma unroll
        for (int j = 0; j < KMAX; ++j) {  // the plan the first launch ran with (mirrors depth_lm_kernel)
            st.cand[j] = r;
            r = radius_accept(r, 1.0);
        }
    }
    st.launches += 1;
    const int base_hist = st.n_hist;
    double cost = 0.5 * sums[0];
    double x_norm = sqrt(sums[1]);
    int accepted_in_batch = 0;  // candidates 0..accepted_in_batch-1 were accepted in sequence
    if (first && (n == 0 || sums[2] <= kGradientTol)) st.termination = RSDSFM_TERM_GRADIENT;
#pragma unroll
    for (int j = 0; j < KMAX; ++j) {
        if (!(j < used_K && st.termination < 0)) break;
        if (st.iteration >= kMaxIter) {
            st.termination = RSDSFM_TERM_MAX_ITER;
            break;
        }
        if (st.radius <= kMinRadius) {
            st.termination = RSDSFM_TERM_MIN_RADIUS;
            break;
        }
        if (st.cand[j] != st.radius) break;  // speculation no longer matches the trust-region state: replan
        const double* s = sums + 3 + 5 * j;
        st.iteration += 1;
        const double model_change = s[1];
        const double ccost = 0.5 * s[0];
        if (!(model_change > 0.0)) {  // HandleInvalidStep
            st.num_unsuccessful += 1;
            st.invalid_run += 1;
            if (st.invalid_run >= kMaxInvalid) {
                st.termination = RSDSFM_TERM_FAILURE;
                break;
            }
            st.radius *= 0.5;
            break;
        }
        st.invalid_run = 0;
        const double step_norm = sqrt(s[2]);
        if (step_norm <= kParameterTol * (x_norm + kParameterTol)) {
            st.termination = RSDSFM_TERM_PARAMETER;
            break;
        }
        const double cost_change = cost - ccost;
        if (fabs(cost_change) <= kFunctionTol * cost) {
            st.termination = RSDSFM_TERM_FUNCTION;
            break;
        }
        const double rel = cost_change / model_change;
        if (rel > kMinRelDecrease) {  // HandleSuccessfulStep
            hist[st.n_hist] = st.radius;
            st.n_hist += 1;
            accepted_in_batch = j + 1;
            cost = ccost;
            x_norm = sqrt(s[3]);
            st.radius = radius_accept(st.radius, rel);
            st.decrease_factor = 2.0;
            st.num_successful += 1;
            if (s[4] <= kGradientTol) {
                st.termination = RSDSFM_TERM_GRADIENT;
                break;
            }
        } else {  // HandleUnsuccessfulStep
            st.num_unsuccessful += 1;
            st.radius = st.radius / st.decrease_factor;
            st.decrease_factor *= 2.0;
            break;
        }
    }
    st.cost = cost;
    // what does the output buffer hold after the launch that produced these sums?
    if (used_write <= accepted_in_batch) st.rho_holds = base_hist + used_write;
    else st.rho_holds = -1;
    if (st.termination < 0 && st.iteration >= kMaxIter) st.termination = RSDSFM_TERM_MAX_ITER;
    if (st.termination < 0 && st.radius <= kMinRadius) st.termination = RSDSFM_TERM_MIN_RADIUS;
    st.next_launch = launch_id + 1;
    if (st.termination >= 0) {
        st.predict = st.n_hist < KMAX ? st.n_hist : KMAX;
        if (st.rho_holds == st.n_hist) {
            st.status = 1;
        } else {
            st.status = 2;  // launch `next_launch` replays the accepted steps and writes the result
            st.K = 0;
            st.write_which = 0;
        }
    } else {  // continue: speculate the next KMAX iterations from the current trust-region state
        st.status = 0;
        st.K = KMAX;
        st.write_which = 0;
        double r = st.radius;
#pragma unroll
        for (int j = 0; j < KMAX; ++j) {
            st.cand[j] = r;
            r = radius_accept(r, 1.0);
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// fused LM kernel
// ---------------------------------------------------------------------------------------------------
struct LmPlanLds {
    int n_hist, K, write_which;
    double inv_hist[kMaxIter];  // 1 / radius of each accepted step
    double inv_cand[KMAX];      // 1 / radius of each speculated step
    __device__ __forceinline__ double inv_hist_at(int h) const { return inv_hist[h]; }
};

// register-resident copy of the plan (for kernels whose main loop must not touch LDS through compiler-emitted
// ds_reads, see depth_lm_dma_kernel); accepted steps beyond the 4th (rare) are read from `hist_lds`
struct LmPlanReg {
    int n_hist, K, write_which;
    double inv_cand[KMAX];
    double h0, h1, h2, h3;
    const double* hist_lds;
    __device__ __forceinline__ double inv_hist_at(int h) const {
        return h == 0 ? h0 : (h == 1 ? h1 : (h == 2 ? h2 : (h == 3 ? h3 : hist_lds[h])));
    }
};

// plan of launch 0 of a solve with the shape known at compile time (no accepted steps yet, K0 speculated iterations):
// lm_pixel becomes one straight-line block, which lets the compiler interleave the independent pixels of a lane.
template <int K0>
struct LmPlanFirstK {
    static constexpr int n_hist = 0;
    static constexpr int K = K0;
    int write_which;
    double inv_cand[KMAX];
    __device__ __forceinline__ double inv_hist_at(int) const { return 0.0; }
    __device__ __forceinline__ void load(const LmPlanLds& l) {
        write_which = l.write_which;
#pragma unroll
        for (int j = 0; j < KMAX; ++j) inv_cand[j] = l.inv_cand[j];
    }
};
using LmPlanFirst = LmPlanFirstK<KMAX>;

// plan of a pure replay with the number of accepted steps known at compile time (scoring: the accepted trust-region radii of a
// finished hypothesis, nothing speculated): lm_pixel becomes straight-line code and the pixel chains of a lane interleave
template <int NH>
struct LmPlanReplay {
    static constexpr int n_hist = NH;
    static constexpr int K = 0;
    static constexpr int write_which = 0;
    double ih[NH > 0 ? NH : 1];   // 1 / radius of each accepted step
    double inv_cand[KMAX];        // (never read: K == 0)
    __device__ __forceinline__ double inv_hist_at(int h) const { return ih[h]; }
};

// ht + clamp(diag) / radius: the damped 1x1 normal matrix of one pixel (inv_radius = 1 / radius)
__device__ __forceinline__ double lm_denominator(double ht, double diag, double inv_radius) {
#if RSDSFM_FUSED
    return __builtin_fma(diag, inv_radius, ht);
#else
    const double lam = diag * inv_radius;
    return ht + lam;
#endif
}

// one pixel through the planned LM trajectory; returns the state selected by write_which.
// Arithmetic mirrors oracle/rsdsfm_oracle.c rso_estimate_inverse_depths (mode 1) operation for operation.
struct NoHook {
    __device__ __forceinline__ void operator()(int, double, const PixelModel&) const {}
};

// `hook(j, rho_j, model)` is called with every speculated iterate (j = 0 .. K-1: the state after j+1 accepted steps)
// CORE: the Jacobi scaling through the in-range cores of sqrt and the reciprocal (device_math.hpp: the same bits for an argument in
// range); *worst tracks the range tests (sqrt_range_track) and the caller has the solve run again with CORE = false when one failed
template <bool CORE, class Plan, class Hook = NoHook>
__device__ __forceinline__ double lm_pixel_t(double x, double y, double ux, double uy, double al, double ak,
                                             const Pose& pose, double two_over, const Plan& plan,
                                             double (&acc)[NS], const Hook& hook = Hook(), PixelModel* model_out = nullptr, uint32_t* worst = nullptr) {
    PixelModel m;
    m.init(x, y, ux, uy, al, ak, pose, two_over);
    if (model_out) *model_out = m;
    double s;  // Jacobi scaling (iteration 0 Jacobian)
    if (CORE) {
        const double jj = dot2(m.J0, m.J0, m.J1, m.J1);
        s = rcp_core(1.0 + sqrt_core_z(jj, *worst));
    } else {
        s = 1.0 / (1.0 + sqrt(dot2(m.J0, m.J0, m.J1, m.J1)));
    }
    const double jt0 = m.J0 * s, jt1 = m.J1 * s;
    const double ht = dot2(jt0, jt0, jt1, jt1);
    const double diag = clampd(ht, kMinLmDiag, kMaxLmDiag);
    double rho = 1.0;  // nonlinearRefinement.cc:140
    double r0, r1;
    m.residual(rho, r0, r1);
    for (int h = 0; h < plan.n_hist; ++h) {  // replay the accepted steps
        const double gt = dot2(jt0, r0, jt1, r1);
        const double step = -(gt / lm_denominator(ht, diag, plan.inv_hist_at(h)));
        rho = mad(step, s, rho);
        m.residual(rho, r0, r1);
    }
    double out = rho;
    if (plan.K > 0) {
        acc[0] = acc_sq2(acc[0], r0, r1);
        acc[1] = acc_sq(acc[1], rho);
        acc[2] = fmax(acc[2], fabs(dot2(m.J0, r0, m.J1, r1)));
    }
#pragma unroll
    for (int j = 0; j < KMAX; ++j) {
        if (j < plan.K) {
            const double gt = dot2(jt0, r0, jt1, r1);
            const double step = -(gt / lm_denominator(ht, diag, plan.inv_cand[j]));
            const double m0 = jt0 * step, m1 = jt1 * step;
#if RSDSFM_FUSED
            acc[3 + 5 * j + 1] = __builtin_fma(-m0, __builtin_fma(m0, 0.5, r0), __builtin_fma(-m1, __builtin_fma(m1, 0.5, r1), acc[3 + 5 * j + 1]));
#else
            acc[3 + 5 * j + 1] -= m0 * (r0 + m0 / 2.0) + m1 * (r1 + m1 / 2.0);
#endif
            const double cand = mad(step, s, rho);
            const double dx = rho - cand;
            acc[3 + 5 * j + 2] = acc_sq(acc[3 + 5 * j + 2], dx);
            m.residual(cand, r0, r1);
            acc[3 + 5 * j + 0] = acc_sq2(acc[3 + 5 * j + 0], r0, r1);
            acc[3 + 5 * j + 3] = acc_sq(acc[3 + 5 * j + 3], cand);
            acc[3 + 5 * j + 4] = fmax(acc[3 + 5 * j + 4], fabs(dot2(m.J0, r0, m.J1, r1)));
            rho = cand;
            hook(j, cand, m);
            if (plan.write_which == j + 1) out = cand;
        }
    }
    return out;
}

template <class Plan, class Hook = NoHook>
__device__ __forceinline__ double lm_pixel(double x, double y, double ux, double uy, double al, double ak,
                                           const Pose& pose, double two_over, const Plan& plan,
                                           double (&acc)[NS], const Hook& hook = Hook(), PixelModel* model_out = nullptr) {
    return lm_pixel_t<false>(x, y, ux, uy, al, ak, pose, two_over, plan, acc, hook, model_out);
}

__device__ __forceinline__ bool is_max_slot(int s) { return s == 2 || (s >= 3 && ((s - 3) % 5) == 4); }


}  // namespace rsdsfm
