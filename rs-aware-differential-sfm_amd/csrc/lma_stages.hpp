// lma_stages.hpp -- the stages behind a pixel pass on the analytic LM trajectory (lma_common.hpp), shared by the RANSAC's depth solves
// (ransac_lma_kernels.hip) and the dense depth solve (depth_lma_kernels.hip):
//   lma_rows_stage  one workgroup (256 threads): fixed-order reduction of the pixel pass's partial rows of one solve / hypothesis, and the listed
//                   pixels (guards a / b) on the reference's exact recurrence, sorted by pixel index -> the row the decide stage consumes
//   lma_decide      Ceres' trust-region loop on the closed forms of that row (one lane), guard (c)
#pragma once

#include "lma_common.hpp"

namespace rsdsfm {

constexpr int kLB = 256;  // workgroup size of the stages (and of the pixel passes)

struct PixIn {
    double x, y, ux, uy, al, ak;
};
// (32-bit BYTE offsets from uniform bases: the loads take the scalar base + 32-bit vector offset form -- no 64-bit address arithmetic per load)
__device__ __forceinline__ PixIn load_pix(const double2* __restrict__ q, const double2* __restrict__ u, const double* __restrict__ alpha,
                                          const double* __restrict__ alpha_k, unsigned i) {
    const unsigned o16 = i << 4, o8 = i << 3;
    const double2 qq = *reinterpret_cast<const double2*>(reinterpret_cast<const char*>(q) + o16);
    const double2 uu = *reinterpret_cast<const double2*>(reinterpret_cast<const char*>(u) + o16);
    PixIn p;
    p.x = qq.x, p.y = qq.y, p.ux = uu.x, p.uy = uu.y;
    p.al = *reinterpret_cast<const double*>(reinterpret_cast<const char*>(alpha) + o8);
    p.ak = *reinterpret_cast<const double*>(reinterpret_cast<const char*>(alpha_k) + o8);
    return p;
}

// ---------------------------------------------------------------------------------------------------
// rows: reduction of the partial rows + the listed pixels
// ---------------------------------------------------------------------------------------------------
// (256 threads, hypothesis t) -> row[kLmaRow] in LDS `s_row`
__device__ __forceinline__ void lma_rows_stage(const double* __restrict__ partials, int nblocks, int T, int t, const double2* __restrict__ q,
                                               const double2* __restrict__ u, const double* __restrict__ alpha, const double* __restrict__ alpha_k,
                                               const Pose& pose, const LmaCand& cd, const int* __restrict__ irr_count,
                                               const int* __restrict__ irr_list, double* s_row) {
    __shared__ double s_red3[768];
    __shared__ int s_list[kLmaListCap], s_sorted[kLmaListCap];
    constexpr int LBATCH = 64;       // listed pixels per round (one wave walks them; lists are short)
    __shared__ double s_x[LBATCH][2 + 5 * kLmaKP + 2 * kLmaNC + 2 + 1];  // per listed pixel of the current batch: its terms of the row (+ 1: bank padding)
    const int tid = threadIdx.x;
    const int raw = irr_count[t];  // (requested here: its round trip overlaps the rows')
    // ---- the workgroups' partial rows of hypothesis t: [nblocks][kLmaSlots] doubles, contiguous.  Thread `tid` walks the flat array with the
    // stride 768 = 64 rows (a multiple of kLmaSlots and of the workgroup): its three elements per stride keep their slots, every load is
    // coalesced, 12 are in flight; then slot sl adds its 64 (thread, m) partial sums in a fixed order.
    {
        static_assert(768 % kLmaSlots == 0 && 768 % kLB == 0, "the stride keeps every thread on its slots");
        const double* __restrict__ base = partials + (int64_t)t * nblocks * kLmaSlots;
        const int total = nblocks * kLmaSlots;
        double acc[3] = {0.0, 0.0, 0.0};
        bool mx[3];
#pragma unroll
        for (int m = 0; m < 3; ++m) mx[m] = ((tid + kLB * m) % kLmaSlots) == kLmaG;
        constexpr int UB = 20;  // strides per batch: 60 loads in flight per thread -- 1280 rows in ONE round trip (the stage is bound by their latency)
        for (int f0 = 0; f0 < total; f0 += UB * 768) {
            double vv[UB][3];
#pragma unroll
            for (int j = 0; j < UB; ++j)
#pragma unroll
                for (int m = 0; m < 3; ++m) {
                    const int f = f0 + j * 768 + m * kLB + tid;
                    vv[j][m] = f < total ? base[f] : 0.0;  // (sums: + 0.0; the maximum is of absolute values)
                }
#pragma unroll
            for (int j = 0; j < UB; ++j)
#pragma unroll
                for (int m = 0; m < 3; ++m) acc[m] = mx[m] ? fmax(acc[m], vv[j][m]) : acc[m] + vv[j][m];
        }
#pragma unroll
        for (int m = 0; m < 3; ++m) s_red3[m * kLB + tid] = acc[m];
        __syncthreads();
        if (tid < kLmaSlots) {
            double r = s_red3[tid];
            for (int e = tid + kLmaSlots; e < 768; e += kLmaSlots) r = tid == kLmaG ? fmax(r, s_red3[e]) : r + s_red3[e];
            // A B C D E G -> row[0..5]; the fused scores -> row[kLmaRowScore ..]
            if (tid < 6) s_row[tid] = r;
            else s_row[kLmaRowScore + (tid - 6)] = r;
        }
        if (tid >= 6 && tid < kLmaRowScore) s_row[tid] = 0.0;
    }
    // ---- the listed pixels, sorted by pixel index (rank sort: the indices of a hypothesis are distinct)
    const int nl = min(raw, kLmaListCap);
    for (int i = tid; i < nl; i += kLB) s_list[i] = irr_list[(int64_t)t * kLmaListCap + i];
    __syncthreads();
    for (int i = tid; i < nl; i += kLB) {
        const int mine = s_list[i];
        int rank = 0;
        for (int j = 0; j < nl; ++j) rank += s_list[j] < mine ? 1 : 0;
        s_sorted[rank] = mine;
    }
    __syncthreads();
    if (tid == 0 && raw > kLmaListCap) s_row[7] = 1.0;  // overflow: guard (c)
    if (nl == 0) return;  // (uniform)
    const double two_over = 2.0 / (2.0 + pose.k);
    const LmaPlan& plan = cd.plan;
    constexpr int XW = 2 + 5 * kLmaKP + 2 * kLmaNC + 2;
    for (int base = 0; base < nl; base += LBATCH) {
        const int e = base + tid;
        double* xr = s_x[tid < LBATCH ? tid : 0];
        if (tid >= LBATCH) {
        } else if (e < nl) {
            const int64_t i = s_sorted[e];  // (a pixel index of the whole input: 64-bit addressing)
            const double2 qq = q[i], uu = u[i];
            PixIn px;
            px.x = qq.x, px.y = qq.y, px.ux = uu.x, px.uy = uu.y, px.al = alpha[i], px.ak = alpha_k[i];
            const LmaPx v = lma_pixel(px.x, px.y, px.ux, px.uy, px.al, px.ak, pose, two_over);  // the bits the pixel pass saw
            LmxWalk wk;
            lmx_walk(px.x, px.y, px.ux, px.uy, px.al, px.ak, pose, two_over, plan, kLmaKP, wk);
            // clamped: its exact terms enter the row, its frozen closed-form terms (a = |r(1)|^2, rho* = 1) leave it
            xr[0] = v.clamped ? wk.c0 : 0.0;
            xr[1] = v.clamped ? wk.g0 : 0.0;
#pragma unroll
            for (int k = 0; k < kLmaKP; ++k) {
                xr[2 + 5 * k + 0] = v.clamped ? wk.m[k] : 0.0;
                xr[2 + 5 * k + 1] = v.clamped ? wk.s2[k] : 0.0;
                xr[2 + 5 * k + 2] = v.clamped ? wk.c[k] : 0.0;
                xr[2 + 5 * k + 3] = v.clamped ? wk.x2[k] : 0.0;
                xr[2 + 5 * k + 4] = v.clamped ? wk.g[k] : 0.0;
            }
            // scores: the exact iterate's in place of the closed form's, at every fused iterate
#pragma unroll
            for (int c = 0; c < kLmaNC; ++c) {
                double dc = 0.0, de = 0.0;
                if (c < cd.nc) {
                    bool in_a;
                    double err_a;
                    lma_score(v, cd.phi2[c], cd.tol2, in_a, err_a);
                    const double ex = point_error(px.x, px.y, px.ux, px.uy, px.al, px.ak, pose, two_over, wk.rho[cd.steps[c]]);
                    const bool in_x = ex < cd.tol;
                    dc = (in_x ? 1.0 : 0.0) - (in_a ? 1.0 : 0.0);
                    de = cd.count_only ? 0.0 : (in_x ? ex : 0.0) - err_a;
                }
                xr[2 + 5 * kLmaKP + 2 * c] = dc;
                xr[2 + 5 * kLmaKP + 2 * c + 1] = de;
            }
            xr[XW - 2] = v.clamped ? v.a : 0.0;
            xr[XW - 1] = v.clamped ? 1.0 : 0.0;
        } else {
            for (int j = 0; j < XW; ++j) xr[j] = 0.0;
        }
        __syncthreads();
        // column j of the batch, in pixel order (one thread per column: the batch is short)
        if (tid < XW) {
            const int cntb = min(LBATCH, nl - base);
            const bool is_max = tid == 1 || (tid >= 2 && tid < 2 + 5 * kLmaKP && ((tid - 2) % 5) == 4);
            double r = 0.0;
            for (int j = 0; j < cntb; ++j) r = is_max ? fmax(r, s_x[j][tid]) : r + s_x[j][tid];
            if (tid < 2) s_row[kLmaRowX0 + tid] = is_max ? fmax(s_row[kLmaRowX0 + tid], r) : s_row[kLmaRowX0 + tid] + r;
            else if (tid < 2 + 5 * kLmaKP) s_row[kLmaRowXk + (tid - 2)] = is_max ? fmax(s_row[kLmaRowXk + (tid - 2)], r) : s_row[kLmaRowXk + (tid - 2)] + r;
            else if (tid < XW - 2) s_row[kLmaRowScore + (tid - 2 - 5 * kLmaKP)] += r;
            else if (tid == XW - 2) s_row[0] -= r;  // A' = A - sum of the clamped pixels' frozen terms
            else {
                s_row[3] -= r;  // D' = D - clamped pixels
                s_row[6] += r;
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------
// decide: the trust-region loop on the closed forms
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool lma_band(double x, double thr) { return fabs(x - thr) <= kLmaBand * fabs(thr); }  // (false for NaN)

// row: the hypothesis' row summed over the ranks.  Returns the guard that tripped (0: none) and fills st / hist / score.
__device__ __forceinline__ int lma_decide(const double* row, int64_t n, const LmaCand& cd, const LmaPlan& plan, bool nan_pose, LmScal& st, double* hist, bool& scored,
                                          double& count, double& err, double* phi_out = nullptr) {
    const double A = row[0], B = row[1], C = row[2], D = row[3], E = row[4], G = row[5];
    const bool listed_walk = row[6] != 0.0;  // clamped pixels walk the planned radii beside the closed form
    int fallback = 0;
    if (row[7] != 0.0) fallback = 8;
    if (__builtin_isinf(A) || __builtin_isinf(B) || __builtin_isinf(C) || __builtin_isinf(D) || __builtin_isinf(E) || __builtin_isinf(G)) fallback = 1;
    st.status = 1;
    st.restart = 0;
    st.n_hist = 0;
    st.K = 0;
    st.write_which = 0;
    st.iteration = 0;
    st.num_successful = 0;
    st.num_unsuccessful = 0;
    st.invalid_run = 0;
    st.termination = -1;
    st.rho_holds = -1;
    st.launches = 1;
    st.next_launch = 1;
    st.predict = 0;
    st.radius = kInitialRadius;
    st.decrease_factor = 2.0;
    double phi = 1.0;
    double XC = row[kLmaRowX0], Xg = row[kLmaRowX0 + 1];
    double cost = 0.5 * ((A + B) + XC), x_norm = sqrt((double)n);
    double gmax = fmax(G, Xg);
    st.initial_cost = cost;
    bool on_plan = true;  // the state is the planned iterate st.n_hist and the radius is the planned one
    if (lma_band(gmax, kGradientTol)) fallback = 2;
    if (n == 0 || gmax <= kGradientTol) st.termination = RSDSFM_TERM_GRADIENT;
    while (st.termination < 0 && !fallback) {
        if (st.iteration >= kMaxIter) {
            st.termination = RSDSFM_TERM_MAX_ITER;
            break;
        }
        if (st.radius <= kMinRadius) {
            st.termination = RSDSFM_TERM_MIN_RADIUS;
            break;
        }
        st.iteration += 1;
        const int kk = st.n_hist;
        on_plan = on_plan && kk < kLmaKP && st.radius == plan.radius[kk];
        if (listed_walk && !on_plan && B == B) {  // (NaN sums -- a NaN pixel -- make every step invalid whatever the listed pixels add)
            fallback = 9;
            break;
        }
        double psi, phic;
        if (on_plan) psi = plan.psi[kk], phic = plan.phi[kk + 1];  // (the same bits: the plan was computed with lma_phi_step -- no divisions on the serial path)
        else lma_phi_step(st.radius, phi, psi, phic);
        const double p2 = phi * phi, pc2 = phic * phic;
        const double* xk = row + kLmaRowXk + 5 * (kk < kLmaKP ? kk : 0);
        const double XM = listed_walk ? xk[0] : 0.0, XS = listed_walk ? xk[1] : 0.0, XCc = listed_walk ? xk[2] : 0.0;
        const double XX = listed_walk ? xk[3] : 0.0, Xgc = listed_walk ? xk[4] : 0.0;
        const double model_change = __builtin_fma(B * p2, psi * (1.0 - 0.5 * psi), XM);
        const double stepsq = __builtin_fma(C * p2, psi * psi, XS);
        if (!(model_change > 0.0)) {  // HandleInvalidStep
            if (model_change == model_change) {
                fallback = 3;
                break;
            }
            st.num_unsuccessful += 1;
            st.invalid_run += 1;
            if (st.invalid_run >= kMaxInvalid) {
                st.termination = RSDSFM_TERM_FAILURE;
                break;
            }
            st.radius *= 0.5;
            continue;
        }
        if (model_change < 1e-25 * cost) {
            fallback = 3;
            break;
        }
        st.invalid_run = 0;
        const double step_norm = sqrt(stepsq);
        const double ptol = kParameterTol * (x_norm + kParameterTol);
        if (lma_band(step_norm, ptol)) {
            fallback = 4;
            break;
        }
        if (step_norm <= ptol) {
            st.termination = RSDSFM_TERM_PARAMETER;
            break;
        }
        const double cost_change = 0.5 * __builtin_fma(B, p2 - pc2, XC - XCc);
        if (lma_band(fabs(cost_change), kFunctionTol * cost)) {
            fallback = 5;
            break;
        }
        if (fabs(cost_change) <= kFunctionTol * cost) {
            st.termination = RSDSFM_TERM_FUNCTION;
            break;
        }
        const double rel = cost_change / model_change;
        if (!(rel > 0.95)) {
            fallback = 6;
            break;
        }
        // HandleSuccessfulStep
        hist[st.n_hist] = st.radius;
        st.n_hist += 1;
        phi = phic;
        XC = XCc;
        Xg = Xgc;
        cost = 0.5 * (__builtin_fma(B, pc2, A) + XC);
        x_norm = sqrt(__builtin_fma(C, pc2, __builtin_fma(2.0 * E, phi, D)) + XX);
        gmax = fmax(G * phi, Xg);
        st.radius = radius_accept(st.radius, rel);
        st.decrease_factor = 2.0;
        st.num_successful += 1;
        if (lma_band(gmax, kGradientTol)) {
            fallback = 2;
            break;
        }
        if (gmax <= kGradientTol) st.termination = RSDSFM_TERM_GRADIENT;
    }
    st.cost = cost;
    st.rho_holds = -1;
    if (phi_out) *phi_out = phi;
    scored = false;
    count = err = 0.0;
    if (fallback) return fallback;
    // the score of the final iterate, where the pixel pass fused it (the iterate after n_hist steps ON THE PLAN)
    if (nan_pose) {  // every error is NaN: no inlier at any iterate
        scored = true;
        return 0;
    }
    const double phi2 = phi * phi;
    for (int c = 0; c < cd.nc; ++c)
        if (cd.steps[c] == st.n_hist && cd.phi2[c] == phi2) {
            count = row[kLmaRowScore + 2 * c];
            err = row[kLmaRowScore + 2 * c + 1];
            scored = true;
        }
    return 0;
}

}  // namespace rsdsfm
