// refine_kernels.hip -- joint nonlinear refinement of (v, w[, k], rho_i) on MI355X (gfx950).
//
// Replaces nonlinear_refinement::nonLinearRefinement (reference nonlinearRefinement.cc:183-252): one Ceres problem
// over the inliers, trust-region LM with DENSE_SCHUR.  The rho_i are the 1x1 e-blocks, (v, w[, k]) the f-blocks,
// so every LM iteration is two streaming passes over the inliers plus two single-workgroup kernels:
//
//   refine_schur_kernel    per inlier: residual + analytic 2x(NP+1) Jacobian (Jacobi-scaled), LM-damped 1x1
//                          e-block inverse, contribution to F^T F, F^T E (E^T E)^-1 E^T F, F^T b, ... (70 sums for
//                          NP = 7), DPP/LDS block reduction -> partials[block][NSCHUR]
//   refine_solve_kernel    fixed-order reduction, reduced NP x NP system + LM diagonal, Cholesky, candidate (v,w,k)
//   refine_backsub_kernel  per inlier: back-substitution of rho_i, model cost change, candidate residual / cost and
//                          (speculatively) the gradient / norms at the candidate -> partials[block][NBACK]
//   refine_decide_kernel   fixed-order reduction + the Ceres trust-region decisions (accept / reject / converge)
//
// rho lives in two device buffers that swap on acceptance (no copy).  All state is device-resident
// (RefineState); the host only polls the termination flag every few iterations.
// Arithmetic mirrors oracle/rsdsfm_oracle.c rso_refine operation for operation (per-inlier terms bit-identical;
// the global sums differ only in summation order).
#include "refine_common.hpp"

namespace rsdsfm {


// ---------------------------------------------------------------------------------------------------
// iteration zero
// ---------------------------------------------------------------------------------------------------
// gathers the flow of each inlier (nonlinearRefinement.cc:209-213; quirk Q2: flow(., rank) unless gathered), sets
// rho = 1/z, Jacobi scale of the rho columns, and the iteration-zero sums.
template <int NP>
__global__ __launch_bounds__(kFB) void refine_init_kernel(const double2* __restrict__ flow, int64_t n_flow, int64_t m,
                                                         const double* __restrict__ inl, const double* __restrict__ alpha,
                                                         const double* __restrict__ alpha_k,
                                                         const int64_t* __restrict__ inlier_idx, int flow_index_mode,
                                                         const RefineState* __restrict__ st, double4* __restrict__ xyuv,
                                                         double* __restrict__ beta_out, double* __restrict__ rho0,
                                                         double* __restrict__ srho, double* __restrict__ partials,
                                                         int* __restrict__ bad_index, int want_zsum) {
    using CT = Counts<NP>;
    __shared__ double s_red[kFB / 64][CT::NINIT];
    const PassShape ps = pass_shape(st, m);
    if (!ps.live) return;
    m = ps.m;
    double p[7];
#pragma unroll
    for (int c = 0; c < 7; ++c) p[c] = st->p[c];
    double acc[CT::NINIT];
#pragma unroll
    for (int s = 0; s < CT::NINIT; ++s) acc[s] = 0.0;
    const int64_t stride = (int64_t)ps.grid * kFB;
    for (int64_t i = (int64_t)blockIdx.x * kFB + threadIdx.x; i < m; i += stride) {
        int64_t fi = (flow_index_mode == RSDSFM_FLOW_GATHERED) ? inlier_idx[i] : i;
        if (fi < 0 || fi >= n_flow) {
            *bad_index = 1;
            fi = 0;
        }
        const double2 f = flow[fi];
        const double x = inl[3 * i], y = inl[3 * i + 1];
        // the per-inlier constants of every later pass as ONE 32-byte record (the passes are HBM-bound: x, y no longer come out
        // of the 24-byte (x, y, z) triples, and with k fixed beta replaces alpha and alpha_k: 56 instead of 72 bytes per inlier)
        xyuv[i] = make_double4(x, y, f.x, f.y);
        if (NP == 6) beta_out[i] = beta_of(alpha[i], alpha_k[i], p[6]);
        const double rho = 1.0 / inl[3 * i + 2];
        rho0[i] = rho;
        RJ<NP> o;
        resid_jac<NP>(x, y, f.x, f.y, alpha[i], alpha_k[i], p, rho, o);
        acc[0] += o.r[0] * o.r[0] + o.r[1] * o.r[1];
#pragma unroll
        for (int c = 0; c < NP; ++c) {
            acc[1 + c] += o.Jp[0][c] * o.Jp[0][c] + o.Jp[1][c] * o.Jp[1][c];
            acc[1 + NP + c] += o.Jp[0][c] * o.r[0] + o.Jp[1][c] * o.r[1];
        }
        srho[i] = 1.0 / (1.0 + sqrt(o.Jr[0] * o.Jr[0] + o.Jr[1] * o.Jr[1]));
        acc[CT::INIT_MAX] = fmax(acc[CT::INIT_MAX], fabs(o.Jr[0] * o.r[0] + o.Jr[1] * o.r[1]));
        acc[CT::INIT_MAX + 1] += rho * rho;
        if (want_zsum) acc[CT::INIT_MAX + 2] += 1.0 / rho;  // z as refine_finish_kernel forms it
    }
    block_reduce_store<CT::NINIT>(acc, CT::INIT_MAX, s_red, partials + (int64_t)blockIdx.x * CT::NINIT);
}

template <int NP>
__global__ __launch_bounds__(kFB) void refine_init_decide_kernel(const double* __restrict__ partials, int nblocks,
                                                                RefineState* st, int64_t m, const int64_t* __restrict__ m_dev = nullptr) {
    using CT = Counts<NP>;
    __shared__ double s_red[kFB / 64][CT::NINIT];
    __shared__ double s[CT::NINIT];
    if (nblocks < 0) nblocks = st->grid, m = st->m;  // device-resident shape (see pass_shape)
    if (m_dev) m = *m_dev;  // (column-tiled solve ahead of the host's read of the RANSAC result: the GLOBAL inlier count, device-resident)
    reduce_partials<CT::NINIT>(partials, nblocks, CT::INIT_MAX, s_red, s);
    if (threadIdx.x == 0) {
        double gmax = s[CT::INIT_MAX], xsq = s[CT::INIT_MAX + 1];
        for (int c = 0; c < NP; ++c) {
            st->sp[c] = 1.0 / (1.0 + sqrt(s[1 + c]));
            if (fabs(s[1 + NP + c]) > gmax) gmax = fabs(s[1 + NP + c]);
            xsq += st->p[c] * st->p[c];
        }
        st->cost = 0.5 * s[0];
        st->zsum = s[CT::INIT_MAX + 2];
        st->initial_cost = st->cost;
        st->gmax = gmax;
        st->x_norm = sqrt(xsq);
        st->radius = kInitialRadius;
        st->decrease_factor = 2.0;
        st->iteration = 0;
        st->invalid_run = 0;
        st->num_successful = 0;
        st->num_unsuccessful = 0;
        st->cur = 0;
        st->solve_ok = 0;
        st->termination = (m == 0 || gmax <= kGradientTol) ? RSDSFM_TERM_GRADIENT : -1;
    }
}

// ---------------------------------------------------------------------------------------------------
// pass 1: Schur complement sums
// ---------------------------------------------------------------------------------------------------
// one inlier's contribution to the 2 TRI + 2 NP sums, from its residual / Jacobian `o` at the point the sums are taken at, its Jacobi
// scale `sr`, the parameter columns' scales `sp` and 1 / radius
template <int NP>
__device__ __forceinline__ void schur_accumulate(const RJ<NP>& o, double sr, const double (&sp)[NP], double inv_radius,
                                                 double (&acc)[NP * (NP + 1) + 2 * NP]) {
    constexpr int TRI = NP * (NP + 1) / 2;
    const double E0 = o.Jr[0] * sr, E1 = o.Jr[1] * sr;
    const double ht = E0 * E0 + E1 * E1;
    const double lam = clampd(ht, kMinLmDiag, kMaxLmDiag) * inv_radius;
    const double ete_inv = 1.0 / (ht + lam);
    const double Etb = E0 * o.r[0] + E1 * o.r[1];
    double F0[NP], F1[NP], EtF[NP];
#pragma unroll
    for (int c = 0; c < NP; ++c) {
        F0[c] = o.Jp[0][c] * sp[c];
        F1[c] = o.Jp[1][c] * sp[c];
        EtF[c] = E0 * F0[c] + E1 * F1[c];
    }
    int tri = 0;
#pragma unroll
    for (int a = 0; a < NP; ++a) {
        acc[2 * TRI + a] += F0[a] * o.r[0] + F1[a] * o.r[1];
        acc[2 * TRI + NP + a] += EtF[a] * (ete_inv * Etb);
#pragma unroll
        for (int b = a; b < NP; ++b) {
            acc[tri] += F0[a] * F0[b] + F1[a] * F1[b];
            acc[TRI + tri] += EtF[a] * (ete_inv * EtF[b]);
            ++tri;
        }
    }
}

template <int NP>
__global__ __launch_bounds__(kFB) void refine_schur_kernel(int64_t m, const double4* __restrict__ xyuv,
                                                          const double* __restrict__ beta_in, const double* __restrict__ alpha,
                                                          const double* __restrict__ alpha_k,
                                                          const double* __restrict__ rho_a, const double* __restrict__ rho_b,
                                                          const double* __restrict__ srho, const RefineState* __restrict__ st,
                                                          double* __restrict__ partials) {
    using CT = Counts<NP>;
    __shared__ double s_red[kFB / 64][CT::NSCHUR];
    if (st->termination >= 0 || st->iteration >= kMaxIter || st->radius <= kMinRadius) return;
    const PassShape ps = pass_shape(st, m);
    if (!ps.live) return;
    m = ps.m;
    double p[7], sp[NP];
#pragma unroll
    for (int c = 0; c < 7; ++c) p[c] = st->p[c];
#pragma unroll
    for (int c = 0; c < NP; ++c) sp[c] = st->sp[c];
    const double inv_radius = 1.0 / st->radius;
    const double* __restrict__ rho = st->cur ? rho_b : rho_a;
    double acc[CT::NSCHUR];
#pragma unroll
    for (int s = 0; s < CT::NSCHUR; ++s) acc[s] = 0.0;
    const int64_t stride = (int64_t)ps.grid * kFB;
    for (int64_t i = (int64_t)blockIdx.x * kFB + threadIdx.x; i < m; i += stride) {
        const double4 c4 = xyuv[i];
        RJ<NP> o;
        if (NP == 6)
            resid_jac_beta<NP>(c4.x, c4.y, c4.z, c4.w, beta_in[i], 0.0, p, rho[i], o);
        else
            resid_jac<NP>(c4.x, c4.y, c4.z, c4.w, alpha[i], alpha_k[i], p, rho[i], o);
        schur_accumulate<NP>(o, srho[i], sp, inv_radius, acc);
    }
    block_reduce_store<CT::NSCHUR>(acc, -1, s_red, partials + (int64_t)blockIdx.x * CT::NSCHUR);
}

// the serial tail of the reduced solve (one lane): damped Schur complement from the NSCHUR sums `s`, Cholesky, candidate parameters
template <int NP>
__device__ __forceinline__ void solve_serial(RefineState* st, const double* s, const double (&p_cur)[7], const double (&sp_cur)[NP], double radius) {
    using CT = Counts<NP>;
    st->iteration += 1;
    const double inv_radius = 1.0 / radius;
    // reduced system in registers (all loops fully unrolled: static indices, no LDS round trips in the serial chain)
    double sv[CT::NSCHUR];
#pragma unroll
    for (int i = 0; i < CT::NSCHUR; ++i) sv[i] = s[i];
    double S[NP][NP], rhs[NP], yv[NP], yp[NP];
    {
        int tri = 0;
#pragma unroll
        for (int a = 0; a < NP; ++a) {
            rhs[a] = sv[2 * CT::TRI + a] - sv[2 * CT::TRI + NP + a];
#pragma unroll
            for (int b = a; b < NP; ++b) {
                double sab = sv[tri] - sv[CT::TRI + tri];
                if (a == b) sab += clampd(sv[tri], kMinLmDiag, kMaxLmDiag) * inv_radius;  // D_f^2
                S[a][b] = sab;
                S[b][a] = sab;
                ++tri;
            }
        }
    }
    // dense Cholesky solve (mirrors the oracle's chol_solve operation for operation)
    bool ok = true;
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        if (ok) {
            double d = S[j][j];
#pragma unroll
            for (int t = 0; t < j; ++t) d -= S[j][t] * S[j][t];
            if (!(d > 0.0)) {
                ok = false;
            } else {
                d = sqrt(d);
                S[j][j] = d;
#pragma unroll
                for (int i = j + 1; i < NP; ++i) {
                    double sacc = S[i][j];
#pragma unroll
                    for (int t = 0; t < j; ++t) sacc -= S[i][t] * S[j][t];
                    S[i][j] = sacc / d;
                }
            }
        }
    }
    if (ok) {
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            double sacc = rhs[i];
#pragma unroll
            for (int t = 0; t < i; ++t) sacc -= S[i][t] * yv[t];
            yv[i] = sacc / S[i][i];
        }
#pragma unroll
        for (int i = NP - 1; i >= 0; --i) {
            double sacc = yv[i];
#pragma unroll
            for (int t = i + 1; t < NP; ++t) sacc -= S[t][i] * yp[t];
            yp[i] = sacc / S[i][i];
        }
        double stepsq = 0.0;
        double pc_new[7];
#pragma unroll
        for (int c = 0; c < 7; ++c) pc_new[c] = p_cur[c];
#pragma unroll
        for (int c = 0; c < NP; ++c) {
            const double step = -yp[c];
            const double pc = p_cur[c] + step * sp_cur[c];
            pc_new[c] = pc;
            const double dx = p_cur[c] - pc;
            stepsq += dx * dx;
        }
#pragma unroll
        for (int c = 0; c < 7; ++c) st->pc[c] = pc_new[c];
#pragma unroll
        for (int c = 0; c < NP; ++c) st->yp[c] = yp[c];
        st->stepsq_p = stepsq;
    }
    st->solve_ok = ok ? 1 : 0;
}

// reduced system + Cholesky (one workgroup; the solve itself runs on one lane: NP <= 7)
template <int NP>
__global__ __launch_bounds__(kFB) void refine_solve_kernel(const double* __restrict__ partials, int nblocks,
                                                          RefineState* st) {
    using CT = Counts<NP>;
    __shared__ double s_red[kFB / 64][CT::NSCHUR];
    __shared__ double s[CT::NSCHUR];
    if (st->termination >= 0) return;
    if (st->iteration >= kMaxIter) {  // top-of-loop checks of TrustRegionMinimizer
        if (threadIdx.x == 0) st->termination = RSDSFM_TERM_MAX_ITER;
        return;
    }
    if (st->radius <= kMinRadius) {
        if (threadIdx.x == 0) st->termination = RSDSFM_TERM_MIN_RADIUS;
        return;
    }
    // the state the serial tail needs is requested before the reduction so that its latency is hidden
    double p_cur[7], sp_cur[NP];
    double radius = 0.0;
    if (threadIdx.x == 0) {
        radius = st->radius;
#pragma unroll
        for (int c = 0; c < 7; ++c) p_cur[c] = st->p[c];
#pragma unroll
        for (int c = 0; c < NP; ++c) sp_cur[c] = st->sp[c];
    }
    reduce_partials<CT::NSCHUR>(partials, nblocks >= 0 ? nblocks : st->grid, -1, s_red, s);
    if (threadIdx.x == 0) solve_serial<NP>(st, s, p_cur, sp_cur, radius);
}

// ---------------------------------------------------------------------------------------------------
// pass 2: back-substitution + candidate evaluation
// ---------------------------------------------------------------------------------------------------
template <int NP>
__global__ __launch_bounds__(kFB) void refine_backsub_kernel(int64_t m, const double4* __restrict__ xyuv,
                                                            const double* __restrict__ beta_in, const double* __restrict__ alpha,
                                                            const double* __restrict__ alpha_k, double* __restrict__ rho_a,
                                                            double* __restrict__ rho_b, const double* __restrict__ srho,
                                                            const RefineState* __restrict__ st, double* __restrict__ partials, int want_zsum) {
    using CT = Counts<NP>;
    __shared__ double s_red[kFB / 64][CT::NBACK];
    if (st->termination >= 0 || !st->solve_ok) return;
    const PassShape ps = pass_shape(st, m);
    if (!ps.live) return;
    m = ps.m;
    double p[7], pc[7], sp[NP], yp[NP];
#pragma unroll
    for (int c = 0; c < 7; ++c) {
        p[c] = st->p[c];
        pc[c] = st->pc[c];
    }
#pragma unroll
    for (int c = 0; c < NP; ++c) {
        sp[c] = st->sp[c];
        yp[c] = st->yp[c];
    }
    const double inv_radius = 1.0 / st->radius;
    const double* __restrict__ rho = st->cur ? rho_b : rho_a;
    double* __restrict__ cand = st->cur ? rho_a : rho_b;
    double acc[CT::NBACK];
#pragma unroll
    for (int s = 0; s < CT::NBACK; ++s) acc[s] = 0.0;
    const int64_t stride = (int64_t)ps.grid * kFB;
    for (int64_t i = (int64_t)blockIdx.x * kFB + threadIdx.x; i < m; i += stride) {
        const double4 c4 = xyuv[i];
        const double x = c4.x, y = c4.y;
        const double rh = rho[i];
        // beta at the current and at the candidate parameters: read once when k is fixed, recomputed from alpha / alpha_k otherwise
        double be, dbe = 0.0, bec, dbec = 0.0;
        if (NP == 6) {
            be = bec = beta_in[i];
        } else {
            const double al = alpha[i], ak = alpha_k[i];
            be = beta_of(al, ak, p[6]), dbe = dbeta_of(al, ak, p[6]);
            bec = beta_of(al, ak, pc[6]), dbec = dbeta_of(al, ak, pc[6]);
        }
        RJ<NP> o;
        resid_jac_beta<NP>(x, y, c4.z, c4.w, be, dbe, p, rh, o);
        const double sr = srho[i];
        const double E0 = o.Jr[0] * sr, E1 = o.Jr[1] * sr;
        const double ht = E0 * E0 + E1 * E1;
        const double lam = clampd(ht, kMinLmDiag, kMaxLmDiag) * inv_radius;
        const double ete_inv = 1.0 / (ht + lam);
        const double Etb = E0 * o.r[0] + E1 * o.r[1];
        double Fy0 = 0.0, Fy1 = 0.0;
#pragma unroll
        for (int c = 0; c < NP; ++c) {
            Fy0 += o.Jp[0][c] * sp[c] * yp[c];
            Fy1 += o.Jp[1][c] * sp[c] * yp[c];
        }
        const double ye = ete_inv * (Etb - (E0 * Fy0 + E1 * Fy1));
        const double step_e = -ye;
        const double m0 = -Fy0 + E0 * step_e, m1 = -Fy1 + E1 * step_e;
        acc[0] -= m0 * (o.r[0] + m0 / 2.0) + m1 * (o.r[1] + m1 / 2.0);
        const double cd = rh + step_e * sr;
        cand[i] = cd;
        const double dx = rh - cd;
        acc[1] += dx * dx;
        RJ<NP> oc;
        resid_jac_beta<NP>(x, y, c4.z, c4.w, bec, dbec, pc, cd, oc);
        const double c2 = oc.r[0] * oc.r[0] + oc.r[1] * oc.r[1];
        acc[2] += c2;
        // quantities of HandleSuccessfulStep at the candidate (used only if the step is accepted)
        acc[3] += c2;
#pragma unroll
        for (int c = 0; c < NP; ++c) acc[4 + c] += oc.Jp[0][c] * oc.r[0] + oc.Jp[1][c] * oc.r[1];
        acc[CT::BACK_MAX] = fmax(acc[CT::BACK_MAX], fabs(oc.Jr[0] * oc.r[0] + oc.Jr[1] * oc.r[1]));
        acc[CT::BACK_MAX + 1] += cd * cd;
        if (want_zsum) acc[CT::BACK_MAX + 2] += 1.0 / cd;
    }
    block_reduce_store<CT::NBACK>(acc, CT::BACK_MAX, s_red, partials + (int64_t)blockIdx.x * CT::NBACK);
}

// the serial part of the trust-region decision (one lane): `s` = the NBACK sums (read only when the reduced system had factored);
// returns 1 when the step was ACCEPTED and the solve goes on (the state then sits at the candidate, with the radius Ceres' rule gives it)
template <int NP>
__device__ __forceinline__ int decide_serial(RefineState* st, const double* s, int solve_ok, double* __restrict__ trace, int trace_rows) {
    using CT = Counts<NP>;
    const int successful_before = st->num_successful;
    // optional trace (rsdsfm_set_refine_trace): one row of kRefineTraceCols doubles per LM iteration, see include/rsdsfm.h
    double* tr = (trace && st->iteration >= 1 && st->iteration <= trace_rows) ? trace + (int64_t)(st->iteration - 1) * kRefineTraceCols : nullptr;
    const double model_change = solve_ok ? s[0] : 0.0;
    if (tr) {
        tr[0] = (double)st->iteration;
        tr[1] = st->cost;
        tr[2] = tr[4] = tr[6] = __builtin_nan("");
        tr[3] = model_change;
        tr[5] = st->radius;
    }
    if (!solve_ok || !(model_change > 0.0)) {  // HandleInvalidStep
        if (tr) tr[7] = RSDSFM_TRACE_INVALID;
        st->num_unsuccessful += 1;
        st->invalid_run += 1;
        if (st->invalid_run >= kMaxInvalid) {
            st->termination = RSDSFM_TERM_FAILURE;
            return 0;
        }
        st->radius *= 0.5;
        return 0;
    }
    st->invalid_run = 0;
    const double step_norm = sqrt(st->stepsq_p + s[1]);
    const double ccost = 0.5 * s[2];
    if (tr) {
        tr[2] = ccost;
        tr[6] = step_norm;
    }
    if (step_norm <= kParameterTol * (st->x_norm + kParameterTol)) {
        if (tr) tr[7] = RSDSFM_TRACE_PARAMETER_TOL;
        st->termination = RSDSFM_TERM_PARAMETER;
        return 0;
    }
    const double cost_change = st->cost - ccost;
    if (fabs(cost_change) <= kFunctionTol * st->cost) {
        if (tr) tr[7] = RSDSFM_TRACE_FUNCTION_TOL;
        st->termination = RSDSFM_TERM_FUNCTION;
        return 0;
    }
    const double rel = cost_change / model_change;
    if (tr) tr[4] = rel;
    if (rel > kMinRelDecrease) {  // HandleSuccessfulStep
        double gmax = s[CT::BACK_MAX], xsq = s[CT::BACK_MAX + 1];
        for (int c = 0; c < 7; ++c) st->p[c] = st->pc[c];
        for (int c = 0; c < NP; ++c) {
            xsq += st->p[c] * st->p[c];
            if (fabs(s[4 + c]) > gmax) gmax = fabs(s[4 + c]);
        }
        st->cur ^= 1;
        st->zsum = s[CT::BACK_MAX + 2];
        st->cost = 0.5 * s[3];
        st->gmax = gmax;
        st->x_norm = sqrt(xsq);
        st->radius = radius_accept(st->radius, rel);
        st->decrease_factor = 2.0;
        st->num_successful += 1;
        if (gmax <= kGradientTol) st->termination = RSDSFM_TERM_GRADIENT;
        if (tr) tr[7] = gmax <= kGradientTol ? RSDSFM_TRACE_ACCEPTED_GRADIENT_TOL : RSDSFM_TRACE_ACCEPTED;
    } else {  // HandleUnsuccessfulStep
        if (tr) tr[7] = RSDSFM_TRACE_REJECTED;
        st->num_unsuccessful += 1;
        st->radius = st->radius / st->decrease_factor;
        st->decrease_factor *= 2.0;
    }
    return st->termination < 0 && st->num_successful > successful_before ? 1 : 0;
}

template <int NP>
__global__ __launch_bounds__(kFB) void refine_decide_kernel(const double* __restrict__ partials, int nblocks,
                                                           RefineState* st, double* __restrict__ trace, int trace_rows) {
    using CT = Counts<NP>;
    __shared__ double s_red[kFB / 64][CT::NBACK];
    __shared__ double s[CT::NBACK];
    if (st->termination >= 0) return;
    const int solve_ok = st->solve_ok;
    if (solve_ok) reduce_partials<CT::NBACK>(partials, nblocks >= 0 ? nblocks : st->grid, CT::BACK_MAX, s_red, s);
    if (threadIdx.x != 0) return;
    (void)decide_serial<NP>(st, s, solve_ok, trace, trace_rows);
}

// nonlinearRefinement.cc:244-248
__global__ __launch_bounds__(kFB) void refine_finish_kernel(int64_t m, const double* __restrict__ inl,
                                                           const double* __restrict__ rho_a, const double* __restrict__ rho_b,
                                                           const RefineState* __restrict__ st, double* __restrict__ inl_out,
                                                           double* __restrict__ zpartials, RefineState* __restrict__ state_host,
                                                           const int* __restrict__ bad_index) {
    __shared__ double s_red[kFB / 64];
    if (state_host && blockIdx.x == 0) {  // frame solve: the state (+ bad-index flag) travels to host-mapped memory with this launch
        static_assert(sizeof(RefineState) % 8 == 0, "copied as 8-byte words");
        for (int i = threadIdx.x; i < (int)(sizeof(RefineState) / 8); i += kFB)
            reinterpret_cast<double*>(state_host)[i] = reinterpret_cast<const double*>(st)[i];
        if (threadIdx.x == 0) *reinterpret_cast<int*>(reinterpret_cast<char*>(state_host) + sizeof(RefineState)) = *bad_index;
    }
    const double* __restrict__ rho = st->cur ? rho_b : rho_a;
    if (m < 0) m = st->m;  // device-resident count: a plain copy pass, any grid will do
    const int64_t stride = (int64_t)gridDim.x * kFB;
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * kFB + threadIdx.x; i < m; i += stride) {
        inl_out[3 * i] = inl[3 * i];
        inl_out[3 * i + 1] = inl[3 * i + 1];
        const double z = 1.0 / rho[i];
        inl_out[3 * i + 2] = z;
        acc += z;
    }
    if (zpartials) {  // frame solve: the per-workgroup sums of z the mean-z sign test needs (main.cc:466-472), saving that stage's own pass
        const double r = wave_sum(acc);
        if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = r;
        __syncthreads();
        if (threadIdx.x == 0) {
            double t = s_red[0];
            for (int w2 = 1; w2 < kFB / 64; ++w2) t += s_red[w2];
            zpartials[blockIdx.x] = t;
        }
    }
}

// Start state of a refinement that is enqueued before the host has read the RANSAC result (frame solve): pose of the best trial and
// the inlier count from the device-resident RansacBest, logical grid by the launchers' rule (refine_grid below).
__global__ void refine_state_from_best_kernel(const RansacBest* __restrict__ best, RefineState* __restrict__ st, int np, int cap,
                                              int* __restrict__ bad_index) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    RefineState z;
    memset(&z, 0, sizeof(z));
    z.np = np;
    for (int i = 0; i < 3; ++i) {
        z.p[i] = best->hyp[3 + i];      // v
        z.p[3 + i] = best->hyp[i];      // w
    }
    z.p[6] = best->hyp[6];
    z.termination = -1;
    z.radius = kInitialRadius;
    z.need_schur = 1;  // the first slot of the iteration loop is the Schur pass of iteration 1
    // (a RANSAC that is not over -- ransac_pick_kernel -- : no inliers, and every kernel of this refinement leaves at once)
    const int64_t m = best->undecided ? 0 : best->num_inliers_scan;
    z.m = m;
    int64_t b = (m + kFB - 1) / kFB;
    if (b < 1) b = 1;
    if (b > cap) {
        const int64_t iters = (b + cap - 1) / cap;
        b = (b + iters - 1) / iters;
    }
    z.grid = (int)b;
    *st = z;
    bad_index[0] = 0;
    bad_index[4] = bad_index[5] = bad_index[6] = 0;  // (the list counters of the radius-factorised path: refine_rf_counters)
}

// ---------------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------------
static inline int refine_grid_cap(const Ctx* c) { return c->num_cus * kWorkgroupsPerCu; }
static inline int refine_grid(const Ctx* c, int64_t m) {
    int64_t b = (m + kFB - 1) / kFB;
    const int64_t cap = (int64_t)c->num_cus * kWorkgroupsPerCu;
    if (b < 1) b = 1;
    if (b > cap) {
        const int64_t iters = (b + cap - 1) / cap;
        b = (b + iters - 1) / iters;
    }
    return (int)b;
}

// (a workgroup's row: the init sums, the Schur sums, the back-substitution sums, or a slot's Schur | back-substitution sums -- the widest)
// (the buffer also holds a second set of rows -- the prologue of a slot's pass reads the previous pass's rows while the workgroups write
// their own -- and the two states the slot kernels of a chunk alternate between: sized by the cap, whatever m)
static inline int refine_partials_half(const Ctx* c) { return refine_grid_cap(c) * SlotRow<7>::NW; }
constexpr int kStateDoubles = (int)((sizeof(RefineState) + 63) / 64 * 8);  // one state, padded to 64 bytes, in doubles
int refine_partials_doubles_cap(const Ctx* c) { return 2 * refine_partials_half(c) + 2 * kStateDoubles + 8 + refine_rf_extra_doubles(); }
int refine_partials_half_doubles(const Ctx* c) { return refine_partials_half(c); }
int refine_state_doubles() { return kStateDoubles; }
int refine_partials_doubles(const Ctx* c, int64_t m) { (void)m; return refine_partials_doubles_cap(c); }
static inline double* rows_buffer(const Ctx* c, const RefineBuffers& B, int which) { return B.partials + (size_t)(which & 1) * refine_partials_half(c); }
static inline RefineState* chunk_state(const Ctx* c, const RefineBuffers& B, int which) {
    return reinterpret_cast<RefineState*>(B.partials + 2 * (size_t)refine_partials_half(c) + (size_t)(which & 1) * kStateDoubles);
}
// slot j of a chunk starts from the published state (j = 0) or from what slot j - 1 left, and leaves its own in the other buffer
static inline const RefineState* slot_state_in(const Ctx* c, const RefineBuffers& B, int j) { return j == 0 ? B.state : chunk_state(c, B, j - 1); }

int refine_state_from_best_launch(Ctx* c, const RansacBest* d_best, const RefineBuffers& B, int np) {
    hipLaunchKernelGGL(refine_state_from_best_kernel, dim3(1), dim3(64), 0, c->stream, d_best, B.state, np, refine_grid_cap(c), B.bad_index);
    RSDSFM_HIP_CHECK(c, hipGetLastError());
    return RSDSFM_OK;
}

template <int NP>
static int refine_init_t(Ctx* c, const RefineBuffers& B) {
    const int grid = B.m_on_device ? refine_grid_cap(c) : refine_grid(c, B.m);
    const int64_t m_arg = B.m_on_device ? -1 : B.m;
    hipLaunchKernelGGL(refine_init_kernel<NP>, dim3(grid), dim3(kFB), 0, c->stream, reinterpret_cast<const double2*>(B.flow), B.n_flow,
                       m_arg, B.inl, B.alpha, B.alpha_k, B.inlier_idx, B.flow_index_mode, B.state, reinterpret_cast<double4*>(B.uu),
                       B.beta, B.rho_a, B.srho, B.partials, B.bad_index, B.want_zsum ? 1 : 0);
    RSDSFM_HIP_CHECK(c, hipGetLastError());
    hipLaunchKernelGGL(refine_init_decide_kernel<NP>, dim3(1), dim3(kFB), 0, c->stream, B.partials, B.m_on_device ? -1 : grid, B.state, m_arg);
    RSDSFM_HIP_CHECK(c, hipGetLastError());
    return RSDSFM_OK;
}

// ---- row-tiled stages (one shard of the inliers per rank; see dist.py TiledFrameSolve) ----
// the shard's partials reduced to one row (the all-gather payload); the decide kernels then reduce the gathered
// [ranks][NV] array exactly as they reduce per-workgroup partials
template <int NV>
__global__ __launch_bounds__(kFB) void refine_row_kernel(const double* __restrict__ partials, int nblocks, int max_slot,
                                                        double* __restrict__ row, const RefineState* __restrict__ st = nullptr) {
    __shared__ double s_red[kFB / 64][NV];
    __shared__ double s[NV];
    if (nblocks < 0) nblocks = st->grid;  // device-resident shape (see pass_shape)
    reduce_partials<NV>(partials, nblocks, max_slot, s_red, s);
    if (threadIdx.x < NV) row[threadIdx.x] = s[threadIdx.x];
}

template <int NP>
static int refine_stage_rows_t(Ctx* c, const RefineBuffers& B, int stage, double* row) {
    using CT = Counts<NP>;
    const int grid = B.m_on_device ? refine_grid_cap(c) : refine_grid(c, B.m);
    if (stage == 0) {
        hipLaunchKernelGGL(refine_init_kernel<NP>, dim3(grid), dim3(kFB), 0, c->stream, reinterpret_cast<const double2*>(B.flow),
                           B.n_flow, B.m_on_device ? (int64_t)-1 : B.m, B.inl, B.alpha, B.alpha_k, B.inlier_idx, B.flow_index_mode, B.state,
                           reinterpret_cast<double4*>(B.uu), B.beta, B.rho_a, B.srho, B.partials, B.bad_index, B.want_zsum ? 1 : 0);
        RSDSFM_HIP_CHECK(c, hipGetLastError());
        hipLaunchKernelGGL(refine_row_kernel<CT::NINIT>, dim3(1), dim3(kFB), 0, c->stream, B.partials, B.m_on_device ? -1 : grid, CT::INIT_MAX, row,
                           static_cast<const RefineState*>(B.state));
    } else if (B.m_on_device) {
        return fail(c, RSDSFM_ERR_INVALID, "two-stage refinement protocol: the inlier count must be known on the host");
    } else if (stage == 1) {
        hipLaunchKernelGGL(refine_schur_kernel<NP>, dim3(grid), dim3(kFB), 0, c->stream, B.m,
                           reinterpret_cast<const double4*>(B.uu), B.beta, B.alpha, B.alpha_k, B.rho_a, B.rho_b, B.srho, B.state, B.partials);
        RSDSFM_HIP_CHECK(c, hipGetLastError());
        hipLaunchKernelGGL(refine_row_kernel<CT::NSCHUR>, dim3(1), dim3(kFB), 0, c->stream, B.partials, grid, -1, row);
    } else {
        hipLaunchKernelGGL(refine_backsub_kernel<NP>, dim3(grid), dim3(kFB), 0, c->stream, B.m,
                           reinterpret_cast<const double4*>(B.uu), B.beta, B.alpha, B.alpha_k, B.rho_a, B.rho_b, B.srho, B.state, B.partials, B.want_zsum ? 1 : 0);
        RSDSFM_HIP_CHECK(c, hipGetLastError());
        hipLaunchKernelGGL(refine_row_kernel<CT::NBACK>, dim3(1), dim3(kFB), 0, c->stream, B.partials, grid, CT::BACK_MAX, row);
    }
    RSDSFM_HIP_CHECK(c, hipGetLastError());
    return RSDSFM_OK;
}

template <int NP>
static int refine_stage_apply_t(Ctx* c, const RefineBuffers& B, int stage, const double* rows_all, int nranks, int64_t m_total, const int64_t* m_total_dev) {
    if (stage == 0)
        hipLaunchKernelGGL(refine_init_decide_kernel<NP>, dim3(1), dim3(kFB), 0, c->stream, rows_all, nranks, B.state, m_total, m_total_dev);
    else if (stage == 1)
        hipLaunchKernelGGL(refine_solve_kernel<NP>, dim3(1), dim3(kFB), 0, c->stream, rows_all, nranks, B.state);
    else
        hipLaunchKernelGGL(refine_decide_kernel<NP>, dim3(1), dim3(kFB), 0, c->stream, rows_all, nranks, B.state, c->d_refine_trace, c->refine_trace_rows);
    RSDSFM_HIP_CHECK(c, hipGetLastError());
    return RSDSFM_OK;
}

// ---------------------------------------------------------------------------------------------------
// column-tiled solve: ONE exchange per LM iteration
// ---------------------------------------------------------------------------------------------------
// The two-stage protocol above exchanges twice per iteration (Schur rows -> solve; back-substitution rows -> decision) because the next
// Schur sums depend on the decision: on the point (candidate or not) and on the radius Ceres' rule derives from the step quality.  A
// SLOT merges them: its pass runs the back-substitution of iteration i AND, at the candidate it has just formed, the Schur sums of
// iteration i + 1 for the radius an accepted step of quality >= 0.937 gets (radius_accept(radius, 1): x 3, the common case); one row
// [NSCHUR | NBACK] per rank travels; the apply stage decides iteration i and -- when the step was accepted with exactly that radius --
// solves iteration i + 1 from the speculated sums, which are the very numbers the Schur pass would produce (same per-inlier code, same
// thread mapping, same reductions).  Otherwise (rejected, invalid, or another radius) it sets RefineState::need_schur and the NEXT slot is a
// plain Schur pass.  The host enqueues slots without knowing which kind each will be: the kernels read it from the state.
template <int NP>
__device__ __forceinline__ void slot_apply_body(RefineState* st, const double* __restrict__ rows_all, int nranks, double* __restrict__ trace, int trace_rows);

// The streaming pass of a slot, with the single-workgroup stage of the PREVIOUS slot in its prologue: every workgroup copies the state it
// starts from (st_in) to LDS, reduces the previous slot's rows and runs the decision / reduced solve on that copy -- the same code on the
// same numbers in every workgroup, no hand-off between workgroups -- and goes on with the result; workgroup 0 also leaves it in st_out for
// the next kernel (another buffer than st_in: a slower workgroup may still be reading that).  A launch and a pass over an idle chip less
// per LM iteration than a stage kernel of its own between the passes, which is what VERDICT r3's item 3 was after; the hand-off variants
// (fences in round 2, write-through stores + ticket in round 4) lost because of the hand-off, which this form does not have.
template <int NP>
__global__ __launch_bounds__(kFB) void refine_slot_pass_kernel(int64_t m, const double4* __restrict__ xyuv, const double* __restrict__ beta_in,
                                                              const double* __restrict__ alpha, const double* __restrict__ alpha_k,
                                                              double* __restrict__ rho_a, double* __restrict__ rho_b, const double* __restrict__ srho,
                                                              const RefineState* __restrict__ st_in, RefineState* __restrict__ st_out,
                                                              const double* __restrict__ rows_prev, int nrows_prev, double* __restrict__ partials,
                                                              int want_zsum, double* __restrict__ trace, int trace_rows) {
    using CT = Counts<NP>;
    using SR = SlotRow<NP>;
    __shared__ double s_redS[kFB / 64][CT::NSCHUR];
    __shared__ double s_redB[kFB / 64][CT::NBACK];
    __shared__ RefineState s_state;
    state_to_lds(&s_state, st_in);
    RefineState* st = &s_state;
    if (st->termination < 0 && st->pending_apply)
        slot_apply_body<NP>(st, rows_prev, nrows_prev >= 0 ? nrows_prev : st->grid, blockIdx.x == 0 ? trace : nullptr, trace_rows);
    __syncthreads();
    if (threadIdx.x == 0) st->pending_apply = st->termination < 0 ? 1 : 0;  // (this pass runs: its rows -- or its top-of-loop exit -- wait for an apply)
    __syncthreads();
    if (blockIdx.x == 0) state_from_lds(st_out, st);
    if (st->termination >= 0) return;
    const PassShape ps = pass_shape(st, m);
    if (!ps.live) return;
    m = ps.m;
    double* row = partials + (int64_t)blockIdx.x * SR::NW;
    double p[7], sp[NP];
#pragma unroll
    for (int c = 0; c < 7; ++c) p[c] = uniform_d(st->p[c]);
#pragma unroll
    for (int c = 0; c < NP; ++c) sp[c] = uniform_d(st->sp[c]);
    const double* __restrict__ rho = st->cur ? rho_b : rho_a;
    const int64_t stride = (int64_t)ps.grid * kFB;
    double accS[CT::NSCHUR];
#pragma unroll
    for (int s = 0; s < CT::NSCHUR; ++s) accS[s] = 0.0;
    if (st->need_schur) {  // a plain Schur pass at the current state (refine_schur_kernel's loop)
        if (st->iteration >= kMaxIter || st->radius <= kMinRadius) return;  // (the apply stage records the termination)
        const double inv_radius = 1.0 / st->radius;
        for (int64_t i = (int64_t)blockIdx.x * kFB + threadIdx.x; i < m; i += stride) {
            const double4 c4 = xyuv[i];
            RJ<NP> o;
            if (NP == 6)
                resid_jac_beta<NP>(c4.x, c4.y, c4.z, c4.w, beta_in[i], 0.0, p, rho[i], o);
            else
                resid_jac<NP>(c4.x, c4.y, c4.z, c4.w, alpha[i], alpha_k[i], p, rho[i], o);
            schur_accumulate<NP>(o, srho[i], sp, inv_radius, accS);
        }
        block_reduce_store<CT::NSCHUR>(accS, -1, s_redS, row);
        return;
    }
    if (!st->solve_ok) return;
    // the back-substitution of this iteration (refine_backsub_kernel's loop) + the Schur sums of the next one at the candidate
    double pc[7], yp[NP];
#pragma unroll
    for (int c = 0; c < 7; ++c) pc[c] = uniform_d(st->pc[c]);
#pragma unroll
    for (int c = 0; c < NP; ++c) yp[c] = uniform_d(st->yp[c]);
    const double inv_radius = 1.0 / st->radius;
    const double inv_radius_spec = 1.0 / radius_accept(st->radius, 1.0);
    const bool spec = SR::SPECULATES && st->spec_miss_run < 2;  // (the apply stage reads the same word before it updates it)
    double* __restrict__ cand = st->cur ? rho_a : rho_b;
    double acc[CT::NBACK];
#pragma unroll
    for (int s = 0; s < CT::NBACK; ++s) acc[s] = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * kFB + threadIdx.x; i < m; i += stride) {
        const double4 c4 = xyuv[i];
        const double x = c4.x, y = c4.y;
        const double rh = rho[i];
        double be, dbe = 0.0, bec, dbec = 0.0;
        if (NP == 6) {
            be = bec = beta_in[i];
        } else {
            const double al = alpha[i], ak = alpha_k[i];
            be = beta_of(al, ak, p[6]), dbe = dbeta_of(al, ak, p[6]);
            bec = beta_of(al, ak, pc[6]), dbec = dbeta_of(al, ak, pc[6]);
        }
        RJ<NP> o;
        resid_jac_beta<NP>(x, y, c4.z, c4.w, be, dbe, p, rh, o);
        const double sr = srho[i];
        const double E0 = o.Jr[0] * sr, E1 = o.Jr[1] * sr;
        const double ht = E0 * E0 + E1 * E1;
        const double lam = clampd(ht, kMinLmDiag, kMaxLmDiag) * inv_radius;
        const double ete_inv = 1.0 / (ht + lam);
        const double Etb = E0 * o.r[0] + E1 * o.r[1];
        double Fy0 = 0.0, Fy1 = 0.0;
#pragma unroll
        for (int c = 0; c < NP; ++c) {
            Fy0 += o.Jp[0][c] * sp[c] * yp[c];
            Fy1 += o.Jp[1][c] * sp[c] * yp[c];
        }
        const double ye = ete_inv * (Etb - (E0 * Fy0 + E1 * Fy1));
        const double step_e = -ye;
        const double m0 = -Fy0 + E0 * step_e, m1 = -Fy1 + E1 * step_e;
        acc[0] -= m0 * (o.r[0] + m0 / 2.0) + m1 * (o.r[1] + m1 / 2.0);
        const double cd = rh + step_e * sr;
        cand[i] = cd;
        const double dx = rh - cd;
        acc[1] += dx * dx;
        RJ<NP> oc;
        resid_jac_beta<NP>(x, y, c4.z, c4.w, bec, dbec, pc, cd, oc);
        const double c2 = oc.r[0] * oc.r[0] + oc.r[1] * oc.r[1];
        acc[2] += c2;
        acc[3] += c2;
#pragma unroll
        for (int c = 0; c < NP; ++c) acc[4 + c] += oc.Jp[0][c] * oc.r[0] + oc.Jp[1][c] * oc.r[1];
        acc[CT::BACK_MAX] = fmax(acc[CT::BACK_MAX], fabs(oc.Jr[0] * oc.r[0] + oc.Jr[1] * oc.r[1]));
        acc[CT::BACK_MAX + 1] += cd * cd;
        if (want_zsum) acc[CT::BACK_MAX + 2] += 1.0 / cd;
        if (SR::SPECULATES) {  // (compile-time: the NP = 7 pass carries one evaluation or the other, never both)
            if (spec) schur_accumulate<NP>(oc, sr, sp, inv_radius_spec, accS);  // (oc: exactly what the next Schur pass would evaluate at the accepted candidate)
        }
    }
    if (spec) {
        block_reduce_store<CT::NSCHUR>(accS, -1, s_redS, row);
        __syncthreads();
    }
    block_reduce_store<CT::NBACK>(acc, CT::BACK_MAX, s_redB, row + SR::OFF_BACK);
}

// the shard's slot partials [workgroups][NW] reduced to one row [NW]: the two column ranges with the reductions of their own stages
template <int NP>
__global__ __launch_bounds__(kFB) void refine_slot_row_kernel(const double* __restrict__ partials, int nblocks, double* __restrict__ row,
                                                             const RefineState* __restrict__ st = nullptr) {
    using CT = Counts<NP>;
    using SR = SlotRow<NP>;
    __shared__ double s_redS[kFB / 64][CT::NSCHUR];
    __shared__ double s_redB[kFB / 64][CT::NBACK];
    __shared__ double sS[CT::NSCHUR];
    __shared__ double sB[CT::NBACK];
    if (nblocks < 0) nblocks = st->grid;  // device-resident shape (see pass_shape)
    reduce_partials<CT::NSCHUR>(partials, nblocks, -1, s_redS, sS, SR::NW, 0);
    reduce_partials<CT::NBACK>(partials, nblocks, CT::BACK_MAX, s_redB, sB, SR::NW, SR::OFF_BACK);
    if (threadIdx.x < CT::NSCHUR) row[threadIdx.x] = sS[threadIdx.x];
    if (threadIdx.x < CT::NBACK) row[SR::OFF_BACK + threadIdx.x] = sB[threadIdx.x];
    if (threadIdx.x == 0 && (CT::NBACK & 1)) row[SR::NW - 1] = 0.0;
}

// the apply stage of a slot (replicated on every rank of the column-tiled solve): rows_all = the gathered [nranks][NW] rows, or the
// workgroups' partials of a single context.  (Measured without gain at 1280x720: reducing the Schur columns beside the back-substitution
// columns on a second half of the workgroup ahead of the decision -- slower, with 512 and with 256 rows --; working on a copy of the
// state in LDS.  What did help is fewer rows: see kFB.)
template <int NP>
__device__ __forceinline__ void slot_apply_body(RefineState* st, const double* __restrict__ rows_all, int nranks, double* __restrict__ trace, int trace_rows) {
    using CT = Counts<NP>;
    using SR = SlotRow<NP>;
    __shared__ double s_gS[ReduceShape<CT::NSCHUR>::G][CT::NSCHUR];
    __shared__ double s_gB[ReduceShape<CT::NBACK>::G][CT::NBACK];
    __shared__ double s_aS[kFB / 64][CT::NSCHUR];
    __shared__ double s_aB[kFB / 64][CT::NBACK];
    __shared__ double sS[CT::NSCHUR];
    __shared__ double sB[CT::NBACK];
    __shared__ int s_do_solve;
    const int tid = threadIdx.x;
    const int was_schur = st->need_schur;
    const int solve_ok = st->solve_ok;
    const int spec_on = SR::SPECULATES && st->spec_miss_run < 2;  // (whether the pass speculated at all)
    if (tid == 0) s_do_solve = was_schur;
    // the sums: the back-substitution columns the decision needs and -- ahead of the decision -- the Schur columns the reduced solve needs
    // when the slot was a Schur pass or its speculation applies; one round trip to the rows for both (each in reduce_partials' order)
    const bool needB = !was_schur && solve_ok, wantS = was_schur || (spec_on && solve_ok);
    bool haveS = false, fused = false;
    if constexpr (SR::SPECULATES) {  // (with k refined a slot carries one range or the other, never both)
        fused = reduce_two_ranges_groups<CT::NBACK, CT::NSCHUR>(rows_all, nranks, SR::NW, SR::OFF_BACK, CT::BACK_MAX, needB, s_gB, 0, -1, wantS, s_gS, tid);
        if (fused) {
            __syncthreads();
            if (needB) reduce_partials_slots<CT::NBACK>(s_gB, CT::BACK_MAX, sB, tid);
            if (wantS) reduce_partials_slots<CT::NSCHUR>(s_gS, -1, sS, tid);
            __syncthreads();
            haveS = wantS;
        }
    }
    if (!fused && needB) reduce_partials<CT::NBACK>(rows_all, nranks, CT::BACK_MAX, s_aB, sB, SR::NW, SR::OFF_BACK);
    if (!was_schur && tid == 0) {  // the decision of the iteration whose back-substitution this slot carried
        const double r_spec = radius_accept(st->radius, 1.0);  // (what the pass speculated with: the radius BEFORE the decision)
        const int accepted = decide_serial<NP>(st, sB, solve_ok, trace, trace_rows);
        // the speculated Schur sums are the next iteration's iff the state moved to the candidate with exactly that radius
        const int applies = (accepted && st->radius == r_spec) ? 1 : 0;
        st->spec_miss_run = applies ? 0 : min(st->spec_miss_run + 1, 2);
        s_do_solve = applies && spec_on;
        st->need_schur = (st->termination < 0 && !s_do_solve) ? 1 : 0;
    }
    __syncthreads();
    if (tid == 0) st->slots += 1;
    const int do_solve = s_do_solve, termination = st->termination, iteration = st->iteration;
    const double radius = st->radius;
    __syncthreads();  // (every thread holds the decision's outcome before lane 0 goes on writing the state)
    if (!do_solve || termination >= 0) return;
    if (iteration >= kMaxIter) {  // top-of-loop checks of TrustRegionMinimizer (refine_solve_kernel)
        if (tid == 0) st->termination = RSDSFM_TERM_MAX_ITER;
        return;
    }
    if (radius <= kMinRadius) {
        if (tid == 0) st->termination = RSDSFM_TERM_MIN_RADIUS;
        return;
    }
    if (!haveS) reduce_partials<CT::NSCHUR>(rows_all, nranks, -1, s_aS, sS, SR::NW, 0);
    if (tid == 0) {
        double p_cur[7], sp_cur[NP];
#pragma unroll
        for (int c = 0; c < 7; ++c) p_cur[c] = st->p[c];
#pragma unroll
        for (int c = 0; c < NP; ++c) sp_cur[c] = st->sp[c];
        solve_serial<NP>(st, sS, p_cur, sp_cur, radius);
        st->need_schur = 0;
    }
}

// the single-workgroup stage on its own: behind the LAST pass of a chunk (st_in -> st_out = the published state the output pass, the caller's
// tail and the host read), and behind every exchange of a custom transport that drives the slots stage by stage
template <int NP>
__global__ __launch_bounds__(kFB) void refine_slot_apply_kernel(const double* __restrict__ rows_all, int nranks, const RefineState* __restrict__ st_in,
                                                               RefineState* __restrict__ st_out, double* __restrict__ trace, int trace_rows) {
    __shared__ RefineState s_state;
    state_to_lds(&s_state, st_in);
    RefineState* st = &s_state;
    if (st->termination < 0 && st->pending_apply) slot_apply_body<NP>(st, rows_all, nranks >= 0 ? nranks : st->grid, trace, trace_rows);
    __syncthreads();
    if (threadIdx.x == 0) st->pending_apply = 0;
    __syncthreads();
    state_from_lds(st_out, st);
}

// One SLOT of the iteration loop (the slot kernels below): a streaming pass and the single-workgroup stage behind it.  The first slot of a
// solve is the Schur pass of iteration 1 + its reduced solve; every later one carries the back-substitution of iteration i together with the
// Schur sums of iteration i + 1 speculated at the candidate, then the decision of iteration i and -- when the speculation applies, the
// common case -- the reduced solve of iteration i + 1: two launches and ONE read of the inliers per LM iteration where the four-kernel
// sequence (refine_schur / solve / backsub / decide, still the stage protocol of rsdsfm_tile_refine_*) takes four and two.  A slot whose
// speculation did not apply is followed by a plain Schur slot; the kernels read which kind from the state.
template <int NP>
static int refine_iter_t(Ctx* c, const RefineBuffers& B, int j, int chunk) {
    const int grid = B.m_on_device ? refine_grid_cap(c) : refine_grid(c, B.m);
    const int64_t m_arg = B.m_on_device ? -1 : B.m;
    const int nb_arg = B.m_on_device ? -1 : grid;
    hipLaunchKernelGGL(refine_slot_pass_kernel<NP>, dim3(grid), dim3(kFB), 0, c->stream, m_arg, reinterpret_cast<const double4*>(B.uu), B.beta,
                       B.alpha, B.alpha_k, B.rho_a, B.rho_b, B.srho, slot_state_in(c, B, j), chunk_state(c, B, j), rows_buffer(c, B, j - 1), nb_arg,
                       rows_buffer(c, B, j), B.want_zsum ? 1 : 0, c->d_refine_trace, c->refine_trace_rows);
    RSDSFM_HIP_CHECK(c, hipGetLastError());
    // the stage on its own: behind the last pass of a chunk (into the published state), and -- while several pairs of a sequence share the
    // GPU, or on request -- behind every pass (in place: the next pass then finds nothing pending and its prologue only copies the state)
    const bool separate = c->refine_stage_mode == 2 || (c->refine_stage_mode == 0 && (c->refine_stage_separate || frames_in_flight(c) > 1));
    if (j == chunk - 1 || separate) {
        hipLaunchKernelGGL(refine_slot_apply_kernel<NP>, dim3(1), dim3(kFB), 0, c->stream, rows_buffer(c, B, j), nb_arg, chunk_state(c, B, j),
                           j == chunk - 1 ? B.state : chunk_state(c, B, j), c->d_refine_trace, c->refine_trace_rows);
        RSDSFM_HIP_CHECK(c, hipGetLastError());
    }
    return RSDSFM_OK;
}

template <int NP>
static int refine_slot_rows_t(Ctx* c, const RefineBuffers& B, double* row, int j, const double* rows_all_prev, int nranks) {
    // (device-resident inlier count -- the column-tiled solve ahead of the host's read of the RANSAC result --: the launch grid is the cap, the
    // logical grid and the count are RefineState's, as in the single-context frame solve)
    const int grid = B.m_on_device ? refine_grid_cap(c) : refine_grid(c, B.m);
    hipLaunchKernelGGL(refine_slot_pass_kernel<NP>, dim3(grid), dim3(kFB), 0, c->stream, B.m_on_device ? (int64_t)-1 : B.m, reinterpret_cast<const double4*>(B.uu), B.beta, B.alpha,
                       B.alpha_k, B.rho_a, B.rho_b, B.srho, slot_state_in(c, B, j), chunk_state(c, B, j), rows_all_prev, nranks, rows_buffer(c, B, 0),
                       B.want_zsum ? 1 : 0, c->d_refine_trace, c->refine_trace_rows);
    RSDSFM_HIP_CHECK(c, hipGetLastError());
    hipLaunchKernelGGL(refine_slot_row_kernel<NP>, dim3(1), dim3(kFB), 0, c->stream, rows_buffer(c, B, 0), B.m_on_device ? -1 : grid, row,
                       static_cast<const RefineState*>(B.state));
    RSDSFM_HIP_CHECK(c, hipGetLastError());
    return RSDSFM_OK;
}
int refine_slot_row_doubles(int np) { return np == 7 ? SlotRow<7>::NW : SlotRow<6>::NW; }
int refine_slot_partials_doubles(const Ctx* c, int64_t m) { return refine_partials_doubles(c, m); }
int refine_slot_rows_launch(Ctx* c, const RefineBuffers& B, int np, double* row, int j, const double* rows_all_prev, int nranks) {
    return np == 7 ? refine_slot_rows_t<7>(c, B, row, j, rows_all_prev, nranks) : refine_slot_rows_t<6>(c, B, row, j, rows_all_prev, nranks);
}
int refine_slot_apply_launch(Ctx* c, const RefineBuffers& B, int np, const double* rows_all, int nranks, int chunk) {
    if (np == 7)
        hipLaunchKernelGGL(refine_slot_apply_kernel<7>, dim3(1), dim3(kFB), 0, c->stream, rows_all, nranks, chunk_state(c, B, chunk - 1), B.state,
                           c->d_refine_trace, c->refine_trace_rows);
    else
        hipLaunchKernelGGL(refine_slot_apply_kernel<6>, dim3(1), dim3(kFB), 0, c->stream, rows_all, nranks, chunk_state(c, B, chunk - 1), B.state,
                           c->d_refine_trace, c->refine_trace_rows);
    RSDSFM_HIP_CHECK(c, hipGetLastError());
    return RSDSFM_OK;
}

int refine_stage_row_doubles(int np, int stage) {
    if (np == 7) return stage == 0 ? Counts<7>::NINIT : stage == 1 ? Counts<7>::NSCHUR : Counts<7>::NBACK;
    return stage == 0 ? Counts<6>::NINIT : stage == 1 ? Counts<6>::NSCHUR : Counts<6>::NBACK;
}
int refine_stage_rows_launch(Ctx* c, const RefineBuffers& B, int np, int stage, double* row) {
    return np == 7 ? refine_stage_rows_t<7>(c, B, stage, row) : refine_stage_rows_t<6>(c, B, stage, row);
}
int refine_stage_apply_launch(Ctx* c, const RefineBuffers& B, int np, int stage, const double* rows_all, int nranks, int64_t m_total, const int64_t* m_total_dev) {
    return np == 7 ? refine_stage_apply_t<7>(c, B, stage, rows_all, nranks, m_total, m_total_dev)
                   : refine_stage_apply_t<6>(c, B, stage, rows_all, nranks, m_total, m_total_dev);
}

int refine_init_launch(Ctx* c, const RefineBuffers& B, int np) { return np == 7 ? refine_init_t<7>(c, B) : refine_init_t<6>(c, B); }
int refine_iter_launch(Ctx* c, const RefineBuffers& B, int np, int j, int chunk) { return np == 7 ? refine_iter_t<7>(c, B, j, chunk) : refine_iter_t<6>(c, B, j, chunk); }

int refine_finish_grid(const Ctx* c, const RefineBuffers& B) { return B.m_on_device ? refine_grid_cap(c) : refine_grid(c, B.m); }

int refine_finish_launch(Ctx* c, const RefineBuffers& B, double* inl_out) {
    if (!B.m_on_device && B.m == 0 && !B.zpartials && !B.state_host) return RSDSFM_OK;
    hipLaunchKernelGGL(refine_finish_kernel, dim3(refine_finish_grid(c, B)), dim3(kFB), 0, c->stream,
                       B.m_on_device ? (int64_t)-1 : B.m, B.inl, B.rho_a, B.rho_b, B.state, inl_out, B.zpartials, B.state_host, B.bad_index);
    RSDSFM_HIP_CHECK(c, hipGetLastError());
    return RSDSFM_OK;
}

}  // namespace rsdsfm
