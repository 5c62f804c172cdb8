// depth_kernels.hip -- dense per-pixel inverse-depth solve on MI355X (gfx950).
//
// Replaces nonlinear_refinement::estimateInverseDepths (reference nonlinearRefinement.cc:109-180): N
// independent 1-D least-squares problems r_i(rho_i) = u_i - beta_i (a_i rho_i + b_i) for a fixed pose.
//
//  * depth_closed_form_kernel : exact optimum (one undamped Gauss-Newton step from rho = 1).
//  * depth_lm_kernel          : emulation of the Ceres 1.14 trust-region Levenberg-Marquardt the
//    reference runs.  Ceres' accept / converge decisions are GLOBAL (summed cost, step norm, max
//    gradient), so one launch speculatively evaluates up to KMAX consecutive LM iterations per pixel
//    (radius x3 each, which is what an exact-model step produces) and block-reduces the 5 per-iteration sums;
//    depth_lm_decide_kernel (one workgroup) then reduces the per-workgroup partials in a fixed order and
//    runs the trust-region state machine (lm_advance) on them.  The launch also writes the most likely final iterate, so the
//    common case costs one streaming pass: 48 B read + 8 B written per pixel.  The LM diagonal is formed as
//    clamp(diag) * (1/radius) (see the oracle for why), so the inner loop has one fp64 division per iteration.
//
// HBM-bound streaming, one lane per pixel PAIR so that every global access is a 16-byte vector
// (q: 2 x dwordx4, u: 2 x dwordx4, alpha / alpha_k / rho: dwordx4).  No LDS tiling (no reuse), no MFMA.
#include "device_math.hpp"
#include "rsdsfm_internal.hpp"

namespace rsdsfm {

// ---------------------------------------------------------------------------------------------------
// closed form
// ---------------------------------------------------------------------------------------------------
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

__global__ __launch_bounds__(kDepthBlock) void depth_closed_form_kernel(const double2* __restrict__ q,
                                                                        const double2* __restrict__ u,
                                                                        const double2* __restrict__ alpha2,
                                                                        const double2* __restrict__ alpha_k2,
                                                                        int64_t n, Pose pose,
                                                                        double2* __restrict__ rho2) {
    const double two_over = 2.0 / (2.0 + pose.k);
    const int64_t npairs = n >> 1;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < npairs; p += stride) {
        double2 qa = q[2 * p], qb = q[2 * p + 1];
        double2 ua = u[2 * p], ub = u[2 * p + 1];
        double2 al = alpha2[p], ak = alpha_k2[p];
        double2 out;
        out.x = closed_form_rho(qa.x, qa.y, ua.x, ua.y, al.x, ak.x, pose, two_over);
        out.y = closed_form_rho(qb.x, qb.y, ub.x, ub.y, al.y, ak.y, pose, two_over);
        rho2[p] = out;
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
        const int64_t i = n - 1;
        double2 qa = q[i], ua = u[i];
        const double* alpha = reinterpret_cast<const double*>(alpha2);
        const double* alpha_k = reinterpret_cast<const double*>(alpha_k2);
        reinterpret_cast<double*>(rho2)[i] = closed_form_rho(qa.x, qa.y, ua.x, ua.y, alpha[i], alpha_k[i], pose, two_over);
    }
}

// ---------------------------------------------------------------------------------------------------
// Ceres trust-region state machine on speculative sums
// ---------------------------------------------------------------------------------------------------
// sums layout: [0] = sum |r|^2 of the current state, [1] = sum rho^2, [2] = max |J.r| ; then for candidate
// j (0-based): base = 3 + 5 j : [sum |r(cand)|^2, model_cost_change, sum step^2, sum cand^2, max |J.r(cand)|]
// `first`: this is the launch of iteration zero (current state = rho == 1, nothing accepted yet).
// `used_K`, `used_write` : the plan the launch that produced `sums` ran with.
__device__ void lm_advance(LmState& st, const double* sums, int64_t n, int first, int used_K, int used_write,
                           int launch_id) {
    if (first) {
        st.status = 0;
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
    for (int j = 0; j < used_K && st.termination < 0; ++j) {
        if (st.iteration >= kMaxIter) {
            st.termination = RSDSFM_TERM_MAX_ITER;
            break;
        }
        if (st.radius < kMinRadius) {
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
            st.hist[st.n_hist] = st.radius;
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
    if (st.termination < 0 && st.radius < kMinRadius) st.termination = RSDSFM_TERM_MIN_RADIUS;
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
};

// one pixel through the planned LM trajectory; returns the state selected by write_which.
// Arithmetic mirrors oracle/rsdsfm_oracle.c rso_estimate_inverse_depths (mode 1) operation for operation.
__device__ __forceinline__ double lm_pixel(double x, double y, double ux, double uy, double al, double ak,
                                           const Pose& pose, double two_over, const LmPlanLds& plan,
                                           double (&acc)[NS]) {
    PixelModel m;
    m.init(x, y, ux, uy, al, ak, pose, two_over);
    const double s = 1.0 / (1.0 + sqrt(m.J0 * m.J0 + m.J1 * m.J1));  // Jacobi scaling (iteration 0 Jacobian)
    const double jt0 = m.J0 * s, jt1 = m.J1 * s;
    const double ht = jt0 * jt0 + jt1 * jt1;
    const double diag = clampd(ht, kMinLmDiag, kMaxLmDiag);
    double rho = 1.0;  // nonlinearRefinement.cc:140
    double r0, r1;
    m.residual(rho, r0, r1);
    for (int h = 0; h < plan.n_hist; ++h) {  // replay the accepted steps
        const double lam = diag * plan.inv_hist[h];
        const double gt = jt0 * r0 + jt1 * r1;
        const double step = -(gt / (ht + lam));
        rho = rho + step * s;
        m.residual(rho, r0, r1);
    }
    double out = rho;
    if (plan.K > 0) {
        acc[0] += r0 * r0 + r1 * r1;
        acc[1] += rho * rho;
        acc[2] = fmax(acc[2], fabs(m.J0 * r0 + m.J1 * r1));
    }
#pragma unroll
    for (int j = 0; j < KMAX; ++j) {
        if (j < plan.K) {
            const double lam = diag * plan.inv_cand[j];
            const double gt = jt0 * r0 + jt1 * r1;
            const double step = -(gt / (ht + lam));
            const double m0 = jt0 * step, m1 = jt1 * step;
            acc[3 + 5 * j + 1] -= m0 * (r0 + m0 / 2.0) + m1 * (r1 + m1 / 2.0);
            const double cand = rho + step * s;
            const double dx = rho - cand;
            acc[3 + 5 * j + 2] += dx * dx;
            m.residual(cand, r0, r1);
            acc[3 + 5 * j + 0] += r0 * r0 + r1 * r1;
            acc[3 + 5 * j + 3] += cand * cand;
            acc[3 + 5 * j + 4] = fmax(acc[3 + 5 * j + 4], fabs(m.J0 * r0 + m.J1 * r1));
            rho = cand;
            if (plan.write_which == j + 1) out = cand;
        }
    }
    return out;
}

__device__ __forceinline__ bool is_max_slot(int s) { return s == 2 || (s >= 3 && ((s - 3) % 5) == 4); }

// launch_id 0 = the launch of LM iteration zero (fresh state, built-in plan); launch_id > 0 acts only if the
// state machine designated exactly this launch (next_launch) to continue (status 0) or to apply (status 2).
__global__ __launch_bounds__(kDepthBlock) void depth_lm_kernel(const double2* __restrict__ q,
                                                               const double2* __restrict__ u,
                                                               const double2* __restrict__ alpha2,
                                                               const double2* __restrict__ alpha_k2, int64_t n,
                                                               Pose pose, double2* __restrict__ rho2,
                                                               const LmState* __restrict__ state,
                                                               double* __restrict__ partials, int launch_id) {
    __shared__ LmPlanLds plan;
    __shared__ double s_red[kDepthBlock / 64][NS];
    const int tid = threadIdx.x;
    if (launch_id == 0) {
        if (tid == 0) {
            plan.n_hist = 0;
            plan.K = KMAX;
            // speculate which iterate is final: what the previous solve on this context accepted (1 at start)
            const int pr = state->predict;
            plan.write_which = (pr >= 0 && pr <= KMAX) ? pr : 1;
            double r = kInitialRadius;
            for (int j = 0; j < KMAX; ++j) {
                plan.inv_cand[j] = 1.0 / r;
                r = radius_accept(r, 1.0);
            }
        }
    } else {
        const int status = state->status;
        if (status == 1 || state->next_launch != launch_id) return;  // finished, or not this launch's turn
        if (tid == 0) {
            plan.n_hist = state->n_hist;
            plan.K = (status == 2) ? 0 : state->K;
            plan.write_which = (status == 2) ? 0 : state->write_which;
        }
        if (tid < kMaxIter) plan.inv_hist[tid] = 1.0 / state->hist[tid];
        if (tid < KMAX) plan.inv_cand[tid] = 1.0 / state->cand[tid];
    }
    __syncthreads();

    const double two_over = 2.0 / (2.0 + pose.k);
    double acc[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) acc[s] = 0.0;

    const int64_t npairs = n >> 1;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + tid; p < npairs; p += stride) {
        double2 qa = q[2 * p], qb = q[2 * p + 1];
        double2 ua = u[2 * p], ub = u[2 * p + 1];
        double2 al = alpha2[p], ak = alpha_k2[p];
        double2 out;
        out.x = lm_pixel(qa.x, qa.y, ua.x, ua.y, al.x, ak.x, pose, two_over, plan, acc);
        out.y = lm_pixel(qb.x, qb.y, ub.x, ub.y, al.y, ak.y, pose, two_over, plan, acc);
        rho2[p] = out;
    }
    if ((n & 1) && blockIdx.x == 0 && tid == 0) {
        const int64_t i = n - 1;
        double2 qa = q[i], ua = u[i];
        const double* alpha = reinterpret_cast<const double*>(alpha2);
        const double* alpha_k = reinterpret_cast<const double*>(alpha_k2);
        reinterpret_cast<double*>(rho2)[i] = lm_pixel(qa.x, qa.y, ua.x, ua.y, alpha[i], alpha_k[i], pose, two_over, plan, acc);
    }
    if (plan.K == 0) return;  // apply-only launch: no sums

    // ---- workgroup partial (DPP wave reduction, then the 4 waves in order), one row of `partials` per workgroup ----
    const int lane = tid & 63, wv = tid >> 6;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        double r = is_max_slot(s) ? wave_max(acc[s]) : wave_sum(acc[s]);
        if (lane == 0) s_red[wv][s] = r;
    }
    __syncthreads();
    if (tid < NS) {
        double r = s_red[0][tid];
        for (int w2 = 1; w2 < kDepthBlock / 64; ++w2) r = is_max_slot(tid) ? fmax(r, s_red[w2][tid]) : r + s_red[w2][tid];
        partials[(int64_t)blockIdx.x * NS + tid] = r;
    }
}

// One workgroup: reduces the per-workgroup partials of launch `launch_id` in block order and advances the
// trust-region state machine.  Acts only if that launch actually speculated (status 0 and its turn).
__global__ __launch_bounds__(kDecideBlock) void depth_lm_decide_kernel(const double* __restrict__ partials, int nblocks,
                                                                       LmState* state, int64_t n, int launch_id) {
    __shared__ double s_red[kDecideBlock / 64][NS];
    __shared__ double s_sums[NS];
    __shared__ LmState s_state;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (launch_id > 0 && (state->status != 0 || state->next_launch != launch_id)) return;
    double fin[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) fin[s] = 0.0;
    for (int b = tid; b < nblocks; b += kDecideBlock) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            double v = partials[(int64_t)b * NS + s];
            fin[s] = is_max_slot(s) ? fmax(fin[s], v) : fin[s] + v;
        }
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        double r = is_max_slot(s) ? wave_max(fin[s]) : wave_sum(fin[s]);
        if (lane == 0) s_red[wv][s] = r;
    }
    {   // LDS copy of the state: the state machine indexes it dynamically (keep it out of registers / scratch)
        const int nwords = (int)(sizeof(LmState) / sizeof(int32_t));
        const int32_t* src = reinterpret_cast<const int32_t*>(state);
        int32_t* dst = reinterpret_cast<int32_t*>(&s_state);
        for (int i = tid; i < nwords; i += kDecideBlock) dst[i] = src[i];
    }
    __syncthreads();
    if (tid < NS) {
        double r = s_red[0][tid];
        for (int w2 = 1; w2 < kDecideBlock / 64; ++w2) r = is_max_slot(tid) ? fmax(r, s_red[w2][tid]) : r + s_red[w2][tid];
        s_sums[tid] = r;
    }
    __syncthreads();
    if (tid == 0) {
        const int used_K = (launch_id == 0) ? KMAX : s_state.K;
        const int pr = s_state.predict;
        const int used_write = (launch_id == 0) ? ((pr >= 0 && pr <= KMAX) ? pr : 1) : s_state.write_which;
        lm_advance(s_state, s_sums, n, launch_id == 0, used_K, used_write, launch_id);
    }
    __syncthreads();
    {
        const int nwords = (int)(sizeof(LmState) / sizeof(int32_t));
        int32_t* dst = reinterpret_cast<int32_t*>(state);
        const int32_t* src = reinterpret_cast<const int32_t*>(&s_state);
        for (int i = tid; i < nwords; i += kDecideBlock) dst[i] = src[i];
    }
}

// ---------------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------------
// balanced grid: every thread runs the same number of grid-stride iterations (no half-idle second sweep)
static inline int depth_grid(int64_t n, int max_blocks) {
    int64_t npairs = n >> 1;
    int64_t blocks = (npairs + kDepthBlock - 1) / kDepthBlock;
    if (blocks < 1) blocks = 1;
    if (blocks > max_blocks) {
        int64_t iters = (blocks + max_blocks - 1) / max_blocks;
        blocks = (blocks + iters - 1) / iters;
    }
    return (int)blocks;
}

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

int depth_closed_form_launch(Ctx* c, const double* q, const double* u, const double* a, const double* ak,
                             int64_t n, const Pose& pose, double* rho) {
    if (n == 0) return RSDSFM_OK;
    if (!aligned16(q) || !aligned16(u) || !aligned16(a) || !aligned16(ak) || !aligned16(rho))
        return fail(c, RSDSFM_ERR_INVALID, "device pointers must be 16-byte aligned");
    const int grid = depth_grid(n, 2048);
    hipLaunchKernelGGL(depth_closed_form_kernel, dim3(grid), dim3(kDepthBlock), 0, c->stream,
                       reinterpret_cast<const double2*>(q), reinterpret_cast<const double2*>(u),
                       reinterpret_cast<const double2*>(a), reinterpret_cast<const double2*>(ak), n, pose,
                       reinterpret_cast<double2*>(rho));
    RSDSFM_HIP_CHECK(c, hipGetLastError());
    return RSDSFM_OK;
}

int depth_lm_launch(Ctx* c, const double* q, const double* u, const double* a, const double* ak, int64_t n,
                    const Pose& pose, double* rho, int launch_id) {
    if (!aligned16(q) || !aligned16(u) || !aligned16(a) || !aligned16(ak) || !aligned16(rho))
        return fail(c, RSDSFM_ERR_INVALID, "device pointers must be 16-byte aligned");
    const int grid = depth_grid(n, kDepthMaxBlocks);
    hipLaunchKernelGGL(depth_lm_kernel, dim3(grid), dim3(kDepthBlock), 0, c->stream,
                       reinterpret_cast<const double2*>(q), reinterpret_cast<const double2*>(u),
                       reinterpret_cast<const double2*>(a), reinterpret_cast<const double2*>(ak), n, pose,
                       reinterpret_cast<double2*>(rho), c->d_lm, c->d_partials, launch_id);
    RSDSFM_HIP_CHECK(c, hipGetLastError());
    return RSDSFM_OK;
}

int depth_lm_decide_launch(Ctx* c, int64_t n, int launch_id) {
    const int grid = depth_grid(n, kDepthMaxBlocks);
    hipLaunchKernelGGL(depth_lm_decide_kernel, dim3(1), dim3(kDecideBlock), 0, c->stream, c->d_partials, grid, c->d_lm, n,
                       launch_id);
    RSDSFM_HIP_CHECK(c, hipGetLastError());
    return RSDSFM_OK;
}

}  // namespace rsdsfm
