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
#include "lm_common.hpp"
#include "rsdsfm_internal.hpp"

namespace rsdsfm {

// ---------------------------------------------------------------------------------------------------
// closed form
// ---------------------------------------------------------------------------------------------------
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

// fixed-order reduction of this context's per-workgroup partials (the last depth_lm_kernel launch over n points)
// into ONE row of NS sums -- the unit the row-tiled multi-GPU driver all-gathers
__global__ __launch_bounds__(kDecideBlock) void depth_lm_reduce_kernel(const double* __restrict__ partials, int nblocks,
                                                                       double* __restrict__ row) {
    __shared__ double s_red[kDecideBlock / 64][NS];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
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
    __syncthreads();
    if (tid < NS) {
        double r = s_red[0][tid];
        for (int w2 = 1; w2 < kDecideBlock / 64; ++w2) r = is_max_slot(tid) ? fmax(r, s_red[w2][tid]) : r + s_red[w2][tid];
        row[tid] = r;
    }
}

int depth_lm_reduce_launch(Ctx* c, int64_t n, double* d_row) {
    const int grid = depth_grid(n, kDepthMaxBlocks);
    hipLaunchKernelGGL(depth_lm_reduce_kernel, dim3(1), dim3(kDecideBlock), 0, c->stream, c->d_partials, grid, d_row);
    RSDSFM_HIP_CHECK(c, hipGetLastError());
    return RSDSFM_OK;
}

// decide on caller-provided rows (e.g. the all-gathered per-rank rows, in rank order)
int depth_lm_decide_rows_launch(Ctx* c, const double* d_rows, int nrows, int64_t n_total, int launch_id) {
    hipLaunchKernelGGL(depth_lm_decide_kernel, dim3(1), dim3(kDecideBlock), 0, c->stream, d_rows, nrows, c->d_lm, n_total, launch_id);
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
