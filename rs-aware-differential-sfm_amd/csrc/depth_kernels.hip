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
#include <hip/hip_ext.h>
#include <string.h>

#include <algorithm>

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

// Wave reduction of the NS per-lane sums into red[NS] (valid after the caller's barrier).  A thread of the depth solve only
// handles a couple of pixel pairs, so 18 DPP butterflies (~320 instructions) per wave were ~20 % of the kernel's
// instruction count.  The KMAX + 1 max slots keep the DPP butterfly; the 14 sum slots are transposed through LDS in two
// rounds of 7 (row stride 65: conflict-free): every lane stores its values, lane (slot, half) adds the 32 values of its
// half in lane order, the halves are added -- a fixed order.  A wave's LDS operations execute in order; the
// wave_barrier only pins the compiler's ordering.
constexpr int kTRows = 7;
constexpr int kTStride = 65;
__device__ __forceinline__ void wave_reduce_sums(const double (&acc)[NS], double* red, double* Tw, double (*half_buf)[kTRows], int lane) {
#pragma unroll
    for (int s = 0; s < NS; ++s)
        if (is_max_slot(s)) {
            const double r = wave_max(acc[s]);
            if (lane == 0) red[s] = r;
        }
    constexpr int kSums = NS - (KMAX + 1);
    static_assert(kSums == 2 * kTRows, "two rounds of kTRows sum slots");
#pragma unroll
    for (int round = 0; round < 2; ++round) {
        int kk = 0;
#pragma unroll
        for (int s = 0; s < NS; ++s)
            if (!is_max_slot(s)) {
                if (kk / kTRows == round) Tw[(kk % kTRows) * kTStride + lane] = acc[s];
                ++kk;
            }
        __builtin_amdgcn_wave_barrier();
        const int sl = lane & 31, hf = lane >> 5;
        if (sl < kTRows) {
            const double* row = Tw + sl * kTStride + hf * 32;
            double part = row[0];
#pragma unroll
            for (int j = 1; j < 32; ++j) part += row[j];
            half_buf[hf][sl] = part;
        }
        __builtin_amdgcn_wave_barrier();
        if (lane < kTRows) {
            int slot = 0, k2 = 0;
#pragma unroll
            for (int s = 0; s < NS; ++s)
                if (!is_max_slot(s)) {
                    if (k2 == round * kTRows + lane) slot = s;
                    ++k2;
                }
            red[slot] = half_buf[0][lane] + half_buf[1][lane];
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// streamed-once operands: non-temporal 16-byte accesses (measured: the memory-only skeleton of the batched kernel runs at
// 40 us instead of 44 us per 4 pairs with them, the full kernel 51 instead of 53 us)
typedef double d2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double2 nt_load(const double2* p) {
    d2v v = __builtin_nontemporal_load(reinterpret_cast<const d2v*>(p));
    return make_double2(v.x, v.y);
}
__device__ __forceinline__ void nt_store(double2* p, double2 o) {
    d2v v;
    v.x = o.x;
    v.y = o.y;
    __builtin_nontemporal_store(v, reinterpret_cast<d2v*>(p));
}

// launch_id 0 = the launch of LM iteration zero (fresh state, built-in plan); launch_id > 0 acts only if the
// state machine designated exactly this launch (next_launch) to continue (status 0) or to apply (status 2).
// FIRST = 1: launch 0 of a solve (always a full speculative pass); FIRST = 0: follow-up launches (apply / continue /
// no-op).  Same code; the template only gives the two roles distinct kernel names in profiles.
// CORE (launch 0 only): the Jacobi scaling 1 / (1 + sqrt(J.J)) through the in-range cores of sqrt and the reciprocal (device_math.hpp:
// the same bits for an argument in range, 12 instructions per pixel less).  A thread that met an argument out of range (a vanishing
// Jacobian, a non-finite flow) stores the launch's epoch into the context's flag word; the follow-up launch then leaves the solve
// unfinished with LmScal::restart set and rsdsfm_depth_finish_dev runs it again with the standard functions (capi.hip).
template <int FIRST, bool CORE = false>
__global__ __launch_bounds__(kDepthBlock) void depth_lm_kernel(const double2* __restrict__ q,
                                                               const double2* __restrict__ u,
                                                               const double2* __restrict__ alpha2,
                                                               const double2* __restrict__ alpha_k2, int64_t n,
                                                               Pose pose, double2* __restrict__ rho2,
                                                               const LmState* __restrict__ state,
                                                               double* __restrict__ partials, int launch_id,
                                                               int* __restrict__ predict_used, int core_epoch) {
    static_assert(!CORE || FIRST, "the function cores run in launch 0 only");
    __shared__ LmPlanLds plan;
    __shared__ double s_red[kDepthBlock / 64][NS];
    __shared__ double s_T[kDepthBlock / 64][kTRows * kTStride];
    __shared__ double s_half[kDepthBlock / 64][2][kTRows];
    const int tid = threadIdx.x;
    if (FIRST) {  // == (launch_id == 0)
        if (tid == 0) {
            plan.n_hist = 0;
            plan.K = KMAX;
            // speculate which iterate is final: what the previous solve on this context accepted (1 at start)
            const int pr = state->predict;
            if (blockIdx.x == 0 && predict_used) *predict_used = pr;  // for depth_lm_decide_apply_kernel
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
    if constexpr (FIRST) {  // launch 0: compile-time plan shape, operands streamed once
        LmPlanFirst pf;
        pf.load(plan);
        uint32_t worst = 0;  // CORE: the range tests of this thread's function-core arguments (sqrt_range_track)
        for (int64_t p = (int64_t)blockIdx.x * blockDim.x + tid; p < npairs; p += stride) {
            const double2 qa = nt_load(q + 2 * p), qb = nt_load(q + 2 * p + 1);
            const double2 ua = nt_load(u + 2 * p), ub = nt_load(u + 2 * p + 1);
            const double2 al = nt_load(alpha2 + p), ak = nt_load(alpha_k2 + p);
            double2 out;
            out.x = lm_pixel_t<CORE>(qa.x, qa.y, ua.x, ua.y, al.x, ak.x, pose, two_over, pf, acc, NoHook(), nullptr, &worst);
            out.y = lm_pixel_t<CORE>(qb.x, qb.y, ub.x, ub.y, al.y, ak.y, pose, two_over, pf, acc, NoHook(), nullptr, &worst);
            nt_store(rho2 + p, out);
        }
        if (CORE && worst >= kSqrtRangeKeys) predict_used[1] = core_epoch;  // (benign race: every writer stores the same word)
    } else {
        for (int64_t p = (int64_t)blockIdx.x * blockDim.x + tid; p < npairs; p += stride) {
            double2 qa = q[2 * p], qb = q[2 * p + 1];
            double2 ua = u[2 * p], ub = u[2 * p + 1];
            double2 al = alpha2[p], ak = alpha_k2[p];
            double2 out;
            out.x = lm_pixel(qa.x, qa.y, ua.x, ua.y, al.x, ak.x, pose, two_over, plan, acc);
            out.y = lm_pixel(qb.x, qb.y, ub.x, ub.y, al.y, ak.y, pose, two_over, plan, acc);
            rho2[p] = out;
        }
    }
    if ((n & 1) && blockIdx.x == 0 && tid == 0) {
        const int64_t i = n - 1;
        double2 qa = q[i], ua = u[i];
        const double* alpha = reinterpret_cast<const double*>(alpha2);
        const double* alpha_k = reinterpret_cast<const double*>(alpha_k2);
        reinterpret_cast<double*>(rho2)[i] = lm_pixel(qa.x, qa.y, ua.x, ua.y, alpha[i], alpha_k[i], pose, two_over, plan, acc);
    }
    if (plan.K == 0) return;  // apply-only launch: no sums

    // ---- workgroup partial (wave reduction, then the 4 waves in order), one row of `partials` per workgroup ----
    const int lane = tid & 63, wv = tid >> 6;
    wave_reduce_sums(acc, s_red[wv], s_T[wv], s_half[wv], lane);
    __syncthreads();
    if (tid < NS) {
        double r = s_red[0][tid];
        for (int w2 = 1; w2 < kDepthBlock / 64; ++w2) r = is_max_slot(tid) ? fmax(r, s_red[w2][tid]) : r + s_red[w2][tid];
        partials[(int64_t)blockIdx.x * NS + tid] = r;
    }
}

// ---------------------------------------------------------------------------------------------------
// Follow-up of launch 0 in the fast path: decision + apply in ONE launch (replaces depth_lm_decide_kernel followed by
// depth_lm_kernel<0>).  Launch 0 always starts from the initial trust-region state, so its decision depends only on the
// sums, n and the predictor value launch 0 used (`predict_used`): every workgroup reduces the partial rows itself (same
// fixed order, ~64 KB from L2) and runs the state machine redundantly -- all workgroups obtain the same state without any
// inter-workgroup communication, and nobody reads the global state during the kernel, so workgroup 0 can store it.
//   result already written by launch 0 (predictor right, the steady state)  -> return
//   finished but a different iterate is final                              -> replay the accepted steps, write rho
//   more LM iterations needed (rare)                                       -> return; rsdsfm_depth_finish_dev continues
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ void decide_apply_body(const double2* __restrict__ q, const double2* __restrict__ u,
                                                  const double2* __restrict__ alpha2, const double2* __restrict__ alpha_k2, int64_t n,
                                                  const Pose& pose, double2* __restrict__ rho2, LmState* state,
                                                  const double* __restrict__ partials, int nrows,
                                                  const int* __restrict__ predict_used, int core_epoch) {
    __shared__ LmPlanLds plan;
    __shared__ double s_red[kDepthBlock / 64][NS];
    __shared__ double s_T[kDepthBlock / 64][kTRows * kTStride];
    __shared__ double s_half[kDepthBlock / 64][2][kTRows];
    __shared__ double s_sums[NS];
    __shared__ double s_hist[kMaxIter];
    __shared__ int s_status;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int pr = *predict_used;
    // launch 0 ran its Jacobi scaling through the in-range function cores and met an argument out of range: its sums do not count
    const bool core_miss = core_epoch != 0 && predict_used[1] == core_epoch;
    double fin[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) fin[s] = 0.0;
    for (int b = tid; b < nrows; b += kDepthBlock) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const double v = partials[(int64_t)b * NS + s];
            fin[s] = is_max_slot(s) ? fmax(fin[s], v) : fin[s] + v;
        }
    }
    wave_reduce_sums(fin, s_red[wv], s_T[wv], s_half[wv], lane);
    __syncthreads();
    if (tid < NS) {
        double r = s_red[0][tid];
        for (int w2 = 1; w2 < kDepthBlock / 64; ++w2) r = is_max_slot(tid) ? fmax(r, s_red[w2][tid]) : r + s_red[w2][tid];
        s_sums[tid] = r;
    }
    __syncthreads();
    if (tid == 0) {
        LmScal st = {};
        st.predict = pr;
        const int used_write = (pr >= 0 && pr <= KMAX) ? pr : 1;
        lm_advance(st, s_hist, s_sums, n, 1, KMAX, used_write, 0);
        if (core_miss) {  // unfinished, whatever the sums said: rsdsfm_depth_finish_dev starts the solve over with the standard functions
            st.status = 0;
            st.restart = 1;
            st.termination = -1;
            st.next_launch = 1;
        }
        s_status = st.status;
        plan.n_hist = st.n_hist;
        plan.K = 0;
        plan.write_which = 0;
        if (st.status == 2) {  // this launch writes the result: the stored state says so
            st.status = 1;
            st.rho_holds = st.n_hist;
            st.launches += 1;
            st.next_launch = 2;
        }
        if (blockIdx.x == 0) {
            *static_cast<LmScal*>(state) = st;
            for (int h = 0; h < st.n_hist; ++h) state->hist[h] = s_hist[h];
        }
    }
    __syncthreads();
    if (s_status != 2) return;
    if (tid < plan.n_hist) plan.inv_hist[tid] = 1.0 / s_hist[tid];
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
}


__global__ __launch_bounds__(kDepthBlock) void depth_lm_decide_apply_kernel(const double2* __restrict__ q, const double2* __restrict__ u,
                                                                            const double2* __restrict__ alpha2,
                                                                            const double2* __restrict__ alpha_k2, int64_t n, Pose pose,
                                                                            double2* __restrict__ rho2, LmState* state,
                                                                            const double* __restrict__ partials, int nrows,
                                                                            const int* __restrict__ predict_used, int core_epoch) {
    decide_apply_body(q, u, alpha2, alpha_k2, n, pose, rho2, state, partials, nrows, predict_used, core_epoch);
}

// ---------------------------------------------------------------------------------------------------
// Batched fast path: up to kDepthBatchMax INDEPENDENT solves (frame pairs) per launch, blockIdx.y = solve.  The problem
// descriptors travel by value in the kernel arguments (no descriptor memory, nothing to upload per call); every solve
// keeps its own state, partial rows and predictor word (those of its context), so the per-solve protocol is exactly
// that of the single launches and rsdsfm_depth_finish_dev continues any of them individually.  One launch over several
// pairs amortises the ramp-up / tail of the streaming pass and the launch floor of the follow-up.
// ---------------------------------------------------------------------------------------------------
struct DepthBatchItem {
    const double2 *q, *u, *a2, *ak2;
    double2* rho2;
    int64_t n;
    Pose pose;
    LmState* state;
    double* partials;
    int* predict_used;  // [0] the predictor value launch 0 used, [1] the range-flag word of the function cores
    int core_epoch;     // != 0: this solve's launch 0 runs the function cores; the value a thread stores into predict_used[1] on a miss
};
struct DepthBatchArgs {
    int count;
    DepthBatchItem item[kDepthBatchMax];
};

template <bool CORE>
__global__ __launch_bounds__(kDepthBlock) void depth_lm_batch_kernel(DepthBatchArgs args) {
    __shared__ LmPlanLds plan;
    __shared__ double s_red[kDepthBlock / 64][NS];
    __shared__ double s_T[kDepthBlock / 64][kTRows * kTStride];
    __shared__ double s_half[kDepthBlock / 64][2][kTRows];
    const DepthBatchItem& it = args.item[blockIdx.y];
    const int tid = threadIdx.x;
    if (tid == 0) {
        plan.n_hist = 0;
        plan.K = KMAX;
        const int pr = it.state->predict;
        if (blockIdx.x == 0) *it.predict_used = pr;
        plan.write_which = (pr >= 0 && pr <= KMAX) ? pr : 1;
        double r = kInitialRadius;
        for (int j = 0; j < KMAX; ++j) {
            plan.inv_cand[j] = 1.0 / r;
            r = radius_accept(r, 1.0);
        }
    }
    __syncthreads();
    LmPlanFirst pf;  // launch 0: plan shape known at compile time -> the loop body is one basic block
    pf.load(plan);
    const Pose pose = it.pose;
    const double2 *q = it.q, *u = it.u, *alpha2 = it.a2, *alpha_k2 = it.ak2;
    double2* rho2 = it.rho2;
    const int64_t n = it.n;
    const double two_over = 2.0 / (2.0 + pose.k);
    double acc[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) acc[s] = 0.0;
    const int64_t npairs = n >> 1;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    uint32_t worst = 0;  // CORE: the range tests of this thread's function-core arguments (sqrt_range_track)
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + tid; p < npairs; p += stride) {
        const double2 qa = nt_load(q + 2 * p), qb = nt_load(q + 2 * p + 1);
        const double2 ua = nt_load(u + 2 * p), ub = nt_load(u + 2 * p + 1);
        const double2 al = nt_load(alpha2 + p), ak = nt_load(alpha_k2 + p);
        double2 out;
        out.x = lm_pixel_t<CORE>(qa.x, qa.y, ua.x, ua.y, al.x, ak.x, pose, two_over, pf, acc, NoHook(), nullptr, &worst);
        out.y = lm_pixel_t<CORE>(qb.x, qb.y, ub.x, ub.y, al.y, ak.y, pose, two_over, pf, acc, NoHook(), nullptr, &worst);
        nt_store(rho2 + p, out);
    }
    // (a batch runs the cores only when every solve of it does: core_epoch != 0 for all items)
    if (CORE && worst >= kSqrtRangeKeys) it.predict_used[1] = it.core_epoch;  // (benign race: every writer stores the same word)
    if ((n & 1) && blockIdx.x == 0 && tid == 0) {
        const int64_t i = n - 1;
        double2 qa = q[i], ua = u[i];
        const double* alpha = reinterpret_cast<const double*>(alpha2);
        const double* alpha_k = reinterpret_cast<const double*>(alpha_k2);
        reinterpret_cast<double*>(rho2)[i] = lm_pixel(qa.x, qa.y, ua.x, ua.y, alpha[i], alpha_k[i], pose, two_over, pf, acc);
    }
    const int lane = tid & 63, wv = tid >> 6;
    wave_reduce_sums(acc, s_red[wv], s_T[wv], s_half[wv], lane);
    __syncthreads();
    if (tid < NS) {
        double r = s_red[0][tid];
        for (int w2 = 1; w2 < kDepthBlock / 64; ++w2) r = is_max_slot(tid) ? fmax(r, s_red[w2][tid]) : r + s_red[w2][tid];
        it.partials[(int64_t)blockIdx.x * NS + tid] = r;
    }
}

__global__ __launch_bounds__(kDepthBlock) void depth_lm_decide_apply_batch_kernel(DepthBatchArgs args, int nrows) {
    const DepthBatchItem& it = args.item[blockIdx.y];
    decide_apply_body(it.q, it.u, it.a2, it.ak2, it.n, it.pose, it.rho2, it.state, it.partials, nrows, it.predict_used, it.core_epoch);
}

// ---------------------------------------------------------------------------------------------------
// Variant 2: launch 0 with the trust-region decision fused into its tail.  Every workgroup publishes its row of
// partials (agent-scope release), takes a ticket on one of 32 arrival counters and, when it completes a counter, on the
// root counter; the workgroup that completes the root is the last one: it acquires, reduces all rows in the same fixed
// order as depth_lm_decide_kernel, runs lm_advance and resets the counters.  Two-level counters keep the serialised
// atomics per address at <= 32 (a single 1024-way ticket was measured at ~15 us).  No workgroup ever waits.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kDepthBlock) void depth_lm_fused_kernel(const double2* __restrict__ q, const double2* __restrict__ u,
                                                                     const double2* __restrict__ alpha2,
                                                                     const double2* __restrict__ alpha_k2, int64_t n, Pose pose,
                                                                     double2* __restrict__ rho2, LmState* state,
                                                                     double* partials, unsigned* tickets) {
    __shared__ LmPlanLds plan;
    __shared__ double s_red[kDepthBlock / 64][NS];
    __shared__ double s_sums[NS];
    __shared__ int s_last;
    const int tid = threadIdx.x;
    if (tid == 0) {
        plan.n_hist = 0;
        plan.K = KMAX;
        const int pr = state->predict;
        plan.write_which = (pr >= 0 && pr <= KMAX) ? pr : 1;
        double r = kInitialRadius;
        for (int j = 0; j < KMAX; ++j) {
            plan.inv_cand[j] = 1.0 / r;
            r = radius_accept(r, 1.0);
        }
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
    const int lane = tid & 63, wv = tid >> 6;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        double r = is_max_slot(s) ? wave_max(acc[s]) : wave_sum(acc[s]);
        if (lane == 0) s_red[wv][s] = r;
    }
    __syncthreads();
    // ---- arrival ----
    // The row is published with RETURNING agent-scope atomic exchanges (performed at the memory side, coherent across the
    // XCDs' L2s); consuming the returned values makes the wave wait until they have been performed, so the ticket that
    // follows is ordered after the row without an agent-scope release fence (which writes back the whole L2: measured
    // 5x slower than the separate decide kernel).
    unsigned long long* prow = reinterpret_cast<unsigned long long*>(partials) + (int64_t)blockIdx.x * NS;
    unsigned long long old = 0ull;
    if (tid < NS) {
        double r = s_red[0][tid];
        for (int w2 = 1; w2 < kDepthBlock / 64; ++w2) r = is_max_slot(tid) ? fmax(r, s_red[w2][tid]) : r + s_red[w2][tid];
        old = __hip_atomic_exchange(prow + tid, (unsigned long long)__double_as_longlong(r), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // every lane of wave 0 observes its returned value (forces the s_waitcnt); the ballot keeps the dependency alive
    const bool seen = __builtin_amdgcn_ballot_w64(old == 0xFFFFFFFFFFFFFFFFull) != 0xFFFFFFFFFFFFFFFFull;
    if (tid == 0) {
        const unsigned nb = gridDim.x;
        const unsigned g = blockIdx.x & 31u;
        const unsigned gsize = (nb >> 5) + (g < (nb & 31u) ? 1u : 0u);  // workgroups with blockIdx % 32 == g
        const unsigned ngroups = nb < 32u ? nb : 32u;
        int last = 0;
        if (seen && __hip_atomic_fetch_add(&tickets[1 + g], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gsize - 1u) {
            if (__hip_atomic_fetch_add(&tickets[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == ngroups - 1u) last = 1;
        }
        s_last = last;
    }
    __syncthreads();
    if (!s_last) return;
    if (tid < 33) tickets[tid] = 0u;  // ready for the next launch (stream order separates launches)
    LmScal st;
    if (tid == 0) st = *static_cast<const LmScal*>(state);
    const int nblocks = gridDim.x;
    const unsigned long long* vp = reinterpret_cast<const unsigned long long*>(partials);
    double fin[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) fin[s] = 0.0;
    for (int b = tid; b < nblocks; b += kDepthBlock) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            // agent-scope loads: served from the memory side, never from a stale line of this XCD's L2
            const double v = __longlong_as_double((long long)__hip_atomic_load(vp + (int64_t)b * NS + s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
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
        for (int w2 = 1; w2 < kDepthBlock / 64; ++w2) r = is_max_slot(tid) ? fmax(r, s_red[w2][tid]) : r + s_red[w2][tid];
        s_sums[tid] = r;
    }
    __syncthreads();
    if (tid == 0) {
        const int pr = st.predict;
        const int used_write = (pr >= 0 && pr <= KMAX) ? pr : 1;
        lm_advance(st, state->hist, s_sums, n, true, KMAX, used_write, 0);
        *static_cast<LmScal*>(state) = st;
    }
}

// ---------------------------------------------------------------------------------------------------
// LDS-DMA variant of the fused LM kernel (selected per context: rsdsfm_set_depth_variant)
// ---------------------------------------------------------------------------------------------------
// Same arithmetic and outputs as depth_lm_kernel, different data movement: every WAVE streams its own 128-point
// tiles HBM -> LDS with `global_load_lds_dwordx4` (no VGPR staging) into a private double buffer, issuing the loads of
// tile i+1 before it computes tile i and waiting with a COUNTED s_waitcnt vmcnt(6) (the 6 loads of the next tile
// may stay in flight; loads retire in order).  HBM latency is thereby hidden behind the fp64 work without spending
// registers on prefetch.  No workgroup barrier and NO compiler-emitted LDS read inside the loop (hipcc would put an
// s_waitcnt vmcnt(0) in front of it and drain the pipeline): the plan lives in registers and the tile is read
// with ds_read_b128/b64 in one asm statement that carries its own lgkmcnt(0).
// Measured on MI355X (1280x720, round 1): the kernel is VALU-bound (~13 us of fp64 issue with no memory traffic at
// all), so this variant gains only ~5 % over the plain one (19.9 vs 21.0 us per back-to-back launch); it is kept
// as a tested option and becomes the better choice as soon as the arithmetic per pixel shrinks.
// Tile layout in a wave buffer (6144 B): q[128] @0, u[128] @2048, alpha[128] @4096, alpha_k[128] @5120; lane l owns
// points l and l+64 of the tile (conflict-free 16-byte / 8-byte LDS reads).
constexpr int kTilePoints = 128;
constexpr int kTileBytes = 6144;
constexpr int kDmaLdsHeader = 1024;  // s_red [4][NS] + inv_hist[kMaxIter]
constexpr int kDmaLdsBytes = kDmaLdsHeader + (kDepthBlock / 64) * 2 * kTileBytes;

__device__ __forceinline__ void dma16(const void* gsrc_lane, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc_lane,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

__device__ __forceinline__ void issue_tile(const double2* __restrict__ q, const double2* __restrict__ u,
                                           const double2* __restrict__ alpha2, const double2* __restrict__ alpha_k2,
                                           int64_t tile, int lane, char* buf) {
    const int64_t p0 = tile * kTilePoints;
    const char* gq = reinterpret_cast<const char*>(q + p0) + lane * 16;
    const char* gu = reinterpret_cast<const char*>(u + p0) + lane * 16;
    const char* ga = reinterpret_cast<const char*>(alpha2) + p0 * 8 + lane * 16;
    const char* gk = reinterpret_cast<const char*>(alpha_k2) + p0 * 8 + lane * 16;
    dma16(gq, buf);
    dma16(gq + 1024, buf + 1024);
    dma16(gu, buf + 2048);
    dma16(gu + 1024, buf + 3072);
    dma16(ga, buf + 4096);
    dma16(gk, buf + 5120);
}

template <int FIRST>
__global__ __launch_bounds__(kDepthBlock) void depth_lm_dma_kernel(const double2* __restrict__ q,
                                                                   const double2* __restrict__ u,
                                                                   const double2* __restrict__ alpha2,
                                                                   const double2* __restrict__ alpha_k2, int64_t n,
                                                                   Pose pose, double2* __restrict__ rho2,
                                                                   const LmState* __restrict__ state,
                                                                   double* __restrict__ partials, int launch_id) {
    extern __shared__ __attribute__((aligned(16))) char lds[];  // ONE LDS object (cdna guide: a second one costs vmcnt(0)s)
    double(*s_red)[NS] = reinterpret_cast<double(*)[NS]>(lds);
    double* s_inv_hist = reinterpret_cast<double*>(lds + 640);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform: the tile loop runs on scalar branches
    LmPlanReg plan;
    plan.hist_lds = s_inv_hist;
    plan.h0 = plan.h1 = plan.h2 = plan.h3 = 0.0;
    if (launch_id == 0) {
        plan.n_hist = 0;
        plan.K = KMAX;
        const int pr = state->predict;
        plan.write_which = (pr >= 0 && pr <= KMAX) ? pr : 1;
        double r = kInitialRadius;
#pragma unroll
        for (int j = 0; j < KMAX; ++j) {
            plan.inv_cand[j] = 1.0 / r;
            r = radius_accept(r, 1.0);
        }
    } else {
        const int status = state->status;
        if (status == 1 || state->next_launch != launch_id) return;  // finished, or not this launch's turn
        plan.n_hist = state->n_hist;
        plan.K = (status == 2) ? 0 : state->K;
        plan.write_which = (status == 2) ? 0 : state->write_which;
#pragma unroll
        for (int j = 0; j < KMAX; ++j) plan.inv_cand[j] = 1.0 / state->cand[j];
        plan.h0 = 1.0 / state->hist[0];
        plan.h1 = 1.0 / state->hist[1];
        plan.h2 = 1.0 / state->hist[2];
        plan.h3 = 1.0 / state->hist[3];
        if (tid < kMaxIter) s_inv_hist[tid] = 1.0 / state->hist[tid];
        __syncthreads();
    }
    const double two_over = 2.0 / (2.0 + pose.k);
    double acc[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) acc[s] = 0.0;

    char* wbuf = lds + kDmaLdsHeader + wv * (2 * kTileBytes);
    const unsigned lds_a16 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)wbuf + lane * 16;
    const unsigned lds_a8 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)wbuf + lane * 8;
    const int64_t ntiles = n / kTilePoints;
    const int64_t nwaves = (int64_t)gridDim.x * (kDepthBlock / 64);
    int64_t tile = (int64_t)blockIdx.x * (kDepthBlock / 64) + wv;
    double* rho = reinterpret_cast<double*>(rho2);
    int cur = 0;
    if (tile < ntiles) issue_tile(q, u, alpha2, alpha_k2, tile, lane, wbuf);
    while (tile < ntiles) {
        const int64_t next = tile + nwaves;
        if (next < ntiles) {
            issue_tile(q, u, alpha2, alpha_k2, next, lane, wbuf + (cur ^ 1) * kTileBytes);
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");  // tile `tile` has landed; the 6 loads of `next` stay in flight
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        double2 qa, qb, ua, ub;
        double ala, alb, aka, akb;
        const unsigned a16 = lds_a16 + cur * kTileBytes, a8 = lds_a8 + cur * kTileBytes;
        asm volatile(
            "ds_read_b128 %0, %8\n\t"
            "ds_read_b128 %1, %8 offset:1024\n\t"
            "ds_read_b128 %2, %8 offset:2048\n\t"
            "ds_read_b128 %3, %8 offset:3072\n\t"
            "ds_read_b64 %4, %9 offset:4096\n\t"
            "ds_read_b64 %5, %9 offset:4608\n\t"
            "ds_read_b64 %6, %9 offset:5120\n\t"
            "ds_read_b64 %7, %9 offset:5632\n\t"
            "s_waitcnt lgkmcnt(0)"
            : "=&v"(qa), "=&v"(qb), "=&v"(ua), "=&v"(ub), "=&v"(ala), "=&v"(alb), "=&v"(aka), "=&v"(akb)
            : "v"(a16), "v"(a8)
            : "memory");
        const int64_t p0 = tile * kTilePoints;
        const double r0 = lm_pixel(qa.x, qa.y, ua.x, ua.y, ala, aka, pose, two_over, plan, acc);
        const double r1 = lm_pixel(qb.x, qb.y, ub.x, ub.y, alb, akb, pose, two_over, plan, acc);
        rho[p0 + lane] = r0;
        rho[p0 + 64 + lane] = r1;
        tile = next;
        cur ^= 1;
    }
    // remainder (< 128 points): first wave of workgroup 0, ordinary loads
    if (blockIdx.x == 0 && wv == 0) {
        const double* alpha = reinterpret_cast<const double*>(alpha2);
        const double* alpha_k = reinterpret_cast<const double*>(alpha_k2);
        for (int64_t i = ntiles * kTilePoints + lane; i < n; i += 64) {
            const double2 qq = q[i], uu = u[i];
            rho[i] = lm_pixel(qq.x, qq.y, uu.x, uu.y, alpha[i], alpha_k[i], pose, two_over, plan, acc);
        }
    }
    if (plan.K == 0) return;  // apply-only launch: no sums

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
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (launch_id > 0 && (state->status != 0 || state->next_launch != launch_id)) return;
    // the scalar state is fetched now so that its global round trip overlaps the loads of the partials
    LmScal st;
    if (tid == 0) st = *static_cast<const LmScal*>(state);
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
        s_sums[tid] = r;
    }
    __syncthreads();
    if (tid == 0) {  // the state machine runs on a register copy of the scalar state (no LDS / scratch round trips)
        const int pr = st.predict;
        const int used_K = (launch_id == 0) ? KMAX : st.K;
        const int used_write = (launch_id == 0) ? ((pr >= 0 && pr <= KMAX) ? pr : 1) : st.write_which;
        lm_advance(st, state->hist, s_sums, n, launch_id == 0, used_K, used_write, launch_id);
        *static_cast<LmScal*>(state) = st;
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

// grid of the LDS-DMA variant: 3 workgroups per CU (3 x 50 KB of LDS)
static inline int depth_dma_grid(const Ctx* c, int64_t n) {
    const int64_t ntiles = n / kTilePoints;
    int64_t blocks = (ntiles + 3) / 4;
    const int64_t cap = (int64_t)c->num_cus * 3;
    if (blocks < 1) blocks = 1;
    if (blocks > cap) blocks = cap;
    if (blocks > kDepthMaxBlocks) blocks = kDepthMaxBlocks;
    return (int)blocks;
}

int depth_lm_grid(const Ctx* c, int64_t n) { return c->depth_variant == 1 ? depth_dma_grid(c, n) : depth_grid(n, kDepthMaxBlocks); }

// launch 0 of a solve takes the in-range function cores when the caller allows it (`core`: the fast path of variant 0, whose follow-up
// launch checks the flag) and the context has not just had to start a solve over; the epoch a miss is reported with is c->depth_epoch
static inline bool depth_use_cores(const Ctx* c) { return !RSDSFM_FUSED && c->ransac_math_mode == 0 && c->depth_standard_math == 0; }

int depth_lm_launch(Ctx* c, const double* q, const double* u, const double* a, const double* ak, int64_t n,
                    const Pose& pose, double* rho, int launch_id, bool core) {
    if (!aligned16(q) || !aligned16(u) || !aligned16(a) || !aligned16(ak) || !aligned16(rho))
        return fail(c, RSDSFM_ERR_INVALID, "device pointers must be 16-byte aligned");
    const int grid = depth_lm_grid(c, n);
    const double2 *q2 = reinterpret_cast<const double2*>(q), *u2 = reinterpret_cast<const double2*>(u),
                  *a2 = reinterpret_cast<const double2*>(a), *ak2 = reinterpret_cast<const double2*>(ak);
    double2* rho2 = reinterpret_cast<double2*>(rho);
    if (c->depth_variant == 1) {
        static bool attr_set = false;
        if (!attr_set) {
            RSDSFM_HIP_CHECK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(depth_lm_dma_kernel<1>),
                                                    hipFuncAttributeMaxDynamicSharedMemorySize, kDmaLdsBytes));
            RSDSFM_HIP_CHECK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(depth_lm_dma_kernel<0>),
                                                    hipFuncAttributeMaxDynamicSharedMemorySize, kDmaLdsBytes));
            attr_set = true;
        }
        if (launch_id == 0)
            hipLaunchKernelGGL(depth_lm_dma_kernel<1>, dim3(grid), dim3(kDepthBlock), kDmaLdsBytes, c->stream, q2, u2, a2, ak2, n, pose,
                               rho2, c->d_lm, c->d_partials, launch_id);
        else
            hipLaunchKernelGGL(depth_lm_dma_kernel<0>, dim3(grid), dim3(kDepthBlock), kDmaLdsBytes, c->stream, q2, u2, a2, ak2, n, pose,
                               rho2, c->d_lm, c->d_partials, launch_id);
    } else {
        int* predict_used = reinterpret_cast<int*>(c->d_tickets + 40);
        if (launch_id == 0) {
            bool use = false;
            if (core) {
                if (c->depth_standard_math > 0) c->depth_standard_math -= 1;  // (one more solve on the standard functions behind a restart)
                else use = depth_use_cores(c);
            }
            if (use) {
                c->depth_epoch = c->depth_epoch >= 0x3fffffff ? 1 : c->depth_epoch + 1;
                hipLaunchKernelGGL((depth_lm_kernel<1, true>), dim3(grid), dim3(kDepthBlock), 0, c->stream, q2, u2, a2, ak2, n, pose, rho2, c->d_lm,
                                   c->d_partials, launch_id, predict_used, c->depth_epoch);
                c->depth_core_launch = c->depth_epoch;
            } else {
                hipLaunchKernelGGL((depth_lm_kernel<1, false>), dim3(grid), dim3(kDepthBlock), 0, c->stream, q2, u2, a2, ak2, n, pose, rho2, c->d_lm,
                                   c->d_partials, launch_id, predict_used, 0);
                c->depth_core_launch = 0;
            }
        } else {
            hipLaunchKernelGGL((depth_lm_kernel<0, false>), dim3(grid), dim3(kDepthBlock), 0, c->stream, q2, u2, a2, ak2, n, pose, rho2, c->d_lm,
                               c->d_partials, launch_id, static_cast<int*>(nullptr), 0);
        }
    }
    RSDSFM_HIP_CHECK(c, hipGetLastError());
    return RSDSFM_OK;
}

// The follow-up launch decides in EVERY workgroup (redundantly), so in the steady state (nothing to apply) a smaller grid does
// less redundant work: measured 69.5 / 75.2 / 78.9 / 80.5 / 81.4 Gpix/s for 256 / 128 / 64 / 32 / 8 workgroups per pair.  The rare
// apply pass (predictor miss: first solve of a context, or a change of the data regime) grid-strides over the pixels with
// whatever grid it gets; 32 workgroups keep it at ~0.1 ms for 1280x720.
constexpr int kApplyGrid = 32;

// fast-path follow-up of launch 0: decision + apply (variant 0 only; the rows are those of the launch-0 grid)
int depth_lm_decide_apply_launch(Ctx* c, const double* q, const double* u, const double* a, const double* ak, int64_t n,
                                 const Pose& pose, double* rho) {
    const int grid = depth_lm_grid(c, n);
    hipLaunchKernelGGL(depth_lm_decide_apply_kernel, dim3(std::min(grid, kApplyGrid)), dim3(kDepthBlock), 0, c->stream, reinterpret_cast<const double2*>(q),
                       reinterpret_cast<const double2*>(u), reinterpret_cast<const double2*>(a), reinterpret_cast<const double2*>(ak), n, pose,
                       reinterpret_cast<double2*>(rho), c->d_lm, c->d_partials, grid, reinterpret_cast<const int*>(c->d_tickets + 40),
                       c->depth_core_launch);  // (the epoch of the launch 0 in front of it when that ran the function cores, else 0)
    RSDSFM_HIP_CHECK(c, hipGetLastError());
    return RSDSFM_OK;
}

// batched fast path over `count` contexts that share one stream (launched on c[0]'s stream).  Workgroups per solve: with
// several solves per launch the chip is full anyway, and fewer, longer workgroups amortise the reduction epilogue (measured at
// 4 x 1280x720: 80.6 / 82.4 / 81.4 / 81.7 Gpix/s for caps of 512 / 300 / 256 / 180 -> 450 / 300 / 225 / 180 workgroups per pair)
constexpr int kDepthBatchBlocks = 300;
static_assert(kDepthBatchBlocks <= kDepthMaxBlocks, "partial rows are sized for kDepthMaxBlocks");
int depth_lm_batch_launch(Ctx* const* cs, int count, const double* const* q, const double* const* u, const double* const* a,
                          const double* const* ak, const int64_t* n, const Pose* poses, double* const* rho, int launch0_only) {
    Ctx* c0 = cs[0];
    DepthBatchArgs args;
    memset(&args, 0, sizeof(args));
    args.count = count;
    int grid = 1;
    // the function cores in launch 0 when EVERY solve of the batch may take them (one kernel variant per launch) and a follow-up
    // launch is there to check the flags (a launch0_only caller has none)
    bool core = !launch0_only;
    for (int i = 0; i < count; ++i) core = core && depth_use_cores(cs[i]) && cs[i]->depth_variant == 0;
    if (!launch0_only)
        for (int i = 0; i < count; ++i)
            if (cs[i]->depth_standard_math > 0) cs[i]->depth_standard_math -= 1;  // (one more solve on the standard functions behind a restart)
    for (int i = 0; i < count; ++i) {
        if (!aligned16(q[i]) || !aligned16(u[i]) || !aligned16(a[i]) || !aligned16(ak[i]) || !aligned16(rho[i]))
            return fail(c0, RSDSFM_ERR_INVALID, "device pointers must be 16-byte aligned");
        DepthBatchItem& it = args.item[i];
        it.q = reinterpret_cast<const double2*>(q[i]);
        it.u = reinterpret_cast<const double2*>(u[i]);
        it.a2 = reinterpret_cast<const double2*>(a[i]);
        it.ak2 = reinterpret_cast<const double2*>(ak[i]);
        it.rho2 = reinterpret_cast<double2*>(rho[i]);
        it.n = n[i];
        it.pose = poses[i];
        it.state = cs[i]->d_lm;
        it.partials = cs[i]->d_partials;
        it.predict_used = reinterpret_cast<int*>(cs[i]->d_tickets + 40);
        if (core) {
            cs[i]->depth_epoch = cs[i]->depth_epoch >= 0x3fffffff ? 1 : cs[i]->depth_epoch + 1;
            it.core_epoch = cs[i]->depth_epoch;
        }
        cs[i]->depth_core_launch = it.core_epoch;
        grid = std::max(grid, depth_grid(n[i], kDepthBatchBlocks));
    }
    // rsdsfm_set_profiling on the batch's first context: the launch is bracketed by the DISPATCH's own start / stop timestamps
    // (hipExtLaunchKernelGGL events = what rocprofv3 --kernel-trace reports; events recorded around the launch would add the gap)
    const bool prof = c0->profile && c0->ev_prof[0] && c0->ev_prof[1];
    hipEvent_t ev0 = prof ? c0->ev_prof[0] : nullptr, ev1 = prof ? c0->ev_prof[1] : nullptr;
    if (core)
        hipExtLaunchKernelGGL(depth_lm_batch_kernel<true>, dim3(grid, count), dim3(kDepthBlock), 0, c0->stream, ev0, ev1, 0, args);
    else
        hipExtLaunchKernelGGL(depth_lm_batch_kernel<false>, dim3(grid, count), dim3(kDepthBlock), 0, c0->stream, ev0, ev1, 0, args);
    RSDSFM_HIP_CHECK(c0, hipGetLastError());
    if (prof) c0->prof_pending = true, c0->prof_what = 1;
    if (launch0_only) return RSDSFM_OK;
    hipLaunchKernelGGL(depth_lm_decide_apply_batch_kernel, dim3(std::min(grid, kApplyGrid), count), dim3(kDepthBlock), 0, c0->stream, args, grid);
    RSDSFM_HIP_CHECK(c0, hipGetLastError());
    return RSDSFM_OK;
}

int depth_lm_fused_launch(Ctx* c, const double* q, const double* u, const double* a, const double* ak, int64_t n,
                          const Pose& pose, double* rho) {
    if (!aligned16(q) || !aligned16(u) || !aligned16(a) || !aligned16(ak) || !aligned16(rho))
        return fail(c, RSDSFM_ERR_INVALID, "device pointers must be 16-byte aligned");
    const int grid = depth_lm_grid(c, n);
    hipLaunchKernelGGL(depth_lm_fused_kernel, dim3(grid), dim3(kDepthBlock), 0, c->stream, reinterpret_cast<const double2*>(q),
                       reinterpret_cast<const double2*>(u), reinterpret_cast<const double2*>(a), reinterpret_cast<const double2*>(ak), n, pose,
                       reinterpret_cast<double2*>(rho), c->d_lm, c->d_partials, c->d_tickets);
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
    const int grid = depth_lm_grid(c, n);
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
    const int grid = depth_lm_grid(c, n);
    hipLaunchKernelGGL(depth_lm_decide_kernel, dim3(1), dim3(kDecideBlock), 0, c->stream, c->d_partials, grid, c->d_lm, n,
                       launch_id);
    RSDSFM_HIP_CHECK(c, hipGetLastError());
    return RSDSFM_OK;
}

}  // namespace rsdsfm
