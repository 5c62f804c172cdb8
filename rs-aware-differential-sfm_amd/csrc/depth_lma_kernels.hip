// depth_lma_kernels.hip -- the dense per-pixel inverse-depth solve (reference nonlinearRefinement.cc:109-180: N independent 1-D problems
// under ONE Ceres trust-region loop) on the ANALYTIC LM TRAJECTORY (lma_common.hpp), MI355X (gfx950).  The fast path of
// rsdsfm_estimate_inverse_depths(_batch)_dev in RSDSFM_DEPTH_CERES_LM mode since round 5; the iterate-by-iterate kernels (depth_kernels.hip)
// are what a guard falls back to and what rsdsfm_set_lm_arithmetic(1) selects.
//
//   depth_lma_batch_kernel               ONE streaming pass per solve (up to 8 solves per launch): per pixel the closed form's quantities
//                                        (~50 operations instead of ~190), the solve's five sums + gradient maximum, and rho of the PREDICTED
//                                        final iterate (rho* + e0 phi: where the context's previous solve ended) -- 48 B read + 8 B written
//                                        per pixel; pixels whose LM diagonal is clamped (guard a) go to the solve's list
//   depth_lma_decide_apply_batch_kernel  every workgroup, redundantly: reduction of the partial rows, the listed pixels on the reference's exact
//                                        recurrence, Ceres' loop on the closed forms (lma_stages.hpp) -> the state; then the listed pixels get
//                                        their exact rho, and -- only when the prediction was wrong -- every pixel its rho of the iterate
//                                        that IS final.  A guard that trips (c) leaves the solve unfinished with LmScal::restart = 2 and
//                                        rsdsfm_depth_finish_dev runs it again iterate by iterate.
// HBM-bound: 56 B/pixel, ~5 us of arithmetic per 4 x 1280x720 under ~37 us of memory traffic.
#include <hip/hip_ext.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>

#include "lma_common.hpp"
#include "lma_stages.hpp"
#include "rsdsfm_internal.hpp"

namespace rsdsfm {

namespace {

typedef double d2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double2 nt_load2(const double2* p) {
    d2v v = __builtin_nontemporal_load(reinterpret_cast<const d2v*>(p));
    return make_double2(v.x, v.y);
}
__device__ __forceinline__ void nt_store2(double2* p, double2 o) {
    d2v v;
    v.x = o.x;
    v.y = o.y;
    __builtin_nontemporal_store(v, reinterpret_cast<d2v*>(p));
}

struct DepthLmaItem {
    const double2 *q, *u, *a2, *ak2;
    double2* rho2;
    int64_t n;
    Pose pose;
    LmState* state;
    double* partials;   // [workgroups][kLmaSlots]
    int* predict_used;  // [0]: accepted steps of the iterate launch 0 wrote
    int* list;          // [0], [1]: counters of the two parities; then [2][kLmaListCap] pixel indices
    int parity;         // which half of the list this solve uses (the follow-up launch zeroes the other half's counter)
};
struct DepthLmaArgs {
    int count;
    LmaCand cd;  // (no fused iterates: nc = 0; carries the plan)
    DepthLmaItem item[kDepthBatchMax];
};

__device__ __forceinline__ void depth_list_push(int* list, int parity, int64_t i) {
    const int pos = atomicAdd(&list[parity], 1);
    if (pos < kLmaListCap) list[2 + parity * kLmaListCap + pos] = (int)i;
}

// rho of a pixel at the iterate after `steps` accepted steps ON THE PLAN: the closed form, or for a clamped pixel the exact recurrence
__device__ __forceinline__ double depth_lma_rho(double x, double y, double ux, double uy, double al, double ak, const Pose& pose, double two_over,
                                                const LmaPlan& plan, int steps, double phi) {
    if (steps == 0) return 1.0;  // (a solve without an accepted step leaves the start value untouched: exactly 1.0, whatever the pixel holds -- the reference's)
    const LmaPx v = lma_pixel(x, y, ux, uy, al, ak, pose, two_over);
    if (!v.clamped) return __builtin_fma(v.e0, phi, v.rhos);
    LmxWalk wk;
    lmx_walk(x, y, ux, uy, al, ak, pose, two_over, plan, steps < kLmaKP ? steps : kLmaKP, wk);
    return wk.rho[steps < kLmaKP ? steps : kLmaKP];
}

}  // namespace

template <int U>
__global__ __launch_bounds__(kDepthBlock) void depth_lma_batch_kernel(DepthLmaArgs args) {
    __shared__ double s_red[kDepthBlock / 64][6];
    const DepthLmaItem& it = args.item[blockIdx.y];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    // the iterate this launch writes: where the context's previous solve ended (a branch predictor: the follow-up launch rewrites rho
    // when it was wrong)
    const int pr = it.state->predict;
    const int ps = (pr >= 0 && pr <= kLmaKP) ? pr : 1;
    if (blockIdx.x == 0 && tid == 0) *it.predict_used = ps;
    const double phi = args.cd.plan.phi[ps];
    const Pose pose = it.pose;
    const double2 *q = it.q, *u = it.u, *alpha2 = it.a2, *alpha_k2 = it.ak2;
    double2* rho2 = it.rho2;
    const int64_t n = it.n;
    const double two_over = 2.0 / (2.0 + pose.k);
    double A = 0.0, B = 0.0, C = 0.0, D = 0.0, E = 0.0, G = 0.0;
    auto pixel = [&](double x, double y, double ux, double uy, double al, double ak, int64_t i) -> double {
        const LmaPx v = lma_pixel(x, y, ux, uy, al, ak, pose, two_over);
        A += v.a;
        B += v.ge;
        C = __builtin_fma(v.e0, v.e0, C);
        D = __builtin_fma(v.rhos, v.rhos, D);
        E = __builtin_fma(v.rhos, v.e0, E);
        G = lma_max_abs(G, v.g);
        if (v.clamped) depth_list_push(it.list, it.parity, i);  // guard (a): rare; its rho comes from the follow-up launch
        return __builtin_fma(v.e0, phi, v.rhos);
    };
    const int64_t npairs = n >> 1;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    // U pixel pairs per thread and iteration, all their loads issued before the first result is needed
    int64_t p = (int64_t)blockIdx.x * blockDim.x + tid;
    for (; p + (U - 1) * stride < npairs; p += U * stride) {
        double2 qa[U], qb[U], ua[U], ub[U], al[U], ak[U];
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const int64_t pj = p + j * stride;
            qa[j] = nt_load2(q + 2 * pj), qb[j] = nt_load2(q + 2 * pj + 1);
            ua[j] = nt_load2(u + 2 * pj), ub[j] = nt_load2(u + 2 * pj + 1);
            al[j] = nt_load2(alpha2 + pj), ak[j] = nt_load2(alpha_k2 + pj);
        }
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const int64_t pj = p + j * stride;
            double2 out;
            out.x = pixel(qa[j].x, qa[j].y, ua[j].x, ua[j].y, al[j].x, ak[j].x, 2 * pj);
            out.y = pixel(qb[j].x, qb[j].y, ub[j].x, ub[j].y, al[j].y, ak[j].y, 2 * pj + 1);
            nt_store2(rho2 + pj, out);
        }
    }
    for (; p < npairs; p += stride) {
        const double2 qa = nt_load2(q + 2 * p), qb = nt_load2(q + 2 * p + 1);
        const double2 ua = nt_load2(u + 2 * p), ub = nt_load2(u + 2 * p + 1);
        const double2 al = nt_load2(alpha2 + p), ak = nt_load2(alpha_k2 + p);
        double2 out;
        out.x = pixel(qa.x, qa.y, ua.x, ua.y, al.x, ak.x, 2 * p);
        out.y = pixel(qb.x, qb.y, ub.x, ub.y, al.y, ak.y, 2 * p + 1);
        nt_store2(rho2 + p, out);
    }
    if ((n & 1) && blockIdx.x == 0 && tid == 0) {
        const int64_t i = n - 1;
        const double2 qa = q[i], ua = u[i];
        const double* alpha = reinterpret_cast<const double*>(alpha2);
        const double* alpha_k = reinterpret_cast<const double*>(alpha_k2);
        reinterpret_cast<double*>(rho2)[i] = pixel(qa.x, qa.y, ua.x, ua.y, alpha[i], alpha_k[i], i);
    }
    // workgroup partial: lanes (DPP), then the waves in order
    A = wave_sum(A), B = wave_sum(B), C = wave_sum(C), D = wave_sum(D), E = wave_sum(E), G = wave_max(G);
    if (lane == 0) s_red[wv][0] = A, s_red[wv][1] = B, s_red[wv][2] = C, s_red[wv][3] = D, s_red[wv][4] = E, s_red[wv][5] = G;
    __syncthreads();
    if (tid < kLmaSlots) {
        double r = 0.0;
        if (tid < 6) {
            r = s_red[0][tid];
            for (int w2 = 1; w2 < kDepthBlock / 64; ++w2) r = tid == kLmaG ? fmax(r, s_red[w2][tid]) : r + s_red[w2][tid];
        }
        it.partials[(int64_t)blockIdx.x * kLmaSlots + tid] = r;
    }
}

// decision on the solve's row (LDS, every workgroup redundantly) -> the state; then the rho of the listed pixels, and of every pixel when the
// prediction was wrong.  n_decide: the points the ROW stands for (all ranks' shards in the row-tiled solve; it.n otherwise)
__device__ __forceinline__ void depth_lma_finish(const DepthLmaItem& it, const LmaCand& cd, const double* s_row, int64_t n_decide) {
    __shared__ double s_hist[kMaxIter];
    __shared__ double s_phi;
    __shared__ int s_mode, s_steps;  // s_mode: 0 = nothing but the listed pixels to write, 1 = every pixel, -1 = unfinished (a guard tripped)
    const int tid = threadIdx.x;
    const double* alpha = reinterpret_cast<const double*>(it.a2);
    const double* alpha_k = reinterpret_cast<const double*>(it.ak2);
    const int* cnt = it.list + it.parity;
    const int* idx = it.list + 2 + it.parity * kLmaListCap;
    if (tid == 0) {
        LmScal st;
        bool scored;
        double count, err, phi = 1.0;
        const int fb = lma_decide(s_row, n_decide, cd, cd.plan, false, st, s_hist, scored, count, err, &phi);
        const int ps = *it.predict_used;
        st.launches = 2;
        st.next_launch = 2;
        st.predict = st.n_hist <= kLmaKP ? st.n_hist : kLmaKP;
        if (fb) {  // unfinished: the caller starts the solve over iterate by iterate
            st.status = 0;
            st.restart = 2;
            st.termination = -1;
            st.next_launch = 1;
            st.iteration = fb;  // (which guard: diagnostics)
            s_mode = -1;
        } else {
            st.status = 1;
            st.rho_holds = st.n_hist;
            s_mode = (st.n_hist == ps && phi == cd.plan.phi[ps]) ? 0 : 1;
        }
        s_phi = phi;
        s_steps = st.n_hist;
        if (blockIdx.x == 0) {
            *static_cast<LmScal*>(it.state) = st;
            for (int h = 0; h < st.n_hist; ++h) it.state->hist[h] = s_hist[h];
            it.list[it.parity ^ 1] = 0;  // the other half's counter: the context's next solve starts with an empty list
        }
    }
    __syncthreads();
    const int mode = s_mode, steps = s_steps;
    if (mode < 0) return;
    const double phi = s_phi;
    const Pose pose = it.pose;
    const double two_over = 2.0 / (2.0 + pose.k);
    double* rho = reinterpret_cast<double*>(it.rho2);
    if (mode == 0) {  // the prediction held: only the listed (clamped) pixels still need their rho -- the exact recurrence's
        if (blockIdx.x != 0) return;
        const int nl = min(*cnt, kLmaListCap);
        for (int e = tid; e < nl; e += kDepthBlock) {
            const int64_t i = idx[e];
            const double2 qq = it.q[i], uu = it.u[i];
            rho[i] = depth_lma_rho(qq.x, qq.y, uu.x, uu.y, alpha[i], alpha_k[i], pose, two_over, cd.plan, steps, phi);
        }
        return;
    }
    // another iterate is final: every pixel gets its rho of that one
    const int64_t n = it.n;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + tid; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const double2 qq = it.q[i], uu = it.u[i];
        rho[i] = depth_lma_rho(qq.x, qq.y, uu.x, uu.y, alpha[i], alpha_k[i], pose, two_over, cd.plan, steps, phi);
    }
}

__global__ __launch_bounds__(kDepthBlock) void depth_lma_decide_apply_batch_kernel(DepthLmaArgs args, int nrows) {
    static_assert(kDepthBlock == kLB, "the stages are written for 256 threads");
    __shared__ double s_row[kLmaRow];
    const DepthLmaItem& it = args.item[blockIdx.y];
    lma_rows_stage(it.partials, nrows, 1, 0, it.q, it.u, reinterpret_cast<const double*>(it.a2), reinterpret_cast<const double*>(it.ak2), it.pose, args.cd,
                   it.list + it.parity, it.list + 2 + it.parity * kLmaListCap, s_row);
    __syncthreads();
    depth_lma_finish(it, args.cd, s_row, it.n);
}

// row-tiled solve (one shard of the points per rank): the shard's row (one workgroup) -> all-gather -> the rows of all ranks added in rank
// order, the decision replicated on every rank (and in every workgroup), the shard's rho
__global__ __launch_bounds__(kDepthBlock) void depth_lma_rows_kernel(DepthLmaArgs args, int nrows, double* __restrict__ row_out) {
    __shared__ double s_row[kLmaRow];
    const DepthLmaItem& it = args.item[0];
    lma_rows_stage(it.partials, nrows, 1, 0, it.q, it.u, reinterpret_cast<const double*>(it.a2), reinterpret_cast<const double*>(it.ak2), it.pose, args.cd,
                   it.list + it.parity, it.list + 2 + it.parity * kLmaListCap, s_row);
    __syncthreads();
    if (threadIdx.x < kLmaRow) row_out[threadIdx.x] = s_row[threadIdx.x];
}
__global__ __launch_bounds__(kDepthBlock) void depth_lma_decide_apply_rows_kernel(DepthLmaArgs args, const double* __restrict__ rows_all, int nranks,
                                                                                 int64_t n_total) {
    __shared__ double s_row[kLmaRow];
    const int j = threadIdx.x;
    if (j < kLmaRow) {
        const bool is_max = j == kLmaG || j == 7 || j == kLmaRowX0 + 1 || (j >= kLmaRowXk && j < kLmaRowScore && ((j - kLmaRowXk) % 5) == 4);
        double r = rows_all[j];
        for (int rk = 1; rk < nranks; ++rk) r = is_max ? fmax(r, rows_all[(int64_t)rk * kLmaRow + j]) : r + rows_all[(int64_t)rk * kLmaRow + j];
        s_row[j] = r;
    }
    __syncthreads();
    depth_lma_finish(args.item[0], args.cd, s_row, n_total);
}

// ---------------------------------------------------------------------------------------------------
// launcher
// ---------------------------------------------------------------------------------------------------
// workgroups per solve and pixel pairs per thread and iteration, measured on the memory-bound kernel (us per launch / HBM fraction; 1280x720 pairs):
// 4 pairs per launch: 300 x 1: 41.9 / 0.616, 512 x 1: 40.9 / 0.631, 512 x 2: 41.0 / 0.630, 256 x 2: 42.9 / 0.601, 200 x 2: 46.6 / 0.553;
// 8 pairs per launch: 300 x 1: 80.6 / 0.640, 512 x 1: 78.0 / 0.662, 512 x 2: 76.5 / 0.675
constexpr int kDepthLmaBlocks = 512;
constexpr int kDepthLmaUnroll = 2;
constexpr int kDepthLmaApplyGrid = 32;
static_assert(kDepthLmaBlocks <= kDepthMaxBlocks && kLmaSlots <= NS, "partial rows fit the context's partial buffer");

static inline bool aligned16p(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// may this context's next dense depth solve take the analytic fast path?
// (no hold behind a guard, unlike the RANSAC: there both arithmetics return the same bits in everything but diagnostics; here rho itself differs
// in its last digits between them, and a result must not depend on what the context solved before -- every solve tries the analytic path and a
// solve whose guards trip is run again iterate by iterate: a function of its inputs alone)
bool depth_lma_allowed(const Ctx* c, int64_t n) { return c->lm_arithmetic == 0 && c->depth_variant == 0 && n <= (int64_t)INT32_MAX; }

static int depth_lma_item(Ctx* c0, Ctx* c, const double* q, const double* u, const double* a, const double* ak, int64_t n, const Pose& pose, double* rho,
                          bool new_solve, DepthLmaItem& it) {
    if (!aligned16p(q) || !aligned16p(u) || !aligned16p(a) || !aligned16p(ak) || !aligned16p(rho)) return fail(c0, RSDSFM_ERR_INVALID, "device pointers must be 16-byte aligned");
    if (!c->d_lma_list) {
        const size_t bytes = sizeof(int) * (2 + 2 * (size_t)kLmaListCap);
        RSDSFM_HIP_CHECK(c0, hipMalloc(reinterpret_cast<void**>(&c->d_lma_list), bytes));
        RSDSFM_HIP_CHECK(c0, hipMemsetAsync(c->d_lma_list, 0, bytes, c0->stream));
        c->depth_lma_parity = 0;
    }
    if (new_solve) c->depth_lma_parity ^= 1;
    it.q = reinterpret_cast<const double2*>(q);
    it.u = reinterpret_cast<const double2*>(u);
    it.a2 = reinterpret_cast<const double2*>(a);
    it.ak2 = reinterpret_cast<const double2*>(ak);
    it.rho2 = reinterpret_cast<double2*>(rho);
    it.n = n;
    it.pose = pose;
    it.state = c->d_lm;
    it.partials = c->d_partials;
    it.predict_used = reinterpret_cast<int*>(c->d_tickets + 40);
    it.list = c->d_lma_list;
    it.parity = c->depth_lma_parity;
    c->depth_core_launch = 0;
    return RSDSFM_OK;
}
static void depth_lma_args_init(DepthLmaArgs& args, int count) {
    memset(&args, 0, sizeof(args));
    args.count = count;
    args.cd.nc = 0;
    args.cd.plan = lma_plan();
    for (int c2 = 0; c2 < kLmaNC; ++c2) args.cd.steps[c2] = 1, args.cd.phi2[c2] = 0.0;
}
static int depth_lma_blocks(int64_t n) {
    int64_t blocks = (n / 2 + kDepthBlock - 1) / kDepthBlock;
    return (int)std::max<int64_t>(1, std::min<int64_t>(blocks, kDepthLmaBlocks));
}

// launch 0 + the follow-up launch of `count` solves (contexts sharing one stream); every context must satisfy depth_lma_allowed
int depth_lma_batch_launch(Ctx* const* cs, int count, const double* const* q, const double* const* u, const double* const* a,
                           const double* const* ak, const int64_t* n, const Pose* poses, double* const* rho) {
    Ctx* c0 = cs[0];
    DepthLmaArgs args;
    depth_lma_args_init(args, count);
    int grid = 1;
    for (int i = 0; i < count; ++i) {
        const int rc = depth_lma_item(c0, cs[i], q[i], u[i], a[i], ak[i], n[i], poses[i], rho[i], true, args.item[i]);
        if (rc != RSDSFM_OK) return rc;
        grid = std::max(grid, depth_lma_blocks(n[i]));
    }
    // rsdsfm_set_profiling on the batch's first context: the launch is bracketed by the DISPATCH's own start / stop timestamps
    const bool prof = c0->profile && c0->ev_prof[0] && c0->ev_prof[1];
    hipEvent_t ev0 = prof ? c0->ev_prof[0] : nullptr, ev1 = prof ? c0->ev_prof[1] : nullptr;
    hipExtLaunchKernelGGL(depth_lma_batch_kernel<kDepthLmaUnroll>, dim3(grid, count), dim3(kDepthBlock), 0, c0->stream, ev0, ev1, 0, args);
    RSDSFM_HIP_CHECK(c0, hipGetLastError());
    if (prof) c0->prof_pending = true, c0->prof_what = 1;
    hipLaunchKernelGGL(depth_lma_decide_apply_batch_kernel, dim3(std::min(grid, kDepthLmaApplyGrid), count), dim3(kDepthBlock), 0, c0->stream, args, grid);
    RSDSFM_HIP_CHECK(c0, hipGetLastError());
    return RSDSFM_OK;
}

// row-tiled solve, this rank's shard: launch 0 + the shard's row (depth_lma_row_doubles() doubles: the all-gather payload) ...
int depth_lma_row_doubles() { return kLmaRow; }
int depth_lma_shard_launch(Ctx* c, const double* q, const double* u, const double* a, const double* ak, int64_t n, const Pose& pose, double* rho, double* d_row) {
    DepthLmaArgs args;
    depth_lma_args_init(args, 1);
    const int rc = depth_lma_item(c, c, q, u, a, ak, n, pose, rho, true, args.item[0]);
    if (rc != RSDSFM_OK) return rc;
    const int grid = depth_lma_blocks(n);
    hipLaunchKernelGGL(depth_lma_batch_kernel<kDepthLmaUnroll>, dim3(grid, 1), dim3(kDepthBlock), 0, c->stream, args);
    RSDSFM_HIP_CHECK(c, hipGetLastError());
    hipLaunchKernelGGL(depth_lma_rows_kernel, dim3(1), dim3(kDepthBlock), 0, c->stream, args, grid, d_row);
    RSDSFM_HIP_CHECK(c, hipGetLastError());
    return RSDSFM_OK;
}
// ... and, on the gathered rows of all ranks, the replicated decision + this shard's rho
int depth_lma_shard_finish_launch(Ctx* c, const double* q, const double* u, const double* a, const double* ak, int64_t n, const Pose& pose, double* rho,
                                  const double* d_rows_all, int nranks, int64_t n_total) {
    DepthLmaArgs args;
    depth_lma_args_init(args, 1);
    const int rc = depth_lma_item(c, c, q, u, a, ak, n, pose, rho, false, args.item[0]);
    if (rc != RSDSFM_OK) return rc;
    hipLaunchKernelGGL(depth_lma_decide_apply_rows_kernel, dim3(std::min(depth_lma_blocks(n), kDepthLmaApplyGrid)), dim3(kDepthBlock), 0, c->stream, args, d_rows_all,
                       nranks, n_total);
    RSDSFM_HIP_CHECK(c, hipGetLastError());
    return RSDSFM_OK;
}

}  // namespace rsdsfm
