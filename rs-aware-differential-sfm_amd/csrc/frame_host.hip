// frame_host.hip -- rsdsfm_solve_frame_dev: the solver part of the reference's evaluateSingleRun() (main.cc:398-522) as
// ONE call on device-resident buffers: flatten + alpha -> RANSAC -> nonlinear refinement -> sign flip + depth map ->
// per-scanline pose table; and rsdsfm_solve_frames_dev: a SEQUENCE of frame pairs through one context (BASELINE configs[4]).
// Pure orchestration of the stage entry points (no extra kernels).
//
// One solve is two halves.  frame_begin enqueues everything that can be enqueued before the host has to look at a result -- in the
// common case the WHOLE chain: the minimal solver straight from the flow image with the flatten beside it on a second stream, round 0
// of the hypothesis-batched depth solves, the speculated final stage, the start of the refinement, its first chunk of LM iterations,
// the output pass, the depth map and the pose table -- and returns without waiting.  frame_finish waits, checks what the speculation
// assumed, and drives whatever is left (more LM rounds, further refinement chunks, a frame with dropped pixels).  The single solve
// runs the halves back to back; the sequence solve begins the next pairs on other streams ("lanes") between them, so that the
// latency-bound kernels of one pair (the minimal solver: 50 waves for 186 us; the single-workgroup decide / solve stages) run beside
// the streaming kernels of another.  One host thread, one context, per-pair results identical to the single solve.
#include <stdlib.h>
#include <string.h>

#include <atomic>

#include <algorithm>
#include <new>

#include "rsdsfm_internal.hpp"

namespace rsdsfm {

// Frame solves between their begin and the end of their finish, per device and over ALL contexts of the process: a solve that is not alone on
// its GPU (several host threads with a context each; the lanes of a sequence) gives the refinement's single-workgroup stage launches of its own
// instead of the redundant prologue (Ctx::refine_stage_mode, refine_kernels.hip) -- a scheduling choice, never a result.
static std::atomic<int> g_frames_in_flight[64];
int frames_in_flight(const Ctx* c) { return g_frames_in_flight[c->device & 63].load(std::memory_order_relaxed); }

int alpha_ones_launch(Ctx* c, double* d_alpha, int64_t n);
int flatten_device(Ctx* c, const double* d_img, int32_t rows, int32_t cols, int32_t col0, double fx, double fy, double cx, double cy, double gamma,
                   double thr, double* d_q, double* d_u, double* d_alpha, double* d_alpha_k, int64_t* n_out);

// everything one frame solve keeps between its two halves (one per context / lane, at a fixed address: the tails capture it)
struct FrameRun {
    rsdsfm_frame_job job;
    rsdsfm_frame_params prm;
    size_t N = 0;
    int64_t n = 0;
    double *d_q = nullptr, *d_u = nullptr, *d_a = nullptr, *d_ak = nullptr, *d_in_a = nullptr, *d_in_ak = nullptr, *d_inl = nullptr, *d_inl_ref = nullptr;
    int64_t* d_idx = nullptr;
    uint8_t* d_mask = nullptr;
    int32_t* d_ys = nullptr;
    double *d_zpartials = nullptr, *d_zheader = nullptr;
    char* d_refine_ws = nullptr;
    rsdsfm_ransac_out ro;
    RansacRun ransac;
    RefineRun refine;
    Minimal9Direct direct;
    DenseFlatten dense;
    bool refinement_enqueued = false, ahead = false, side_flatten = false, dense_in_launch = false, open = false;
    bool counted = false;  // this run is in g_frames_in_flight (begin counted it, nothing has taken it out yet)
    int rc_begin = RSDSFM_OK;
    int64_t m_known = -1;  // the inlier count once the host has it; until then the kernels read it from the refinement's state
    RefineTail tail;
    RansacSpecTail spec_tail;
    std::function<int()> join;
};

namespace {

// the reserved tail of the context's pinned block (kPinnedTail bytes): state read-back | point count ... depth-map header
// (pointers into the pinned block are taken when they are used: the RANSAC may still grow the block)
inline double* header_host(Ctx* c) { return reinterpret_cast<double*>(static_cast<char*>(c->h_pinned) + c->pinned_bytes - 64); }
inline RefineState* state_host(Ctx* c) { return reinterpret_cast<RefineState*>(static_cast<char*>(c->h_pinned) + c->pinned_bytes - kPinnedTail); }
inline int64_t* count_host(Ctx* c) { return reinterpret_cast<int64_t*>(static_cast<char*>(c->h_pinned) + c->pinned_bytes - kPinnedTail / 2); }
static_assert(sizeof(RefineState) + sizeof(int) <= kPinnedTail / 2, "reserved tail: state read-back | point count ... depth-map header");

// work enqueued on the context's second stream for the duration of a scope
struct StreamSwap {
    Ctx* c;
    hipStream_t main;
    StreamSwap(Ctx* ctx, hipStream_t other) : c(ctx), main(ctx->stream) { c->stream = other; }
    ~StreamSwap() { c->stream = main; }
};

int ensure_side_stream(Ctx* c) {
    if (c->aux_stream) return RSDSFM_OK;
    RSDSFM_HIP_CHECK(c, hipStreamCreateWithFlags(&c->aux_stream, hipStreamNonBlocking));
    RSDSFM_HIP_CHECK(c, hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
    RSDSFM_HIP_CHECK(c, hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
    return RSDSFM_OK;
}

// takes a run out of the count exactly once: at the end of its finish, when it is begun again without having been finished (a caller that
// abandoned it on an error), or when its context goes away with the run still open
static void frame_uncount(const Ctx* c, FrameRun* F) {
    if (F && F->counted) {
        F->counted = false;
        g_frames_in_flight[c->device & 63].fetch_sub(1, std::memory_order_relaxed);
    }
}
struct FrameInFlightEnd {
    const Ctx* c;
    FrameRun* F;
    ~FrameInFlightEnd() { frame_uncount(c, F); }
};

int frame_begin(Ctx* c, FrameRun* F) {
    const rsdsfm_frame_job& J = F->job;
    const rsdsfm_frame_params* prm = &F->prm;
    frame_uncount(c, F);  // (a run that was begun and never finished)
    F->open = false;
    F->rc_begin = RSDSFM_OK;
    if (J.rows <= 0 || J.cols <= 0 || !J.d_flow_img || !J.d_depth_map_colmajor) return fail(c, RSDSFM_ERR_INVALID, "bad arguments");
    if (prm->flow_index_mode != RSDSFM_FLOW_COMPAT_RANK && prm->flow_index_mode != RSDSFM_FLOW_GATHERED) return fail(c, RSDSFM_ERR_INVALID, "unknown flow_index_mode");
    if (prm->struct_bytes != 0 && prm->struct_bytes != (int32_t)sizeof(rsdsfm_frame_params))
        return fail(c, RSDSFM_ERR_INVALID, "rsdsfm_frame_params: struct_bytes is neither 0 nor sizeof(rsdsfm_frame_params) -- caller built against another header (use rsdsfm_frame_params_init)");
    const int32_t rows = J.rows, cols = J.cols;
    const double fx = J.fx, fy = J.fy, cx = J.cx, cy = J.cy, gamma = J.gamma;
    const size_t N = (size_t)rows * (size_t)cols;
    F->N = N;
    // frame buffers live in the context's frame arena (separate from the per-stage workspace)
    const size_t ncells = (size_t)flatten_cells(rows, cols);
    const size_t need = 2 * Arena::need(16 * N) + 4 * Arena::need(8 * N) + 2 * Arena::need(24 * N) + Arena::need(8 * N) + Arena::need(N) +
                        Arena::need(4 * N) + Arena::need(8 * 1024) + Arena::need(64) + Arena::need(refine_workspace_bytes(c, (int64_t)N, true)) +
                        2 * Arena::need(sizeof(int64_t) * (ncells + 2048)) + Arena::need(64) + 4096;
    if (need > c->frame_bytes) {
        RSDSFM_HIP_CHECK(c, hipStreamSynchronize(c->stream));
        if (c->d_frame) RSDSFM_HIP_CHECK(c, hipFree(c->d_frame));
        c->d_frame = nullptr;
        c->frame_bytes = 0;
        RSDSFM_HIP_CHECK(c, hipMalloc(&c->d_frame, need));
        c->frame_bytes = need;
    }
    Arena fa(c->d_frame);
    F->d_q = fa.take<double>(2 * N);
    F->d_u = fa.take<double>(2 * N);
    F->d_a = fa.take<double>(N);
    F->d_ak = fa.take<double>(N);
    F->d_in_a = fa.take<double>(N);
    F->d_in_ak = fa.take<double>(N);
    F->d_inl = fa.take<double>(3 * N);
    F->d_inl_ref = fa.take<double>(3 * N);
    F->d_idx = fa.take<int64_t>(N);
    F->d_mask = fa.take<uint8_t>(N);
    F->d_ys = fa.take<int32_t>(N);
    F->d_zpartials = fa.take<double>(1024);  // scratch of the depth-map stage when it is enqueued behind the refinement
    F->d_zheader = fa.take<double>(8);
    // buffers of a refinement that is enqueued while the RANSAC still owns the stage workspace (see below)
    F->d_refine_ws = fa.take<char>(refine_workspace_bytes(c, (int64_t)N, true));
    // scan scratch of a flatten that runs beside the minimal solver (which clears / fills the head of the stage workspace meanwhile)
    int64_t* d_flat_counts = fa.take<int64_t>(ncells);
    int64_t* d_flat_offsets = fa.take<int64_t>(ncells + 2048);  // (+ the segment totals of the two-level scan)

    // The flatten is enqueued WITHOUT waiting for its point count: a dense optical flow gives every pixel a flow vector, so the RANSAC
    // (sampler, grids) is set up for n = rows * cols right away; the real count arrives in host-mapped memory and is checked at the
    // RANSAC's own wait.  If pixels were dropped (zero flow below the threshold) the stages are simply run again with the real count.
    // (Only behind a frame that WAS dense: a sequence of frames with pixels without flow -- e.g. the ground-truth flow of a synthetic
    // example with void pixels -- waits for the count instead of paying for a discarded RANSAC every time.  A hint like the
    // others: it decides what is enqueued when, never a result.)
    // With every pixel kept, point i of the list IS pixel (column i / rows, row i % rows): the minimal solver forms its 9 T sampled
    // points straight from the flow image (Minimal9Direct: the flatten's own expressions) and no longer waits for the flatten, which
    // runs BESIDE it on the context's second stream and is joined in front of the first pass over all points.
    int rc = ensure_pinned(c, ransac_pinned_bytes(prm->ransac_trials));  // (sized up front: pointers into the block stay valid)
    if (rc != RSDSFM_OK) return rc;
    int64_t n = (int64_t)N;
    F->side_flatten = false;
    if (c->frame_dense_hint) {
        *count_host(c) = -1;
        int side = prm->ransac_trials > 0 ? c->frame_side_flatten : 0;
        if (side == 3 && prm->ransac_trials > c->num_cus * 2) side = 0;  // (the solver then runs one hypothesis per LANE: no spare workgroups to speak of)
        F->dense_in_launch = false;
        if (side) {
            F->side_flatten = true;
            F->direct = Minimal9Direct();
            F->direct.img = J.d_flow_img, F->direct.rows = rows, F->direct.cols = cols, F->direct.alpha_ones = prm->use_global_shutter_mode ? 1 : 0;
            F->direct.fx = fx, F->direct.fy = fy, F->direct.cx = cx, F->direct.cy = cy, F->direct.gamma = gamma;
        }
        if (side == 3) {
            // the flatten INSIDE the minimal solver's launch: the first T workgroups solve, the others flatten the dense frame to its
            // known positions and count the pixels they had to drop (minimal9_flatten_kernel); no scan, no second stream, no join
            if (!c->d_flat_counters) {
                RSDSFM_HIP_CHECK(c, hipMalloc(reinterpret_cast<void**>(&c->d_flat_counters), 2 * sizeof(unsigned long long)));
                RSDSFM_HIP_CHECK(c, hipMemsetAsync(c->d_flat_counters, 0, 2 * sizeof(unsigned long long), c->stream));
            }
            F->dense = DenseFlatten();
            F->dense.thr = prm->flow_threshold;
            F->dense.d_q = F->d_q, F->dense.d_u = F->d_u, F->dense.d_alpha = F->d_a, F->dense.d_alpha_k = F->d_ak;
            F->dense.d_counters = c->d_flat_counters;
            F->dense.total_out = count_host(c);
            F->dense_in_launch = true;
        } else if (side == 2) {
            // the flatten BEHIND the minimal solver on the context's stream: the solver (one wave per hypothesis, ~186 us) is the first
            // kernel of the solve, and while it runs the host enqueues everything else -- the short flatten kernels then start back to
            // back instead of as fast as the host can enqueue them
            F->join = [c, F, d_flat_counts, d_flat_offsets]() -> int {
                const rsdsfm_frame_job& J2 = F->job;
                int rc2 = flatten_launch(c, J2.d_flow_img, J2.rows, J2.cols, 0, J2.fx, J2.fy, J2.cx, J2.cy, J2.gamma, F->prm.flow_threshold, F->d_q, F->d_u, F->d_a,
                                         F->d_ak, d_flat_counts, d_flat_offsets, count_host(c), nullptr);
                if (rc2 == RSDSFM_OK && F->prm.use_global_shutter_mode) rc2 = alpha_ones_launch(c, F->d_a, (int64_t)F->N);
                return rc2;
            };
        } else if (side == 1) {
            // the flatten BESIDE the minimal solver on the context's second stream, joined in front of the first pass over all points
            rc = ensure_side_stream(c);
            if (rc != RSDSFM_OK) return rc;
            RSDSFM_HIP_CHECK(c, hipEventRecord(c->ev_fork, c->stream));  // (the caller's uploads on the context's stream come first)
            RSDSFM_HIP_CHECK(c, hipStreamWaitEvent(c->aux_stream, c->ev_fork, 0));
            {
                StreamSwap on_side(c, c->aux_stream);
                rc = flatten_launch(c, J.d_flow_img, rows, cols, 0, fx, fy, cx, cy, gamma, prm->flow_threshold, F->d_q, F->d_u, F->d_a, F->d_ak,
                                    d_flat_counts, d_flat_offsets, count_host(c), nullptr);
                if (rc == RSDSFM_OK && prm->use_global_shutter_mode) rc = alpha_ones_launch(c, F->d_a, n);  // main.cc:441-444: alpha *= 0; alpha += 1
            }
            if (rc != RSDSFM_OK) return rc;
            RSDSFM_HIP_CHECK(c, hipEventRecord(c->ev_join, c->aux_stream));
            F->join = [c]() -> int {
                RSDSFM_HIP_CHECK(c, hipStreamWaitEvent(c->stream, c->ev_join, 0));
                return RSDSFM_OK;
            };
        } else {
            rc = flatten_enqueue(c, J.d_flow_img, rows, cols, fx, fy, cx, cy, gamma, prm->flow_threshold, F->d_q, F->d_u, F->d_a, F->d_ak, count_host(c));
            if (rc != RSDSFM_OK) return rc;
        }
    } else {
        rc = flatten_device(c, J.d_flow_img, rows, cols, 0, fx, fy, cx, cy, gamma, prm->flow_threshold, F->d_q, F->d_u, F->d_a, F->d_ak, &n);
        if (rc != RSDSFM_OK) return rc;
        *count_host(c) = n;
    }
    if (prm->use_global_shutter_mode && !F->side_flatten) {  // main.cc:441-444: alpha *= 0; alpha += 1
        rc = alpha_ones_launch(c, F->d_a, n);
        if (rc != RSDSFM_OK) return rc;
    }
    F->n = n;
    memset(&F->ro, 0, sizeof(F->ro));
    F->ro.inlier_idx = F->d_idx;
    F->ro.inliers = F->d_inl;
    F->ro.alpha = F->d_in_a;
    F->ro.alpha_k = F->d_in_ak;
    F->ro.mask = F->d_mask;
    // ---- RANSAC, refinement, depth map + pose table: three stages, ONE host wait in the common case ------------------------------
    // Each later stage only needs the device-resident result of the one before it, so it is enqueued BEHIND that stage before the host
    // has read anything:
    //  * the depth map and the pose table behind the refinement's output pass (they read v / w / k from the refinement's state and
    //    report {flipped, v'} through host-mapped memory, written by zsum_decide_kernel);
    //  * the refinement's start (state from the device-resident RansacBest, iteration zero, the first chunk of LM iterations, output
    //    pass, and that tail) behind the RANSAC's speculated final stage -- its buffers live in the frame arena, since the RANSAC
    //    may still need the stage workspace for further rounds.
    // If the RANSAC's speculation did not hold (more LM rounds / a scoring pass) the refinement starts over from the host-side result;
    // if LM iterations remain after a chunk the tail runs again behind the next output pass.  None of this changes a result.
    F->m_known = -1;
    F->tail = [c, F](const RefineBuffers& B) -> int {
        const rsdsfm_frame_job& J2 = F->job;
        const RefineState* st = B.state;
        const int64_t m_arg = F->m_known >= 0 ? F->m_known : F->n;
        const int64_t* m_dev = F->m_known >= 0 ? nullptr : &st->m;
        PoseTableOut pt;  // the pose table of (v', w, k) is written by the kernel that decides the sign of v
        if (J2.d_R_rows9_or_null && J2.d_t_rows3_or_null)
            pt.R = J2.d_R_rows9_or_null, pt.t = J2.d_t_rows3_or_null, pt.rows = J2.rows, pt.gamma = J2.gamma, pt.wk_dev = st->p + 3;
        // (the sums of z come from the refinement's output pass: one entry per workgroup of that launch)
        return depth_map_slab_launch(c, F->d_inl_ref, m_arg, F->d_zpartials, refine_finish_grid(c, B), m_arg, nullptr, J2.fx, J2.fy, J2.cx, J2.cy, J2.rows,
                                     0, J2.cols, J2.d_depth_map_colmajor, nullptr, F->d_ys, F->d_zheader, header_host(c), st->p, m_dev, &pt);
    };
    F->spec_tail = [c, F](const RansacBest* d_best) -> int {
        return refine_begin(c, F->d_u, F->n, F->n, F->d_inl, F->d_in_a, F->d_in_ak, F->d_idx, nullptr, nullptr, 0.0, F->prm.use_acceleration_mode,
                            F->prm.flow_index_mode, F->d_inl_ref, &F->tail, d_best, F->d_refine_ws, &F->refine, state_host(c), F->d_zpartials);
    };
    F->refinement_enqueued = false;
    // (the refinement goes behind the SPECULATED final stage unless that stage did not count in the last two RANSACs: data whose
    // hypotheses need further LM rounds every time -- noise-free flow -- would pay ~26 launches that leave at once per frame; an isolated
    // miss -- one DeepFlow-like pair in ten has a hypothesis with three accepted steps -- does not switch it off.  Either way the
    // refinement is enqueued behind the definitive final stage from the device-resident result, without a host round trip.)
    F->ahead = prm->use_refinement && c->ransac_spec_miss < 2;
    F->open = true;
    g_frames_in_flight[c->device & 63].fetch_add(1, std::memory_order_relaxed);
    F->counted = true;
    F->rc_begin = ransac_begin(c, F->d_q, F->d_u, F->d_a, F->d_ak, n, prm->use_acceleration_mode, prm->ransac_trials, prm->ransac_tol, nullptr, J.seed,
                               prm->depth_mode, prm->k_sign_mode, &F->ro, prm->use_refinement ? &F->spec_tail : nullptr, &F->refinement_enqueued, &F->ransac,
                               F->side_flatten ? &F->direct : nullptr, F->side_flatten && !F->dense_in_launch ? &F->join : nullptr,
                               F->dense_in_launch ? &F->dense : nullptr, F->ahead, true);
    return RSDSFM_OK;  // (an error of the speculated run may only mean that n was wrong: frame_finish sorts that out)
}

int frame_finish(Ctx* c, FrameRun* F, rsdsfm_frame_result* res) {
    if (!F->open) return fail(c, RSDSFM_ERR_INVALID, "no frame solve in flight");
    F->open = false;
    FrameInFlightEnd in_flight_end_{c, F};  // (until this function returns: the solve's kernels occupy the GPU while the host waits)
    const rsdsfm_frame_job& J = F->job;
    const rsdsfm_frame_params* prm = &F->prm;
    memset(res, 0, sizeof(*res));
    int rc = F->rc_begin;
    if (rc == RSDSFM_OK) rc = ransac_finish(c, &F->ransac);
    if (rc != RSDSFM_OK) RSDSFM_HIP_CHECK(c, hipStreamSynchronize(c->stream));  // (an error of the speculated run may only mean n was wrong)
    if (F->side_flatten && c->aux_stream) RSDSFM_HIP_CHECK(c, hipStreamSynchronize(c->aux_stream));  // (joined long ago, unless the run failed in front of the join)
    int64_t n = F->n;
    if (*count_host(c) < 0) {  // the flatten behind the minimal solver was never enqueued: the run failed in front of it
        c->frame_dense_hint = 1;
        return rc != RSDSFM_OK ? rc : fail(c, RSDSFM_ERR_NUMERIC, "frame solve: the point count did not arrive");
    }
    bool counted = true;  // the run above is the one that counts (its scheduling hints are kept)
    if (*count_host(c) != n) {  // pixels without flow were dropped: everything behind the flatten ran on the wrong point count
        if (F->dense_in_launch) {
            // ... and the dense flatten wrote every kept pixel to the position it would have in a frame without holes: the general
            // flatten (count, scan, scatter) has to run after all
            int64_t n_real = 0;
            rc = flatten_device(c, J.d_flow_img, J.rows, J.cols, 0, J.fx, J.fy, J.cx, J.cy, J.gamma, prm->flow_threshold, F->d_q, F->d_u, F->d_a, F->d_ak, &n_real);
            if (rc == RSDSFM_OK && prm->use_global_shutter_mode) rc = alpha_ones_launch(c, F->d_a, n_real);
            if (rc != RSDSFM_OK) return rc;
            *count_host(c) = n_real;
        }
        n = *count_host(c);
        F->n = n;
        F->refinement_enqueued = false;
        counted = false;
        rc = ransac_begin(c, F->d_q, F->d_u, F->d_a, F->d_ak, n, prm->use_acceleration_mode, prm->ransac_trials, prm->ransac_tol, nullptr, J.seed,
                          prm->depth_mode, prm->k_sign_mode, &F->ro, prm->use_refinement ? &F->spec_tail : nullptr, &F->refinement_enqueued, &F->ransac, nullptr, nullptr,
                          nullptr, F->ahead, true);
        if (rc == RSDSFM_OK) rc = ransac_finish(c, &F->ransac);
        counted = rc == RSDSFM_OK;
    }
    c->frame_dense_hint = n == (int64_t)F->N ? 1 : 0;
    if (rc != RSDSFM_OK) return rc;
    if (counted) ransac_commit_hints(c, F->ransac);
    const rsdsfm_ransac_out& ro = F->ro;
    res->n_points = n;
    res->num_inliers = ro.num_inliers;
    res->best_trial = ro.best_trial;
    memcpy(res->ransac_w, ro.w, sizeof(ro.w));
    memcpy(res->ransac_v, ro.v, sizeof(ro.v));
    res->ransac_k = ro.k;
    double v[3], w[3], k = ro.k;
    for (int i = 0; i < 3; ++i) v[i] = ro.v[i], w[i] = ro.w[i];
    double* d_final = F->d_inl;
    int flipped = 0;
    F->m_known = ro.num_inliers;
    if (prm->use_refinement) {
        bool exact = false;
        if (F->refinement_enqueued) {
            rc = refine_poll(c, &F->refine, v, w, &k, &res->refine_summary);
            exact = rc == kRcRefineRestartExact;  // (a guard of the radius-factorised path: again from the host-side RANSAC result, iterate by iterate)
        }
        if (!F->refinement_enqueued || exact)
            rc = refine_device(c, F->d_u, n, ro.num_inliers, F->d_inl, F->d_in_a, F->d_in_ak, F->d_idx, v, w, k, prm->use_acceleration_mode,
                               prm->flow_index_mode, F->d_inl_ref, v, w, &k, &res->refine_summary, &F->tail, F->d_zpartials, exact);
        if (rc != RSDSFM_OK) return rc;
        d_final = F->d_inl_ref;
        const double* h_header = header_host(c);
        flipped = h_header[0] != 0.0;
        v[0] = h_header[1], v[1] = h_header[2], v[2] = h_header[3];
    } else {
        // depth map and, behind it on the stream, the pose table of the (possibly sign-flipped) final motion: one synchronisation
        rc = depth_map_device(c, d_final, ro.num_inliers, v, J.fx, J.fy, J.cx, J.cy, J.rows, J.cols, J.d_depth_map_colmajor, nullptr, F->d_ys, &flipped, w,
                              k, J.gamma, J.d_R_rows9_or_null, J.d_t_rows3_or_null);
        if (rc != RSDSFM_OK) return rc;
    }
    res->flipped = flipped;
    memcpy(res->v, v, sizeof(v));
    memcpy(res->w, w, sizeof(w));
    res->k = k;
    res->d_inliers = d_final;
    res->d_inlier_idx = F->d_idx;
    res->d_scanline = F->d_ys;
    return RSDSFM_OK;
}

FrameRun* frame_run_of(Ctx* c) {
    if (!c->frame_run) c->frame_run = new (std::nothrow) FrameRun();
    return static_cast<FrameRun*>(c->frame_run);
}

}  // namespace

void frame_release(Ctx* c) {
    frame_uncount(c, static_cast<FrameRun*>(c->frame_run));  // (a context destroyed with a run open)
    delete static_cast<FrameRun*>(c->frame_run);
    c->frame_run = nullptr;
    for (rsdsfm_ctx* lane : c->lanes) rsdsfm_destroy(lane);
    c->lanes.clear();
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    if (c->ev_seq) (void)hipEventDestroy(c->ev_seq);
    if (c->aux_stream) (void)hipStreamDestroy(c->aux_stream);
    if (c->d_flat_counters) (void)hipFree(c->d_flat_counters);
    c->d_flat_counters = nullptr;
    c->ev_fork = c->ev_join = c->ev_seq = nullptr;
    c->aux_stream = nullptr;
}

}  // namespace rsdsfm

using namespace rsdsfm;

extern "C" {

void rsdsfm_frame_params_init(rsdsfm_frame_params* p) {
    if (!p) return;
    memset(p, 0, sizeof(*p));
    p->ransac_trials = 5;            // main.cc:304
    p->use_acceleration_mode = 0;    // main.cc:306
    p->use_refinement = 1;           // main.cc:307
    p->depth_mode = RSDSFM_DEPTH_CERES_LM;
    p->k_sign_mode = RSDSFM_K_COMPAT;
    p->flow_index_mode = RSDSFM_FLOW_COMPAT_RANK;  // main.cc:457
    p->use_global_shutter_mode = 0;  // main.cc:305
    p->struct_bytes = (int32_t)sizeof(rsdsfm_frame_params);
    p->ransac_tol = 0.05;            // main.cc:310
    p->flow_threshold = 1e-10;       // main.cc:311
    p->seed = 1;
}

int rsdsfm_solve_frame_dev(rsdsfm_ctx* ctx, const double* d_flow_img, int32_t rows, int32_t cols, double fx, double fy, double cx,
                           double cy, double gamma, const rsdsfm_frame_params* prm, double* d_depth_map, double* d_R_rows9,
                           double* d_t_rows3, rsdsfm_frame_result* res) {
    if (!ctx) return RSDSFM_ERR_INVALID;
    Ctx* c = &ctx->c;
    DeviceGuard device_guard_(c);
    if (!prm || !res) return fail(c, RSDSFM_ERR_INVALID, "bad arguments");
    FrameRun* F = frame_run_of(c);
    if (!F) return fail(c, RSDSFM_ERR_INVALID, "out of host memory");
    F->prm = *prm;
    F->job = rsdsfm_frame_job{d_flow_img, rows, cols, fx, fy, cx, cy, gamma, d_depth_map, d_R_rows9, d_t_rows3, prm->seed};
    int rc = frame_begin(c, F);
    if (rc != RSDSFM_OK) return rc;
    return frame_finish(c, F, res);
}

int rsdsfm_set_sequence_lanes(rsdsfm_ctx* ctx, int32_t lanes) {
    if (!ctx) return RSDSFM_ERR_INVALID;
    Ctx* c = &ctx->c;
    DeviceGuard device_guard_(c);
    if (lanes < 0 || lanes > 16) return fail(c, RSDSFM_ERR_INVALID, "lanes must be 0 (default) .. 16");
    c->seq_lanes = lanes;
    return RSDSFM_OK;
}

int rsdsfm_set_frame_side_flatten(rsdsfm_ctx* ctx, int mode) {
    if (!ctx) return RSDSFM_ERR_INVALID;
    if (mode < 0 || mode > 3) return fail(&ctx->c, RSDSFM_ERR_INVALID, "mode must be 0 (flatten first), 1 (flatten on a second stream), 2 (flatten behind the minimal solver) or 3 (flatten inside the solver's launch)");
    ctx->c.frame_side_flatten = mode;
    return RSDSFM_OK;
}

// A sequence of frame pairs through ONE context, one host thread (BASELINE configs[4], "sequence throughput mode").  Pair i runs on
// lane i % L: lane 0 is the context itself, the others are contexts of their own (stream, workspace, scheduling hints) owned by it.
// Up to L pairs are in flight: the host begins pair i (frame_begin: enqueues its whole speculated chain, no wait) and only then
// finishes pair i - L + 1 (frame_finish: waits for it).  Every pair's results are those of rsdsfm_solve_frame_dev (which lane, which
// neighbours and which hints a pair meets decides when its kernels run, never what they compute).
int rsdsfm_solve_frames_dev(rsdsfm_ctx* ctx, const rsdsfm_frame_job* jobs, int32_t count, const rsdsfm_frame_params* prm, rsdsfm_frame_result* results) {
    if (!ctx) return RSDSFM_ERR_INVALID;
    Ctx* c = &ctx->c;
    DeviceGuard device_guard_(c);
    if (count < 0 || !prm || (count > 0 && (!jobs || !results))) return fail(c, RSDSFM_ERR_INVALID, "bad arguments");
    if (count == 0) return RSDSFM_OK;
    const int L = std::max(1, std::min<int>(count, c->seq_lanes > 0 ? c->seq_lanes : kSequenceLanesDefault));
    while ((int)c->lanes.size() < L - 1) {
        rsdsfm_ctx* lane = nullptr;
        int rc = rsdsfm_create(&lane, c->device, nullptr);
        if (rc != RSDSFM_OK) return fail(c, rc, "could not create a sequence lane");
        c->lanes.push_back(lane);
    }
    auto lane_ctx = [&](int l) -> Ctx* { return l == 0 ? c : &c->lanes[(size_t)l - 1]->c; };
    // what the caller enqueued on the context's stream (e.g. the upload of the flow images) comes before every lane's work
    if (L > 1) {
        if (!c->ev_seq) RSDSFM_HIP_CHECK(c, hipEventCreateWithFlags(&c->ev_seq, hipEventDisableTiming));
        RSDSFM_HIP_CHECK(c, hipEventRecord(c->ev_seq, c->stream));
    }
    for (int l = 1; l < L; ++l) {
        Ctx* lc = lane_ctx(l);
        RSDSFM_HIP_CHECK(c, hipStreamWaitEvent(lc->stream, c->ev_seq, 0));
        lc->ransac_k0 = c->ransac_k0;
        lc->ransac_math_mode = c->ransac_math_mode;
        lc->lm_arithmetic = c->lm_arithmetic;
        lc->refine_arithmetic = c->refine_arithmetic;
        lc->lma_count_only_force = c->lma_count_only_force;
        lc->frame_side_flatten = c->frame_side_flatten;
        lc->refine_stage_mode = c->refine_stage_mode;
    }
    // several pairs in flight share the GPU: the refinement's single-workgroup stage gets launches of its own (Ctx::refine_stage_mode)
    for (int l = 0; l < L; ++l) lane_ctx(l)->refine_stage_separate = L > 1;
    int first_error = RSDSFM_OK;
    int in_flight[16];  // pair index each lane is working on, -1 = none
    for (int l = 0; l < 16; ++l) in_flight[l] = -1;
    auto finish_lane = [&](int l) {
        const int i = in_flight[l];
        if (i < 0) return;
        in_flight[l] = -1;
        Ctx* lc = lane_ctx(l);
        int rc = frame_finish(lc, static_cast<FrameRun*>(lc->frame_run), &results[i]);
        if (rc != RSDSFM_OK && first_error == RSDSFM_OK) {
            first_error = rc;
            if (lc != c) c->err = "pair " + std::to_string(i) + ": " + lc->err;
        }
    };
    for (int i = 0; i < count && first_error == RSDSFM_OK; ++i) {
        const int l = i % L;
        finish_lane(l);  // pair i - L, if any
        if (first_error != RSDSFM_OK) break;
        Ctx* lc = lane_ctx(l);
        FrameRun* F = frame_run_of(lc);
        if (!F) {
            first_error = fail(c, RSDSFM_ERR_INVALID, "out of host memory");
            break;
        }
        F->prm = *prm;
        F->job = jobs[i];
        int rc = frame_begin(lc, F);
        if (rc != RSDSFM_OK) {
            first_error = rc;
            if (lc != c) c->err = "pair " + std::to_string(i) + ": " + lc->err;
            break;
        }
        in_flight[l] = i;
    }
    // drain in pair order
    for (;;) {
        int best = -1;
        for (int l = 0; l < L; ++l)
            if (in_flight[l] >= 0 && (best < 0 || in_flight[l] < in_flight[best])) best = l;
        if (best < 0) break;
        finish_lane(best);
    }
    for (int l = 0; l < L; ++l) lane_ctx(l)->refine_stage_separate = false;  // (single solves on this context: the stage back in the prologue)
    return first_error;
}

int rsdsfm_set_refine_stage(rsdsfm_ctx* ctx, int mode) {
    if (!ctx) return RSDSFM_ERR_INVALID;
    Ctx* c = &ctx->c;
    if (mode < 0 || mode > 2) return fail(c, RSDSFM_ERR_INVALID, "refine stage: 0 (automatic), 1 (in the next pass's prologue) or 2 (a launch of its own)");
    c->refine_stage_mode = mode;
    return RSDSFM_OK;
}

}  // extern "C"
