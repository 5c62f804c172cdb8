// frame_host.hip -- rsdsfm_solve_frame_dev: the solver part of the reference's evaluateSingleRun() (main.cc:398-522) as
// ONE call on device-resident buffers: flatten + alpha -> RANSAC -> nonlinear refinement -> sign flip + depth map ->
// per-scanline pose table.  Pure orchestration of the stage entry points (no extra kernels).
#include <stdlib.h>
#include <string.h>

#include "rsdsfm_internal.hpp"

namespace rsdsfm {
int ransac_device(Ctx* c, const double* d_q, const double* d_u, const double* d_a, const double* d_ak, int64_t n, int use_alpha_k, int T,
                  double tol, const int32_t* h_samples, uint64_t seed, int depth_mode, int k_sign_mode, rsdsfm_ransac_out* out,
                  const RansacSpecTail* spec_tail, bool* spec_tail_held);
int refine_device(Ctx* c, const double* d_flow, int64_t n_flow, int64_t m, const double* d_inl, const double* d_alpha,
                  const double* d_alpha_k, const int64_t* d_inlier_idx, const double v_in[3], const double w_in[3], double k_in,
                  int const_acceleration, int flow_index_mode, double* d_inl_out, double v_out[3], double w_out[3], double* k_out,
                  rsdsfm_lm_summary* summary, const RefineTail* tail, double* d_zpartials);
int alpha_ones_launch(Ctx* c, double* d_alpha, int64_t n);
}  // namespace rsdsfm

using namespace rsdsfm;

extern "C" {

int rsdsfm_solve_frame_dev(rsdsfm_ctx* ctx, const double* d_flow_img, int32_t rows, int32_t cols, double fx, double fy, double cx,
                           double cy, double gamma, const rsdsfm_frame_params* prm, double* d_depth_map, double* d_R_rows9,
                           double* d_t_rows3, rsdsfm_frame_result* res) {
    if (!ctx) return RSDSFM_ERR_INVALID;
    Ctx* c = &ctx->c;
    DeviceGuard device_guard_(c);
    if (!prm || !res || rows <= 0 || cols <= 0 || !d_flow_img || !d_depth_map) return fail(c, RSDSFM_ERR_INVALID, "bad arguments");
    if (prm->flow_index_mode != RSDSFM_FLOW_COMPAT_RANK && prm->flow_index_mode != RSDSFM_FLOW_GATHERED) return fail(c, RSDSFM_ERR_INVALID, "unknown flow_index_mode");
    const size_t N = (size_t)rows * (size_t)cols;
    // frame buffers live in the context's frame arena (separate from the per-stage workspace)
    const size_t need = 2 * Arena::need(16 * N) + 4 * Arena::need(8 * N) + 2 * Arena::need(24 * N) + Arena::need(8 * N) + Arena::need(N) +
                        Arena::need(4 * N) + Arena::need(8 * 1024) + Arena::need(64) + Arena::need(refine_workspace_bytes(c, (int64_t)N, true)) + 4096;
    if (need > c->frame_bytes) {
        RSDSFM_HIP_CHECK(c, hipStreamSynchronize(c->stream));
        if (c->d_frame) RSDSFM_HIP_CHECK(c, hipFree(c->d_frame));
        c->d_frame = nullptr;
        c->frame_bytes = 0;
        RSDSFM_HIP_CHECK(c, hipMalloc(&c->d_frame, need));
        c->frame_bytes = need;
    }
    Arena fa(c->d_frame);
    double* d_q = fa.take<double>(2 * N);
    double* d_u = fa.take<double>(2 * N);
    double* d_a = fa.take<double>(N);
    double* d_ak = fa.take<double>(N);
    double* d_in_a = fa.take<double>(N);
    double* d_in_ak = fa.take<double>(N);
    double* d_inl = fa.take<double>(3 * N);
    double* d_inl_ref = fa.take<double>(3 * N);
    int64_t* d_idx = fa.take<int64_t>(N);
    uint8_t* d_mask = fa.take<uint8_t>(N);
    int32_t* d_ys = fa.take<int32_t>(N);
    double* d_zpartials = fa.take<double>(1024);  // scratch of the depth-map stage when it is enqueued behind the refinement
    double* d_zheader = fa.take<double>(8);
    // buffers of a refinement that is enqueued while the RANSAC still owns the stage workspace (see below)
    char* d_refine_ws = fa.take<char>(refine_workspace_bytes(c, (int64_t)N, true));

    memset(res, 0, sizeof(*res));
    // The flatten is enqueued WITHOUT waiting for its point count: a dense optical flow gives every pixel a flow vector, so the RANSAC
    // (sampler, grids) is set up for n = rows * cols right away and the stream runs on from the flatten into the minimal solver; the
    // real count arrives in host-mapped memory and is checked at the RANSAC's own wait.  If pixels were dropped (zero flow below the
    // threshold) the stages are simply run again with the real count -- the slower path every frame took before.
    int rc = ensure_pinned(c, ransac_pinned_bytes(prm->ransac_trials));  // (sized up front: pointers into the block stay valid)
    if (rc != RSDSFM_OK) return rc;
    // (Only behind a frame that WAS dense: a sequence of frames with pixels without flow -- e.g. the ground-truth flow of a synthetic
    // example with void pixels -- waits for the count as before instead of paying for a discarded RANSAC every time.  A hint like the
    // others: it decides what is enqueued when, never a result.)
    int64_t* h_n = reinterpret_cast<int64_t*>(static_cast<char*>(c->h_pinned) + c->pinned_bytes - kPinnedTail / 2);
    int64_t n = (int64_t)N;
    if (c->frame_dense_hint) {
        *h_n = -1;
        rc = flatten_enqueue(c, d_flow_img, rows, cols, fx, fy, cx, cy, gamma, prm->flow_threshold, d_q, d_u, d_a, d_ak, h_n);
        if (rc != RSDSFM_OK) return rc;
    } else {
        rc = rsdsfm_flatten_dev(ctx, d_flow_img, rows, cols, fx, fy, cx, cy, gamma, prm->flow_threshold, d_q, d_u, d_a, d_ak, &n);
        if (rc != RSDSFM_OK) return rc;
        h_n = reinterpret_cast<int64_t*>(static_cast<char*>(c->h_pinned) + c->pinned_bytes - kPinnedTail / 2);
        *h_n = n;
    }
    if (prm->use_global_shutter_mode) {  // main.cc:441-444: alpha *= 0; alpha += 1
        rc = alpha_ones_launch(c, d_a, n);
        if (rc != RSDSFM_OK) return rc;
    }
    rsdsfm_ransac_out ro;
    memset(&ro, 0, sizeof(ro));
    ro.inlier_idx = d_idx;
    ro.inliers = d_inl;
    ro.alpha = d_in_a;
    ro.alpha_k = d_in_ak;
    ro.mask = d_mask;
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
    double v[3] = {0, 0, 0}, w[3] = {0, 0, 0}, k = 0;
    double* d_final = d_inl;
    int flipped = 0;
    int64_t m_known = -1;  // the inlier count once the host has it; until then the kernels read it from the refinement's state
    // (pointers into the pinned block are taken when they are used: the RANSAC may still grow the block)
    auto header_host = [&]() { return reinterpret_cast<double*>(static_cast<char*>(c->h_pinned) + c->pinned_bytes - 64); };
    auto state_host = [&]() { return reinterpret_cast<RefineState*>(static_cast<char*>(c->h_pinned) + c->pinned_bytes - kPinnedTail); };
    static_assert(sizeof(RefineState) + sizeof(int) <= kPinnedTail / 2, "reserved tail: state read-back | point count ... depth-map header");
    const RefineTail tail = [&](const RefineBuffers& B) -> int {
        const RefineState* st = B.state;
        const int64_t m_arg = m_known >= 0 ? m_known : n;
        const int64_t* m_dev = m_known >= 0 ? nullptr : &st->m;
        PoseTableOut pt;  // the pose table of (v', w, k) is written by the kernel that decides the sign of v
        if (d_R_rows9 && d_t_rows3) pt.R = d_R_rows9, pt.t = d_t_rows3, pt.rows = rows, pt.gamma = gamma, pt.wk_dev = st->p + 3;
        // (the sums of z come from the refinement's output pass: one entry per workgroup of that launch)
        return depth_map_slab_launch(c, d_inl_ref, m_arg, d_zpartials, refine_finish_grid(c, B), m_arg, nullptr, fx, fy, cx, cy, rows, 0, cols,
                                     d_depth_map, nullptr, d_ys, d_zheader, header_host(), st->p, m_dev, &pt);
    };
    RefineRun run;
    const RansacSpecTail spec_tail = [&](const RansacBest* d_best) -> int {
        return refine_begin(c, d_u, n, n, d_inl, d_in_a, d_in_ak, d_idx, nullptr, nullptr, 0.0, prm->use_acceleration_mode, prm->flow_index_mode,
                            d_inl_ref, &tail, d_best, d_refine_ws, &run, state_host(), d_zpartials);
    };
    bool refinement_enqueued = false;
    // (the refinement is only enqueued ahead where the previous RANSAC's speculated final stage held: data whose hypotheses need further
    // LM rounds every time -- noise-free flow -- would pay for a discarded refinement chunk per frame)
    const bool ahead = prm->use_refinement && c->ransac_spec_held_hint != 0;
    rc = ransac_device(c, d_q, d_u, d_a, d_ak, n, prm->use_acceleration_mode, prm->ransac_trials, prm->ransac_tol, nullptr, prm->seed,
                       prm->depth_mode, prm->k_sign_mode, &ro, ahead ? &spec_tail : nullptr, &refinement_enqueued);
    if (rc != RSDSFM_OK) RSDSFM_HIP_CHECK(c, hipStreamSynchronize(c->stream));  // (an error of the speculated run may only mean n was wrong)
    if (*h_n != n) {  // pixels without flow were dropped: everything behind the flatten ran on the wrong point count
        n = *h_n;
        refinement_enqueued = false;
        rc = ransac_device(c, d_q, d_u, d_a, d_ak, n, prm->use_acceleration_mode, prm->ransac_trials, prm->ransac_tol, nullptr, prm->seed,
                           prm->depth_mode, prm->k_sign_mode, &ro, ahead ? &spec_tail : nullptr, &refinement_enqueued);
    }
    c->frame_dense_hint = n == (int64_t)N ? 1 : 0;
    if (rc != RSDSFM_OK) return rc;
    res->n_points = n;
    res->num_inliers = ro.num_inliers;
    res->best_trial = ro.best_trial;
    memcpy(res->ransac_w, ro.w, sizeof(ro.w));
    memcpy(res->ransac_v, ro.v, sizeof(ro.v));
    res->ransac_k = ro.k;
    for (int i = 0; i < 3; ++i) v[i] = ro.v[i], w[i] = ro.w[i];
    k = ro.k;
    m_known = ro.num_inliers;
    if (prm->use_refinement) {
        if (refinement_enqueued)
            rc = refine_poll(c, &run, v, w, &k, &res->refine_summary);
        else
            rc = refine_device(c, d_u, n, ro.num_inliers, d_inl, d_in_a, d_in_ak, d_idx, v, w, k, prm->use_acceleration_mode, prm->flow_index_mode,
                               d_inl_ref, v, w, &k, &res->refine_summary, &tail, d_zpartials);
        if (rc != RSDSFM_OK) return rc;
        d_final = d_inl_ref;
        const double* h_header = header_host();
        flipped = h_header[0] != 0.0;
        v[0] = h_header[1], v[1] = h_header[2], v[2] = h_header[3];
    } else {
        // depth map and, behind it on the stream, the pose table of the (possibly sign-flipped) final motion: one synchronisation
        rc = depth_map_device(c, d_final, ro.num_inliers, v, fx, fy, cx, cy, rows, cols, d_depth_map, nullptr, d_ys, &flipped, w, k, gamma,
                              d_R_rows9, d_t_rows3);
        if (rc != RSDSFM_OK) return rc;
    }
    res->flipped = flipped;
    memcpy(res->v, v, sizeof(v));
    memcpy(res->w, w, sizeof(w));
    res->k = k;
    res->d_inliers = d_final;
    res->d_inlier_idx = d_idx;
    res->d_scanline = d_ys;
    return RSDSFM_OK;
}

}  // extern "C"
