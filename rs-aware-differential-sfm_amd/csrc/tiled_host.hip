// tiled_host.hip -- stage-level C ABI of the row-tiled whole-frame solve.
//
// A frame is split into column slabs, one per rank (the reference's flatten order is column-major, main.cc:398-444,
// so the concatenation of the slabs' point lists in rank order IS the reference's point list).  Every stage below
// works on the caller's shard and ends in a small fixed-size "row" of sums; the caller all-gathers the rows in rank
// order (RCCL) and hands the gathered [ranks][...] array to the matching decide stage, which every rank runs
// identically -- the same kernels that reduce per-workgroup partials in the single-GPU solve reduce per-rank rows
// here, so all ranks take identical decisions and no rank ever owns "the" state.  The only other exchanges are the
// 9 T sampled points (before minimal9) and one all-gather of the depth-map slabs.
//
//   minimal.cc:209-306  ransac              -> rsdsfm_minimal9_dev, rsdsfm_tile_ransac_*
//   nonlinearRefinement.cc:183-252          -> rsdsfm_tile_refine_*
//   main.cc:466-509     sign fix, depth map -> rsdsfm_tile_zsum_dev, rsdsfm_tile_depth_map_dev
#include <string.h>

#include <algorithm>
#include <new>

#include "rsdsfm_internal.hpp"

using namespace rsdsfm;

namespace {

int ensure_tile(Ctx* c, size_t bytes) {
    if (bytes <= c->tile_bytes) return RSDSFM_OK;
    RSDSFM_HIP_CHECK(c, hipStreamSynchronize(c->stream));
    if (c->d_tile) (void)hipFree(c->d_tile);
    c->d_tile = nullptr;
    c->tile_bytes = 0;
    RSDSFM_HIP_CHECK(c, hipMalloc(&c->d_tile, bytes));
    c->tile_bytes = bytes;
    return RSDSFM_OK;
}

inline bool bad_depth_mode(int m) { return m != RSDSFM_DEPTH_CLOSED_FORM && m != RSDSFM_DEPTH_CERES_LM; }

}  // namespace

extern "C" {

int rsdsfm_sample_indices(int64_t n, int32_t iterations, uint64_t seed, int32_t* samples) {
    if (n < 9 || n > (int64_t)INT32_MAX || iterations < 0 || (iterations > 0 && !samples)) return RSDSFM_ERR_INVALID;
    sample_indices(n, iterations, seed, samples);
    return RSDSFM_OK;
}

int rsdsfm_minimal9_dev(rsdsfm_ctx* ctx, const double* d_q9, const double* d_u9, const double* d_alpha9, const double* d_alpha_k9,
                        int32_t count, int use_alpha_k, int k_sign_mode, double* d_hyp) {
    if (!ctx) return RSDSFM_ERR_INVALID;
    Ctx* c = &ctx->c;
    DeviceGuard device_guard_(c);
    if (count < 0) return fail(c, RSDSFM_ERR_INVALID, "negative count");
    if (count == 0) return RSDSFM_OK;
    if (!d_q9 || !d_u9 || !d_alpha9 || !d_alpha_k9 || !d_hyp) return fail(c, RSDSFM_ERR_INVALID, "null device pointer");
    return minimal9_launch(c, d_q9, d_u9, d_alpha9, d_alpha_k9, nullptr, count, use_alpha_k, k_sign_mode, d_hyp);
}

int rsdsfm_minimal9_probe_dev(rsdsfm_ctx* ctx, const double* d_q9, const double* d_u9, const double* d_alpha9, const double* d_alpha_k9,
                              int32_t count, int use_alpha_k, int k_sign_mode, int use_cores, double* d_hyp, double* d_probe4) {
    if (!ctx) return RSDSFM_ERR_INVALID;
    Ctx* c = &ctx->c;
    DeviceGuard device_guard_(c);
    if (count < 1 || count > c->num_cus * 2) return fail(c, RSDSFM_ERR_INVALID, "probe: 1 <= count <= 2 x CUs (the wave-per-hypothesis solver)");
    if (!d_q9 || !d_u9 || !d_alpha9 || !d_alpha_k9 || !d_hyp || !d_probe4) return fail(c, RSDSFM_ERR_INVALID, "null device pointer");
    Minimal9Direct dir;
    dir.probe = d_probe4;
    if (use_cores) {
        if (!c->d_core_flag) {
            RSDSFM_HIP_CHECK(c, hipMalloc(reinterpret_cast<void**>(&c->d_core_flag), 64));
            RSDSFM_HIP_CHECK(c, hipMemsetAsync(c->d_core_flag, 0, 64, c->stream));
        }
        c->core_epoch = c->core_epoch >= 0x3fffffff ? 1 : c->core_epoch + 1;
        dir.core_flag = c->d_core_flag;
        dir.core_epoch = c->core_epoch;
    }
    return minimal9_launch(c, d_q9, d_u9, d_alpha9, d_alpha_k9, nullptr, count, use_alpha_k, k_sign_mode, d_hyp, nullptr, 0, &dir);
}

size_t rsdsfm_tile_lm_state_bytes(void) { return sizeof(LmState); }
size_t rsdsfm_tile_best_bytes(void) { return sizeof(RansacBest); }
int32_t rsdsfm_tile_ransac_row_size(void) { return ransac_rows_doubles(); }
int32_t rsdsfm_tile_ransac_batch(void) { return kRansacBatch; }

int rsdsfm_tile_ransac_lm_rows_dev(rsdsfm_ctx* ctx, const double* d_q, const double* d_u, const double* d_alpha,
                                   const double* d_alpha_k, int64_t n, const double* d_hyp, int32_t count, const void* d_states,
                                   int32_t round, double tolerance, double* d_rows) {
    if (!ctx) return RSDSFM_ERR_INVALID;
    Ctx* c = &ctx->c;
    DeviceGuard device_guard_(c);
    if (n < 0 || count < 1 || count > kRansacBatch || round < 0) return fail(c, RSDSFM_ERR_INVALID, "bad arguments (count must be 1..128 per call)");
    if (!d_hyp || !d_states || !d_rows || (n > 0 && (!d_q || !d_u || !d_alpha || !d_alpha_k))) return fail(c, RSDSFM_ERR_INVALID, "null device pointer");
    int rc = ensure_ws(c, Arena::need(sizeof(double) * (size_t)ransac_lm_partials_doubles(c, n, count)) + 1024);
    if (rc != RSDSFM_OK) return rc;
    return ransac_lm_rows_launch(c, d_q, d_u, d_alpha, d_alpha_k, n, d_hyp, count, static_cast<const LmState*>(d_states),
                                 static_cast<double*>(c->d_ws), round, tolerance, d_rows);
}

int rsdsfm_tile_ransac_decide_dev(rsdsfm_ctx* ctx, const double* d_rows_all, int32_t nranks, int32_t count, void* d_states,
                                  int64_t n_total, int32_t round, int32_t* d_flags, int32_t* d_scored, double* d_trial_count,
                                  double* d_trial_err) {
    if (!ctx) return RSDSFM_ERR_INVALID;
    Ctx* c = &ctx->c;
    DeviceGuard device_guard_(c);
    if (nranks < 1 || count < 1 || round < 0 || n_total < 0) return fail(c, RSDSFM_ERR_INVALID, "bad arguments");
    if (!d_rows_all || !d_states || !d_flags || !d_scored || !d_trial_count || !d_trial_err) return fail(c, RSDSFM_ERR_INVALID, "null device pointer");
    return ransac_decide_rows_launch(c, d_rows_all, nranks, count, static_cast<LmState*>(d_states), n_total, round, d_flags, d_scored,
                                     d_trial_count, d_trial_err);
}

int rsdsfm_tile_ransac_score_rows_dev(rsdsfm_ctx* ctx, const double* d_q, const double* d_u, const double* d_alpha,
                                      const double* d_alpha_k, int64_t n, const double* d_hyp, int32_t count, const void* d_states,
                                      int depth_mode, double tolerance, const int32_t* d_scored, double* d_rows) {
    if (!ctx) return RSDSFM_ERR_INVALID;
    Ctx* c = &ctx->c;
    DeviceGuard device_guard_(c);
    if (n < 0 || count < 1 || count > kRansacBatch || bad_depth_mode(depth_mode)) return fail(c, RSDSFM_ERR_INVALID, "bad arguments");
    if (!d_hyp || !d_states || !d_rows || (n > 0 && (!d_q || !d_u || !d_alpha || !d_alpha_k))) return fail(c, RSDSFM_ERR_INVALID, "null device pointer");
    int rc = ensure_ws(c, Arena::need(sizeof(double) * (size_t)ransac_lm_partials_doubles(c, n, count)) + 1024);
    if (rc != RSDSFM_OK) return rc;
    return ransac_score_rows_launch(c, d_q, d_u, d_alpha, d_alpha_k, n, d_hyp, count, static_cast<const LmState*>(d_states), depth_mode,
                                    tolerance, d_scored, static_cast<double*>(c->d_ws), d_rows);
}

int rsdsfm_tile_ransac_score_merge_dev(rsdsfm_ctx* ctx, const double* d_rows_all, int32_t nranks, int32_t count,
                                       const int32_t* d_scored, double* d_trial_count, double* d_trial_err) {
    if (!ctx) return RSDSFM_ERR_INVALID;
    Ctx* c = &ctx->c;
    DeviceGuard device_guard_(c);
    if (nranks < 1 || count < 1 || !d_rows_all || !d_trial_count || !d_trial_err) return fail(c, RSDSFM_ERR_INVALID, "bad arguments");
    return ransac_score_merge_launch(c, d_rows_all, nranks, count, d_scored, d_trial_count, d_trial_err);
}

int rsdsfm_tile_ransac_pick_dev(rsdsfm_ctx* ctx, const double* d_trial_count, const double* d_trial_err, int32_t iterations,
                                const double* d_hyp, void* d_best) {
    if (!ctx) return RSDSFM_ERR_INVALID;
    Ctx* c = &ctx->c;
    DeviceGuard device_guard_(c);
    if (iterations < 0 || !d_best || (iterations > 0 && (!d_trial_count || !d_trial_err || !d_hyp))) return fail(c, RSDSFM_ERR_INVALID, "bad arguments");
    return ransac_pick_launch(c, d_trial_count, d_trial_err, iterations, d_hyp, static_cast<RansacBest*>(d_best));
}

int rsdsfm_tile_ransac_final_dev(rsdsfm_ctx* ctx, const double* d_q, const double* d_u, const double* d_alpha, const double* d_alpha_k,
                                 int64_t n, void* d_best, const void* d_states, int depth_mode, double tolerance, double* d_inv_depth,
                                 uint8_t* d_mask, int64_t* d_inlier_idx, double* d_inliers, double* d_out_alpha,
                                 double* d_out_alpha_k, rsdsfm_ransac_out* out) {
    if (!ctx) return RSDSFM_ERR_INVALID;
    Ctx* c = &ctx->c;
    DeviceGuard device_guard_(c);
    if (n < 0 || !d_best || !d_states || !out || bad_depth_mode(depth_mode)) return fail(c, RSDSFM_ERR_INVALID, "bad arguments");
    if (n > 0 && (!d_q || !d_u || !d_alpha || !d_alpha_k || !d_inv_depth || !d_mask)) return fail(c, RSDSFM_ERR_INVALID, "null device pointer");
    int rc = ensure_ws(c, 2 * Arena::need(sizeof(int64_t) * 2048) + 1024);
    if (rc != RSDSFM_OK) return rc;
    rc = ensure_pinned(c, sizeof(RansacBest) + 64);
    if (rc != RSDSFM_OK) return rc;
    Arena ws(c->d_ws);
    int64_t* d_bcounts = ws.take<int64_t>(2048);
    int64_t* d_boffs = ws.take<int64_t>(2048);
    RansacBest* best = static_cast<RansacBest*>(d_best);
    rc = ransac_final_launch(c, d_q, d_u, d_alpha, d_alpha_k, n, best, static_cast<const LmState*>(d_states), depth_mode, tolerance,
                             d_inv_depth, d_mask, d_bcounts, d_boffs, d_inlier_idx, d_inliers, d_out_alpha, d_out_alpha_k);
    if (rc != RSDSFM_OK) return rc;
    RansacBest* h = static_cast<RansacBest*>(c->h_pinned);
    RSDSFM_HIP_CHECK(c, hipMemcpyAsync(h, best, sizeof(RansacBest), hipMemcpyDeviceToHost, c->stream));
    RSDSFM_HIP_CHECK(c, hipStreamSynchronize(c->stream));
    // global winner (identical on every rank) + this shard's inlier count
    out->best_trial = h->best_trial;
    out->inlier_error = h->inlier_error;
    memcpy(out->w, &h->hyp[0], 3 * sizeof(double));
    memcpy(out->v, &h->hyp[3], 3 * sizeof(double));
    out->k = h->hyp[6];
    out->num_inliers = h->num_inliers_scan;  // of THIS shard; the global count is best.num_inliers == sum over ranks
    return RSDSFM_OK;
}

int64_t rsdsfm_tile_ransac_global_inliers(rsdsfm_ctx* ctx, const void* d_best) {
    if (!ctx || !d_best) return -1;
    Ctx* c = &ctx->c;
    DeviceGuard device_guard_(c);
    if (ensure_pinned(c, sizeof(RansacBest) + 64) != RSDSFM_OK) return -1;
    RansacBest* h = static_cast<RansacBest*>(c->h_pinned);
    if (hipMemcpyAsync(h, d_best, sizeof(RansacBest), hipMemcpyDeviceToHost, c->stream) != hipSuccess) return -1;
    if (hipStreamSynchronize(c->stream) != hipSuccess) return -1;
    return h->num_inliers;
}

// ---------------------------------------------------------------------------------------------------
// refinement
// ---------------------------------------------------------------------------------------------------
int rsdsfm_tile_refine_begin_dev(rsdsfm_ctx* ctx, const double* d_flow, int64_t n_flow, int64_t m, const double* d_inl,
                                 const double* d_alpha, const double* d_alpha_k, const int64_t* d_inlier_idx, const double v_in[3],
                                 const double w_in[3], double k_in, int const_acceleration, int flow_index_mode) {
    if (!ctx) return RSDSFM_ERR_INVALID;
    Ctx* c = &ctx->c;
    DeviceGuard device_guard_(c);
    if (m < 0 || n_flow < 0 || !v_in || !w_in) return fail(c, RSDSFM_ERR_INVALID, "bad arguments");
    // RSDSFM_FLOW_GATHERED: d_flow = the shard's flow list, column d_inlier_idx[i] (shard-local) belongs to inlier i.
    // RSDSFM_FLOW_COMPAT_RANK (the reference's rank-indexed flow, quirk Q2): the CALLER has fetched the columns of the GLOBAL flow list
    // at this shard's global inlier ranks [prefix, prefix + m) -- they live on the shards in front of it -- and d_flow holds them:
    // column i belongs to inlier i, n_flow >= m (dist.TiledFrameSolve does that exchange; the native driver does it itself).
    if (flow_index_mode != RSDSFM_FLOW_GATHERED && flow_index_mode != RSDSFM_FLOW_COMPAT_RANK) return fail(c, RSDSFM_ERR_INVALID, "unknown flow_index_mode");
    if (flow_index_mode == RSDSFM_FLOW_COMPAT_RANK && n_flow < m) return fail(c, RSDSFM_ERR_INVALID, "rank-indexed flow: d_flow must hold one column per inlier of the shard");
    if (m > 0 && (!d_flow || !d_inl || !d_alpha || !d_alpha_k || (flow_index_mode == RSDSFM_FLOW_GATHERED && !d_inlier_idx))) return fail(c, RSDSFM_ERR_INVALID, "null device pointer");
    const int np = const_acceleration ? 7 : 6;
    const size_t M = (size_t)std::max<int64_t>(m, 1);
    const size_t npart = (size_t)refine_partials_doubles(c, m);
    int rc = ensure_tile(c, Arena::need(sizeof(RefineState)) + Arena::need(32 * M) + 4 * Arena::need(8 * M) + Arena::need(8 * npart) + Arena::need(64) + 1024);
    if (rc != RSDSFM_OK) return rc;
    rc = ensure_pinned(c, sizeof(RefineState) + 64);
    if (rc != RSDSFM_OK) return rc;
    if (!c->tile_session) c->tile_session = new (std::nothrow) RefineBuffers();
    if (!c->tile_session) return fail(c, RSDSFM_ERR_INVALID, "out of host memory");
    RefineBuffers& B = *static_cast<RefineBuffers*>(c->tile_session);
    Arena ws(c->d_tile);
    B.flow = d_flow;
    B.n_flow = n_flow;
    B.m = m;
    B.inl = d_inl;
    B.alpha = d_alpha;
    B.alpha_k = d_alpha_k;
    B.inlier_idx = d_inlier_idx;
    B.flow_index_mode = flow_index_mode;
    B.state = ws.take<RefineState>(1);
    B.uu = ws.take<double>(4 * M);
    B.beta = ws.take<double>(M);
    B.rho_a = ws.take<double>(M);
    B.rho_b = ws.take<double>(M);
    B.srho = ws.take<double>(M);
    B.partials = ws.take<double>(npart);
    B.bad_index = ws.take<int>(1);
    c->tile_np = np;
    RefineState* hs = static_cast<RefineState*>(c->h_pinned);
    memset(hs, 0, sizeof(RefineState));
    hs->np = np;
    for (int i = 0; i < 3; ++i) {
        hs->p[i] = v_in[i];
        hs->p[3 + i] = w_in[i];
    }
    hs->p[6] = k_in;
    hs->termination = -1;
    hs->radius = kInitialRadius;
    RSDSFM_HIP_CHECK(c, hipMemcpyAsync(B.state, hs, sizeof(RefineState), hipMemcpyHostToDevice, c->stream));
    RSDSFM_HIP_CHECK(c, hipMemsetAsync(B.bad_index, 0, sizeof(int), c->stream));
    rc = refine_trace_reset(c);
    if (rc != RSDSFM_OK) return rc;
    RSDSFM_HIP_CHECK(c, hipStreamSynchronize(c->stream));  // the pinned block is reused by the polls
    return RSDSFM_OK;
}

int32_t rsdsfm_tile_refine_row_size(int const_acceleration, int32_t stage) {
    if (stage < 0 || stage > 2) return -1;
    return refine_stage_row_doubles(const_acceleration ? 7 : 6, stage);
}

int rsdsfm_tile_refine_rows_dev(rsdsfm_ctx* ctx, int32_t stage, double* d_row) {
    if (!ctx) return RSDSFM_ERR_INVALID;
    Ctx* c = &ctx->c;
    DeviceGuard device_guard_(c);
    if (!c->tile_session || c->tile_np == 0) return fail(c, RSDSFM_ERR_INVALID, "no open refinement session");
    if (stage < 0 || stage > 2 || !d_row) return fail(c, RSDSFM_ERR_INVALID, "bad arguments");
    return refine_stage_rows_launch(c, *static_cast<RefineBuffers*>(c->tile_session), c->tile_np, stage, d_row);
}

int rsdsfm_tile_refine_apply_dev(rsdsfm_ctx* ctx, int32_t stage, const double* d_rows_all, int32_t nranks, int64_t m_total) {
    if (!ctx) return RSDSFM_ERR_INVALID;
    Ctx* c = &ctx->c;
    DeviceGuard device_guard_(c);
    if (!c->tile_session || c->tile_np == 0) return fail(c, RSDSFM_ERR_INVALID, "no open refinement session");
    if (stage < 0 || stage > 2 || !d_rows_all || nranks < 1 || m_total < 0) return fail(c, RSDSFM_ERR_INVALID, "bad arguments");
    return refine_stage_apply_launch(c, *static_cast<RefineBuffers*>(c->tile_session), c->tile_np, stage, d_rows_all, nranks, m_total);
}

int rsdsfm_tile_refine_poll(rsdsfm_ctx* ctx, double v_out[3], double w_out[3], double* k_out, rsdsfm_lm_summary* summary) {
    if (!ctx) return RSDSFM_ERR_INVALID;
    Ctx* c = &ctx->c;
    DeviceGuard device_guard_(c);
    if (!c->tile_session || c->tile_np == 0) return fail(c, RSDSFM_ERR_INVALID, "no open refinement session");
    RefineBuffers& B = *static_cast<RefineBuffers*>(c->tile_session);
    RefineState* hs = static_cast<RefineState*>(c->h_pinned);
    int* h_bad = reinterpret_cast<int*>(static_cast<char*>(c->h_pinned) + sizeof(RefineState));
    RSDSFM_HIP_CHECK(c, hipMemcpyAsync(hs, B.state, sizeof(RefineState), hipMemcpyDeviceToHost, c->stream));
    RSDSFM_HIP_CHECK(c, hipMemcpyAsync(h_bad, B.bad_index, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    RSDSFM_HIP_CHECK(c, hipStreamSynchronize(c->stream));
    if (*h_bad) return fail(c, RSDSFM_ERR_INVALID, "flow index out of range (bad inlier_idx)");
    if (v_out && w_out)
        for (int i = 0; i < 3; ++i) {
            v_out[i] = hs->p[i];
            w_out[i] = hs->p[3 + i];
        }
    if (k_out) *k_out = hs->p[6];
    if (summary) {
        summary->num_iterations = hs->iteration;
        summary->num_successful_steps = hs->num_successful;
        summary->num_unsuccessful_steps = hs->num_unsuccessful;
        summary->termination = hs->termination;  // -1 while the solve is still running
        summary->initial_cost = hs->initial_cost;
        summary->final_cost = hs->cost;
        summary->final_radius = hs->radius;
    }
    return RSDSFM_OK;
}

int rsdsfm_tile_refine_finish_dev(rsdsfm_ctx* ctx, double* d_inl_out) {
    if (!ctx) return RSDSFM_ERR_INVALID;
    Ctx* c = &ctx->c;
    DeviceGuard device_guard_(c);
    if (!c->tile_session || c->tile_np == 0) return fail(c, RSDSFM_ERR_INVALID, "no open refinement session");
    RefineBuffers& B = *static_cast<RefineBuffers*>(c->tile_session);
    if (B.m > 0 && !d_inl_out) return fail(c, RSDSFM_ERR_INVALID, "null device pointer");
    int rc = refine_finish_launch(c, B, d_inl_out);
    c->tile_np = 0;
    return rc;
}

// ---------------------------------------------------------------------------------------------------
// sign fix + depth map
// ---------------------------------------------------------------------------------------------------
int rsdsfm_tile_zsum_dev(rsdsfm_ctx* ctx, const double* d_inl, int64_t m, double* d_zsum) {
    if (!ctx) return RSDSFM_ERR_INVALID;
    Ctx* c = &ctx->c;
    DeviceGuard device_guard_(c);
    if (m < 0 || !d_zsum || (m > 0 && !d_inl)) return fail(c, RSDSFM_ERR_INVALID, "bad arguments");
    int rc = ensure_ws(c, Arena::need(8 * 1024) + 1024);
    if (rc != RSDSFM_OK) return rc;
    return zsum_row_launch(c, d_inl, m, static_cast<double*>(c->d_ws), d_zsum);
}

int rsdsfm_tile_depth_map_dev(rsdsfm_ctx* ctx, double* d_inl, int64_t m, const double* d_zsums_all, int32_t nranks, int64_t m_total,
                              double v_inout[3], double fx, double fy, double cx, double cy, int32_t rows, int32_t col0,
                              int32_t slab_cols, double* d_depth_slab, int32_t* d_xs, int32_t* d_ys, int* flipped) {
    if (!ctx) return RSDSFM_ERR_INVALID;
    Ctx* c = &ctx->c;
    DeviceGuard device_guard_(c);
    if (m < 0 || rows < 0 || col0 < 0 || slab_cols < 0 || nranks < 1 || m_total < m || !v_inout || !d_zsums_all) return fail(c, RSDSFM_ERR_INVALID, "bad arguments");
    const size_t npix = (size_t)rows * (size_t)slab_cols;
    if ((m > 0 && !d_inl) || (npix > 0 && !d_depth_slab)) return fail(c, RSDSFM_ERR_INVALID, "null device pointer");
    int rc = ensure_ws(c, Arena::need(64) + 1024);
    if (rc != RSDSFM_OK) return rc;
    rc = ensure_pinned(c, 64);
    if (rc != RSDSFM_OK) return rc;
    Arena ws(c->d_ws);
    double* d_header = ws.take<double>(4);
    rc = depth_map_slab_launch(c, d_inl, m, d_zsums_all, nranks, m_total, v_inout, fx, fy, cx, cy, rows, col0, slab_cols, d_depth_slab,
                               d_xs, d_ys, d_header);
    if (rc != RSDSFM_OK) return rc;
    double* h_header = static_cast<double*>(c->h_pinned);
    RSDSFM_HIP_CHECK(c, hipMemcpyAsync(h_header, d_header, 4 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    RSDSFM_HIP_CHECK(c, hipStreamSynchronize(c->stream));
    if (flipped) *flipped = h_header[0] != 0.0;
    v_inout[0] = h_header[1];
    v_inout[1] = h_header[2];
    v_inout[2] = h_header[3];
    return RSDSFM_OK;
}

}  // extern "C"
