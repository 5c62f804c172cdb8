// refine_host.hip -- host orchestration + C ABI of nonlinear_refinement::nonLinearRefinement
// (reference nonlinearRefinement.cc:183-252) over the kernels of refine_kernels.hip.
#include <string.h>

#include <algorithm>

#include "rsdsfm_internal.hpp"

namespace rsdsfm {

// all pointers DEVICE (d_inlier_idx may be null in compat mode); v/w/k and summary on the host
// tail (optional): work of the CALLER that only needs the refinement's device-resident result -- the refined inliers in d_inl_out and
// (v, w, k) in RefineState::p -- and is enqueued behind the output pass, BEFORE the host waits for the state: the frame solve hands
// in its depth-map + pose-table stage, so that the one synchronisation at the end of the refinement covers that stage as well
// (one host round trip with an idle GPU less per frame).  If LM iterations remain after a chunk the tail simply runs again behind
// the next output pass; what it computed on the unfinished state is overwritten.
//
// The refinement runs in two phases.  refine_begin lays out the buffers and enqueues the start state, iteration zero, the first chunk of
// LM iterations, the output pass and the tail; refine_poll waits for the state and enqueues further chunks until the solve has
// terminated.  refine_device = both.  The frame solve calls refine_begin BEFORE it has read the RANSAC result (d_best != null: the
// start pose and the inlier count come from the device-resident RansacBest, `m` is only an upper bound, the buffers live in
// `ws_base` instead of the stage workspace the RANSAC still owns) and refine_poll once the RANSAC's own synchronisation has passed.
int refine_begin(Ctx* c, const double* d_flow, int64_t n_flow, int64_t m, const double* d_inl, const double* d_alpha,
                 const double* d_alpha_k, const int64_t* d_inlier_idx, const double v_in[3], const double w_in[3], double k_in,
                 int const_acceleration, int flow_index_mode, double* d_inl_out, const RefineTail* tail, const RansacBest* d_best,
                 void* ws_base, RefineRun* run, RefineState* hs_prefetch, double* d_zpartials, bool exact) {
    if (m < 0 || n_flow < 0 || (!d_best && (!v_in || !w_in))) return fail(c, RSDSFM_ERR_INVALID, "bad arguments");
    if (flow_index_mode != RSDSFM_FLOW_COMPAT_RANK && flow_index_mode != RSDSFM_FLOW_GATHERED) return fail(c, RSDSFM_ERR_INVALID, "unknown flow_index_mode");
    if (flow_index_mode == RSDSFM_FLOW_GATHERED && m > 0 && !d_inlier_idx) return fail(c, RSDSFM_ERR_INVALID, "gathered mode needs inlier_idx");
    if (m > 0 && (!d_flow || !d_inl || !d_alpha || !d_alpha_k || !d_inl_out)) return fail(c, RSDSFM_ERR_INVALID, "null device pointer");
    const int np = const_acceleration ? 7 : 6;
    const size_t M = (size_t)std::max<int64_t>(m, 1);
    const size_t npart = (size_t)(d_best ? refine_partials_doubles_cap(c) : refine_partials_doubles(c, m));
    int rc = RSDSFM_OK;
    if (!ws_base) {
        rc = ensure_ws(c, refine_workspace_bytes(c, m, d_best != nullptr));
        if (rc != RSDSFM_OK) return rc;
        ws_base = c->d_ws;
    }
    rc = ensure_pinned(c, sizeof(RefineState) + 64);
    if (rc != RSDSFM_OK) return rc;
    Arena ws(ws_base);
    RefineBuffers& B = run->B;
    B = RefineBuffers();
    B.flow = d_flow;
    B.n_flow = n_flow;
    B.m = m;
    B.m_on_device = d_best != nullptr;
    B.inl = d_inl;
    B.alpha = d_alpha;
    B.alpha_k = d_alpha_k;
    B.inlier_idx = d_inlier_idx;
    B.flow_index_mode = flow_index_mode;
    // state and the bad-index flag share one block, laid out like the pinned host copy: one upload and one read-back cover both
    static_assert(sizeof(RefineState) % sizeof(int) == 0, "flag follows the state");
    char* state_block = ws.take<char>(sizeof(RefineState) + 64);
    B.state = reinterpret_cast<RefineState*>(state_block);
    B.uu = ws.take<double>(4 * M);
    B.beta = ws.take<double>(M);
    B.rho_a = ws.take<double>(M);
    B.rho_b = ws.take<double>(M);
    B.srho = ws.take<double>(M);
    B.partials = ws.take<double>(npart);
    B.bad_index = reinterpret_cast<int*>(state_block + sizeof(RefineState));
    B.zpartials = d_zpartials;
    run->np = np;
    // the radius-factorised path (refine_rf_kernels.hip) unless the context asks for the reference's arithmetic or this solve is run again
    // behind a tripped guard
    run->rf = !exact && c->lm_arithmetic == 0 && c->refine_arithmetic == 0;
    if (run->rf) c->refine_rf_runs += 1;
    run->d_inl_out = d_inl_out;
    run->tail = tail;
    run->launched = 0;
    run->hint_prev = c->refine_iters_hint;
    run->hs = hs_prefetch ? hs_prefetch : static_cast<RefineState*>(c->h_pinned);
    run->prefetch = hs_prefetch != nullptr;
    run->prefetched = false;
    B.state_host = hs_prefetch;  // prefetch: the output pass writes the state there itself
    if (d_best) {
        rc = refine_state_from_best_launch(c, d_best, B, np);
        if (rc != RSDSFM_OK) return rc;
    } else {
        RefineState* hs = static_cast<RefineState*>(c->h_pinned);
        int* h_bad = reinterpret_cast<int*>(static_cast<char*>(c->h_pinned) + sizeof(RefineState));
        memset(hs, 0, sizeof(RefineState));
        hs->np = np;
        for (int i = 0; i < 3; ++i) {
            hs->p[i] = v_in[i];
            hs->p[3 + i] = w_in[i];
        }
        hs->p[6] = k_in;
        hs->termination = -1;
        hs->radius = kInitialRadius;
        hs->need_schur = 1;  // the first slot of the iteration loop is the Schur pass of iteration 1
        memset(h_bad, 0, kRefineStateBlockTail);  // bad-index flag + the list counters of the radius-factorised path
        RSDSFM_HIP_CHECK(c, hipMemcpyAsync(B.state, hs, sizeof(RefineState) + kRefineStateBlockTail, hipMemcpyHostToDevice, c->stream));
    }
    rc = refine_trace_reset(c);
    if (rc != RSDSFM_OK) return rc;
    if (!run->rf) {  // (the radius-factorised path's first slot IS iteration zero)
        rc = refine_init_launch(c, B, np);
        if (rc != RSDSFM_OK) return rc;
    }
    // The iteration loop is enqueued in chunks of SLOTS (refine_kernels.hip: a streaming pass + the single-workgroup stage behind it; a solve
    // of `it` LM iterations consumes it + 1 of them, one more for every step that was rejected or got another radius than the speculated
    // one).  The kernels of a finished solve return immediately, but an empty slot still costs two launches (~9 us at 1280x720) and a chunk
    // that is too short a host round trip plus a second output pass and tail (~40 us).  The first chunk is 7 slots: DeepFlow-like pairs
    // take 3 / 4 / 5 / 6 iterations in 4 / 33 / 46 / 17 % of the solves and one in five has a step whose speculation does not apply
    // (tools/refine_slots.py: 4 .. 8 slots, 7 cover 96 %; 6 cover 73 % and cost 2 us more per solve on average; following the previous solve's
    // count more closely was measured 1 % slower, the counts vary from pair to pair) -- except behind a refinement that ended within 3
    // slots (noise-free data, e.g. ground-truth flow, ends after ONE iteration), where the first chunk is that count + 1, and behind one
    // that took more than 8 (acceleration mode: 7 .. 38, two per iteration that did not speculate or whose speculation did not apply), where it
    // is that count + 1 as well, at most 28.  The chunking changes when the host looks at the state, never what the kernels compute.
    const int hp = run->hint_prev;
    run->chunk = (hp >= 0 && hp <= 3) ? hp + 1 : (hp > 8 ? std::min(hp + 1, 28) : 7);
    // (radius-factorised path: a solve of `it` candidate evaluations consumes exactly it + 1 slots -- 4 .. 7 on DeepFlow-like pairs, 14 .. 16 in
    // acceleration mode --, the same rule fits)
    return refine_enqueue_chunk(c, run);
}

// one chunk of slots of the iteration loop, the output pass and the caller's tail
int refine_enqueue_chunk(Ctx* c, RefineRun* run) {
    static const bool watch = getenv("RSDSFM_SYNC_WATCHDOG_S") != nullptr;
    if (watch) {  // (diagnostics: what sync_stream's watchdog prints when this context's stream does not come back)
        const RefineBuffers B = run->B;
        const int launched = run->launched, chunk = run->chunk, rf = run->rf ? 1 : 0;
        set_sync_dumper([B, launched, chunk, rf](Ctx* cc) {
            fprintf(stderr, "[rsdsfm] last refinement chunk enqueued: slots [%d, %d), radius-factorised %d\n", launched, launched + chunk, rf);
            refine_rf_debug_dump(cc, B);
        });
    }
    if (run->rf) {
        // Slot g's pass carries the stage of slot g - 1 in its prologue.  The chunk is CLOSED by one more pass whose prologue runs the stage of
        // the chunk's last slot and publishes the state: on a solve that has ended -- the common case -- its workgroups leave right behind the
        // prologue (~6 us); on one that has not, it simply is the next slot.  There is NO stage kernel of its own on this path: a single
        // 512-thread workgroup holding 222 registers per lane, launched behind every pass while the other lanes of a sequence keep the chip busy
        // (what the iterate-by-iterate path does with its 72-register stage kernel while several solves share the GPU), wedged the GPU on the
        // MI355X boxes -- 3 of 3 sequence runs against 0 of 5 with the stage in the prologues (tools/seq_hang_probe.py, HISTORY.md).
        // rsdsfm_set_refine_stage therefore only concerns the iterate-by-iterate kernels.
        for (int i = 0; i <= run->chunk; ++i) {
            const int g = run->launched + i;
            int rc = refine_rf_pass_launch(c, run->B, run->np, g, run->launched, nullptr, 0, -1, nullptr, i == run->chunk);
            if (rc != RSDSFM_OK) return rc;
        }
        run->launched += 1;  // (the closing pass has a slot index of its own, whether or not it had anything left to do)
    } else {
        for (int i = 0; i < run->chunk; ++i) {
            int rc = refine_iter_launch(c, run->B, run->np, i, run->chunk);
            if (rc != RSDSFM_OK) return rc;
        }
    }
    run->launched += run->chunk;
    // the output pass is enqueued before the host knows whether the solve has finished (the common case: <= 5 iterations),
    // which saves a host round trip with an idle GPU; if iterations remain it simply runs again after the next chunk
    int rc = refine_finish_launch(c, run->B, run->d_inl_out);
    if (rc != RSDSFM_OK) return rc;
    // the tail only pays behind a chunk that can be the last one: where the previous solve needed clearly more slots than are enqueued so
    // far (acceleration mode: ~14) it is left out, and refine_poll enqueues it should the solve end early after all.  (One more than
    // enqueued is not "clearly": DeepFlow-like pairs vary by one or two from pair to pair, and a tail that was enqueued in vain is ~20 us
    // of kernels where a missing one is a ~35 us host round trip with an idle GPU.)
    run->tail_done = false;
    if (run->tail && !(run->hint_prev > run->launched + 1)) {
        rc = (*run->tail)(run->B);
        run->tail_done = true;
    }
    if (rc != RSDSFM_OK) return rc;
    if (run->prefetch) run->prefetched = true;  // (refine_finish_kernel has written the state to run->hs)
    return rc;
}

int refine_poll(Ctx* c, RefineRun* run, double v_out[3], double w_out[3], double* k_out, rsdsfm_lm_summary* summary) {
    if (!v_out || !w_out || !k_out) return fail(c, RSDSFM_ERR_INVALID, "bad arguments");
    RefineState* hs = run->hs;
    int* h_bad = reinterpret_cast<int*>(reinterpret_cast<char*>(hs) + sizeof(RefineState));
    for (bool first = true;; first = false) {
        if (!(first && run->prefetch && run->prefetched)) {  // (otherwise the read-back was enqueued with the chunk and the caller has waited)
            if (!run->prefetch) RSDSFM_HIP_CHECK(c, hipMemcpyAsync(hs, run->B.state, sizeof(RefineState) + sizeof(int), hipMemcpyDeviceToHost, c->stream));
            if (int rcs = sync_stream(c, "refine_poll line 168")) return rcs;
        }
        if (*h_bad) return fail(c, RSDSFM_ERR_INVALID, "flow index out of range (flow has fewer columns than inliers / bad inlier_idx)");
        if (hs->termination == kTermRestartExact) {  // a guard of the radius-factorised path: the caller runs the solve again, iterate by iterate
            c->refine_rf_restarts += 1;
            c->refine_rf_last_guard = hs->rf_guard;
            return kRcRefineRestartExact;
        }
        if (hs->termination >= 0) {
            if (run->tail && !run->tail_done) {  // ended earlier than the previous solve: the tail was not behind this chunk
                int rc = (*run->tail)(run->B);
                if (rc != RSDSFM_OK) return rc;
                run->tail_done = true;
                if (int rcs = sync_stream(c, "refine_poll line 181")) return rcs;
            }
            break;
        }
        if (run->launched > 8 * kMaxIter + 16) return fail(c, RSDSFM_ERR_NUMERIC, "refinement did not terminate");
        // later chunks: what the previous solve still needed at this point, between 2 and 8 slots (a DeepFlow-like solve that outlives its 7
        // slots needs one or two more; acceleration mode varies by tens from pair to pair)
        run->chunk = std::min(8, std::max(2, run->hint_prev - run->launched));
        int rc = refine_enqueue_chunk(c, run);
        if (rc != RSDSFM_OK) return rc;
    }
    c->refine_iters_hint = hs->slots;
    if (run->rf) c->refine_rf_resolves += hs->rf_resolves;
    for (int i = 0; i < 3; ++i) {
        v_out[i] = hs->p[i];
        w_out[i] = hs->p[3 + i];
    }
    *k_out = hs->p[6];
    if (summary) {
        summary->num_iterations = hs->iteration;
        summary->num_successful_steps = hs->num_successful;
        summary->num_unsuccessful_steps = hs->num_unsuccessful;
        summary->termination = hs->termination;
        summary->initial_cost = hs->initial_cost;
        summary->final_cost = hs->cost;
        summary->final_radius = hs->radius;
    }
    return RSDSFM_OK;
}

size_t refine_workspace_bytes(const Ctx* c, int64_t m, bool m_on_device) {
    const size_t M = (size_t)std::max<int64_t>(m, 1);
    const size_t npart = (size_t)(m_on_device ? refine_partials_doubles_cap(c) : refine_partials_doubles(c, m));
    return Arena::need(sizeof(RefineState) + 64) + Arena::need(32 * M) + 4 * Arena::need(8 * M) + Arena::need(8 * npart) + Arena::need(64) + 1024;
}

int refine_device(Ctx* c, const double* d_flow, int64_t n_flow, int64_t m, const double* d_inl, const double* d_alpha,
                  const double* d_alpha_k, const int64_t* d_inlier_idx, const double v_in[3], const double w_in[3], double k_in,
                  int const_acceleration, int flow_index_mode, double* d_inl_out, double v_out[3], double w_out[3], double* k_out,
                  rsdsfm_lm_summary* summary, const RefineTail* tail, double* d_zpartials, bool exact) {
    if (!v_out || !w_out || !k_out) return fail(c, RSDSFM_ERR_INVALID, "bad arguments");
    const double v0[3] = {v_in[0], v_in[1], v_in[2]}, w0[3] = {w_in[0], w_in[1], w_in[2]};  // (v_in may alias v_out)
    for (int attempt = 0;; ++attempt) {
        RefineRun run;
        int rc = refine_begin(c, d_flow, n_flow, m, d_inl, d_alpha, d_alpha_k, d_inlier_idx, v0, w0, k_in, const_acceleration, flow_index_mode,
                              d_inl_out, tail, nullptr, nullptr, &run, nullptr, d_zpartials, exact || attempt > 0);
        if (rc != RSDSFM_OK) return rc;
        rc = refine_poll(c, &run, v_out, w_out, k_out, summary);
        if (rc != kRcRefineRestartExact || attempt > 0) return rc == kRcRefineRestartExact ? fail(c, RSDSFM_ERR_NUMERIC, "refinement: restart loop") : rc;
    }
}

}  // namespace rsdsfm

using namespace rsdsfm;

extern "C" {

int rsdsfm_refine_dev(rsdsfm_ctx* ctx, const double* d_flow, int64_t n_flow, int64_t m, const double* d_inl, const double* d_alpha,
                      const double* d_alpha_k, const int64_t* d_inlier_idx, const double v_in[3], const double w_in[3], double k_in,
                      int const_acceleration, int flow_index_mode, double* d_inl_out, double v_out[3], double w_out[3], double* k_out,
                      rsdsfm_lm_summary* summary) {
    if (!ctx) return RSDSFM_ERR_INVALID;
    DeviceGuard device_guard_(&ctx->c);
    return refine_device(&ctx->c, d_flow, n_flow, m, d_inl, d_alpha, d_alpha_k, d_inlier_idx, v_in, w_in, k_in, const_acceleration,
                         flow_index_mode, d_inl_out, v_out, w_out, k_out, summary, nullptr, nullptr);
}

// tag: 0, or what rsdsfm_last_ransac_tag returned behind the rsdsfm_ransac whose outputs these arrays are (see include/rsdsfm.h)
static int refine_host(rsdsfm_ctx* ctx, uint64_t tag, const double* flow, int64_t n_flow, int64_t m, const double* inl, const double* alpha,
                       const double* alpha_k, const int64_t* inlier_idx, const double v_in[3], const double w_in[3], double k_in,
                       int const_acceleration, int flow_index_mode, double* inl_out, double v_out[3], double w_out[3], double* k_out,
                       rsdsfm_lm_summary* summary) {
    if (!ctx) return RSDSFM_ERR_INVALID;
    Ctx* c = &ctx->c;
    DeviceGuard device_guard_(c);
    if (m < 0 || n_flow < 0) return fail(c, RSDSFM_ERR_INVALID, "bad arguments");
    if (m > 0 && (!flow || !inl || !alpha || !alpha_k || !inl_out)) return fail(c, RSDSFM_ERR_INVALID, "null pointer");
    const size_t M = (size_t)m, NF = (size_t)n_flow;
    // The RANSAC's device-resident outputs instead of a second upload -- when the tag is the context's last RANSAC's, nothing has taken the staging
    // buffer since, the sizes are that RANSAC's, and 16 probed entries of the caller's arrays still hold what was downloaded / uploaded then (a
    // spot check against arrays rebuilt in between: the contract is that they are unmodified).  The flow is reused when it is the very array the
    // RANSAC was given as u (pointer + size + probes), else uploaded.
    const Ctx::RansacCache& rcache = c->ransac_cache;
    bool cached = tag != 0 && tag == rcache.tag && rcache.stage_gen == c->stage_gen && rcache.m == m && M > 0 && (inlier_idx != nullptr || flow_index_mode == RSDSFM_FLOW_COMPAT_RANK);
    for (int j = 0; cached && j < 16; ++j) cached = memcmp(&rcache.inl_probe[j], &inl[(3 * M - 1) * (size_t)j / 15], sizeof(double)) == 0;
    bool flow_cached = cached && flow == rcache.h_u && n_flow == rcache.n;
    for (int j = 0; flow_cached && j < 16; ++j) flow_cached = memcmp(&rcache.u_probe[j], &flow[(2 * NF - 1) * (size_t)j / 15], sizeof(double)) == 0;
    int rc = RSDSFM_OK;
    const double *d_flow, *d_inl, *d_a, *d_ak;
    const int64_t* d_idx;
    double* d_out;
    if (cached) {
        // (the output and -- if need be -- the flow go to the workspace tail the refinement does not use; the staging buffer stays as it is)
        const size_t extra = Arena::need(24 * M) + (flow_cached ? 0 : Arena::need(16 * NF)) + 1024;
        const size_t ws_ref = refine_workspace_bytes(c, m, false);
        rc = ensure_ws(c, ws_ref + extra);
        if (rc != RSDSFM_OK) return rc;
        Arena xa(static_cast<char*>(c->d_ws) + ((ws_ref + 255) & ~(size_t)255));
        d_out = xa.take<double>(3 * M);
        double* d_fl = flow_cached ? nullptr : xa.take<double>(2 * NF);
        if (!flow_cached && NF && (rc = xfer_h2d(c, d_fl, flow, 16 * NF)) != RSDSFM_OK) return rc;
        d_flow = flow_cached ? rcache.d_u : d_fl;
        d_inl = rcache.d_inl, d_a = rcache.d_alpha, d_ak = rcache.d_alpha_k, d_idx = inlier_idx ? rcache.d_idx : nullptr;
        c->refine_cache_hits += 1;
    } else {
        rc = ensure_stage(c, Arena::need(16 * NF) + 2 * Arena::need(24 * M) + 3 * Arena::need(8 * M) + 2048);
        if (rc != RSDSFM_OK) return rc;
        Arena sa(c->d_stage);
        double* s_flow = sa.take<double>(2 * NF);
        double* s_inl = sa.take<double>(3 * M);
        d_out = sa.take<double>(3 * M);
        double* s_a = sa.take<double>(M);
        double* s_ak = sa.take<double>(M);
        int64_t* s_idx = sa.take<int64_t>(M);
        const XferUp up[5] = {{s_flow, flow, 16 * NF}, {s_inl, inl, 24 * M}, {s_a, alpha, 8 * M}, {s_ak, alpha_k, 8 * M}, {s_idx, inlier_idx, inlier_idx ? 8 * M : 0}};
        if ((rc = xfer_h2d_many(c, up, 5)) != RSDSFM_OK) return rc;
        d_flow = s_flow, d_inl = s_inl, d_a = s_a, d_ak = s_ak, d_idx = inlier_idx ? s_idx : nullptr;
    }
    xfer_trace(cached ? "begin (resident)" : "uploaded");
    rc = refine_device(c, d_flow, n_flow, m, d_inl, d_a, d_ak, d_idx, v_in, w_in, k_in, const_acceleration, flow_index_mode, d_out, v_out, w_out, k_out,
                       summary, nullptr, nullptr);
    if (rc != RSDSFM_OK) return rc;
    xfer_trace("solved");
    if (M && (rc = xfer_d2h(c, inl_out, d_out, 24 * M)) != RSDSFM_OK) return rc;
    RSDSFM_HIP_CHECK(c, hipStreamSynchronize(c->stream));
    xfer_trace("downloaded");
    xfer_trace_dump("rsdsfm_refine");
    return RSDSFM_OK;
}

int rsdsfm_refine(rsdsfm_ctx* ctx, const double* flow, int64_t n_flow, int64_t m, const double* inl, const double* alpha,
                  const double* alpha_k, const int64_t* inlier_idx, const double v_in[3], const double w_in[3], double k_in,
                  int const_acceleration, int flow_index_mode, double* inl_out, double v_out[3], double w_out[3], double* k_out,
                  rsdsfm_lm_summary* summary) {
    return refine_host(ctx, 0, flow, n_flow, m, inl, alpha, alpha_k, inlier_idx, v_in, w_in, k_in, const_acceleration, flow_index_mode, inl_out, v_out, w_out,
                       k_out, summary);
}

int rsdsfm_refine_from_ransac(rsdsfm_ctx* ctx, uint64_t ransac_tag, const double* flow, int64_t n_flow, int64_t m, const double* inl,
                              const double* alpha, const double* alpha_k, const int64_t* inlier_idx, const double v_in[3], const double w_in[3],
                              double k_in, int const_acceleration, int flow_index_mode, double* inl_out, double v_out[3], double w_out[3],
                              double* k_out, rsdsfm_lm_summary* summary) {
    return refine_host(ctx, ransac_tag, flow, n_flow, m, inl, alpha, alpha_k, inlier_idx, v_in, w_in, k_in, const_acceleration, flow_index_mode, inl_out, v_out,
                       w_out, k_out, summary);
}

int rsdsfm_last_ransac_tag(rsdsfm_ctx* ctx, uint64_t* tag, int64_t* cache_hits_or_null) {
    if (!ctx || !tag) return RSDSFM_ERR_INVALID;
    *tag = ctx->c.ransac_cache.stage_gen == ctx->c.stage_gen ? ctx->c.ransac_cache.tag : 0;
    if (cache_hits_or_null) *cache_hits_or_null = ctx->c.refine_cache_hits;
    return RSDSFM_OK;
}

}  // extern "C"
