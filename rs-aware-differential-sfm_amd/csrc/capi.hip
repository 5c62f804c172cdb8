// capi.hip -- extern "C" entry points of librsdsfm_hip.so (see include/rsdsfm.h).
#include <string.h>

#include <algorithm>
#include <new>
#include <vector>

#include <chrono>
#include <thread>

#include "rsdsfm_internal.hpp"

namespace rsdsfm {

int fail(Ctx* c, int code, const char* msg) {
    if (c) c->err = msg;
    return code;
}

int ensure_stage(Ctx* c, size_t bytes) {
    c->stage_gen += 1;  // (whoever calls this is about to overwrite the staging buffer: what rsdsfm_ransac left there is gone)
    if (bytes <= c->stage_bytes) return RSDSFM_OK;
    RSDSFM_HIP_CHECK(c, hipStreamSynchronize(c->stream));
    if (c->d_stage) RSDSFM_HIP_CHECK(c, hipFree(c->d_stage));
    c->d_stage = nullptr;
    c->stage_bytes = 0;
    size_t want = std::max(bytes, (size_t)1 << 20);
    RSDSFM_HIP_CHECK(c, hipMalloc(&c->d_stage, want));
    c->stage_bytes = want;
    return RSDSFM_OK;
}

int ensure_ws(Ctx* c, size_t bytes) {
    if (bytes <= c->ws_bytes) return RSDSFM_OK;
    RSDSFM_HIP_CHECK(c, hipStreamSynchronize(c->stream));
    if (c->d_ws) RSDSFM_HIP_CHECK(c, hipFree(c->d_ws));
    c->d_ws = nullptr;
    c->ws_bytes = 0;
    size_t want = std::max(bytes, (size_t)1 << 20);
    RSDSFM_HIP_CHECK(c, hipMalloc(&c->d_ws, want));
    c->ws_bytes = want;
    return RSDSFM_OK;
}

int ensure_pinned(Ctx* c, size_t bytes) {
    if (bytes <= c->pinned_bytes) return RSDSFM_OK;
    if (c->h_pinned) RSDSFM_HIP_CHECK(c, hipHostFree(c->h_pinned));
    c->h_pinned = nullptr;
    c->pinned_bytes = 0;
    size_t want = (std::max(bytes, (size_t)1 << 16) + 63) & ~(size_t)63;  // (a multiple of 64: users address the END of the block too)
    RSDSFM_HIP_CHECK(c, hipHostMalloc(&c->h_pinned, want, hipHostMallocDefault));
    c->pinned_bytes = want;
    return RSDSFM_OK;
}

// hipStreamSynchronize on the context's stream -- with RSDSFM_SYNC_WATCHDOG_S=<seconds> in the environment (diagnosing a solve that does not
// come back) a polling wait that, once the time is up, says where the host was waiting, lets a registered dumper describe the device-resident
// state through a second stream, and aborts the process
static std::function<void(Ctx*)> g_sync_dumper;
void set_sync_dumper(std::function<void(Ctx*)> f) { g_sync_dumper = std::move(f); }
int sync_stream(Ctx* c, const char* where) {
    static const double limit = getenv("RSDSFM_SYNC_WATCHDOG_S") ? atof(getenv("RSDSFM_SYNC_WATCHDOG_S")) : (getenv("RSDSFM_SYNC_POLL") ? 1e9 : 0.0);
    if (limit <= 0.0) {
        RSDSFM_HIP_CHECK(c, hipStreamSynchronize(c->stream));
        return RSDSFM_OK;
    }
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        const hipError_t e = hipStreamQuery(c->stream);
        if (e == hipSuccess) return RSDSFM_OK;
        if (e != hipErrorNotReady) RSDSFM_HIP_CHECK(c, e);
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > limit) {
            fprintf(stderr, "[rsdsfm] the stream did not come back within %.1f s; the host waits at: %s\n", limit, where);
            if (g_sync_dumper) g_sync_dumper(c);
            fflush(stderr);
            abort();
        }
        std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
}

using StageAlloc = Arena;

void fill_lm_summary(const LmState& st, rsdsfm_lm_summary* s) {
    if (!s) return;
    s->num_iterations = st.iteration;
    s->num_successful_steps = st.num_successful;
    s->num_unsuccessful_steps = st.num_unsuccessful;
    s->termination = st.termination;
    s->initial_cost = st.initial_cost;
    s->final_cost = st.cost;
    s->final_radius = st.radius;
}

static int read_lm_state(Ctx* c) {
    RSDSFM_HIP_CHECK(c, hipMemcpyAsync(c->h_lm, c->d_lm, sizeof(LmState), hipMemcpyDeviceToHost, c->stream));
    RSDSFM_HIP_CHECK(c, hipStreamSynchronize(c->stream));
    return RSDSFM_OK;
}

}  // namespace rsdsfm

using namespace rsdsfm;

#define CTX_OR_FAIL(ctx)                  \
    if (!(ctx)) return RSDSFM_ERR_INVALID; \
    Ctx* c = &(ctx)->c;                    \
    DeviceGuard device_guard_(c)

extern "C" {

#ifndef RSDSFM_FUSED
#define RSDSFM_FUSED 0
#endif
const char* rsdsfm_version(void) {
    // (what the DEFAULT path computes with -- not what the build flag is called: since round 5 the depth solves walk the analytic LM trajectory,
    // since round 6 the joint refinement runs on radius-factorised Schur sums, both with fused multiply-adds and both guarded to the integers
    // of the reference's arithmetic, which rsdsfm_set_lm_arithmetic(1) selects operation for operation)
    return RSDSFM_FUSED ? "rsdsfm-mi355x 0.4.0 (gfx950, fp64; analytic LM trajectory + radius-factorised refinement, guarded to the reference arithmetic's "
                          "integers; rsdsfm_set_lm_arithmetic(1) = iterate by iterate with the FUSED per-pixel model: explicit fmas)"
                        : "rsdsfm-mi355x 0.4.0 (gfx950, fp64; analytic LM trajectory + radius-factorised refinement with fused multiply-adds, guarded to the "
                          "reference arithmetic's integers; rsdsfm_set_lm_arithmetic(1) = iterate by iterate in the reference's arithmetic: no fused multiply-add)";
}
int rsdsfm_fused_arithmetic(void) { return RSDSFM_FUSED; }

int rsdsfm_create(rsdsfm_ctx** out, int device, void* stream_or_null) {
    if (!out) return RSDSFM_ERR_INVALID;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return RSDSFM_ERR_NO_DEVICE;
    if (device < 0 || device >= count) return RSDSFM_ERR_NO_DEVICE;
    rsdsfm_ctx* ctx = new (std::nothrow) rsdsfm_ctx();
    if (!ctx) return RSDSFM_ERR_INVALID;
    Ctx* c = &ctx->c;
    c->device = device;
    DeviceGuard device_guard_(c);  // the caller's current device is restored on return
    {
        int cur = -1;
        if (hipGetDevice(&cur) != hipSuccess || cur != device) {
            delete ctx;
            return RSDSFM_ERR_HIP;
        }
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) c->num_cus = prop.multiProcessorCount;
    if (stream_or_null) {
        c->stream = static_cast<hipStream_t>(stream_or_null);
        c->own_stream = false;
    } else {
        if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
            delete ctx;
            return RSDSFM_ERR_HIP;
        }
        c->own_stream = true;
    }
    bool ok = hipMalloc(&c->d_partials, sizeof(double) * 2048 * 64) == hipSuccess &&
              hipMalloc(&c->d_tickets, sizeof(unsigned) * 64) == hipSuccess &&
              hipMalloc(&c->d_lm, sizeof(LmState)) == hipSuccess &&
              hipHostMalloc(reinterpret_cast<void**>(&c->h_lm), sizeof(LmState), hipHostMallocDefault) == hipSuccess &&
              hipMemset(c->d_tickets, 0, sizeof(unsigned) * 64) == hipSuccess &&
              hipMemset(c->d_lm, 0, sizeof(LmState)) == hipSuccess;
    if (ok) {
        memset(c->h_lm, 0, sizeof(LmState));
        c->h_lm->predict = 1;  // noisy data (the common case inside RANSAC) stops after one accepted step
        c->h_lm->status = 1;
        ok = hipMemcpy(c->d_lm, c->h_lm, sizeof(LmState), hipMemcpyHostToDevice) == hipSuccess;
    }
    if (!ok) {
        rsdsfm_destroy(ctx);
        return RSDSFM_ERR_HIP;
    }
    *out = ctx;
    return RSDSFM_OK;
}

void rsdsfm_destroy(rsdsfm_ctx* ctx) {
    if (!ctx) return;
    Ctx* c = &ctx->c;
    DeviceGuard device_guard_(c);
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    frame_release(c);  // (the sequence lanes, the second stream and its events)
    if (c->d_partials) (void)hipFree(c->d_partials);
    if (c->d_tickets) (void)hipFree(c->d_tickets);
    if (c->d_lm) (void)hipFree(c->d_lm);
    if (c->h_lm) (void)hipHostFree(c->h_lm);
    if (c->d_stage) (void)hipFree(c->d_stage);
    if (c->d_ws) (void)hipFree(c->d_ws);
    if (c->d_frame) (void)hipFree(c->d_frame);
    if (c->d_tile) (void)hipFree(c->d_tile);
    if (c->d_refine_trace) (void)hipFree(c->d_refine_trace);
    if (c->d_core_flag) (void)hipFree(c->d_core_flag);
    delete static_cast<RefineBuffers*>(c->tile_session);
    if (c->h_pinned) (void)hipHostFree(c->h_pinned);
    dist_release(c);
    for (unsigned* p : c->d_claim)
        if (p) (void)hipFree(p);
    if (c->ev_ready) (void)hipEventDestroy(c->ev_ready);
    for (hipEvent_t e : c->ev_prof)
        if (e) (void)hipEventDestroy(e);
    if (c->d_clk_probe) (void)hipFree(c->d_clk_probe);
    if (c->d_lma_list) (void)hipFree(c->d_lma_list);
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    delete ctx;
}

const char* rsdsfm_last_error(const rsdsfm_ctx* ctx) { return ctx ? ctx->c.err.c_str() : "null context"; }

int rsdsfm_synchronize(rsdsfm_ctx* ctx) {
    CTX_OR_FAIL(ctx);
    RSDSFM_HIP_CHECK(c, hipStreamSynchronize(c->stream));
    return RSDSFM_OK;
}

int rsdsfm_set_depth_variant(rsdsfm_ctx* ctx, int variant) {
    CTX_OR_FAIL(ctx);
    if (variant < 0 || variant > 3) return fail(c, RSDSFM_ERR_INVALID, "depth variant must be 0 (default), 1 (LDS-DMA), 2 (decision fused into launch 0) or 3 (separate decide kernel)");
    RSDSFM_HIP_CHECK(c, hipStreamSynchronize(c->stream));
    c->depth_variant = variant;
    return RSDSFM_OK;
}

int rsdsfm_set_profiling(rsdsfm_ctx* ctx, int on) {
    CTX_OR_FAIL(ctx);
    if (on && !c->ev_prof[0]) {
        RSDSFM_HIP_CHECK(c, hipEventCreate(&c->ev_prof[0]));
        RSDSFM_HIP_CHECK(c, hipEventCreate(&c->ev_prof[1]));
        RSDSFM_HIP_CHECK(c, hipMalloc((void**)&c->d_clk_probe, 4 * sizeof(unsigned long long)));
        RSDSFM_HIP_CHECK(c, hipMemsetAsync(c->d_clk_probe, 0, 4 * sizeof(unsigned long long), c->stream));
    }
    c->profile = on != 0;
    c->prof_pending = false;
    return RSDSFM_OK;
}

int rsdsfm_profile_last_ms(rsdsfm_ctx* ctx, const char* what, double* ms) {
    CTX_OR_FAIL(ctx);
    if (what && ms && !strncmp(what, "refine_rf_phase", 15) && ((what[15] >= '0' && what[15] <= '9') || (what[15] >= 'a' && what[15] <= 'f')) && !what[16]) {
        // opt-in phase stamps of the radius-factorised refinement's pass (environment RSDSFM_RF_STAMPS=1): microseconds workgroup 0 spent in
        // 0 the state load, 1 the stage in its prologue, 2 the loop over its inliers, 3 the row reduction, summed over the passes since the
        // last read of record 4 = the number of passes (reading it zeroes all five)
        static unsigned long long h[16];
        if (what[15] == '0') {  // (record 0 reads -- and zeroes -- all of them; 1 .. 7 return what that read found; 5 .. 7 split the stage: rows reduced, decided, solved)
            int rc = refine_rf_read_stamps(c, h);
            if (rc != RSDSFM_OK) return fail(c, RSDSFM_ERR_INVALID, "no phase stamps: set RSDSFM_RF_STAMPS=1 before the first refinement");
        }
        const int k = what[15] <= '9' ? what[15] - '0' : what[15] - 'a' + 10;
        *ms = k == 4 ? (double)h[4] : k >= 8 ? (double)(long long)(h[k] - h[8]) * 0.01 : (double)h[k] * 0.01;  // (8 .. f: when wave k - 8 left its loop, relative to wave 0)
        return RSDSFM_OK;
    }
    const bool clock = what && !strcmp(what, "ransac_lm_round0_clock_mhz");
    const int which = !what ? -1 : clock || !strcmp(what, "ransac_lm_round0") ? 0 : !strcmp(what, "depth_lm_batch") ? 1 : -1;
    if (which < 0 || !ms)
        return fail(c, RSDSFM_ERR_INVALID, "unknown profile record (known: \"ransac_lm_round0\", \"ransac_lm_round0_clock_mhz\", \"depth_lm_batch\")");
    if (!c->prof_pending || c->prof_what != which)
        return fail(c, RSDSFM_ERR_INVALID, "no such profile record: enable rsdsfm_set_profiling and run a RANSAC (LM mode) / a batched dense depth solve first");
    RSDSFM_HIP_CHECK(c, hipEventSynchronize(c->ev_prof[1]));
    if (clock) {
        // the shader clock one workgroup of that launch ran at: clocks of its life / 100 MHz ticks of its life (ransac_lm_kernel clk_probe)
        unsigned long long h[4];
        RSDSFM_HIP_CHECK(c, hipMemcpyAsync(h, c->d_clk_probe, sizeof(h), hipMemcpyDeviceToHost, c->stream));  // (on the context's stream: no null-stream semantics)
        RSDSFM_HIP_CHECK(c, hipStreamSynchronize(c->stream));
        if (h[3] <= h[1] || h[2] <= h[0]) return fail(c, RSDSFM_ERR_INVALID, "the bracketed launch left no clock stamps");
        *ms = (double)(h[2] - h[0]) * 100.0 / (double)(h[3] - h[1]);
        return RSDSFM_OK;
    }
    float f = 0.f;
    RSDSFM_HIP_CHECK(c, hipEventElapsedTime(&f, c->ev_prof[0], c->ev_prof[1]));
    *ms = (double)f;
    return RSDSFM_OK;
}

int rsdsfm_set_refine_trace(rsdsfm_ctx* ctx, int32_t rows) {
    CTX_OR_FAIL(ctx);
    if (rows < 0 || rows > 4096) return fail(c, RSDSFM_ERR_INVALID, "refine trace: 0 <= rows <= 4096");
    if (rows != c->refine_trace_rows) {
        RSDSFM_HIP_CHECK(c, hipStreamSynchronize(c->stream));
        if (c->d_refine_trace) RSDSFM_HIP_CHECK(c, hipFree(c->d_refine_trace));
        c->d_refine_trace = nullptr;
        c->refine_trace_rows = 0;
        if (rows > 0) {
            RSDSFM_HIP_CHECK(c, hipMalloc((void**)&c->d_refine_trace, (size_t)rows * kRefineTraceCols * sizeof(double)));
            c->refine_trace_rows = rows;
        }
    }
    if (c->d_refine_trace)  // all-ones bytes = NaN: "no such iteration"
        RSDSFM_HIP_CHECK(c, hipMemsetAsync(c->d_refine_trace, 0xFF, (size_t)rows * kRefineTraceCols * sizeof(double), c->stream));
    return RSDSFM_OK;
}

int rsdsfm_get_refine_trace(rsdsfm_ctx* ctx, double* out, int32_t rows) {
    CTX_OR_FAIL(ctx);
    if (!out || rows < 1) return fail(c, RSDSFM_ERR_INVALID, "refine trace: null output or rows < 1");
    if (!c->d_refine_trace) return fail(c, RSDSFM_ERR_INVALID, "no refine trace: call rsdsfm_set_refine_trace(ctx, rows) before the refinement");
    if (rows > c->refine_trace_rows) return fail(c, RSDSFM_ERR_INVALID, "refine trace: more rows requested than rsdsfm_set_refine_trace reserved");
    RSDSFM_HIP_CHECK(c, hipMemcpyAsync(out, c->d_refine_trace, (size_t)rows * kRefineTraceCols * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    RSDSFM_HIP_CHECK(c, hipStreamSynchronize(c->stream));
    return RSDSFM_OK;
}

int rsdsfm_set_ransac_speculation(rsdsfm_ctx* ctx, int k0) {
    CTX_OR_FAIL(ctx);
    if (k0 != 0 && k0 != 2 && k0 != KMAX) return fail(c, RSDSFM_ERR_INVALID, "ransac speculation depth must be 0 (automatic, default), 2 or 3");
    c->ransac_k0 = k0;
    return RSDSFM_OK;
}

int rsdsfm_set_ransac_math(rsdsfm_ctx* ctx, int mode) {
    CTX_OR_FAIL(ctx);
    if (mode < 0 || mode > 1) return fail(c, RSDSFM_ERR_INVALID, "ransac math: 0 = in-range function cores with restart (default), 1 = standard functions");
    c->ransac_math_mode = mode;
    return RSDSFM_OK;
}

int rsdsfm_ransac_restarts(rsdsfm_ctx* ctx, int64_t* count) {
    CTX_OR_FAIL(ctx);
    if (!count) return fail(c, RSDSFM_ERR_INVALID, "ransac restarts: null output");
    int64_t total = c->ransac_restarts;
    for (rsdsfm_ctx* lane : c->lanes) total += lane->c.ransac_restarts;
    *count = total;
    return RSDSFM_OK;
}

int rsdsfm_set_lm_arithmetic(rsdsfm_ctx* ctx, int mode) {
    CTX_OR_FAIL(ctx);
    if (mode < 0 || mode > 2)
        return fail(c, RSDSFM_ERR_INVALID, "LM arithmetic: 0 = analytic trajectory with guards (default), 1 = iterate by iterate, 2 = 0 with the frame solve's count-only pass forced");
    c->lm_arithmetic = mode == 1 ? 1 : 0;
    c->lma_count_only_force = mode == 2;
    c->lma_hold = c->lma_unique_run = 0;
    dist_reset_hold(c);
    for (rsdsfm_ctx* lane : c->lanes)
        lane->c.lm_arithmetic = c->lm_arithmetic, lane->c.lma_count_only_force = c->lma_count_only_force, lane->c.lma_hold = lane->c.lma_unique_run = 0;
    return RSDSFM_OK;
}

int rsdsfm_lma_count_only(rsdsfm_ctx* ctx, int64_t* runs, int64_t* lazy_runs) {
    CTX_OR_FAIL(ctx);
    if (!runs) return fail(c, RSDSFM_ERR_INVALID, "lma count only: null output");
    int64_t a = c->lma_count_only_runs, b = c->lma_lazy_runs;
    for (rsdsfm_ctx* lane : c->lanes) a += lane->c.lma_count_only_runs, b += lane->c.lma_lazy_runs;
    *runs = a;
    if (lazy_runs) *lazy_runs = b;
    return RSDSFM_OK;
}

int rsdsfm_lma_restarts(rsdsfm_ctx* ctx, int64_t* count, int32_t* last_guards) {
    CTX_OR_FAIL(ctx);
    if (!count) return fail(c, RSDSFM_ERR_INVALID, "lma restarts: null output");
    int64_t total = c->lma_restarts;
    int guards = c->lma_last_guard;
    for (rsdsfm_ctx* lane : c->lanes) total += lane->c.lma_restarts, guards |= lane->c.lma_last_guard;
    *count = total;
    if (last_guards) *last_guards = guards;
    return RSDSFM_OK;
}

int rsdsfm_set_refine_arithmetic(rsdsfm_ctx* ctx, int mode) {
    CTX_OR_FAIL(ctx);
    if (mode != 0 && mode != 1) return fail(c, RSDSFM_ERR_INVALID, "refine arithmetic: 0 = radius-factorised (default), 1 = iterate by iterate");
    c->refine_arithmetic = mode;
    for (rsdsfm_ctx* lane : c->lanes) lane->c.refine_arithmetic = mode;
    return RSDSFM_OK;
}

int rsdsfm_refine_restarts(rsdsfm_ctx* ctx, int64_t* runs, int64_t* restarts, int64_t* resolves_or_null, int32_t* last_guard_or_null) {
    CTX_OR_FAIL(ctx);
    if (!runs || !restarts) return fail(c, RSDSFM_ERR_INVALID, "refine restarts: null output");
    int64_t n = c->refine_rf_runs, r = c->refine_rf_restarts, s = c->refine_rf_resolves;
    int guard = c->refine_rf_last_guard;
    for (rsdsfm_ctx* lane : c->lanes) {
        n += lane->c.refine_rf_runs, r += lane->c.refine_rf_restarts, s += lane->c.refine_rf_resolves;
        if (lane->c.refine_rf_last_guard) guard = lane->c.refine_rf_last_guard;
    }
    *runs = n, *restarts = r;
    if (resolves_or_null) *resolves_or_null = s;
    if (last_guard_or_null) *last_guard_or_null = guard;
    return RSDSFM_OK;
}

int rsdsfm_depth_restarts(rsdsfm_ctx* ctx, int64_t* count) {
    CTX_OR_FAIL(ctx);
    if (!count) return fail(c, RSDSFM_ERR_INVALID, "depth restarts: null output");
    *count = c->depth_restarts;
    return RSDSFM_OK;
}

int rsdsfm_set_true_flow_search(rsdsfm_ctx* ctx, int mode) {
    CTX_OR_FAIL(ctx);
    if (mode < 0 || mode > 2) return fail(c, RSDSFM_ERR_INVALID, "true-flow search mode: 0 = automatic, 1 = exhaustive, 2 = interval-pruned");
    c->true_flow_exhaustive = mode;
    return RSDSFM_OK;
}

const char* rsdsfm_kernel_name(const char* entry_point) {
    if (!entry_point) return "";
    if (!strcmp(entry_point, "estimate_inverse_depths_lm")) return RSDSFM_FUSED ? "depth_lm_kernel<1, false>" : "depth_lm_kernel<1, true>";
    if (!strcmp(entry_point, "estimate_inverse_depths_closed_form")) return "depth_closed_form_kernel";
    return "";
}

// ---------------------------------------------------------------------------------------------------
// dense depth solve
// ---------------------------------------------------------------------------------------------------
int rsdsfm_estimate_inverse_depths_dev(rsdsfm_ctx* ctx, const double* d_q, const double* d_u, int64_t n,
                                       const double v[3], const double w[3], double k, const double* d_alpha,
                                       const double* d_alpha_k, int depth_mode, double* d_rho) {
    CTX_OR_FAIL(ctx);
    if (n < 0 || !v || !w) return fail(c, RSDSFM_ERR_INVALID, "bad arguments");
    if (n > 0 && (!d_q || !d_u || !d_alpha || !d_alpha_k || !d_rho)) return fail(c, RSDSFM_ERR_INVALID, "null device pointer");
    Pose pose;
    memcpy(pose.v, v, sizeof(pose.v));
    memcpy(pose.w, w, sizeof(pose.w));
    pose.k = k;
    if (depth_mode == RSDSFM_DEPTH_CLOSED_FORM) return depth_closed_form_launch(c, d_q, d_u, d_alpha, d_alpha_k, n, pose, d_rho);
    if (depth_mode != RSDSFM_DEPTH_CERES_LM) return fail(c, RSDSFM_ERR_INVALID, "unknown depth_mode");
    // fixed fast-path sequence: speculate (launch 0) -> decide -> launch 1 (apply / continue / no-op)
    if (c->depth_variant == 2) {  // launch 0 carries the decision in its tail (last workgroup): no separate decide kernel
        int rcf = depth_lm_fused_launch(c, d_q, d_u, d_alpha, d_alpha_k, n, pose, d_rho);
        if (rcf != RSDSFM_OK) return rcf;
        rcf = depth_lm_launch(c, d_q, d_u, d_alpha, d_alpha_k, n, pose, d_rho, 1);
        c->lm_issued_k = 2;
        c->lm_issued_d = 1;
        return rcf;
    }
    if (depth_lma_allowed(c, n)) {  // the analytic LM trajectory (depth_lma_kernels.hip): launch 0 + ONE launch that decides and finishes
        Ctx* cs1[1] = {c};
        const double *q1[1] = {d_q}, *u1[1] = {d_u}, *a1[1] = {d_alpha}, *ak1[1] = {d_alpha_k};
        double* r1[1] = {d_rho};
        int rc0 = depth_lma_batch_launch(cs1, 1, q1, u1, a1, ak1, &n, &pose, r1);
        c->lm_issued_k = 1;
        c->lm_issued_d = 1;
        return rc0;
    }
    if (c->depth_variant == 0) {  // launch 0, then ONE launch that decides and (if needed) applies
        int rc0 = depth_lm_launch(c, d_q, d_u, d_alpha, d_alpha_k, n, pose, d_rho, 0, /*core=*/true);  // (the follow-up launch checks the cores' range flag)
        if (rc0 != RSDSFM_OK) return rc0;
        rc0 = depth_lm_decide_apply_launch(c, d_q, d_u, d_alpha, d_alpha_k, n, pose, d_rho);
        c->lm_issued_k = 1;  // a continuation (status 0) starts with launch 1 in rsdsfm_depth_finish_dev
        c->lm_issued_d = 1;
        return rc0;
    }
    int rc = depth_lm_launch(c, d_q, d_u, d_alpha, d_alpha_k, n, pose, d_rho, 0);
    if (rc != RSDSFM_OK) return rc;
    rc = depth_lm_decide_launch(c, n, 0);
    if (rc != RSDSFM_OK) return rc;
    rc = depth_lm_launch(c, d_q, d_u, d_alpha, d_alpha_k, n, pose, d_rho, 1);
    c->lm_issued_k = 2;
    c->lm_issued_d = 1;
    return rc;
}

static int depth_batch_common(rsdsfm_ctx* const* ctxs, int32_t count, const double* const* d_q, const double* const* d_u, const int64_t* n,
                              const double* v3, const double* w3, const double* k, const double* const* d_alpha,
                              const double* const* d_alpha_k, double* const* d_rho, int launch0_only) {
    if (!ctxs || count < 1 || !ctxs[0]) return RSDSFM_ERR_INVALID;
    Ctx* c = &ctxs[0]->c;
    DeviceGuard device_guard_(c);
    if (count > kDepthBatchMax) return fail(c, RSDSFM_ERR_INVALID, "at most 8 solves per batched launch");
    if (!d_q || !d_u || !n || !v3 || !w3 || !k || !d_alpha || !d_alpha_k || !d_rho) return fail(c, RSDSFM_ERR_INVALID, "null argument array");
    Ctx* cs[kDepthBatchMax];
    Pose poses[kDepthBatchMax];
    for (int i = 0; i < count; ++i) {
        if (!ctxs[i]) return fail(c, RSDSFM_ERR_INVALID, "null context in the batch");
        cs[i] = &ctxs[i]->c;
        if (cs[i]->stream != c->stream || cs[i]->device != c->device) return fail(c, RSDSFM_ERR_INVALID, "all contexts of a batch must share one stream");
        for (int j = 0; j < i; ++j)
            if (cs[j] == cs[i]) return fail(c, RSDSFM_ERR_INVALID, "a context may appear only once in a batch (it owns the solve's state)");
        if (n[i] < 0 || (n[i] > 0 && (!d_q[i] || !d_u[i] || !d_alpha[i] || !d_alpha_k[i] || !d_rho[i]))) return fail(c, RSDSFM_ERR_INVALID, "bad problem in the batch");
        memcpy(poses[i].v, v3 + 3 * i, sizeof(poses[i].v));
        memcpy(poses[i].w, w3 + 3 * i, sizeof(poses[i].w));
        poses[i].k = k[i];
    }
    bool analytic = !launch0_only;  // (a launch0_only caller drives the decide stage itself: the iterate-by-iterate protocol)
    for (int i = 0; i < count; ++i) analytic = analytic && depth_lma_allowed(cs[i], n[i]);
    int rc = analytic ? depth_lma_batch_launch(cs, count, d_q, d_u, d_alpha, d_alpha_k, n, poses, d_rho)
                      : depth_lm_batch_launch(cs, count, d_q, d_u, d_alpha, d_alpha_k, n, poses, d_rho, launch0_only);
    if (rc != RSDSFM_OK) return rc;
    for (int i = 0; i < count; ++i) {
        cs[i]->lm_issued_k = 1;
        cs[i]->lm_issued_d = launch0_only ? 0 : 1;
    }
    return RSDSFM_OK;
}

int rsdsfm_estimate_inverse_depths_batch_dev(rsdsfm_ctx* const* ctxs, int32_t count, const double* const* d_q, const double* const* d_u,
                                             const int64_t* n, const double* v3, const double* w3, const double* k,
                                             const double* const* d_alpha, const double* const* d_alpha_k, double* const* d_rho) {
    return depth_batch_common(ctxs, count, d_q, d_u, n, v3, w3, k, d_alpha, d_alpha_k, d_rho, 0);
}

int rsdsfm_depth_lm_batch_launch_dev(rsdsfm_ctx* const* ctxs, int32_t count, const double* const* d_q, const double* const* d_u,
                                     const int64_t* n, const double* v3, const double* w3, const double* k, const double* const* d_alpha,
                                     const double* const* d_alpha_k, double* const* d_rho) {
    return depth_batch_common(ctxs, count, d_q, d_u, n, v3, w3, k, d_alpha, d_alpha_k, d_rho, 1);
}

int rsdsfm_depth_lm_launch_dev(rsdsfm_ctx* ctx, const double* d_q, const double* d_u, int64_t n, const double v[3],
                               const double w[3], double k, const double* d_alpha, const double* d_alpha_k,
                               double* d_rho, int launch_id) {
    CTX_OR_FAIL(ctx);
    if (n < 0 || !v || !w || launch_id < 0) return fail(c, RSDSFM_ERR_INVALID, "bad arguments");
    if (n > 0 && (!d_q || !d_u || !d_alpha || !d_alpha_k || !d_rho)) return fail(c, RSDSFM_ERR_INVALID, "null device pointer");
    Pose pose;
    memcpy(pose.v, v, sizeof(pose.v));
    memcpy(pose.w, w, sizeof(pose.w));
    pose.k = k;
    return depth_lm_launch(c, d_q, d_u, d_alpha, d_alpha_k, n, pose, d_rho, launch_id);
}

// ---- stage-level entry points of the row-tiled (multi-GPU) dense depth solve ----
int rsdsfm_depth_lm_sums_row_size(void) { return NS; }

int rsdsfm_depth_lm_reduce_dev(rsdsfm_ctx* ctx, int64_t n_shard, double* d_row) {
    CTX_OR_FAIL(ctx);
    if (n_shard < 0 || !d_row) return fail(c, RSDSFM_ERR_INVALID, "bad arguments");
    return depth_lm_reduce_launch(c, n_shard, d_row);
}

int rsdsfm_depth_lm_decide_rows_dev(rsdsfm_ctx* ctx, const double* d_rows, int32_t nrows, int64_t n_total, int launch_id) {
    CTX_OR_FAIL(ctx);
    if (!d_rows || nrows < 1 || n_total < 0 || launch_id < 0) return fail(c, RSDSFM_ERR_INVALID, "bad arguments");
    return depth_lm_decide_rows_launch(c, d_rows, nrows, n_total, launch_id);
}

int rsdsfm_depth_lm_state(rsdsfm_ctx* ctx, int32_t* status, int32_t* next_launch, rsdsfm_lm_summary* summary) {
    CTX_OR_FAIL(ctx);
    int rc = read_lm_state(c);
    if (rc != RSDSFM_OK) return rc;
    // (a launch 0 whose result does not count -- LmScal::restart: a function-core miss, a guard of the analytic trajectory -- is reported as
    // "start over": rsdsfm_depth_finish_dev does that by itself, a caller driving the launches on its own must not continue from its sums)
    const bool start_over = c->h_lm->status == 0 && c->h_lm->restart != 0;
    if (status) *status = start_over ? 3 : c->h_lm->status;
    if (next_launch) *next_launch = start_over ? 0 : c->h_lm->next_launch;
    fill_lm_summary(*c->h_lm, summary);
    return RSDSFM_OK;
}

int rsdsfm_depth_finish_dev(rsdsfm_ctx* ctx, const double* d_q, const double* d_u, int64_t n, const double v[3],
                            const double w[3], double k, const double* d_alpha, const double* d_alpha_k,
                            double* d_rho, rsdsfm_lm_summary* summary, int32_t* extra_launches) {
    CTX_OR_FAIL(ctx);
    if (!v || !w) return fail(c, RSDSFM_ERR_INVALID, "bad arguments");
    Pose pose;
    memcpy(pose.v, v, sizeof(pose.v));
    memcpy(pose.w, w, sizeof(pose.w));
    pose.k = k;
    int extra = 0;
    int rc = read_lm_state(c);
    if (rc != RSDSFM_OK) return rc;
    if (c->h_lm->status == 0 && c->h_lm->restart) {
        // launch 0 ran its Jacobi scaling through the in-range function cores and met an argument out of range (a non-finite flow, a
        // denormal Jacobian: depth_kernels.hip CORE) -- or (restart == 2) it ran on the analytic trajectory and a guard tripped
        // (depth_lma_kernels.hip): the solve starts over iterate by iterate with the standard functions -- launch 0, its decision, and
        // from there the ordinary continuation below (a function-core miss keeps the context on the standard functions for its next 16 solves)
        if (c->h_lm->restart == 2) {
            c->lma_restarts += 1;
            c->lma_last_guard = 1 << std::min(std::max(c->h_lm->iteration, 0), 15);
        } else {
            c->depth_standard_math = 16;
            c->depth_restarts += 1;
        }
        rc = depth_lm_launch(c, d_q, d_u, d_alpha, d_alpha_k, n, pose, d_rho, 0, /*core=*/false);
        if (rc != RSDSFM_OK) return rc;
        rc = depth_lm_decide_launch(c, n, 0);
        if (rc != RSDSFM_OK) return rc;
        c->lm_issued_k = 1;
        c->lm_issued_d = 1;
        extra += 2;
        rc = read_lm_state(c);
        if (rc != RSDSFM_OK) return rc;
    }
    for (;;) {
        const LmState& st = *c->h_lm;
        if (st.status == 1) break;
        if (extra > 4 * kMaxIter + 8) return fail(c, RSDSFM_ERR_NUMERIC, "LM state machine did not terminate");
        if (st.status == 2) {  // result known, needs the designated launch to write it
            if (st.next_launch < c->lm_issued_k) break;  // that launch already ran
            rc = depth_lm_launch(c, d_q, d_u, d_alpha, d_alpha_k, n, pose, d_rho, st.next_launch);
            if (rc != RSDSFM_OK) return rc;
            c->lm_issued_k = st.next_launch + 1;
            ++extra;
            break;
        }
        // status 0: launch `next_launch` must speculate (it may already have, in the fast path) and be decided
        const int id = st.next_launch;
        if (id >= c->lm_issued_k) {
            rc = depth_lm_launch(c, d_q, d_u, d_alpha, d_alpha_k, n, pose, d_rho, id);
            if (rc != RSDSFM_OK) return rc;
            c->lm_issued_k = id + 1;
            ++extra;
        }
        rc = depth_lm_decide_launch(c, n, id);
        if (rc != RSDSFM_OK) return rc;
        c->lm_issued_d = id + 1;
        ++extra;
        rc = read_lm_state(c);
        if (rc != RSDSFM_OK) return rc;
    }
    RSDSFM_HIP_CHECK(c, hipStreamSynchronize(c->stream));
    fill_lm_summary(*c->h_lm, summary);
    if (extra_launches) *extra_launches = extra;
    if (c->h_lm->termination == RSDSFM_TERM_FAILURE) return fail(c, RSDSFM_ERR_NUMERIC, "LM failure (5 consecutive invalid steps)");
    return RSDSFM_OK;
}

int rsdsfm_estimate_inverse_depths(rsdsfm_ctx* ctx, const double* q, const double* u, int64_t n, const double v[3],
                                   const double w[3], double k, const double* alpha, const double* alpha_k,
                                   int depth_mode, double* inv_depth, rsdsfm_lm_summary* summary) {
    CTX_OR_FAIL(ctx);
    if (n < 0 || !v || !w) return fail(c, RSDSFM_ERR_INVALID, "bad arguments");
    if (n > 0 && (!q || !u || !alpha || !alpha_k || !inv_depth)) return fail(c, RSDSFM_ERR_INVALID, "null pointer");
    const size_t N = (size_t)n;
    int rc = ensure_stage(c, 2 * StageAlloc::need(16 * N) + 3 * StageAlloc::need(8 * N) + 1024);
    if (rc != RSDSFM_OK) return rc;
    StageAlloc sa(c->d_stage);
    double* d_q = sa.take<double>(2 * N);
    double* d_u = sa.take<double>(2 * N);
    double* d_a = sa.take<double>(N);
    double* d_ak = sa.take<double>(N);
    double* d_rho = sa.take<double>(N);
    if (n > 0) {
        const XferUp up[4] = {{d_q, q, 16 * N}, {d_u, u, 16 * N}, {d_a, alpha, 8 * N}, {d_ak, alpha_k, 8 * N}};  // (host_xfer.hip)
        if ((rc = xfer_h2d_many(c, up, 4)) != RSDSFM_OK) return rc;
    }
    rc = rsdsfm_estimate_inverse_depths_dev(ctx, d_q, d_u, n, v, w, k, d_a, d_ak, depth_mode, d_rho);
    if (rc != RSDSFM_OK) return rc;
    if (depth_mode == RSDSFM_DEPTH_CERES_LM) {
        rc = rsdsfm_depth_finish_dev(ctx, d_q, d_u, n, v, w, k, d_a, d_ak, d_rho, summary, nullptr);
        if (rc != RSDSFM_OK) return rc;
    } else if (summary) {
        memset(summary, 0, sizeof(*summary));
        summary->num_iterations = 1;
        summary->num_successful_steps = 1;
        summary->termination = RSDSFM_TERM_GRADIENT;
    }
    if (n > 0) RSDSFM_HIP_CHECK(c, hipMemcpyAsync(inv_depth, d_rho, 8 * N, hipMemcpyDeviceToHost, c->stream));
    RSDSFM_HIP_CHECK(c, hipStreamSynchronize(c->stream));
    return RSDSFM_OK;
}


// ---------------------------------------------------------------------------------------------------
// RS scale factors, pose table
// ---------------------------------------------------------------------------------------------------
int rsdsfm_get_alpha(rsdsfm_ctx* ctx, const double* flow_px, int64_t n, double h, double gamma, double* alpha) {
    CTX_OR_FAIL(ctx);
    if (n < 0 || (n > 0 && (!flow_px || !alpha))) return fail(c, RSDSFM_ERR_INVALID, "bad arguments");
    const size_t N = (size_t)n;
    int rc = ensure_stage(c, StageAlloc::need(16 * N) + StageAlloc::need(8 * N));
    if (rc != RSDSFM_OK) return rc;
    StageAlloc sa(c->d_stage);
    double* d_f = sa.take<double>(2 * N);
    double* d_a = sa.take<double>(N);
    if (n == 0) return RSDSFM_OK;
    if ((rc = xfer_h2d(c, d_f, flow_px, 16 * N)) != RSDSFM_OK) return rc;
    rc = alpha_launch(c, d_f, n, h, gamma, d_a);
    if (rc != RSDSFM_OK) return rc;
    RSDSFM_HIP_CHECK(c, hipMemcpyAsync(alpha, d_a, 8 * N, hipMemcpyDeviceToHost, c->stream));
    RSDSFM_HIP_CHECK(c, hipStreamSynchronize(c->stream));
    return RSDSFM_OK;
}

int rsdsfm_get_alpha_k(rsdsfm_ctx* ctx, const double* q_px, const double* flow_px, int64_t n, double h, double gamma,
                       double* alpha_k) {
    CTX_OR_FAIL(ctx);
    if (n < 0 || (n > 0 && (!q_px || !flow_px || !alpha_k))) return fail(c, RSDSFM_ERR_INVALID, "bad arguments");
    const size_t N = (size_t)n;
    int rc = ensure_stage(c, 2 * StageAlloc::need(16 * N) + StageAlloc::need(8 * N));
    if (rc != RSDSFM_OK) return rc;
    StageAlloc sa(c->d_stage);
    double* d_q = sa.take<double>(2 * N);
    double* d_f = sa.take<double>(2 * N);
    double* d_a = sa.take<double>(N);
    if (n == 0) return RSDSFM_OK;
    const XferUp up[2] = {{d_q, q_px, 16 * N}, {d_f, flow_px, 16 * N}};
    if ((rc = xfer_h2d_many(c, up, 2)) != RSDSFM_OK) return rc;
    rc = alpha_k_launch(c, d_q, d_f, n, h, gamma, d_a);
    if (rc != RSDSFM_OK) return rc;
    RSDSFM_HIP_CHECK(c, hipMemcpyAsync(alpha_k, d_a, 8 * N, hipMemcpyDeviceToHost, c->stream));
    RSDSFM_HIP_CHECK(c, hipStreamSynchronize(c->stream));
    return RSDSFM_OK;
}

int rsdsfm_pose_table(rsdsfm_ctx* ctx, const double v[3], const double w[3], double k, double gamma, int32_t rows,
                      double* R, double* t) {
    CTX_OR_FAIL(ctx);
    if (rows < 0 || !v || !w || (rows > 0 && (!R || !t))) return fail(c, RSDSFM_ERR_INVALID, "bad arguments");
    if (rows == 0) return RSDSFM_OK;
    Pose pose;
    memcpy(pose.v, v, sizeof(pose.v));
    memcpy(pose.w, w, sizeof(pose.w));
    pose.k = k;
    const size_t Rr = (size_t)rows;
    int rc = ensure_stage(c, StageAlloc::need(72 * Rr) + StageAlloc::need(24 * Rr));
    if (rc != RSDSFM_OK) return rc;
    StageAlloc sa(c->d_stage);
    double* d_R = sa.take<double>(9 * Rr);
    double* d_t = sa.take<double>(3 * Rr);
    rc = pose_table_launch(c, pose, gamma, rows, d_R, d_t);
    if (rc != RSDSFM_OK) return rc;
    RSDSFM_HIP_CHECK(c, hipMemcpyAsync(R, d_R, 72 * Rr, hipMemcpyDeviceToHost, c->stream));
    RSDSFM_HIP_CHECK(c, hipMemcpyAsync(t, d_t, 24 * Rr, hipMemcpyDeviceToHost, c->stream));
    RSDSFM_HIP_CHECK(c, hipStreamSynchronize(c->stream));
    return RSDSFM_OK;
}

}  // extern "C"
