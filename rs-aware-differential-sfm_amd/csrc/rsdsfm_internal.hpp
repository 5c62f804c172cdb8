// rsdsfm_internal.hpp -- shared definitions of the MI355X (gfx950) solver library.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <functional>
#include <string>
#include <vector>

#include "../../include/rsdsfm.h"

namespace rsdsfm {

// ---------------------------------------------------------------------------------------------------
// Ceres 1.14 Solver::Options defaults used by every solve in nonlinearRefinement.cc (the reference
// only sets linear_solver_type = DENSE_SCHUR, nonlinearRefinement.cc:160-161, :225-226)
// ---------------------------------------------------------------------------------------------------
constexpr int kMaxIter = 50;
constexpr int kRefineTraceCols = RSDSFM_REFINE_TRACE_COLS;
constexpr double kInitialRadius = 1e4;
constexpr double kMaxRadius = 1e16;
constexpr double kMinRadius = 1e-32;
constexpr double kMinRelDecrease = 1e-3;
constexpr double kMinLmDiag = 1e-6;
constexpr double kMaxLmDiag = 1e32;
constexpr double kFunctionTol = 1e-6;
constexpr double kGradientTol = 1e-10;
// (the max|J r| slots of the RANSAC's LM rows travel as the indicator "some pixel is above kGradientTol" -- 1.0 / 0.0, combined with
// fmax and compared with `<= kGradientTol` downstream: ransac_kernels.hip; that encoding needs the tolerance below 1)
static_assert(kGradientTol < 1.0, "the indicator encoding of the gradient-maximum slots needs kGradientTol < 1");
constexpr double kParameterTol = 1e-8;
constexpr int kMaxInvalid = 5;
// analytic LM trajectory, guard (d) (lma_common.hpp): a tie in the inlier count whose error sums differ by less than kLmaTie x count
constexpr double kLmaTie = 1e-11;

// speculative LM batching: one kernel launch evaluates up to KMAX consecutive LM iterations under the
// assumption that every step is accepted with step quality ~1 (radius x3); the decision logic (run by
// the last workgroup to finish) verifies the assumption and replans when it does not hold.
constexpr int KMAX = 3;
constexpr int NS = 3 + 5 * KMAX;  // sums per launch: [2cost, xsq, gmax] + KMAX x [2cost, model, stepsq, xsq, gmax]

struct Pose {
    double v[3];
    double w[3];
    double k;
};

// Device-resident state machine of one emulated Ceres trust-region solve (dense depth problem).
// LmScal: the scalar part (lives in registers while lm_advance runs); LmState adds the accepted-radius history.
struct LmScal {
    int32_t status;  // 0 = running (next launch speculates K more iterations), 1 = done, rho holds the result,
                     // 2 = done, next launch must write the result (apply)
    int32_t n_hist;  // accepted steps so far (their radii are hist[0..n_hist))
    int32_t K;       // candidates to speculate in the next launch (0 = apply only)
    int32_t write_which;  // which state index (0 = after hist, j = after candidate j) the next launch writes
    int32_t iteration;    // Ceres iteration counter
    int32_t num_successful;
    int32_t num_unsuccessful;
    int32_t invalid_run;
    int32_t termination;  // RSDSFM_TERM_* or -1
    int32_t rho_holds;    // the output buffer holds the state after this many accepted steps (-1 = nothing valid)
    int32_t launches;     // speculative launches consumed by the state machine
    int32_t next_launch;  // status 0: id of the launch that must speculate next; status 2: id of the launch that applies
    int32_t predict;      // accepted-step count of the previous solve on this context: the iterate launch 0 of the
                          // NEXT solve writes speculatively (a branch predictor; results never depend on it)
    int32_t restart;      // 1: launch 0 met an argument outside the range of its in-range function cores (depth_kernels.hip CORE): its sums
                          // may differ from the standard functions', the solve is unfinished (status 0) and rsdsfm_depth_finish_dev runs it
                          // again from launch 0 with the standard functions
    double radius;
    double decrease_factor;
    double cost;  // cost of the current state
    double initial_cost;
    double cand[KMAX];
};
struct LmState : LmScal {
    double hist[kMaxIter];
};

// device-resident result header of a RANSAC run
struct RansacBest {
    int32_t best_trial;
    int32_t undecided;         // 1: written by a pick that ran BEHIND round 0 on flags that say the RANSAC is not over (hypotheses still running or unscored): everything enqueued behind it on speculation leaves at once
    int64_t num_inliers;       // from the score of the best trial
    int64_t num_inliers_scan;  // total of the compaction scan (must agree)
    double inlier_error;
    double hyp[8];  // w(3), v(3), k, status of the best trial
    int32_t lma_tie;  // analytic LM trajectory, guard (d): another trial has the winner's inlier count and an error sum within kLmaTie x count of the winner's (set by a pick launched with a tie margin; such a result is also `undecided`)
    int32_t shared_best;   // trials that have the winner's inlier count, the winner included (0: not evaluated -- incomplete scores, no inliers)
    int32_t _pad_shared;
    int32_t lazy_pending;  // count-only analytic pass: several trials share the best inlier count and some of them have no error sum yet (ransac_pick_kernel put them on the scoring pass's list; such a result is also `undecided`)
};
// ransac_pick_kernel, lazy error sums: scored[T] (2 = the trial has a count only), the scoring pass's list and its counter; scored == null: off
struct PickLazy {
    int* scored = nullptr;
    int* list = nullptr;
    int* list_count = nullptr;
};

// The scoring pass behind round 0 is enqueued ahead of the host's flag read while one of the context's last kScoreIdleLimit solves needed
// it.  (Following only the previous solve flip-flopped on DeepFlow-like data: one pair in ~25 has every hypothesis end where round 0
// scores it, and the pair after it then paid for a discarded final stage and refinement, ~280 us, against ~12 us for a pass that finds
// nothing to do.)
constexpr int kScoreIdleLimit = 4;
constexpr int kRansacBatch = 128;  // hypotheses per pixel pass (LDS accumulators: 128 x NS x 8 B = 18 KB)

struct Ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    std::string err;
    // scratch (device)
    double* d_partials = nullptr;  // [max_blocks][NS] per-workgroup partial sums, then [8][NS] group partials
    unsigned* d_tickets = nullptr; // 64 device words: arrival counters of the fused depth launch (variant 2)
    LmState* d_lm = nullptr;       // state machine of the depth solve
    LmState* h_lm = nullptr;       // pinned host copy
    int lm_issued_k = 0;           // depth_lm_kernel launches issued for the current solve
    int true_flow_exhaustive = 0;  // ground-truth flow search: 0 = interval-pruned from 96 scanlines on (default), 1 = every scanline for every pixel, 2 = pruned at any size
    int ransac_fused_base = 2;     // accepted steps after which most hypotheses of the previous solve ended: the iterate round 0 scores
    int ransac_score_idle = kScoreIdleLimit;  // consecutive solves (saturating) that did NOT need the separate scoring pass behind round 0; below the limit the pass is enqueued ahead of the host's flag read
    int refine_iters_hint = -1;    // slots (refine_kernels.hip; LM iterations + 1 as a rule) the context's previous refinement consumed (-1: none yet): length of the first chunk the host enqueues
    int ransac_standard_math = 0;  // > 0: that many of the context's next RANSACs run round 0 with the standard sqrt / reciprocal (set to 16 by a run that met an argument outside the range of the in-range cores and had to start over; ransac_lm_kernel CORE)
    int* d_core_flag = nullptr;    // device word the minimal solver stores its launch epoch in when an SVD operand left the range of the function cores (persistent: never a stale value)
    int core_epoch = 0;
    int ransac_math_mode = 0;      // rsdsfm_set_ransac_math: 0 = in-range cores with restart (default), 1 = always the standard functions
    int64_t ransac_restarts = 0;   // RANSAC runs of this context that started over for that reason (rsdsfm_ransac_restarts)
    // the T depth solves of a RANSAC on the analytic LM trajectory (lma_common.hpp, ransac_lma_kernels.hip): the default; a run whose guards
    // trip starts over on the iterate-by-iterate kernels (ransac_kernels.hip), and the context then stays on those for its next lma_hold solves
    int lm_arithmetic = 0;         // rsdsfm_set_lm_arithmetic: 0 = analytic trajectory with guards (default), 1 = always iterate by iterate
    int lma_hold = 0;              // > 0: that many of the context's next RANSACs run iterate by iterate (set to 16 by a run whose guards tripped; renewed by an iterate-by-iterate run that ends in a tie the analytic arithmetic could not break: noise-free data)
    bool lma_count_only_force = false;  // rsdsfm_set_lm_arithmetic(2): the frame solve's pass is count-only whatever the previous solves saw (tests)
    int64_t lma_count_only_runs = 0, lma_lazy_runs = 0;  // RANSACs whose pass was count-only, and those of them that had to fetch error sums (rsdsfm_lma_count_only)
    int lma_unique_run = 0;        // consecutive RANSACs of this context whose best inlier count no other trial shared (capped at 2): from 2 on the frame solve's pixel pass leaves the error sums out (count-only; a tie there is broken by the scoring pass and resets the run)
    int64_t lma_restarts = 0;      // RANSAC runs of this context that started over because a GLOBAL guard tripped: a tie, the count check (rsdsfm_lma_restarts)
    int64_t lma_handed_over = 0;   // RANSAC runs in which some hypothesis' own guard tripped and that hypothesis went on iterate by iterate
    int lma_last_guard = 0;        // bit set of the guards that tripped last (1 << reason: lma_common.hpp; 1 << 7: tie)
    // the dense depth solve's fast path on the analytic trajectory (depth_lma_kernels.hip): the solve's list of clamped pixels (two halves used
    // alternately: the follow-up launch of one solve zeroes the counter the next one starts from)
    int* d_lma_list = nullptr;
    int depth_lma_parity = 0;
    int lma_cand[2] = {2, 1};      // the two iterates (accepted steps) whose scores the analytic pixel pass fuses: FIXED (a function of nothing: the error sums of a hypothesis that ends at a fused iterate and of one that ends elsewhere come from different arithmetics)
    int ransac_spec_miss = 0;  // consecutive RANSACs (saturating at 2) whose speculated final stage did not count; below 2 the frame solve enqueues the refinement behind the speculated stage
    int frame_dense_hint = 1;  // frame solve: the previous frame kept every pixel (dense flow) -> set the RANSAC up for n = rows * cols without waiting for the count
    int lm_issued_d = 0;           // depth_lm_decide_kernel launches issued for the current solve
    int depth_epoch = 0;           // launch-0 counter of the dense depth solve: the value a thread of launch 0 stores into the context's range-flag word (d_tickets[41]) when an argument left the range of the function cores -- never a stale flag, no clearing pass
    int depth_standard_math = 0;   // > 0: that many of the context's next dense depth solves run launch 0 with the standard sqrt / reciprocal (set to 16 by a solve that had to start over)
    int64_t depth_restarts = 0;    // dense depth solves of this context that started over for that reason
    int depth_core_launch = 0;     // epoch of the current solve's launch 0 when it ran the function cores, else 0 (what its follow-up launch compares the flag word with)
    int prof_what = 0;             // rsdsfm_set_profiling: what the pending record brackets (0 = round 0 of the RANSAC's LM solves, 1 = launch 0 of a batched dense depth solve)
    // staging buffers for the host-pointer API (grown on demand)
    void* d_stage = nullptr;
    size_t stage_bytes = 0;
    // internal workspace of the multi-kernel pipelines (grown on demand)
    void* d_ws = nullptr;
    size_t ws_bytes = 0;
    void* d_frame = nullptr;   // frame buffers of rsdsfm_solve_frame_dev
    size_t frame_bytes = 0;
    void* h_pinned = nullptr;  // small pinned host buffer for result headers
    hipEvent_t ev_ready = nullptr;  // "value ready" event for stages whose host read does not have to wait for the whole stream
    size_t pinned_bytes = 0;
    int num_cus = 256;
    int ransac_k0 = 0;         // LM iterations round 0 of the hypothesis-batched depth solves speculates (rsdsfm_set_ransac_speculation): 2, KMAX, or 0 = follow the previous solve (2 when none of its hypotheses went beyond one accepted step)
    int ransac_not_one_step = -1; // last RANSAC: hypotheses that did not stop after at most one accepted step (-1: no solve yet)
    // opt-in profiling (rsdsfm_set_profiling): HIP events on the context's stream around the dominant kernel of the last RANSAC
    bool profile = false;
    hipEvent_t ev_prof[2] = {nullptr, nullptr};
    unsigned long long* d_clk_probe = nullptr;  // [4] {shader clocks, 100 MHz ticks} at the start and the end of one workgroup of the bracketed ransac_lm_kernel
    bool prof_pending = false;
    // opt-in trace of the joint refinement's LM iterations (rsdsfm_set_refine_trace): rows of kRefineTraceCols doubles, written by refine_decide_kernel
    double* d_refine_trace = nullptr;
    int refine_trace_rows = 0;
    int depth_variant = 0;  // 0 = register-staged depth_lm_kernel, 1 = LDS-DMA depth_lm_dma_kernel, 2 = launch 0 with the decision fused into its tail, 3 = separate decide kernel + follow-up launch (the pre-fusion fast path)
    // row-tiled refinement session (tiled_host.hip): buffers live in d_tile, not in the shared workspace
    void* d_tile = nullptr;
    size_t tile_bytes = 0;
    void* tile_session = nullptr;  // heap RefineBuffers of the open session
    int tile_np = 0;
    // persistent claim maps of the forward-splat kernels (rectify_kernels.hip: claim_map_acquire): 0 = back projection, 1 = depth image
    unsigned* d_claim[3] = {nullptr, nullptr, nullptr};  // 2 = the depth map of the solve (glue_kernels.hip: depth_claim_kernel)
    size_t claim_words[3] = {0, 0, 0};
    unsigned claim_epoch[3] = {0, 0, 0};
    void* dist = nullptr;  // dist_host.hip: communicator / transport + exchange buffers of the native tiled solve
    // frame_host.hip: the frame solve in flight (FrameRun), the second stream that runs the flatten beside the minimal solver, and the
    // lanes (contexts of their own, owned by this one) of the sequence solve
    void* frame_run = nullptr;
    hipStream_t aux_stream = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_seq = nullptr;
    int frame_side_flatten = 3;  // rsdsfm_set_frame_side_flatten: where a dense frame's flatten runs -- 0 first, 1 on aux_stream beside the minimal solver, 2 behind it, 3 INSIDE the solver's launch
    unsigned long long* d_flat_counters = nullptr;  // the two counters of minimal9_flatten_kernel (zero between launches)
    int seq_lanes = 0;               // rsdsfm_set_sequence_lanes (0 = kSequenceLanesDefault)
    // where the refinement's single-workgroup stage runs (rsdsfm_set_refine_stage): 0 = automatic -- in the prologue of the next slot's pass
    // (one launch per slot: the shortest single solve), except while the solve is not alone on its GPU -- several pairs of a sequence in
    // flight, or frame solves of other contexts (frames_in_flight) --, where the stage gets a launch of its own behind every pass (the
    // prologue occupies the WHOLE chip for its ~8 us, a single workgroup leaves it to the other solves' kernels); 1 = always in the
    // prologue; 2 = always a launch of its own.  Never a result.
    int refine_stage_mode = 0;
    bool refine_stage_separate = false;  // (what the automatic mode resolves to for the solve in flight)
    // the joint refinement on radius-factorised Schur sums (refine_rf_kernels.hip; the default while lm_arithmetic == 0): solves that ran on
    // it, solves one of whose guards sent them back to the iterate-by-iterate slot kernels, the guard that tripped last (RfGuard), and the
    // reduced systems solved again from stored sums (rejected / invalid steps) -- rsdsfm_refine_restarts
    int refine_arithmetic = 0;  // rsdsfm_set_refine_arithmetic: 0 = radius-factorised (while lm_arithmetic == 0), 1 = the iterate-by-iterate slot kernels
    int64_t refine_rf_runs = 0, refine_rf_restarts = 0, refine_rf_resolves = 0;
    int refine_rf_last_guard = 0;
    // What the last rsdsfm_ransac (host-pointer form) left on the device, so that rsdsfm_refine_from_ransac can start from it instead of
    // uploading the same 37 + 15 MB again (reference main.cc:447-457 hands nonLinearRefinement the RansacValues and the flow that ransac has
    // just seen): valid while `tag` is what the caller presents and no other host-pointer call has taken the staging buffer since
    struct RansacCache {
        uint64_t tag = 0, stage_gen = 0;
        const double *d_u = nullptr, *d_inl = nullptr, *d_alpha = nullptr, *d_alpha_k = nullptr;
        const int64_t* d_idx = nullptr;
        int64_t n = 0, m = 0;
        const void* h_u = nullptr;   // the caller's u array (identity + spot check decide whether `flow` is that array)
        double u_probe[16] = {0};    // u[probe positions] as uploaded
        double inl_probe[16] = {0};  // inliers[probe positions] as downloaded
    } ransac_cache;
    uint64_t stage_gen = 0, ransac_tag_counter = 0;  // stage_gen: bumped by every ensure_stage
    int64_t refine_cache_hits = 0;
    std::vector<rsdsfm_ctx*> lanes;
};
constexpr int kSequenceLanesDefault = 3;  // measured: 1 / 2 / 3 / 4 / 6 / 8 lanes = 0.91 / 1.19 / 1.31 / 1.21 / 1.31 / 1.27 Gpix/s at 1280x720, T = 50
void dist_release(Ctx* c);
void dist_reset_hold(Ctx* c);
void frame_release(Ctx* c);

constexpr int kDepthBlock = 256;
constexpr int kDepthMaxBlocks = 512;
constexpr int kDecideBlock = 256;
constexpr int kDepthBatchMax = 8;  // independent solves per batched launch (descriptors travel in the kernel arguments)

#define RSDSFM_HIP_CHECK(ctx, expr)                                                            \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess) {                                                                \
            (ctx)->err = std::string(#expr) + ": " + hipGetErrorString(_e);                    \
            return RSDSFM_ERR_HIP;                                                             \
        }                                                                                      \
    } while (0)

// Scoped device guard: hipMalloc / hipHostMalloc / kernel launches follow the calling thread's CURRENT device, not the stream's.
// Every extern "C" entry point therefore makes the context's device current (a new host thread starts on device 0; a caller
// may alternate contexts of different GPUs in one thread) and restores the caller's device when it returns.
struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    explicit DeviceGuard(const Ctx* c) {
        if (hipGetDevice(&prev) == hipSuccess && prev != c->device) switched = hipSetDevice(c->device) == hipSuccess;
    }
    ~DeviceGuard() {
        if (switched) (void)hipSetDevice(prev);
    }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
};

int fail(Ctx* c, int code, const char* msg);
int ensure_stage(Ctx* c, size_t bytes);
// host_xfer.hip: the host-pointer boundary's transfers through a ring of pinned chunks filled / drained by a small thread pool, on the context's
// stream.  xfer_h2d returns once the caller's array has been read, xfer_d2h once the caller's array holds the data.
struct XferItem {
    void* host;
    const void* dev;
    size_t bytes;
};
struct XferUp {
    void* dev;
    const void* host;
    size_t bytes;
};
int xfer_h2d(Ctx* c, void* d_dst, const void* h_src, size_t bytes);
int xfer_h2d_many(Ctx* c, const XferUp* items, int count);
int xfer_d2h(Ctx* c, void* h_dst, const void* d_src, size_t bytes);
int xfer_d2h_many(Ctx* c, const XferItem* items, int count);
// RSDSFM_XFER_TRACE=1: time stamps of a host-pointer call, printed to stderr by xfer_trace_dump at its end
bool xfer_trace_on();
void xfer_trace(const char* label);
void xfer_trace_span(const char* label, double us);
void xfer_trace_dump(const char* what);
int ensure_ws(Ctx* c, size_t bytes);
int sync_stream(Ctx* c, const char* where);  // hipStreamSynchronize(c->stream), with an opt-in watchdog (RSDSFM_SYNC_WATCHDOG_S; capi.hip)
void set_sync_dumper(std::function<void(Ctx*)> f);
int ensure_pinned(Ctx* c, size_t bytes);

// bump allocator over a device arena (256-byte aligned slices)
struct Arena {
    char* base;
    size_t off = 0;
    explicit Arena(void* b) : base(static_cast<char*>(b)) {}
    template <class T>
    T* take(size_t count) {
        T* p = reinterpret_cast<T*>(base + off);
        off += need(count * sizeof(T));
        return p;
    }
    static size_t need(size_t bytes) { return (bytes + 255) & ~(size_t)255; }
};


// depth_kernels.hip
int depth_closed_form_launch(Ctx* c, const double* q, const double* u, const double* a, const double* ak,
                             int64_t n, const Pose& pose, double* rho);
int depth_lm_launch(Ctx* c, const double* q, const double* u, const double* a, const double* ak, int64_t n,
                    const Pose& pose, double* rho, int launch_id, bool core = false);
int depth_lm_decide_launch(Ctx* c, int64_t n, int launch_id);
int depth_lm_decide_apply_launch(Ctx* c, const double* q, const double* u, const double* a, const double* ak, int64_t n,
                                 const Pose& pose, double* rho);
int depth_lm_batch_launch(Ctx* const* cs, int count, const double* const* q, const double* const* u, const double* const* a,
                          const double* const* ak, const int64_t* n, const Pose* poses, double* const* rho, int launch0_only);
int depth_lm_fused_launch(Ctx* c, const double* q, const double* u, const double* a, const double* ak, int64_t n,
                          const Pose& pose, double* rho);
int depth_lm_reduce_launch(Ctx* c, int64_t n, double* d_row);
void fill_lm_summary(const LmState& st, rsdsfm_lm_summary* s);  // capi.hip
int depth_lm_decide_rows_launch(Ctx* c, const double* d_rows, int nrows, int64_t n_total, int launch_id);
// depth_lma_kernels.hip: the fast path on the analytic LM trajectory (launch 0 + the follow-up launch)
bool depth_lma_allowed(const Ctx* c, int64_t n);
int depth_lma_batch_launch(Ctx* const* cs, int count, const double* const* q, const double* const* u, const double* const* a,
                           const double* const* ak, const int64_t* n, const Pose* poses, double* const* rho);
int depth_lma_row_doubles();
int depth_lma_shard_launch(Ctx* c, const double* q, const double* u, const double* a, const double* ak, int64_t n, const Pose& pose, double* rho, double* d_row);
int depth_lma_shard_finish_launch(Ctx* c, const double* q, const double* u, const double* a, const double* ak, int64_t n, const Pose& pose, double* rho,
                                  const double* d_rows_all, int nranks, int64_t n_total);

}  // namespace rsdsfm

namespace rsdsfm {
// glue_kernels.hip
int alpha_launch(Ctx* c, const double* flow_px, int64_t n, double h, double gamma, double* alpha);
int alpha_k_launch(Ctx* c, const double* q_px, const double* flow_px, int64_t n, double h, double gamma, double* alpha_k);
// depth map stage (+ optionally the per-scanline pose table behind it, one synchronisation for both)
int depth_map_device(Ctx* c, double* d_inl, int64_t m, double v_inout[3], double fx, double fy, double cx, double cy, int32_t rows, int32_t cols,
                     double* d_depth_map, int32_t* d_xs, int32_t* d_ys, int* flipped, const double* w_or_null, double k, double gamma,
                     double* d_R_rows9, double* d_t_rows3);
// optional output of zsum_decide_kernel (frame solve): the per-scanline pose table of (v', w, k) with (w, k) read from wk_dev
struct PoseTableOut {
    double* R = nullptr;
    double* t = nullptr;
    int rows = 0;
    double gamma = 0.0;
    const double* wk_dev = nullptr;
};
int pose_table_launch(Ctx* c, const Pose& pose, double gamma, int rows, double* R, double* t, const double* v_dev = nullptr,
                      const double* wk_dev = nullptr);
int64_t flatten_cells(int rows, int cols);
int flatten_launch(Ctx* c, const double* d_img, int rows, int cols, int col0, double fx, double fy, double cx, double cy,
                   double gamma, double thr, double* d_q, double* d_u, double* d_alpha, double* d_alpha_k, int64_t* d_counts,
                   int64_t* d_offsets, int64_t* d_total, hipEvent_t total_ready = nullptr);
int depth_map_launch(Ctx* c, double* d_inl, int64_t m, const double v[3], double fx, double fy, double cx, double cy, int rows,
                     int cols, double* d_depth_map, int32_t* d_xs, int32_t* d_ys, double* d_header, double* d_partials,
                     double* h_header = nullptr, const double* v_dev = nullptr, const int64_t* m_dev = nullptr,
                     const PoseTableOut* pt = nullptr);
int depth_map_slab_launch(Ctx* c, double* d_inl, int64_t m, const double* d_zsums, int nz, int64_t m_total, const double v[3],
                          double fx, double fy, double cx, double cy, int rows, int col0, int ncols, double* d_depth_map,
                          int32_t* d_xs, int32_t* d_ys, double* d_header, double* h_header = nullptr, const double* v_dev = nullptr,
                          const int64_t* m_dev = nullptr, const PoseTableOut* pt = nullptr);
// persistent epoch-tagged claim map `which` of the context (rectify_kernels.hip)
int claim_map_acquire(Ctx* c, int which, size_t npix, unsigned** map, unsigned* tag, unsigned* mask);
int claim_map_reserve(Ctx* c, int which, size_t words);
int zsum_row_launch(Ctx* c, const double* d_inl, int64_t m, double* d_partials, double* d_out);
}  // namespace rsdsfm

namespace rsdsfm {
// minimal9_kernels.hip : hyp_out [T][8] = w(3), v(3), k, status
// Minimal9Direct: the sampled points are formed straight from the flow IMAGE instead of the flattened arrays -- valid when the
// flatten keeps every pixel (dense flow), so that point index i is pixel (column i / rows, row i % rows) of the column-major scan
// (main.cc:398-444); the expressions are the flatten's own (device_math.hpp: flatten_point), so the hypotheses have the same bits.
// The frame solve uses it to run the minimal solver BESIDE the flatten instead of behind it.
struct Minimal9Direct {
    const double* img = nullptr;  // row-major [rows][cols][2] flow image (DEVICE)
    int rows = 0, cols = 0, alpha_ones = 0;  // alpha_ones: the global-shutter override alpha = alpha * 0 + 1 (main.cc:441-444)
    double fx = 0, fy = 0, cx = 0, cy = 0, gamma = 0;
    // (independent of img) non-null: the wave-per-hypothesis SVD runs its rotations through the in-range cores of division, reciprocal
    // and square root (device_math.hpp) and stores core_epoch here when an operand was out of range (the caller then has the
    // hypotheses computed again without it)
    int* core_flag = nullptr;
    int core_epoch = 0;
    // diagnostics (rsdsfm_minimal9_probe_dev; wave-per-hypothesis solver only): per hypothesis {sweeps of the 9x9 Jacobi SVD, rotations it
    // performed, shader clocks the SVD took, shader clocks of the whole hypothesis}
    double* probe = nullptr;
};
int minimal9_launch(Ctx* c, const double* q, const double* u, const double* alpha, const double* alpha_k,
                    const int32_t* samples, int T, int use_alpha_k, int k_sign_mode, double* hyp_out, void* zero_begin = nullptr,
                    size_t zero_bytes = 0, const Minimal9Direct* direct = nullptr);
// the solver (direct mode, one wave per hypothesis) and the flatten of a DENSE frame as two roles of ONE launch (minimal9_kernels.hip);
// d_counters: two zero-initialised device words the launch leaves zero again; *total_out (host-mapped) = kept pixels
struct DenseFlatten {
    double thr = 0.0;
    double *d_q = nullptr, *d_u = nullptr, *d_alpha = nullptr, *d_alpha_k = nullptr;
    unsigned long long* d_counters = nullptr;
    int64_t* total_out = nullptr;
};
int minimal9_flatten_launch(Ctx* c, const int32_t* samples, int T, int use_alpha_k, int k_sign_mode, double* hyp_out, void* zero_begin, size_t zero_bytes,
                            const Minimal9Direct& direct, double thr, double* d_q, double* d_u, double* d_alpha, double* d_alpha_k,
                            unsigned long long* d_counters, int64_t* total_out);
}  // namespace rsdsfm

namespace rsdsfm {
// ransac_kernels.hip
int ransac_pixel_grid(const Ctx* c, int64_t n);
int ransac_lm_partials_doubles(const Ctx* c, int64_t n, int batch);
int ransac_lm_round_launch(Ctx* c, const double* q, const double* u, const double* a, const double* ak, int64_t n,
                           const double* hyp, int T, LmState* states, double* partials, int* flags, int* scored,
                           double* trial_count, double* trial_err, int round, double tol, int k0, int fused_base, bool core_math,
                           const int* m9_core_flag = nullptr, int m9_core_epoch = 0, int* unscored_list = nullptr);
int ransac_score_launch(Ctx* c, const double* q, const double* u, const double* a, const double* ak, int64_t n,
                        const double* hyp, int T, const LmState* states, int depth_mode, double tol, const int* scored,
                        double* partials, double* trial_count, double* trial_err, const int* unscored_list = nullptr, const int* unscored_count = nullptr);
int ransac_pick_launch(Ctx* c, const double* trial_count, const double* trial_err, int T, const double* hyp, RansacBest* best,
                       RansacBest* best_host = nullptr, const int* d_flags = nullptr, int* h_flags = nullptr, int scored_ahead = 0,
                       const double* cnt_rt = nullptr, int cnt_stride = 0, int nranks = 0, int64_t* m_all = nullptr, double tie_margin = 0.0,
                       PickLazy lazy = PickLazy());
int ransac_final_launch(Ctx* c, const double* q, const double* u, const double* a, const double* ak, int64_t n,
                        RansacBest* best, const LmState* states, int depth_mode, double tol, double* rho, uint8_t* mask,
                        int64_t* block_counts, int64_t* block_offsets, int64_t* inlier_idx, double* inliers,
                        double* out_alpha, double* out_alpha_k, RansacBest* best_host = nullptr);
// row-tiled stages
int ransac_rows_doubles();
int ransac_lm_rows_launch(Ctx* c, const double* q, const double* u, const double* a, const double* ak, int64_t n,
                          const double* hyp, int T, const LmState* states, double* partials, int round, double tol, double* rows,
                          int* core_flags = nullptr, const int* m9_flag = nullptr, int m9_epoch = 0);
int ransac_rows_payload_doubles(int T, bool core_trailer);
int ransac_decide_rows_launch(Ctx* c, const double* rows_all, int nranks, int T, LmState* states, int64_t n_total, int round,
                              int* flags, int* scored, double* trial_count, double* trial_err, bool core_trailer = false,
                              double* cnt_rt = nullptr, int cnt_stride = 0);  // cnt_rt[rank][hypothesis]: the ranks' shares of the fused inlier counts
int ransac_score_rows_launch(Ctx* c, const double* q, const double* u, const double* a, const double* ak, int64_t n,
                             const double* hyp, int T, const LmState* states, int depth_mode, double tol, const int* scored,
                             double* partials, double* rows);
int ransac_score_merge_launch(Ctx* c, const double* rows_all, int nranks, int T, const int* scored, double* trial_count,
                              double* trial_err, double* cnt_rt = nullptr, int cnt_stride = 0);
void sample_indices(int64_t n, int T, uint64_t seed, int32_t* out);
// ransac_lma_kernels.hip: the depth solves of a hypothesis batch on the analytic LM trajectory
int64_t ransac_lma_partials_doubles(const Ctx* c, int64_t n, int batch);
size_t ransac_lma_list_ints(int batch);
int ransac_lma_launch(Ctx* c, const double* q, const double* u, const double* a, const double* ak, int64_t n, const double* hyp, int T,
                      LmState* states, double* partials, int* flags, int* scored, double* trial_count, double* trial_err, double tol,
                      const int* cand_steps, int ncand, int* irr_count, int* irr_list, int* unscored_list, int* guard_word,
                      const int* m9_core_flag = nullptr, int m9_core_epoch = 0, bool count_only = false);
int ransac_lma_rows_doubles();
int ransac_lma_rows_launch(Ctx* c, const double* q, const double* u, const double* a, const double* ak, int64_t n, const double* hyp, int T,
                           double* partials, double tol, const int* cand_steps, int ncand, int* irr_count, int* irr_list, double* rows);
int ransac_lma_decide_rows_launch(Ctx* c, const double* rows_all, int nranks, int64_t rank_stride, int T, int64_t n_total, const double* hyp,
                                  LmState* states, int* flags, int* scored, double* trial_count, double* trial_err, double tol, const int* cand_steps,
                                  int ncand, int* unscored_list, int* guard_word, double* cnt_rt, int cnt_stride, const int* m9_core_flag = nullptr,
                                  int m9_core_epoch = 0);
}  // namespace rsdsfm

namespace rsdsfm {
// device-resident state of the joint refinement (refine_kernels.hip)
struct RefineState {
    int32_t np, cur, iteration, invalid_run, num_successful, num_unsuccessful, termination, solve_ok;
    double p[7], pc[7], sp[7], yp[7];
    double radius, decrease_factor, cost, initial_cost, x_norm, gmax, stepsq_p;
    // inlier count and logical grid of the streaming passes when the host does not know them at enqueue time (the frame solve
    // enqueues the refinement before it has read the RANSAC result: refine_state_from_best_kernel); kernels launched with
    // m < 0 / nblocks < 0 take these
    int64_t m;
    int32_t grid;
    // column-tiled solve, one exchange per LM iteration (refine slot kernels): 1 = the next slot is a plain Schur pass at the current state
    // (the first iteration, or the sums speculated with the previous back-substitution do not apply); slots = slots consumed so far
    int32_t need_schur;
    // sum of z = 1 / rho over the inliers at the CURRENT state (what refine_finish_kernel writes), kept by the decide stages when the passes
    // are asked for it (RefineBuffers::want_zsum: the column-tiled solve, whose mean-z sign test main.cc:466-472 then needs no exchange of its own)
    double zsum;
    int32_t slots;
    // consecutive decisions (capped at 2) whose step was not "accepted with the radius the slot's pass speculated on": the back-substitution pass
    // speculates while this is < 2 (DeepFlow-like data: four steps in five apply, misses are isolated; acceleration mode: runs of rejected
    // steps and of qualities 0.65 .. 0.93, where speculating would add a wasted Schur evaluation to every pass)
    int32_t spec_miss_run;
    // a slot's pass has run and its single-workgroup stage has not: the next slot kernel's prologue (or the stage kernel behind the last
    // pass of a chunk) reduces that pass's rows and runs the decision / reduced solve before anything else
    int32_t pending_apply;
    // radius-factorised path (refine_rf_kernels.hip): 1 while the solve runs on it; rf_guard = the guard that sent it back to the iterate-by-
    // iterate slot kernels (termination == kTermRestartExact then; see RfGuard)
    int32_t rf;
    double dp[7];  // that path's parameter step in UNSCALED parameters (pc = p + dp)
    double p0[7];  // the start parameters (the Jacobi scales of the rho columns are taken at them: 1 / (1 + |J_rho(x0)|))
    int32_t rf_guard, rf_resolves;  // rf_resolves: reduced systems solved again from stored sums (rejected / invalid steps: no pass of their own)
};
// RefineState::termination of a radius-factorised refinement one of whose guards tripped: the host runs the solve again on the iterate-by-iterate kernels
constexpr int kTermRestartExact = 64;
struct RefineBuffers {
    const double* flow;  // 2 x n_flow
    int64_t n_flow, m;
    bool m_on_device = false;  // m and the logical grid live in RefineState (refine_state_from_best_kernel); `m` is then an upper bound
    const double* inl;  // 3 x m
    const double* alpha;
    const double* alpha_k;
    const int64_t* inlier_idx;
    int flow_index_mode;
    RefineState* state;
    double* uu;     // 4 x m packed per-inlier constants (x, y, u_x, u_y): one 32-byte record per inlier, written by the init pass
    double* beta;   // m: beta_i = 2 (alpha_i + k alpha_k,i) / (2 + k), precomputed when k is not refined (np == 6)
    double* rho_a;  // the two rho buffers (swap on acceptance)
    double* rho_b;
    double* srho;
    double* partials;
    int* bad_index;
    RefineState* state_host = nullptr;  // frame solve: host-mapped copy of the state (+ flag) written by refine_finish_kernel
    double* zpartials = nullptr;  // frame solve: refine_finish_kernel also leaves its per-workgroup sums of z here (refine_finish_grid entries)
    bool want_zsum = false;       // the streaming passes also sum 1 / rho (one division per inlier and pass) into the last slot of their rows -> RefineState::zsum
};
size_t ransac_pinned_bytes(int T);
// see ransac_device (ransac_host.hip): caller's work enqueued behind the speculated final stage, given the device-resident result
typedef std::function<int(const RansacBest*)> RansacSpecTail;
// one RANSAC in flight (ransac_host.hip: ransac_begin / ransac_finish): everything the resumable state machine keeps between its steps
struct RansacRun {
    int pc = 0;
    // problem
    const double *d_q = nullptr, *d_u = nullptr, *d_a = nullptr, *d_ak = nullptr;
    int64_t n = 0;
    int T = 0, Tn = 1, batch = 1, depth_mode = 0, use_alpha_k = 0, k_sign_mode = 0, k0 = 0, fused_base = 0;
    double tol = 0.0;
    rsdsfm_ransac_out* out = nullptr;
    const RansacSpecTail* spec_tail = nullptr;
    bool* spec_tail_held = nullptr;
    int core_epoch = 0;      // the minimal solver's launch epoch of this run (0: it ran without the cores)
    int* d_unscored = nullptr;  // [batch] indices (within the batch) of the hypotheses that finished without a fused score, in order of arrival; their number is d_flags[1]
    bool core_math = true;   // round 0 through the in-range function cores (ransac_lm_kernel CORE); false after a restart
    bool analytic = false;   // the depth solves on the analytic LM trajectory (ransac_lma_kernels.hip); false after a guard tripped
    bool lma_restarted = false;
    bool count_only = false;  // analytic pass without error sums; ransac_pick_kernel asks for the exact sums of the trials that share the best count
    int lazy_rounds = 0;
    int shared_best = 0;      // RansacBest::shared_best of the definitive pick
    double tie_margin = 0.0;  // what the picks are launched with (ransac_pick_kernel)
    int lma_guard = 0;       // guards that tripped (bit set)
    bool lma_tie_seen = false;  // the final pick of an iterate-by-iterate run saw a tie the analytic arithmetic could not have broken
    int lma_cand[2] = {2, 1}, lma_cand_next[2] = {0, 0};
    int *d_irr_count = nullptr, *d_irr_list = nullptr;
    bool restarted = false;
    bool tail_ahead = true;  // enqueue the caller's tail behind the SPECULATED final stage (otherwise only behind the definitive one)
    const Minimal9Direct* direct = nullptr;
    const DenseFlatten* dense = nullptr;  // with `direct`: the flatten of the dense frame rides in the solver's launch
    const std::function<int()>* after_minimal9 = nullptr;
    std::vector<int32_t> samples;
    // workspace (device) and pinned host views
    double *d_hyp = nullptr, *d_partials = nullptr, *d_tcount = nullptr, *d_terr = nullptr, *d_rho = nullptr;
    char* zero_begin = nullptr;
    size_t zero_bytes = 0;
    LmState* d_states = nullptr;
    int *d_scored = nullptr, *d_flags = nullptr;
    RansacBest* d_best = nullptr;
    int64_t *d_bcounts = nullptr, *d_boffs = nullptr;
    uint8_t* d_mask = nullptr;
    RansacBest* h_best = nullptr;
    int* h_running = nullptr;
    double *h_tcount = nullptr, *h_terr = nullptr, *h_hyp = nullptr;
    LmState* h_states = nullptr;
    int32_t* h_samples_pinned = nullptr;
    // progress
    int b0 = 0, B = 0, round = 0;
    bool need_score = true, final_done = false, spec_scored = false, tail_enqueued = false, spec_final = false;
    // scheduling hints this run leaves for the context's next solve (ransac_commit_hints)
    int not_one_step = 0, fused_base_next = 0, score_hint_next = -1;
    bool hints_ready = false;
};
int ransac_begin(Ctx* c, const double* d_q, const double* d_u, const double* d_a, const double* d_ak, int64_t n, int use_alpha_k, int T,
                 double tol, const int32_t* h_samples, uint64_t seed, int depth_mode, int k_sign_mode, rsdsfm_ransac_out* out,
                 const RansacSpecTail* spec_tail, bool* spec_tail_held, RansacRun* run, const Minimal9Direct* direct,
                 const std::function<int()>* after_minimal9, const DenseFlatten* dense = nullptr, bool tail_ahead = true, bool count_only = false);
int ransac_finish(Ctx* c, RansacRun* run);
void ransac_commit_hints(Ctx* c, const RansacRun& run);
int flatten_enqueue(Ctx* c, const double* d_img, int32_t rows, int32_t cols, double fx, double fy, double cx, double cy, double gamma,
                    double thr, double* d_q, double* d_u, double* d_alpha, double* d_alpha_k, int64_t* h_total);
// see refine_device (refine_host.hip): caller's work enqueued behind the refinement's output pass, given the device-resident state
typedef std::function<int(const RefineBuffers&)> RefineTail;
int refine_finish_grid(const Ctx* c, const RefineBuffers& B);
// one refinement in flight (refine_host.hip: refine_begin / refine_poll)
struct RefineRun {
    RefineBuffers B;
    int np = 6, launched = 0, chunk = 5, hint_prev = -1;
    bool rf = false;  // on the radius-factorised path (refine_rf_kernels.hip); false: the iterate-by-iterate slot kernels
    double* d_inl_out = nullptr;
    const RefineTail* tail = nullptr;
    // pinned host copy of the state (+ bad-index flag).  prefetch: every chunk also enqueues its read-back, and the first refine_poll
    // trusts that the CALLER has synchronised the stream since (the frame solve: the RANSAC's own wait) instead of waiting again
    RefineState* hs = nullptr;
    bool prefetch = false, prefetched = false;
    bool tail_done = false;  // the caller's tail was enqueued behind the last output pass (skipped for chunks that cannot be the last)
};
// the last kPinnedTail bytes of the context's pinned block are reserved for the frame solve (refinement state read-back, depth-map header)
constexpr size_t kPinnedTail = 1024;
size_t refine_workspace_bytes(const Ctx* c, int64_t m, bool m_on_device);
// exact: on the iterate-by-iterate slot kernels whatever the context's arithmetic (a solve that is run again behind a tripped guard)
int refine_begin(Ctx* c, const double* d_flow, int64_t n_flow, int64_t m, const double* d_inl, const double* d_alpha,
                 const double* d_alpha_k, const int64_t* d_inlier_idx, const double v_in[3], const double w_in[3], double k_in,
                 int const_acceleration, int flow_index_mode, double* d_inl_out, const RefineTail* tail, const RansacBest* d_best,
                 void* ws_base, RefineRun* run, RefineState* hs_prefetch, double* d_zpartials, bool exact = false);
// refine_poll's return value when a guard of the radius-factorised path tripped: nothing was written to the outputs; run the solve again with exact = true
constexpr int kRcRefineRestartExact = 1;
int refine_device(Ctx* c, const double* d_flow, int64_t n_flow, int64_t m, const double* d_inl, const double* d_alpha,
                  const double* d_alpha_k, const int64_t* d_inlier_idx, const double v_in[3], const double w_in[3], double k_in,
                  int const_acceleration, int flow_index_mode, double* d_inl_out, double v_out[3], double w_out[3], double* k_out,
                  rsdsfm_lm_summary* summary, const RefineTail* tail, double* d_zpartials, bool exact = false);
int refine_enqueue_chunk(Ctx* c, RefineRun* run);
int frames_in_flight(const Ctx* c);  // frame solves between begin and the end of finish on the context's device, all contexts of the process (frame_host.hip)
int refine_poll(Ctx* c, RefineRun* run, double v_out[3], double w_out[3], double* k_out, rsdsfm_lm_summary* summary);
int refine_partials_doubles(const Ctx* c, int64_t m);
int refine_partials_doubles_cap(const Ctx* c);
int refine_state_from_best_launch(Ctx* c, const RansacBest* d_best, const RefineBuffers& B, int np);
// start of a refinement: NaN-fill the opt-in iteration trace (rsdsfm_set_refine_trace), enqueued on the context's stream
inline int refine_trace_reset(Ctx* c) {
    if (!c->d_refine_trace) return RSDSFM_OK;
    return hipMemsetAsync(c->d_refine_trace, 0xFF, (size_t)c->refine_trace_rows * kRefineTraceCols * sizeof(double), c->stream) == hipSuccess ? RSDSFM_OK : RSDSFM_ERR_HIP;
}
int refine_init_launch(Ctx* c, const RefineBuffers& B, int np);
// refine_rf_kernels.hip: slot g (global index within the solve; 0 = the first pass: iteration zero + the Schur sums of iteration 1) of a
// chunk that started at g_first; the stage behind slot g on its own; the shard's row [sums | list] of the column-tiled solve
int refine_rf_pass_launch(Ctx* c, const RefineBuffers& B, int np, int g, int g_first, const double* rows_all_prev, int nranks, int64_t m_total,
                          const int64_t* m_total_dev, bool publish = false);
int refine_rf_apply_launch(Ctx* c, const RefineBuffers& B, int np, int g, bool to_published, const double* rows_all, int nranks, int64_t m_total,
                           const int64_t* m_total_dev);
int refine_rf_row_launch(Ctx* c, const RefineBuffers& B, int np, int g, double* row);
int refine_rf_row_doubles(int np);
void refine_rf_debug_dump(Ctx* c, const RefineBuffers& B);
int refine_rf_read_stamps(Ctx* c, unsigned long long out[16]);
int refine_rf_extra_doubles();
int refine_partials_half_doubles(const Ctx* c);
int refine_state_doubles();
// the three list counters of the radius-factorised path live behind the state and its bad-index flag (zeroed with them)
inline int* refine_rf_counters(const RefineBuffers& B) { return B.bad_index + 4; }
constexpr size_t kRefineStateBlockTail = 32;  // bytes behind RefineState that travel with it: bad-index flag (+0), list counters (+16, +20, +24)
// slot `j` of a chunk of `chunk` slots (refine_kernels.hip): its pass carries the single-workgroup stage of slot j - 1 in its prologue; the
// last one is followed by that stage on its own, which leaves the state in B.state for the output pass, the caller's tail and the host
int refine_iter_launch(Ctx* c, const RefineBuffers& B, int np, int j, int chunk);
int refine_finish_launch(Ctx* c, const RefineBuffers& B, double* inl_out);
// row-tiled stages: stage 0 = iteration-zero sums, 1 = Schur sums, 2 = back-substitution sums
int refine_stage_row_doubles(int np, int stage);
// one exchange per LM iteration (dist_host.hip): a slot = one streaming pass -> one row of refine_slot_row_doubles(np) doubles per rank
// -> (all-gather) -> refine_slot_apply_launch on every rank
int refine_slot_row_doubles(int np);
int refine_slot_partials_doubles(const Ctx* c, int64_t m);
// column-tiled solve, slot j of a chunk: the pass (prologue: the stage of slot j - 1 on the gathered rows `rows_all_prev`) + the shard's row;
// refine_slot_apply_launch: the stage behind the last exchange of a chunk (-> B.state)
int refine_slot_rows_launch(Ctx* c, const RefineBuffers& B, int np, double* row, int j, const double* rows_all_prev, int nranks);
int refine_slot_apply_launch(Ctx* c, const RefineBuffers& B, int np, const double* rows_all, int nranks, int chunk);
int refine_stage_rows_launch(Ctx* c, const RefineBuffers& B, int np, int stage, double* row);
int refine_stage_apply_launch(Ctx* c, const RefineBuffers& B, int np, int stage, const double* rows_all, int nranks, int64_t m_total,
                              const int64_t* m_total_dev = nullptr);
}  // namespace rsdsfm

namespace rsdsfm {
// rectify_kernels.hip
int back_project_launch(Ctx* c, const unsigned char* d_img, const double* d_depth_cm, const double* d_R, const double* d_t, double fx,
                        double fy, double cx, double cy, int rows, int cols, int mode, int q5_mode, unsigned char* d_gs, float* d_c3d);
int interpolate_cracky_launch(Ctx* c, const unsigned char* d_in, int rows, int cols, int offset, unsigned char* d_out);
int rectify_frame_launch(Ctx* c, const double* d_inl, int64_t m, const unsigned char* d_img, const double* d_depth_cm, const double* d_R,
                         const double* d_t, double fx, double fy, double cx, double cy, int rows, int cols, int mode, int q5_mode, int offset,
                         unsigned char* d_preview, unsigned char* d_gs, float* d_c3d, unsigned char* d_fixed, double* d_partials);
int depth_preview_launch(Ctx* c, const double* d_inl, int64_t m, double fx, double fy, double cx, double cy, int rows, int cols,
                         unsigned char* d_out, double* d_partials);
}  // namespace rsdsfm

// the opaque handle of the C ABI
struct rsdsfm_ctx {
    rsdsfm::Ctx c;
};
