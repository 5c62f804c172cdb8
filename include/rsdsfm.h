/*
 * rsdsfm.h -- C ABI of the MI355X-native rolling-shutter differential-SfM solver.
 *
 * Drop-in boundary for the hot path of ThomasZiegler/RS-aware-differential-SfM: every entry point
 * replaces one free function of the reference's `minimal::` / `nonlinear_refinement::` namespaces (or
 * one block of caller-side glue) and cites it (paths relative to the reference's src/).  The reference
 * has no FFI layer of its own; a maintainer binds these through the header-only C++ mirror in
 * rs-aware-differential-sfm_amd/host/ (see INTEGRATION.md).
 *
 * Conventions: all floating data IEEE fp64; a "2xN" array is N interleaved (x, y) pairs (Eigen
 * column-major Array2Xd), "3xN" is N interleaved triples.  Functions return RSDSFM_OK (0) or a negative
 * error code; rsdsfm_last_error() gives text.  A context binds one HIP device + one stream and is
 * single-owner (one per GPU / thread).  There is NO CPU fallback: without a usable HIP device
 * rsdsfm_create fails.
 *
 * Two families:
 *   host-pointer API  (rsdsfm_xxx)      caller-owned HOST buffers, synchronous -- the reference's semantics.
 *   device API        (rsdsfm_xxx_dev)  caller-owned DEVICE buffers (16-byte aligned), asynchronous on the
 *                                       context's stream -- what bench.py and the multi-GPU driver use.
 */
#ifndef RSDSFM_H
#define RSDSFM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RSDSFM_OK 0
#define RSDSFM_ERR_INVALID (-1)     /* bad argument (null pointer, n < 9, misaligned device pointer ...) */
#define RSDSFM_ERR_HIP (-2)         /* HIP runtime error */
#define RSDSFM_ERR_NO_DEVICE (-3)   /* no usable gfx950 device */
#define RSDSFM_ERR_NUMERIC (-4)     /* solver failure (singular system, no real k, LM failure) */
#define RSDSFM_ERR_PENDING (-5)     /* device LM state machine needs more launches (see rsdsfm_depth_finish_dev) */
#define RSDSFM_ERR_PEER (-6)        /* tiled solve: ANOTHER rank failed while setting up its part; every rank returns together */

/* depth_mode */
#define RSDSFM_DEPTH_CLOSED_FORM 0  /* exact per-pixel least-squares optimum (one undamped GN step from rho=1) */
#define RSDSFM_DEPTH_CERES_LM 1     /* emulation of the Ceres 1.14 trust-region LM the reference runs (default) */

/* k_sign_mode (reference quirk Q4, minimal.cc:72) */
#define RSDSFM_K_COMPAT 0
#define RSDSFM_K_FIXED 1

/* flow_index_mode of rsdsfm_refine (reference quirk Q2, nonlinearRefinement.cc:211-212) */
#define RSDSFM_FLOW_COMPAT_RANK 0
#define RSDSFM_FLOW_GATHERED 1

/* mode of rsdsfm_back_project: per-scanline relative poses (RsFrame::backProject, rsframe.cc:803) or the pose of
 * scanline 0 for every pixel (RsFrame::backProjectGs, rsframe.cc:842) */
#define RSDSFM_BACKPROJECT_RS 0
#define RSDSFM_BACKPROJECT_GS 1

/* q5_mode of rsdsfm_back_project (reference quirk Q5: spaceToPlane scales the y coordinate by f_x, rsframe.cc:639) */
#define RSDSFM_Q5_COMPAT 0
#define RSDSFM_Q5_FIXED 1

/* termination of the emulated Ceres trust-region loop */
#define RSDSFM_TERM_GRADIENT 0
#define RSDSFM_TERM_PARAMETER 1
#define RSDSFM_TERM_FUNCTION 2
#define RSDSFM_TERM_MAX_ITER 3
#define RSDSFM_TERM_FAILURE 4
#define RSDSFM_TERM_MIN_RADIUS 5

typedef struct rsdsfm_ctx rsdsfm_ctx;

typedef struct rsdsfm_lm_summary {
    int32_t num_iterations;
    int32_t num_successful_steps;
    int32_t num_unsuccessful_steps;
    int32_t termination;
    double initial_cost;
    double final_cost;
    double final_radius;
} rsdsfm_lm_summary;

/* replaces RansacValues (minimal.h:57-76) + the extras SURVEY Q2 asks for.  Arrays are caller-owned with
 * capacity n (HOST pointers for rsdsfm_ransac, DEVICE pointers for rsdsfm_ransac_dev); NULL = not wanted
 * (inliers/alpha/alpha_k/inlier_idx/mask/inv_depth may each be NULL). */
typedef struct rsdsfm_ransac_out {
    int64_t num_inliers;
    int32_t best_trial;
    int32_t _pad;
    double w[3], v[3], k;
    double inlier_error;
    int64_t* inlier_idx; /* [n]  index of each inlier in the input arrays                       */
    double* inliers;     /* [3n] (x, y, z = 1/rho)                                               */
    double* alpha;       /* [n]                                                                  */
    double* alpha_k;     /* [n]                                                                  */
    uint8_t* mask;       /* [n]                                                                  */
    double* inv_depth;   /* [n]  dense rho of the best trial                                     */
    int64_t* trial_count; /* [T] HOST, or NULL                                                   */
    double* trial_err;    /* [T] HOST, or NULL.  trial_err / inlier_error are DIAGNOSTICS: in mode 0 (rsdsfm_set_lm_arithmetic) a trial that ends at a fused iterate gets the closed form's sum, another one the exact scoring pass's, and a context that met a tie stays on the iterate-by-iterate kernels for 16 runs -- the sums agree to ~1e-13 relative between the forms, the integers and the winner's depths are identical */
    double* trial_vel;    /* [7T] HOST (w, v, k), or NULL                                        */
    int32_t* trial_steps; /* [T] HOST accepted LM steps of each trial's depth solve, or NULL      */
} rsdsfm_ransac_out;

/* ---------------------------------------------------------------------------------------------------- */
/* context                                                                                                */
/* ---------------------------------------------------------------------------------------------------- */
/* stream_or_null: a hipStream_t to adopt (e.g. torch.cuda.current_stream().cuda_stream) or NULL = own stream */
int rsdsfm_create(rsdsfm_ctx** ctx, int device, void* stream_or_null);
void rsdsfm_destroy(rsdsfm_ctx* ctx);
const char* rsdsfm_last_error(const rsdsfm_ctx* ctx);
const char* rsdsfm_version(void);
/* Which build is loaded.  0: librsdsfm_hip.so -- its ITERATE-BY-ITERATE kernels (rsdsfm_set_lm_arithmetic(1), and what a guard of the
 * default path falls back to) evaluate the per-pixel model with the REFERENCE's arithmetic (no fused multiply-add, as the reference's own
 * x86-64 build, src/CMakeLists.txt:18); the DEFAULT path (analytic LM trajectory, radius-factorised refinement) is the library's own
 * arithmetic with fused multiply-adds in both builds, see rsdsfm_set_lm_arithmetic.  1: the opt-in librsdsfm_hip_fused.so (explicit fmas in the
 * residual / LM step / scoring error / scanline projection: faster, values agree to ~1e-12 relative) */
int rsdsfm_fused_arithmetic(void);
int rsdsfm_synchronize(rsdsfm_ctx* ctx);
/* variant of the LM depth solve's fast path: 0 (default) = launch 0 + ONE follow-up launch that decides and applies
 * (depth_lm_decide_apply_kernel); 1 = per-wave LDS-DMA double buffering in launch 0 (global_load_lds + counted vmcnt) with
 * the separate decide kernel; 2 = the decision fused into the tail of launch 0 (experimental, measured slower); 3 = launch
 * 0, separate decide kernel, follow-up launch (the fast path before the fusion).  Same arithmetic, same results. */
int rsdsfm_set_depth_variant(rsdsfm_ctx* ctx, int variant);
/* LM iterations that round 0 of the hypothesis-batched depth solves inside rsdsfm_ransac* speculates per pixel: 3 decides every
 * hypothesis that ends with <= 2 accepted steps in one pass; 2 is cheaper when every hypothesis stops after ONE accepted step
 * (outlier-dominated data); hypotheses that need more take a continuation round; 0 (default) follows the context's previous solve
 * (2 when none of its hypotheses went beyond one accepted step).  Every result is identical for all settings, bit for bit: round 0
 * also scores the iterate most hypotheses of the context's previous solve ended at, all other hypotheses are scored by a separate
 * pass that adds the inlier errors in the same order. */
int rsdsfm_set_ransac_speculation(rsdsfm_ctx* ctx, int k0);
/* How round 0 of those depth solves evaluates its two square roots and its reciprocal per pixel-hypothesis (reference-arithmetic
 * library only).  0 (default): through the in-range cores of the compiler's own correctly rounded expansions -- the same instructions
 * without the argument rescaling and special-case patching that do nothing for a normal, positive, finite argument (16 of the kernel's 307
 * instructions; bit-identical in range, tools/fastmath_check.hip) -- and a RANSAC that meets an argument outside the range (a zero
 * Jacobian, a zero or non-finite error: not on real data) starts over with the standard functions, as do the context's next 16.
 * 1: always the standard functions.  Every result is identical for both settings, bit for bit.
 * rsdsfm_ransac_restarts: how many RANSAC runs of this context (and its sequence lanes) started over. */
int rsdsfm_set_ransac_math(rsdsfm_ctx* ctx, int mode);
int rsdsfm_ransac_restarts(rsdsfm_ctx* ctx, int64_t* count);
/* The arithmetic of the T dense depth solves inside a RANSAC (minimal.cc:246: estimateInverseDepths per trial, nonlinearRefinement.cc:109-180
 * under Ceres 1.14's trust-region loop), RSDSFM_DEPTH_CERES_LM mode.
 * 0 (default): the ANALYTIC LM TRAJECTORY.  For a fixed pose every residual is linear in its own inverse depth, so the iterates Ceres
 *   walks, and every global quantity its accept / converge tests look at, are closed forms of five sums and a maximum per hypothesis
 *   (csrc/lma_common.hpp): one pass over the pixels, ~70 operations per pixel-hypothesis instead of ~55 per pixel, hypothesis AND
 *   iteration.  Guards keep every integer output (accepted steps, terminations, per-trial inlier counts, winner, inlier mask / index list)
 *   equal to the iterate-by-iterate arithmetic's: pixels whose LM diagonal is clamped, and pixels whose error comes within a margin of the
 *   tolerance, walk the reference's exact recurrence; a global decision within 1e-6 (relative) of its threshold, or a tie in the inlier
 *   count that only rounding noise could break (noise-free data), makes the run start over iterate by iterate, as do the context's next
 *   16 runs.  The winner's dense inverse depths and mask always come from the exact replay of its accepted steps.
 * 1: always iterate by iterate (the reference's arithmetic, operation for operation).
 * Inside rsdsfm_solve_frame*_dev, which does not report the trials' error sums, mode 0 has a second form of the pass: COUNT-ONLY.
 *   minimal.cc:278-285 reads a trial's error sum only to break a tie in the inlier count, so behind two solves of the context whose best
 *   count no other trial shared (a selective tolerance) the pass leaves the two square roots per pixel-hypothesis out (~30 % of its
 *   instructions); should trials share the best count after all, exactly those are scored by the iterate-by-iterate scoring pass (the
 *   reference's arithmetic: the tie is broken as the reference breaks it) and the context goes back to fused error sums.  With a
 *   permissive tolerance (BASELINE's 0.05 admits every pixel under any good hypothesis) the sums decide every solve and stay fused.
 * 2: mode 0 with the count-only form forced (tests and measurements).
 * rsdsfm_lma_restarts: how many RANSAC runs of this context (and its sequence lanes) started over because a guard tripped, and (optional)
 * which guards tripped last, as a bit set (1 << r: r = 1 infinite sum, 2 gradient / 3 model-change / 4 parameter / 5 function tolerance within
 * the band, 6 step quality, 7 tie, 8 list overflow, 9 listed pixels off the tabulated trajectory, 10 count check of the winner's replay). */
int rsdsfm_set_lm_arithmetic(rsdsfm_ctx* ctx, int mode);
int rsdsfm_lma_restarts(rsdsfm_ctx* ctx, int64_t* count, int32_t* last_guards_or_null);
/* The joint refinement (nonlinearRefinement.cc:183-252; rsdsfm_refine*, the refinement inside rsdsfm_solve_frame*_dev and the tiled solve) in
 * mode 0 runs on RADIUS-FACTORISED Schur sums (csrc/refine_rf_kernels.hip): the sums a pass takes at a point serve any trust-region radius,
 * so every LM iteration is ONE pass over the inliers and a rejected / invalid step costs none; its arithmetic is the library's own (fused
 * multiply-adds), and every decision of Ceres' loop that lands within a relative band of its threshold (invalid step, parameter / function /
 * gradient tolerance, step quality, Cholesky pivot, a non-finite sum, more than 64 inliers with an active LM-diagonal clamp) sends THAT solve
 * back to the iterate-by-iterate kernels (the reference's arithmetic), which mode 1 selects outright.
 * rsdsfm_refine_restarts: refinements of this context (and its sequence lanes) that ran on the radius-factorised path, how many of them were
 * sent back, (optional) the reduced systems solved again from stored sums, and the guard that tripped last (1 non-finite sum, 2 gradient /
 * 3 model change / 4 parameter / 5 function tolerance inside its band, 6 step quality, 7 pivot, 8 list overflow, 9 minimum radius). */
int rsdsfm_refine_restarts(rsdsfm_ctx* ctx, int64_t* runs, int64_t* restarts, int64_t* resolves_or_null, int32_t* last_guard_or_null);
/* The refinement's arithmetic on its own: 0 (default) = what rsdsfm_set_lm_arithmetic selects (radius-factorised in mode 0), 1 = the
 * iterate-by-iterate slot kernels whatever that mode is -- e.g. to compare the forms of the RANSAC's depth solves bit for bit behind one and the
 * same refinement.  The same on every rank of a tiled solve. */
int rsdsfm_set_refine_arithmetic(rsdsfm_ctx* ctx, int mode);
/* how many RANSACs of this context (and its sequence lanes) ran the count-only form of the pass, and how many of those had to fetch error sums */
int rsdsfm_lma_count_only(rsdsfm_ctx* ctx, int64_t* runs, int64_t* lazy_runs_or_null);
/* The dense depth solve (rsdsfm_estimate_inverse_depths*, LM mode) takes the same in-range cores in launch 0 (Jacobi scaling) under the same
 * switch (rsdsfm_set_ransac_math); a solve whose launch 0 met an argument out of their range is left unfinished by its follow-up
 * launch and rsdsfm_depth_finish_dev -- which every LM-mode caller runs to obtain the summary -- starts it over with the standard
 * functions: identical results, counted here.  (nonlinearRefinement.cc:109-180: the reference has one code path) */
int rsdsfm_depth_restarts(rsdsfm_ctx* ctx, int64_t* count);
/* Opt-in profiling: while on, rsdsfm_ransac* / rsdsfm_solve_frame_dev bracket the dominant kernel of the whole solve -- round 0
 * of the hypothesis-batched LM depth solves, `ransac_lm_kernel<true, 3, BASE, CORE>` -- with two HIP events on the context's stream (in
 * situ: same launch, same neighbours, same clocks as any other solve).  rsdsfm_profile_last_ms(ctx, "ransac_lm_round0", &ms)
 * returns the duration of that launch in the most recent call.  bench.py's roofline record uses it.
 * rsdsfm_profile_last_ms(ctx, "ransac_lm_round0_clock_mhz", &mhz) returns -- in MHz, through the same out-parameter -- the shader clock
 * one workgroup in the middle of that launch ran at (shader clocks of its life / 100 MHz ticks of its life): the chip lowers its clock
 * under sustained fp64 load, and a roofline priced at the nominal clock overstates the ceiling by that ratio.
 * With profiling on for the FIRST context of a batched dense depth solve (rsdsfm_estimate_inverse_depths_batch_dev), its streaming
 * launch (`depth_lm_batch_kernel`) carries the dispatch's own start / stop timestamps (hipExtLaunchKernel events: what rocprofv3
 * --kernel-trace reports for the kernel, without the gap between launches): rsdsfm_profile_last_ms(ctx, "depth_lm_batch", &ms). */
/* Where the single-workgroup stage of the joint refinement's iteration loop runs (nonlinearRefinement.cc:183-252 has no such notion: Ceres runs
 * the loop on the host).  0 (default) = automatic: in the prologue of the next slot's streaming pass, computed redundantly by every workgroup
 * -- one launch per LM iteration, the shortest single solve -- except when the solve is not alone on its GPU (several lanes of
 * rsdsfm_solve_frames_dev in flight, or frame solves of other contexts of the process on the same device), where it gets a launch of its
 * own behind every pass (the prologue keeps the whole chip busy for its ~8 us; a single workgroup leaves it to the other solves' kernels:
 * +4 % sequence throughput); 1 = always in the prologue; 2 = always a launch of its own.  Identical results in every mode. */
int rsdsfm_set_refine_stage(rsdsfm_ctx* ctx, int mode);
int rsdsfm_set_profiling(rsdsfm_ctx* ctx, int on);
int rsdsfm_profile_last_ms(rsdsfm_ctx* ctx, const char* what, double* ms);
/* name of the HIP kernel that dominates the given entry point (for profiling / roofline reports) */
const char* rsdsfm_kernel_name(const char* entry_point);

/* ---------------------------------------------------------------------------------------------------- */
/* host-pointer API -- one per reference function                                                         */
/* ---------------------------------------------------------------------------------------------------- */
/* minimal::getAlpha  minimal.cc:179-186 */
int rsdsfm_get_alpha(rsdsfm_ctx* ctx, const double* flow_px2n, int64_t n, double h, double gamma, double* alpha_n);
/* minimal::getAlphaK  minimal.cc:188-197 */
int rsdsfm_get_alpha_k(rsdsfm_ctx* ctx, const double* q_px2n, const double* flow_px2n, int64_t n, double h,
                       double gamma, double* alpha_k_n);
/* minimal::calculateVelocities  minimal.cc:36-177  (batch of `count` 9-point hypotheses, one lane each) */
int rsdsfm_calculate_velocities(rsdsfm_ctx* ctx, const double* q_18xT, const double* u_18xT,
                                const double* alpha_9xT, const double* alpha_k_9xT, int32_t count,
                                int use_alpha_k, int k_sign_mode, double* w_3xT, double* v_3xT, double* k_T);
/* nonlinear_refinement::estimateInverseDepths  nonlinearRefinement.cc:109-180
 * (and estimateInverseDepth :55-106 with n = 1).  summary: the LM emulation's counters, termination type, costs and final
 * radius; RSDSFM_DEPTH_CLOSED_FORM reports one successful iteration and does not evaluate the costs (both 0). */
int rsdsfm_estimate_inverse_depths(rsdsfm_ctx* ctx, const double* q2n, const double* u2n, int64_t n,
                                   const double v[3], const double w[3], double k, const double* alpha_n,
                                   const double* alpha_k_n, int depth_mode, double* inv_depth_n,
                                   rsdsfm_lm_summary* summary_or_null);
/* minimal::ransac  minimal.cc:209-306.  samples_9xT: injected 9-index samples (int32, T*9) or NULL = the
 * reference's partial Fisher-Yates sampler driven by splitmix64(seed) instead of srand(time)/rand(). */
int rsdsfm_ransac(rsdsfm_ctx* ctx, const double* q2n, const double* u2n, const double* alpha_n,
                  const double* alpha_k_n, int64_t n, int use_alpha_k, int32_t iterations, double tolerance,
                  const int32_t* samples_9xT_or_null, uint64_t seed, int depth_mode, int k_sign_mode,
                  rsdsfm_ransac_out* out);
/* nonlinear_refinement::nonLinearRefinement  nonlinearRefinement.cc:183-252.  inliers in/out are 3xM. */
int rsdsfm_refine(rsdsfm_ctx* ctx, const double* flow2n, int64_t n_flow, int64_t m, const double* inliers_3m,
                  const double* alpha_m, const double* alpha_k_m, const int64_t* inlier_idx_or_null,
                  const double v_in[3], const double w_in[3], double k_in, int const_acceleration,
                  int flow_index_mode, double* inliers_out_3m, double v_out[3], double w_out[3], double* k_out,
                  rsdsfm_lm_summary* summary_or_null);
/* The reference hands nonLinearRefinement the RansacValues -- and the flow -- that minimal::ransac has just seen (main.cc:447-457,
 * errorMeasure.cpp:140-152).  rsdsfm_ransac leaves exactly those arrays on the device; rsdsfm_last_ransac_tag names them (0: nothing usable),
 * and rsdsfm_refine_from_ransac = rsdsfm_refine that starts from them instead of uploading ~52 MB per 1280x720 pair again -- provided the tag is
 * the context's last RANSAC's, no other host-pointer call of the context came in between, the sizes match and 16 probed entries of the arrays
 * passed here still hold what was transferred then; otherwise it uploads, like rsdsfm_refine.  CONTRACT: arrays passed with a non-zero tag are
 * the unmodified outputs of that rsdsfm_ransac (the probes catch a rebuilt array, not a single edited entry); pass tag 0 after editing them.
 * `flow` is reused too when it is the very array (address and length) the RANSAC was given as u.  The C++ mirror (host/minimal.h,
 * host/nonlinearRefinement.h) carries the tag in RansacValues.  cache_hits: refinements of this context that started from resident arrays. */
int rsdsfm_last_ransac_tag(rsdsfm_ctx* ctx, uint64_t* tag, int64_t* cache_hits_or_null);
int rsdsfm_refine_from_ransac(rsdsfm_ctx* ctx, uint64_t ransac_tag, const double* flow2n, int64_t n_flow, int64_t m, const double* inliers_3m,
                              const double* alpha_m, const double* alpha_k_m, const int64_t* inlier_idx_or_null, const double v_in[3],
                              const double w_in[3], double k_in, int const_acceleration, int flow_index_mode, double* inliers_out_3m,
                              double v_out[3], double w_out[3], double* k_out, rsdsfm_lm_summary* summary_or_null);
/* Trace of the joint refinement's trust-region iterations (the counterpart of Ceres' per-iteration log,
 * `Solver::Summary::iterations` behind nonlinearRefinement.cc:228-230 `FullReport()`; diagnostic, off by default).
 * rsdsfm_set_refine_trace(ctx, rows): rows > 0 makes every following refinement on this context (rsdsfm_refine*, the frame
 * solves) record its first `rows` LM iterations in a context-owned device buffer; 0 switches it off.  The record is written by
 * the kernel that takes the iteration's accept / reject / converge decision; results are unaffected.
 * rsdsfm_get_refine_trace(ctx, out, rows): copies rows x RSDSFM_REFINE_TRACE_COLS doubles of the LAST refinement to the host;
 * row i (iteration i + 1): [0] iteration number, [1] cost before the step, [2] cost of the candidate, [3] model cost change,
 * [4] relative decrease (cost change / model change), [5] trust-region radius the step was computed with, [6] step norm,
 * [7] outcome (RSDSFM_TRACE_*); entries the iteration never computed are NaN, rows beyond the last iteration are NaN. */
#define RSDSFM_REFINE_TRACE_COLS 8
#define RSDSFM_TRACE_REJECTED 0.0              /* relative decrease <= 1e-3: radius shrinks */
#define RSDSFM_TRACE_ACCEPTED 1.0
#define RSDSFM_TRACE_INVALID 2.0               /* the reduced system failed to factor, or model change <= 0 */
#define RSDSFM_TRACE_PARAMETER_TOL 3.0         /* terminated: step norm below the parameter tolerance */
#define RSDSFM_TRACE_FUNCTION_TOL 4.0          /* terminated: |cost change| below the function tolerance (candidate not applied) */
#define RSDSFM_TRACE_ACCEPTED_GRADIENT_TOL 5.0 /* accepted, then terminated on the gradient tolerance */
int rsdsfm_set_refine_trace(rsdsfm_ctx* ctx, int32_t rows);
int rsdsfm_get_refine_trace(rsdsfm_ctx* ctx, double* trace_rows_x_8, int32_t rows);
/* caller glue main.cc:398-444 / errorMeasure.cpp:66-111: column-major scan of the row-major flow image,
 * threshold, normalisation, alpha / alpha_k.  Outputs have capacity rows*cols; *n_out = kept points. */
int rsdsfm_flatten(rsdsfm_ctx* ctx, const double* flow_img_rows_cols_2, int32_t rows, int32_t cols, double fx,
                   double fy, double cx, double cy, double gamma, double flow_threshold, double* q2n,
                   double* u2n, double* alpha_n, double* alpha_k_n, int64_t* n_out);
/* caller glue main.cc:466-509: mean-z sign flip of (z, v), then depth_map(y, x) = z with
 * x = int(fx*qx+cx+.5), y = int(fy*qy+cy+.5); depth_map is rows x cols column-major (Eigen MatrixXd) and
 * is zero-filled first.  xs/ys (int32[m], the scanline index is ys) may be NULL.  *flipped may be NULL. */
int rsdsfm_depth_map(rsdsfm_ctx* ctx, double* inliers_3m_inout, int64_t m, double v_inout[3], double fx,
                     double fy, double cx, double cy, int32_t rows, int32_t cols,
                     double* depth_map_colmajor, int32_t* xs_or_null, int32_t* ys_or_null, int* flipped);
/* RsFrame::setRelativePose  rsframe.cc:771-800: per-scanline R (row-major 3x3) and t */
int rsdsfm_pose_table(rsdsfm_ctx* ctx, const double v[3], const double w[3], double k, double gamma,
                      int32_t rows, double* R_rows9, double* t_rows3);

/* ---------------------------------------------------------------------------------------------------- */
/* device API (asynchronous on the context's stream; pointers are DEVICE memory, 16-byte aligned)          */
/* ---------------------------------------------------------------------------------------------------- */
/* estimateInverseDepths on device-resident inputs.  Enqueues the fixed fast-path launch sequence (in
 * RSDSFM_DEPTH_CERES_LM mode: speculative LM launch 0, the decide kernel, launch 1 = apply / continue / no-op)
 * and returns without synchronising. */
int rsdsfm_estimate_inverse_depths_dev(rsdsfm_ctx* ctx, const double* d_q2n, const double* d_u2n, int64_t n,
                                       const double v[3], const double w[3], double k, const double* d_alpha_n,
                                       const double* d_alpha_k_n, int depth_mode, double* d_inv_depth_n);
/* flatten / depth map / pose table on device-resident buffers (same semantics as the host-pointer variants; the
 * scalar results *n_out, v_inout, *flipped are returned after one synchronisation) */
int rsdsfm_flatten_dev(rsdsfm_ctx* ctx, const double* d_flow_img, int32_t rows, int32_t cols, double fx, double fy,
                       double cx, double cy, double gamma, double flow_threshold, double* d_q2n, double* d_u2n,
                       double* d_alpha_n, double* d_alpha_k_n, int64_t* n_out);
int rsdsfm_depth_map_dev(rsdsfm_ctx* ctx, double* d_inliers_3m_inout, int64_t m, double v_inout[3], double fx, double fy,
                         double cx, double cy, int32_t rows, int32_t cols, double* d_depth_map_colmajor,
                         int32_t* d_xs_or_null, int32_t* d_ys_or_null, int* flipped);
int rsdsfm_pose_table_dev(rsdsfm_ctx* ctx, const double v[3], const double w[3], double k, double gamma, int32_t rows,
                          double* d_R_rows9, double* d_t_rows3);
/* ---- one frame pair end to end (the solver part of evaluateSingleRun, main.cc:398-522) --------------------------- */
typedef struct rsdsfm_frame_params {
    int32_t ransac_trials;        /* main.cc:304 (5); the report used 50                                */
    int32_t use_acceleration_mode; /* main.cc:306                                                       */
    int32_t use_refinement;       /* main.cc:307                                                        */
    int32_t depth_mode;           /* RSDSFM_DEPTH_*                                                     */
    int32_t k_sign_mode;          /* RSDSFM_K_*                                                         */
    int32_t flow_index_mode;      /* RSDSFM_FLOW_*: how nonLinearRefinement indexes the flow.  0 = COMPAT_RANK, what
                                     evaluateSingleRun does (main.cc:457 passes the UN-compacted flow and
                                     nonlinearRefinement.cc:209-212 reads column i for the i-th inlier, quirk Q2);
                                     1 = GATHERED (column inlier_idx[i])                                  */
    int32_t use_global_shutter_mode; /* main.cc:305, :441-444: alpha = 1 for every point (overrides the RS model) */
    int32_t struct_bytes;         /* 0 (zero-initialised struct) or sizeof(rsdsfm_frame_params), as rsdsfm_frame_params_init
                                     sets it; anything else is refused: the caller was built against another layout of this
                                     struct (round 1's was 8 bytes shorter and its ransac_tol sits where this field is)   */
    double ransac_tol;            /* main.cc:310 (0.05)                                                 */
    double flow_threshold;        /* main.cc:311 (1e-10)                                                */
    uint64_t seed;                /* sampler seed (reference: srand(time))                              */
} rsdsfm_frame_params;

/* the reference's constants (main.cc:304-311: 5 trials, tolerance 0.05, flow threshold 1e-10, refinement on, rolling shutter,
 * rank-indexed flow), Ceres-LM depth mode, seed 1, struct_bytes = sizeof */
void rsdsfm_frame_params_init(rsdsfm_frame_params* params);

typedef struct rsdsfm_frame_result {
    int64_t n_points, num_inliers;
    int32_t best_trial, flipped;
    double ransac_w[3], ransac_v[3], ransac_k;
    double w[3], v[3], k;            /* after refinement and sign canonicalisation                       */
    rsdsfm_lm_summary refine_summary;
    const double* d_inliers;         /* DEVICE, 3 x num_inliers (x, y, z), valid until the next call on the context */
    const int64_t* d_inlier_idx;     /* DEVICE                                                            */
    const int32_t* d_scanline;       /* DEVICE, scanline (image row) index of each inlier                 */
} rsdsfm_frame_result;

/* flow image (rows x cols x 2, row-major) resident in HBM -> pose, depth map (rows x cols column-major, DEVICE) and
 * per-scanline pose table (DEVICE, may be NULL).  Runs flatten (+ the global-shutter override of alpha), ransac,
 * nonLinearRefinement (flow indexed as params->flow_index_mode says), the sign flip / depth scatter and
 * setRelativePose back to back on the context's stream.  A zero-initialised rsdsfm_frame_params with the trial count,
 * tolerances and use_refinement set reproduces evaluateSingleRun's call sequence (main.cc:398-522). */
int rsdsfm_solve_frame_dev(rsdsfm_ctx* ctx, const double* d_flow_img, int32_t rows, int32_t cols, double fx, double fy,
                           double cx, double cy, double gamma, const rsdsfm_frame_params* params,
                           double* d_depth_map_colmajor, double* d_R_rows9_or_null, double* d_t_rows3_or_null,
                           rsdsfm_frame_result* result);
/* ---- a SEQUENCE of frame pairs (BASELINE configs[4]: "batched frame pairs, sequence throughput mode") ---------------------
 * evaluateSingleRun (main.cc:302-559) solves one pair per process run; a host that has a whole sequence resident hands the pairs
 * over in ONE call.  The library pipelines them: up to `lanes` pairs are in flight on streams of their own, so the latency-bound
 * kernels of one pair (the 9-point minimal solver, the single-workgroup decide / solve stages) run beside the streaming kernels
 * of another.  One host thread, one context; results[i] is exactly what rsdsfm_solve_frame_dev returns for jobs[i] with
 * params->seed = jobs[i].seed (the device pointers inside results[i] stay valid for the last `lanes` pairs only). */
typedef struct rsdsfm_frame_job {
    const double* d_flow_img;      /* DEVICE, row-major rows x cols x 2 (as rsdsfm_solve_frame_dev)            */
    int32_t rows, cols;
    double fx, fy, cx, cy, gamma;
    double* d_depth_map_colmajor;  /* DEVICE, rows x cols                                                     */
    double* d_R_rows9_or_null;     /* DEVICE pose table, may be NULL                                          */
    double* d_t_rows3_or_null;
    uint64_t seed;                 /* sampler seed of this pair (params->seed is ignored)                     */
} rsdsfm_frame_job;
int rsdsfm_solve_frames_dev(rsdsfm_ctx* ctx, const rsdsfm_frame_job* jobs, int32_t count, const rsdsfm_frame_params* params,
                            rsdsfm_frame_result* results);
/* pairs in flight of rsdsfm_solve_frames_dev: 1 .. 16, 0 = default (3).  Scheduling only. */
int rsdsfm_set_sequence_lanes(rsdsfm_ctx* ctx, int32_t lanes);
/* Where the flatten of a frame runs whose predecessor on the context was dense (every pixel kept, so that point i IS pixel
 * (i / rows, i % rows) and the minimal solver can form its sampled points straight from the flow image):
 *   3 (default) = INSIDE the minimal solver's launch: the first T workgroups solve (one wave per hypothesis, ~186 us of a serial
 *       division / square-root chain with the rest of the GPU idle), the other workgroups flatten the dense frame to its known
 *       positions -- no scan -- and count dropped pixels; a frame that turns out not to be dense runs the general flatten again;
 *   2 = behind the solver on the context's stream, 1 = beside it on a second stream, 0 = in front of it (count, scan, scatter).
 * Measured on MI355X (1280x720, T = 50, medians of 60 solves): 0.963 / 0.986 / 0.995 / 0.977 ms for 3 / 2 / 1 / 0 -- the
 * cross-stream join of mode 1 costs more than the 21 us of flatten kernels it hides.  Scheduling only: identical results. */
int rsdsfm_set_frame_side_flatten(rsdsfm_ctx* ctx, int mode);
/* minimal::ransac on device-resident inputs.  The arrays of `out` (inlier_idx, inliers, alpha, alpha_k, mask,
 * inv_depth) are DEVICE pointers with capacity n (each may be NULL); its trial_* arrays are HOST pointers.
 * samples_9xT_or_null is a HOST pointer.  Synchronises once at the end to return the scalars of `out`. */
int rsdsfm_ransac_dev(rsdsfm_ctx* ctx, const double* d_q2n, const double* d_u2n, const double* d_alpha_n,
                      const double* d_alpha_k_n, int64_t n, int use_alpha_k, int32_t iterations, double tolerance,
                      const int32_t* samples_9xT_or_null, uint64_t seed, int depth_mode, int k_sign_mode,
                      rsdsfm_ransac_out* out);
/* nonLinearRefinement on device-resident inputs (d_inlier_idx may be NULL in compat mode).  v/w/k and the
 * summary are HOST.  Polls the device-resident termination flag every few LM iterations. */
int rsdsfm_refine_dev(rsdsfm_ctx* ctx, const double* d_flow2n, int64_t n_flow, int64_t m, const double* d_inliers_3m,
                      const double* d_alpha_m, const double* d_alpha_k_m, const int64_t* d_inlier_idx_or_null,
                      const double v_in[3], const double w_in[3], double k_in, int const_acceleration,
                      int flow_index_mode, double* d_inliers_out_3m, double v_out[3], double w_out[3], double* k_out,
                      rsdsfm_lm_summary* summary_or_null);
/* Batched fast path: `count` (<= 8) independent dense depth solves (LM mode) in ONE pair of launches -- the streaming pass
 * and the decide-and-apply follow-up run over all of them (grid y = solve), which amortises the launch floor and the
 * ramp-up / tail of the streaming pass.  Every solve uses ITS OWN context (state, partial sums), all created on one
 * stream; the launches go to that stream.  Arrays of `count` entries: device pointers, sizes, v / w as count x 3, k.
 * Afterwards rsdsfm_depth_finish_dev(ctxs[i], ...) completes / verifies solve i exactly as after the single call. */
int rsdsfm_estimate_inverse_depths_batch_dev(rsdsfm_ctx* const* ctxs, int32_t count, const double* const* d_q2n,
                                             const double* const* d_u2n, const int64_t* n, const double* v_count_x3,
                                             const double* w_count_x3, const double* k_count, const double* const* d_alpha_n,
                                             const double* const* d_alpha_k_n, double* const* d_inv_depth_n);
/* launch 0 of the batched fast path alone (depth_lm_batch_kernel): what bench.py brackets with HIP events */
int rsdsfm_depth_lm_batch_launch_dev(rsdsfm_ctx* const* ctxs, int32_t count, const double* const* d_q2n, const double* const* d_u2n,
                                     const int64_t* n, const double* v_count_x3, const double* w_count_x3, const double* k_count,
                                     const double* const* d_alpha_n, const double* const* d_alpha_k_n, double* const* d_inv_depth_n);
/* One launch of the fused LM kernel (building block of the calls around it; also what bench.py brackets with
 * HIP events to time the dominant kernel).  launch_id 0 = the launch of LM iteration zero (fresh state);
 * launch_id > 0 acts only if the device state machine designated that launch. */
int rsdsfm_depth_lm_launch_dev(rsdsfm_ctx* ctx, const double* d_q2n, const double* d_u2n, int64_t n,
                               const double v[3], const double w[3], double k, const double* d_alpha_n,
                               const double* d_alpha_k_n, double* d_inv_depth_n, int launch_id);
/* ---- stage-level entry points of the ROW-TILED (multi-GPU) dense depth solve --------------------------------
 * Every rank owns a contiguous index range (its shard) of the flattened arrays.  Per LM launch `id`:
 *   rsdsfm_depth_lm_launch_dev(shard, id)  ->  rsdsfm_depth_lm_reduce_dev  (one row of NS sums)
 *   all-gather of the rows over ranks (RCCL, rank order)  ->  rsdsfm_depth_lm_decide_rows_dev on EVERY rank
 * so that all ranks run the trust-region state machine on identical sums and take identical decisions;
 * rsdsfm_depth_lm_state (synchronises) tells the driver whether another launch is needed.  The depth shards
 * are finally all-gathered into the full map.  See rs-aware-differential-sfm_amd/dist.py. */
int rsdsfm_depth_lm_sums_row_size(void);
int rsdsfm_depth_lm_reduce_dev(rsdsfm_ctx* ctx, int64_t n_shard, double* d_row_out);
int rsdsfm_depth_lm_decide_rows_dev(rsdsfm_ctx* ctx, const double* d_rows, int32_t nrows, int64_t n_total, int launch_id);
/* status: 0 = launch *next_launch must speculate, 1 = done (result written), 2 = done, launch *next_launch writes it,
 * 3 = the solve must START OVER from launch 0 (*next_launch = 0): its launch 0 ran a fast path whose result does not count -- the
 * in-range function cores met an argument out of their range, or a guard of the analytic LM trajectory tripped -- and only
 * rsdsfm_depth_finish_dev knows how to run it again (it does so by itself); a caller that drives the launches on its own must not go on
 * from the sums of that launch. */
int rsdsfm_depth_lm_state(rsdsfm_ctx* ctx, int32_t* status, int32_t* next_launch, rsdsfm_lm_summary* summary_or_null);
/* Synchronises, drives the device LM state machine to completion if the fast path did not finish it
 * (rare: more than 3 LM iterations or a rejected step) and returns the summary.  Returns the number of
 * EXTRA launches that were needed in *extra_launches (may be NULL). */
int rsdsfm_depth_finish_dev(rsdsfm_ctx* ctx, const double* d_q2n, const double* d_u2n, int64_t n,
                            const double v[3], const double w[3], double k, const double* d_alpha_n,
                            const double* d_alpha_k_n, double* d_inv_depth_n, rsdsfm_lm_summary* summary_or_null,
                            int32_t* extra_launches);

/* ---- stage-level entry points of the ROW-TILED (multi-GPU) WHOLE-FRAME solve -----------------------------------
 * RANSAC (minimal.cc:209-306), nonLinearRefinement (nonlinearRefinement.cc:183-252) and the
 * caller glue (main.cc:398-522) with the image split into COLUMN SLABS, one per rank.  The reference flattens the
 * image column-major (main.cc:398-444), so the slabs' point lists concatenated in rank order are the reference's
 * point list.  Each "*_rows_dev" stage works on the caller's shard and leaves a small row of sums on the device;
 * the driver all-gathers the rows in rank order (RCCL) and passes the gathered [nranks][...] array to the matching
 * decide / apply stage, which EVERY rank runs identically (the kernels that reduce per-workgroup partials in the
 * single-GPU solve reduce the per-rank rows) -- all ranks take identical decisions, nothing is broadcast.
 * All calls are asynchronous on the context's stream unless noted.  Driver: rs-aware-differential-sfm_amd/dist.py
 * (TiledFrameSolve). */
/* flatten of one slab: d_img_slab is the row-major [rows][slab_cols][2] slab whose first column is image column
 * col0; q.x is computed from the GLOBAL column.  Synchronises (returns the shard's point count). */
int rsdsfm_flatten_slab_dev(rsdsfm_ctx* ctx, const double* d_img_slab, int32_t rows, int32_t slab_cols, int32_t col0,
                            double fx, double fy, double cx, double cy, double gamma, double thr, double* d_q2n,
                            double* d_u2n, double* d_alpha_n, double* d_alpha_k_n, int64_t* n_out);
/* host: the deterministic sampler of rsdsfm_ransac (9 partial Fisher-Yates draws per trial, minimal.cc:226-244,
 * splitmix64(seed) in place of rand()); n = GLOBAL point count, samples = 9 * iterations global indices */
int rsdsfm_sample_indices(int64_t n, int32_t iterations, uint64_t seed, int32_t* samples);
/* minimal::calculateVelocities (minimal.cc:36-177) on `count` packed 9-point sets already on the device:
 * d_q9 / d_u9 = count x 9 x 2, d_alpha9 / d_alpha_k9 = count x 9; d_hyp = count x 8: w(3), v(3), k, status */
int rsdsfm_minimal9_dev(rsdsfm_ctx* ctx, const double* d_q9, const double* d_u9, const double* d_alpha9,
                        const double* d_alpha_k9, int32_t count, int use_alpha_k, int k_sign_mode, double* d_hyp);
/* diagnostics: the same solver (one wavefront per hypothesis: count <= 2 x the device's CUs; use_cores: its SVD through the in-range
 * function cores, as inside a RANSAC) which also leaves, per hypothesis, d_probe4[4 t ..] = {sweeps of the 9x9 two-sided Jacobi SVD
 * (minimal.cc:98), rotations it performed, shader clocks of the SVD, shader clocks of the whole hypothesis} -- the launch lasts as
 * long as its slowest hypothesis (tools/svd_spread.py) */
int rsdsfm_minimal9_probe_dev(rsdsfm_ctx* ctx, const double* d_q9, const double* d_u9, const double* d_alpha9, const double* d_alpha_k9,
                              int32_t count, int use_alpha_k, int k_sign_mode, int use_cores, double* d_hyp, double* d_probe4);
size_t rsdsfm_tile_lm_state_bytes(void);   /* bytes of one per-hypothesis LM state (caller zero-fills count of them) */
size_t rsdsfm_tile_best_bytes(void);       /* bytes of the device-resident winner record                             */
int32_t rsdsfm_tile_ransac_row_size(void); /* doubles per hypothesis in an LM sums row                               */
int32_t rsdsfm_tile_ransac_batch(void);    /* max hypotheses per *_rows_dev call (128)                               */
/* round r of the speculative LM depth solves of hypotheses [0, count): shard sums -> d_rows[count][row_size].
 * The rows are opaque to the caller except for how they combine across shards: the decide stage adds the SUM slots in rank order
 * and takes the maximum of the four max|J r| slots.  Those four slots do not carry gradient norms: they hold 1.0 when some pixel
 * of the shard is above Ceres' gradient tolerance (1e-10) and 0.0 otherwise -- the only thing the trust-region loop ever asks of
 * them -- so a custom transport must all-gather (or max-reduce) them as they are and must not rescale them. */
int rsdsfm_tile_ransac_lm_rows_dev(rsdsfm_ctx* ctx, const double* d_q2n, const double* d_u2n, const double* d_alpha_n,
                                   const double* d_alpha_k_n, int64_t n_shard, const double* d_hyp, int32_t count,
                                   const void* d_states, int32_t round, double tolerance, double* d_rows);
/* gathered rows [nranks][count][row_size] -> trust-region decisions; d_flags[0] = hypotheses still running (cleared
 * here), d_flags[1] += hypotheses that need the separate score pass (caller clears it once per batch) */
int rsdsfm_tile_ransac_decide_dev(rsdsfm_ctx* ctx, const double* d_rows_all, int32_t nranks, int32_t count, void* d_states,
                                  int64_t n_total, int32_t round, int32_t* d_flags, int32_t* d_scored, double* d_trial_count,
                                  double* d_trial_err);
/* inlier scores (minimal.cc:255-275) of the not yet scored hypotheses: shard sums -> d_rows[count][2] */
int rsdsfm_tile_ransac_score_rows_dev(rsdsfm_ctx* ctx, const double* d_q2n, const double* d_u2n, const double* d_alpha_n,
                                      const double* d_alpha_k_n, int64_t n_shard, const double* d_hyp, int32_t count,
                                      const void* d_states, int depth_mode, double tolerance, const int32_t* d_scored_or_null,
                                      double* d_rows);
int rsdsfm_tile_ransac_score_merge_dev(rsdsfm_ctx* ctx, const double* d_rows_all, int32_t nranks, int32_t count,
                                       const int32_t* d_scored_or_null, double* d_trial_count, double* d_trial_err);
/* the reference's best-trial rule (minimal.cc:278-285) over all trials -> d_best */
int rsdsfm_tile_ransac_pick_dev(rsdsfm_ctx* ctx, const double* d_trial_count, const double* d_trial_err, int32_t iterations,
                                const double* d_hyp, void* d_best);
/* dense 1/depth + mask of the winner on the shard and its order-preserving compaction (minimal.cc:291-305).
 * Synchronises.  out->num_inliers is the SHARD's count; best_trial, w, v, k, inlier_error are the global winner's.
 * d_inlier_idx holds shard-local indices.  The out arrays of `out` are ignored (pass the device arrays explicitly). */
int rsdsfm_tile_ransac_final_dev(rsdsfm_ctx* ctx, const double* d_q2n, const double* d_u2n, const double* d_alpha_n,
                                 const double* d_alpha_k_n, int64_t n_shard, void* d_best, const void* d_states, int depth_mode,
                                 double tolerance, double* d_inv_depth_n, uint8_t* d_mask_n, int64_t* d_inlier_idx,
                                 double* d_inliers3m, double* d_out_alpha, double* d_out_alpha_k, rsdsfm_ransac_out* out);
int64_t rsdsfm_tile_ransac_global_inliers(rsdsfm_ctx* ctx, const void* d_best); /* synchronises; -1 on error */
/* joint refinement on the shard's inliers.  begin opens a session on the context (flow = the shard's flattened u,
 * inlier_idx shard-local with RSDSFM_FLOW_GATHERED; with RSDSFM_FLOW_COMPAT_RANK -- the reference's rank-indexed flow,
 * main.cc:457 -> nonlinearRefinement.cc:209-212 -- d_flow holds the columns of the GLOBAL flow list at this shard's global
 * inlier ranks, one per inlier, fetched by the caller from the shards in front of it; d_inlier_idx may then be NULL);
 * stage 0 = iteration-zero sums, then per LM iteration stage 1
 * (Schur sums -> reduced solve) and stage 2 (back-substitution sums -> accept / reject / converge):
 *   rows_dev(stage) -> all-gather -> apply_dev(stage) on every rank;  poll (synchronises) reads the state machine
 * (summary->termination == -1 while running); finish writes (x, y, 1/rho) and closes the session. */
int rsdsfm_tile_refine_begin_dev(rsdsfm_ctx* ctx, const double* d_flow2n, int64_t n_flow, int64_t m_shard, const double* d_inl3m,
                                 const double* d_alpha_m, const double* d_alpha_k_m, const int64_t* d_inlier_idx,
                                 const double v_in[3], const double w_in[3], double k_in, int const_acceleration,
                                 int flow_index_mode);
int32_t rsdsfm_tile_refine_row_size(int const_acceleration, int32_t stage);
int rsdsfm_tile_refine_rows_dev(rsdsfm_ctx* ctx, int32_t stage, double* d_row);
int rsdsfm_tile_refine_apply_dev(rsdsfm_ctx* ctx, int32_t stage, const double* d_rows_all, int32_t nranks, int64_t m_total);
int rsdsfm_tile_refine_poll(rsdsfm_ctx* ctx, double v_out[3], double w_out[3], double* k_out, rsdsfm_lm_summary* summary);
int rsdsfm_tile_refine_finish_dev(rsdsfm_ctx* ctx, double* d_inl_out3m);
/* sign fix (main.cc:466-481) + depth map (main.cc:499-508) of one slab: zsum_dev leaves the shard's sum of z on the
 * device; depth_map_dev takes the gathered [nranks] sums, decides the flip from the global mean, flips the shard's z
 * in place and writes the column-major [slab_cols][rows] slab of the map.  Synchronises (flip flag, v). */
int rsdsfm_tile_zsum_dev(rsdsfm_ctx* ctx, const double* d_inl3m, int64_t m_shard, double* d_zsum1);
int rsdsfm_tile_depth_map_dev(rsdsfm_ctx* ctx, double* d_inl3m, int64_t m_shard, const double* d_zsums_all, int32_t nranks,
                              int64_t m_total, double v_inout[3], double fx, double fy, double cx, double cy, int32_t rows,
                              int32_t col0, int32_t slab_cols, double* d_depth_slab, int32_t* d_xs_or_null,
                              int32_t* d_ys_or_null, int* flipped);

/* ---- the column-tiled whole solve driven from inside the library (BASELINE configs[3]) ------------------
 * One process per GPU, one context per process.  Every rank passes ITS column slab of the flow image (slab bounds:
 * rsdsfm_tiled_slab_bounds) and all ranks return the same pose and the full depth map: the call sequence of the
 * rsdsfm_tile_* stages above with the collectives issued on the context's stream -- small all-gathers of the stage rows in
 * rank order, one exact all-reduce of the 9 T sampled points, ONE all-gather of the depth-map slabs -- and 4-5 host
 * synchronisations per solve.  Transport: RCCL (resolved with dlopen at the first rsdsfm_dist_* call; the library loads
 * without it), with the communicator created here from a shared unique id, adopted from the caller, or replaced by
 * caller-provided collectives. */
#define RSDSFM_DIST_ID_BYTES 128
/* rank 0 creates the id (ncclGetUniqueId) and shares its 128 bytes with the other ranks over any channel */
int rsdsfm_dist_unique_id(void* id_128_bytes);
/* collective over the ranks: ncclCommInitRank on the context's device */
int rsdsfm_dist_init(rsdsfm_ctx* ctx, int32_t nranks, int32_t rank, const void* id_128_bytes);
/* use a communicator the caller owns (an ncclComm_t of the same RCCL); it is not destroyed with the context */
int rsdsfm_dist_adopt(rsdsfm_ctx* ctx, void* nccl_comm, int32_t nranks, int32_t rank);
/* caller-provided collectives instead of RCCL.  all_gather: `bytes_per_rank` bytes from d_send into d_recv[rank * bytes_per_rank]
 * of every rank, in rank order; d_send may alias d_recv + rank * bytes_per_rank.  all_reduce: in-place sum of `count` doubles.
 * Both must be ordered after the work already enqueued on `hip_stream` and before what is enqueued next; return 0 on success. */
typedef int (*rsdsfm_all_gather_fn)(void* user, const void* d_send, void* d_recv, size_t bytes_per_rank, void* hip_stream);
typedef int (*rsdsfm_all_reduce_sum_f64_fn)(void* user, double* d_buf, size_t count, void* hip_stream);
int rsdsfm_dist_set_transport(rsdsfm_ctx* ctx, int32_t nranks, int32_t rank, rsdsfm_all_gather_fn all_gather_fn,
                              rsdsfm_all_reduce_sum_f64_fn all_reduce_fn, void* user);
int rsdsfm_dist_finalize(rsdsfm_ctx* ctx);
/* slab of `rank`: image columns [col0, col0 + slab_cols); stride_cols = ceil(cols / nranks) is the common slab stride */
int rsdsfm_tiled_slab_bounds(int32_t cols, int32_t nranks, int32_t rank, int32_t* col0, int32_t* slab_cols, int32_t* stride_cols);

typedef struct rsdsfm_tiled_info {
    int32_t nranks, rank, col0, slab_cols;
    int64_t shard_points, shard_inliers;          /* of this rank's slab                                              */
    int32_t host_syncs, collectives, ransac_rounds;       /* diagnostics of the call                                   */
    int32_t path_flags; /* bit 0: the ranks went ahead on the point counts of a dense frame instead of waiting for the counts exchange
                         * (the previous solve of this shape on the communicator succeeded everywhere, all its slabs dense); bit 1: that
                         * assumption did not hold for this frame and the solve started over through the counts / status exchange; bit 2:
                         * the RANSAC started over with the standard functions (an argument outside the range of the in-range function cores
                         * on some rank; counted by rsdsfm_ransac_restarts as well); bit 3: the RANSAC started over iterate by iterate because a
                         * global guard of the analytic LM trajectory tripped (a tie; counted by rsdsfm_lma_restarts); bit 4: the refinement's start, its
                         * first chunk and the depth-map stage went behind the RANSAC's speculated final stage, from the device-resident winner,
                         * and counted (on typical data the call then waits for the GPU once); bits 8-23: exchanges the refinement's LM iterations took
                         * (one per iteration + one in front of the first + one behind every iteration whose speculated Schur sums did
                         * not apply: a rejected step, or an accepted one of quality < 0.937) */
} rsdsfm_tiled_info;

/* d_img_slab: row-major [rows][slab_cols][2] slab of this rank (DEVICE); cols = width of the WHOLE image.  params->
 * flow_index_mode as in rsdsfm_solve_frame_dev: 0 = RSDSFM_FLOW_COMPAT_RANK, the reference's default (the i-th inlier of the GLOBAL
 * inlier list reads column i of the GLOBAL flow list, main.cc:457 -> nonlinearRefinement.cc:209-212: the columns a rank needs sit on
 * the slabs in front of it and are exchanged inside the call -- one all-gather of the heads of the slabs' flow lists, skipped when
 * every needed column is local), 1 = RSDSFM_FLOW_GATHERED.  d_depth_map: rows x cols column-major (DEVICE), the full map on every rank.
 * result: n_points / num_inliers are GLOBAL counts; d_inliers / d_inlier_idx / d_scanline describe this rank's slab (info->
 * shard_inliers entries, indices local to the slab's point list).  Without a communicator (no rsdsfm_dist_* call) the context
 * is a single rank.  info may be NULL. */
int rsdsfm_solve_frame_tiled_dev(rsdsfm_ctx* ctx, const double* d_img_slab, int32_t rows, int32_t cols, double fx, double fy,
                                 double cx, double cy, double gamma, const rsdsfm_frame_params* params,
                                 double* d_depth_map_colmajor, double* d_R_rows9_or_null, double* d_t_rows3_or_null,
                                 rsdsfm_frame_result* result, rsdsfm_tiled_info* info_or_null);

/* The row-tiled DENSE DEPTH solve (BASELINE configs[3]: "row-tiled ... with an all-gather of the depth map"): minimal::
 * estimateInverseDepths (minimal.cc:170-306) of one frame whose flattened point list is sharded over the ranks by contiguous index
 * ranges.  rsdsfm_tiled_shard_bounds: rank r owns points [i0, i0 + count) with i0 = min(n, r * stride), stride = ceil(n / nranks)
 * rounded up to even (16-byte aligned shard starts).  Host only. */
int rsdsfm_tiled_shard_bounds(int64_t n, int32_t nranks, int32_t rank, int64_t* i0, int64_t* count, int64_t* stride);
/* d_*_shard: this rank's slice of the point arrays (DEVICE; q, u: [count][2], alpha, alpha_k: [count]); d_inv_depth: n_total doubles
 * (DEVICE), the FULL inverse-depth vector on every rank.  Ceres-LM mode all-gathers one row of LM sums per rank per launch (rank
 * order: every rank adds the same numbers in the same order, results are bit-identical to the single-context solve's decisions)
 * and synchronises with the host once in the common case; both modes end with ONE all-gather of the shards.  summary (may be
 * NULL) as rsdsfm_estimate_inverse_depths; info (may be NULL): shard_points, host_syncs, collectives, ransac_rounds = LM kernel
 * launches of this rank.  Without a communicator the context is a single rank. */
int rsdsfm_estimate_inverse_depths_tiled_dev(rsdsfm_ctx* ctx, const double* d_q_shard, const double* d_u_shard, int64_t n_total,
                                             const double v[3], const double w[3], double k, const double* d_alpha_shard,
                                             const double* d_alpha_k_shard, int depth_mode, double* d_inv_depth,
                                             rsdsfm_lm_summary* summary_or_null, rsdsfm_tiled_info* info_or_null);

/* ---- consumers of the solve's output ----------------------------------------------------------
 * Images are 8-bit BGR, row-major rows x cols x 3 (cv::Mat CV_8UC3 as the reference holds them); the depth map is the
 * column-major rows x cols array of rsdsfm_depth_map; R / t the per-scanline table of rsdsfm_pose_table. */
/* RsFrame::backProject / backProjectGs (rsframe.cc:803-878): forward splat of the rolling-shutter image into the
 * global-shutter image; pixels of colour (1,1,1) are skipped; of several source pixels hitting one target the one
 * latest in the reference's scan (y outer, x inner) wins.  coords3d (may be NULL): rows x cols x 3 floats, world
 * point of every processed pixel (RsFrame::get3dCoordinates), 0 for skipped pixels. */
int rsdsfm_back_project(rsdsfm_ctx* ctx, const uint8_t* image_bgr, const double* depth_map_colmajor, const double* R_rows9,
                        const double* t_rows3, double fx, double fy, double cx, double cy, int32_t rows, int32_t cols,
                        int mode, int q5_mode, uint8_t* gs_image_bgr, float* coords3d_or_null);
int rsdsfm_back_project_dev(rsdsfm_ctx* ctx, const uint8_t* d_image_bgr, const double* d_depth_map_colmajor,
                            const double* d_R_rows9, const double* d_t_rows3, double fx, double fy, double cx, double cy,
                            int32_t rows, int32_t cols, int mode, int q5_mode, uint8_t* d_gs_image_bgr,
                            float* d_coords3d_or_null);
/* Camera::interpolateCrackyImage (camera.cc:753-774; main.cc:523 calls it with offset 1).  in and out must not alias. */
int rsdsfm_interpolate_cracky(rsdsfm_ctx* ctx, const uint8_t* image_in_bgr, int32_t rows, int32_t cols, int32_t offset,
                              uint8_t* image_out_bgr);
int rsdsfm_interpolate_cracky_dev(rsdsfm_ctx* ctx, const uint8_t* d_image_in_bgr, int32_t rows, int32_t cols, int32_t offset,
                                  uint8_t* d_image_out_bgr);
/* the 8-bit depth image of evaluateSingleRun (main.cc:480-509): depth_est(y, x) = 10 + int((z - z_min) * 244 /
 * (z_max - z_min)), row-major rows x cols, 0 where no inlier lands; inliers = 3 x m (x, y, z) AFTER the sign fix. */
int rsdsfm_depth_preview(rsdsfm_ctx* ctx, const double* inliers3m, int64_t m, double fx, double fy, double cx, double cy,
                         int32_t rows, int32_t cols, uint8_t* depth_est);
int rsdsfm_depth_preview_dev(rsdsfm_ctx* ctx, const double* d_inliers3m, int64_t m, double fx, double fy, double cx, double cy,
                             int32_t rows, int32_t cols, uint8_t* d_depth_est);
/* main.cc:480-523 in ONE call on device buffers: the 8-bit depth image (rsdsfm_depth_preview_dev), the back projection
 * (rsdsfm_back_project_dev) and the crack interpolation of its result (rsdsfm_interpolate_cracky_dev) -- the same kernels' code and
 * the same bytes, in two launches instead of five: the two claim passes share a launch, and the write pass of the back projection forms
 * the interpolated image from its own tile + halo next to the depth image's write pass (offset <= 2 and cols % 4 == 0; three launches
 * otherwise) -- each of the five is short enough for the launch floor to show.  d_gs_image_bgr and d_fixed_image_bgr must not alias. */
int rsdsfm_rectify_frame_dev(rsdsfm_ctx* ctx, const double* d_inliers3m, int64_t m, const uint8_t* d_image_bgr,
                             const double* d_depth_map_colmajor, const double* d_R_rows9, const double* d_t_rows3, double fx, double fy,
                             double cx, double cy, int32_t rows, int32_t cols, int mode, int q5_mode, int32_t offset, uint8_t* d_depth_est,
                             uint8_t* d_gs_image_bgr, float* d_coords3d_or_null, uint8_t* d_fixed_image_bgr);

/* ---- ground-truth flow between two rolling-shutter frames ---------------------------------------
 * Camera::calculateTrueFlow (camera.cc:209-249) + RsFrame::calculateImageCoordinatesRsFrame (rsframe.cc:740-768).
 * world_{x,y,z}: the unprojection maps of frame 1 (Eigen MatrixXd, column-major rows x cols; all-zero = void pixel);
 * R2 / t2: pose table of the rows2 scanlines of frame 2 (as rsdsfm_pose_table lays it out); flow: rows x cols x 2
 * doubles row-major (cv::Mat_<cv::Point_<double>>); best_row (may be NULL): winning scanline per pixel, -1 = void.
 * rows2 must be >= 1 (the reference would read scanline 0 of an empty frame).
 * The search is exact: by default scanlines are visited in blocks of 32 and a block is skipped only when interval arithmetic over its
 * pose entries proves that none of its scanlines can reach the best |y - i| found so far (winners and flows bit-identical to
 * visiting every scanline).  rsdsfm_set_true_flow_search: 0 = automatic (pruned from 96 scanlines on; default), 1 = the exhaustive
 * loop (the reference's own), 2 = pruned at any size. */
int rsdsfm_set_true_flow_search(rsdsfm_ctx* ctx, int mode);
int rsdsfm_true_flow(rsdsfm_ctx* ctx, const double* world_x, const double* world_y, const double* world_z, int32_t rows,
                     int32_t cols, const double* R2_rows9, const double* t2_rows3, int32_t rows2, double fx, double fy,
                     double cx, double cy, int q5_mode, double* flow, int32_t* best_row_or_null);
int rsdsfm_true_flow_dev(rsdsfm_ctx* ctx, const double* d_world_x, const double* d_world_y, const double* d_world_z,
                         int32_t rows, int32_t cols, const double* d_R2_rows9, const double* d_t2_rows3, int32_t rows2,
                         double fx, double fy, double cx, double cy, int q5_mode, double* d_flow,
                         int32_t* d_best_row_or_null);

/* ---- accuracy metrics ---------------------------------------------------------------------------
 * rotation / translation error of evaluateVelocities (errorMeasure.cpp:178-186): host-only scalar math, no context. */
int rsdsfm_velocity_errors(const double w_est[3], const double v_est[3], const double w_true[3], const double v_true[3],
                           double* w_error, double* v_error);

typedef struct rsdsfm_reprojection_stats {
    double scale;             /* mean accepted ratio estimate / truth (camera.cc:659-667)                */
    double mean_error;        /* Camera::meanReprojectionError's return value (camera.cc:690)            */
    double sum_error;
    int64_t number_outliers;  /* coordinates with |ratio| > 10                                           */
    int64_t scale_inliers;    /* ratios that are non-zero and not NaN                                    */
    int64_t error_inliers;    /* points with finite coordinates and error < 50                           */
} rsdsfm_reprojection_stats;

/* Camera::meanReprojectionError (camera.cc:594-691) and, when error_image is given, Camera::createErrorImage
 * (camera.cc:503-591; error_image(y, x) = (char)int(error * 255 / max_norm + 0.5), rows x cols bytes row-major).
 * est_coords: rows x cols x 3 floats (what rsdsfm_back_project writes); gt_depth / est_depth: column-major rows x cols
 * (a ground-truth depth of exactly 0 falls back to the estimated depth, the reference's planeToSpace default argument,
 * rsframe.cc:657); R_abs / t_abs: ABSOLUTE pose of every scanline after RsFrame::relocatePose.  Synchronises. */
int rsdsfm_reprojection_error(rsdsfm_ctx* ctx, const float* est_coords, const double* gt_depth_colmajor,
                              const double* est_depth_colmajor, const double* R_abs_rows9, const double* t_abs_rows3, double fx,
                              double fy, double cx, double cy, int32_t rows, int32_t cols, double max_norm,
                              rsdsfm_reprojection_stats* stats, uint8_t* error_image_or_null);
int rsdsfm_reprojection_error_dev(rsdsfm_ctx* ctx, const float* d_est_coords, const double* d_gt_depth_colmajor,
                                  const double* d_est_depth_colmajor, const double* d_R_abs_rows9, const double* d_t_abs_rows3,
                                  double fx, double fy, double cx, double cy, int32_t rows, int32_t cols, double max_norm,
                                  rsdsfm_reprojection_stats* stats, uint8_t* d_error_image_or_null);

#ifdef __cplusplus
}
#endif
#endif /* RSDSFM_H */
