/*
 * rsdsfm_oracle.h -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 *
 * Plain-C restatement of the rolling-shutter differential-SfM hot path of
 * ThomasZiegler/RS-aware-differential-SfM (src/minimal.cc, src/nonlinearRefinement.cc and
 * the caller-side glue).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may load this library, and only as the checker / reported CPU baseline.
 *
 * PARITY UNPINNED: the reference ships no tests / golden vectors and cannot be built in this
 * image (it needs Ceres 1.14.0, Eigen 3.3.4, OpenCV 3.4.0 and Boost 1.58, none of which are
 * on disk).  The third-party arithmetic (Eigen JacobiSVD / eigen-solvers, Ceres
 * trust-region LM + DENSE_SCHUR) is restated from the published algorithms of those pinned
 * versions.  What pins this oracle instead: analytic known-answer data and an independent
 * numpy/scipy cross-check (tests/golden/make_golden.py).
 *
 * Layout conventions (those of the reference boundary): all floating data fp64; a
 * "2xN" array is N interleaved (x,y) pairs (Eigen column-major Array2Xd), a "3xN" array is
 * N interleaved triples.
 */
#ifndef RSDSFM_ORACLE_H
#define RSDSFM_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Ceres-style termination types of the emulated trust-region loop. */
enum {
    RSO_TERM_GRADIENT = 0,  /* CONVERGENCE: gradient max-norm <= 1e-10                    */
    RSO_TERM_PARAMETER = 1, /* CONVERGENCE: step norm <= 1e-8 (|x| + 1e-8)                  */
    RSO_TERM_FUNCTION = 2,  /* CONVERGENCE: |cost change| <= 1e-6 cost                      */
    RSO_TERM_MAX_ITER = 3,  /* NO_CONVERGENCE: 50 iterations                                */
    RSO_TERM_FAILURE = 4,   /* 5 consecutive invalid steps / linear-solver failure          */
    RSO_TERM_MIN_RADIUS = 5 /* CONVERGENCE: trust-region radius < 1e-32                     */
};

typedef struct rso_lm_summary {
    int32_t num_iterations;       /* index of the last iteration started (Ceres iterations.size()-1) */
    int32_t num_successful_steps; /* accepted steps (iteration 0 not counted)               */
    int32_t num_unsuccessful_steps;
    int32_t termination;
    double initial_cost;
    double final_cost;
    double final_radius;
} rso_lm_summary;

/* minimal.cc:179-186 */
void rso_get_alpha(const double* flow_px2n, int64_t n, double h, double gamma, double* alpha_n);
/* minimal.cc:188-197 */
void rso_get_alpha_k(const double* q_px2n, const double* flow_px2n, int64_t n, double h, double gamma,
                     double* alpha_k_n);

/* minimal.cc:36-177.  k_sign_mode: 0 = compat (reference quirk Q4: eig(P*inv(Pk))), 1 = fixed (negated).
 * returns 0, or <0 when no real eigenvalue k was found (k is left +inf in the reference). */
int rso_calculate_velocities(const double q[18], const double u[18], const double alpha[9],
                             const double alpha_k[9], int use_alpha_k, int k_sign_mode, double w[3],
                             double v[3], double* k);

/* nonlinearRefinement.cc:32-52 (double instantiation of RsResidual::operator()) */
void rso_residual(double x, double y, double ux, double uy, double alpha, double alpha_k,
                  const double v[3], const double w[3], double k, double rho, double r[2]);

/* nonlinearRefinement.cc:109-180.  mode 0: exact per-pixel least-squares optimum (one undamped
 * Gauss-Newton step from rho=1); mode 1: emulation of the Ceres 1.14 trust-region LM the reference runs;
 * mode 2: the same trust-region loop on the closed-form trajectory (rso_lma_trial below). */
int rso_estimate_inverse_depths(const double* q2n, const double* u2n, int64_t n, const double v[3],
                                const double w[3], double k, const double* alpha_n,
                                const double* alpha_k_n, int mode, double* inv_depth_n,
                                rso_lm_summary* summary);

/* ---- mode 2: the ANALYTIC LM TRAJECTORY, checker of the HIP library's default arithmetic for the dense depth solves --------------
 * (see the comment block in rsdsfm_oracle.c).  rso_estimate_inverse_depths(mode 2) and rso_ransac(depth_mode 2) go through
 * rso_lma_trial: one depth solve on the closed-form trajectory and, with tol >= 0, its inlier score {count, sum of inlier errors,
 * mask}; tol < 0: no scoring.  A solve whose guards trip is computed by mode 1 (stats->fallback).  study != 0 also runs mode 1's
 * exact recurrence for EVERY pixel and reports how far the two arithmetics are apart (test infrastructure of the guards). */
typedef struct rso_lma_stats {
    int64_t listed_clamped;   /* pixels on the exact recurrence because the LM diagonal's clamp may bind (guard a)            */
    int64_t listed_near;      /* pixels scored from the exact iterate because their error is within the margin of tol (guard b) */
    int32_t fallback;         /* (count of) solves a global guard sent to mode 1 (guard c)                                   */
    int32_t fallback_reason;  /* 1 infinite sum, 2 gradient, 3 model change, 4 parameter, 5 function tolerance, 6 step quality, 7 (rso_ransac) a tie in count and error sum */
    double margin_use_max;    /* study: kappa = max |e_exact - e_analytic| / ((2 + |r(1)|^2 + h) / 2) over the unlisted pixels; guard (b) holds while kappa < eta / 2 */
    double rho_diff_max;      /* study: max |rho_exact - rho_analytic| / (1 + |rho|)                                          */
    int64_t flips_unguarded;  /* study: unlisted pixels whose inlier decision differs between the arithmetics (must be 0)    */
    int64_t flips_listed;     /* study: listed pixels whose analytic decision would have differed (what the guard caught)    */
} rso_lma_stats;
int rso_lma_trial(const double* q2n, const double* u2n, const double* alpha_n, const double* alpha_k_n, int64_t n,
                  const double v[3], const double w[3], double k, double tol, double* inv_depth_n_or_null,
                  uint8_t* mask_n_or_null, int64_t* count_or_null, double* err_sum_or_null, rso_lm_summary* summary_or_null,
                  rso_lma_stats* stats_or_null, int study);
/* rho of one pixel after the FIRST LM step (radius: Ceres starts at 1e4) from rho = 1 under mode 1's arithmetic */
double rso_one_lm_step(double x, double y, double ux, double uy, double alpha, double alpha_k, const double v[3], const double w[3], double k,
                       double radius);
/* totals over the trials of the last rso_ransac(depth_mode 2) */
void rso_lma_last_stats(rso_lma_stats* out);

/* minimal.cc:255-275: per-point residual norm, inlier flag; returns count, *err_sum = sum over inliers. */
int64_t rso_score(const double* q2n, const double* u2n, const double* alpha_n, const double* alpha_k_n,
                  int64_t n, const double v[3], const double w[3], double k, const double* inv_depth_n,
                  double tol, uint8_t* mask_n, double* err_sum);

/* The reference sampler (minimal.cc:226-244) with its rand() replaced by splitmix64(seed):
 * persistent index permutation, 9 partial Fisher-Yates draws per trial. */
void rso_sample_indices(int64_t n, int32_t trials, uint64_t seed, int32_t* samples_9xT);

typedef struct rso_ransac_out {
    int64_t num_inliers;
    int32_t best_trial;
    int32_t _pad;
    double w[3], v[3], k;
    double inlier_error;
    int64_t* inlier_idx; /* [n] capacity, first num_inliers valid                          */
    double* inliers;     /* [3n] (x, y, z = 1/rho)                                         */
    double* alpha;       /* [n]                                                            */
    double* alpha_k;     /* [n]                                                            */
    uint8_t* mask;       /* [n]                                                            */
    double* inv_depth;   /* [n] dense rho of the best trial                                */
    int64_t* trial_count; /* [T] or NULL                                                   */
    double* trial_err;    /* [T] or NULL                                                   */
    double* trial_vel;    /* [7T] (w, v, k) or NULL                                        */
    int32_t* trial_steps; /* [T] accepted LM steps of each trial's depth solve, or NULL     */
} rso_ransac_out;

/* minimal.cc:209-306 with injected samples (9 indices per trial). */
int rso_ransac(const double* q2n, const double* u2n, const double* alpha_n, const double* alpha_k_n,
               int64_t n, int use_alpha_k, int32_t iterations, double tol, const int32_t* samples_9xT,
               int depth_mode, int k_sign_mode, rso_ransac_out* out);

/* nonlinearRefinement.cc:183-252.  flow_index_mode 0 = compat (quirk Q2: flow(.,rank)), 1 = gathered
 * (flow(., inlier_idx[rank])).  inliers_out is 3xM. */
/* sensitivity study only (tests/test_oracle_sensitivity.py): replaces one recalled detail of the Ceres 1.14 loop by an alternative
 * reading (which: 0 function-tolerance placement, 1 Jacobi scaling, 2 diagonal clamp, 3 literal D^2, 4 radius rule, 5 strict <,
 * 6 the sign of Eigen's SVD null vector);
 * value 0 = the pinned oracle.  Returns 0, or -1 for an unknown switch. */
int rso_set_variant(int which, int value);
/* test diagnostics: per-iteration trace of the next rso_refine calls (rows x 8 doubles, caller NaN-fills; NULL = off) */
void rso_set_refine_trace(double* buf, int rows);
int rso_refine(const double* flow2n, int64_t n_flow, int64_t m, const double* inliers_3m,
               const double* alpha_m, const double* alpha_k_m, const int64_t* inlier_idx_or_null,
               const double v_in[3], const double w_in[3], double k_in, int const_acceleration,
               int flow_index_mode, double* inliers_out_3m, double v_out[3], double w_out[3],
               double* k_out, rso_lm_summary* summary);

/* the same solve in the product's default arithmetic (radius-factorised Schur sums, refine_rf_kernels.hip), restated; *guard_out = 0 or the
 * guard at which the product would leave that arithmetic; *resolves_out = reduced systems solved again from stored sums */
int rso_refine_rf(const double* flow2n, int64_t n_flow, int64_t m, const double* inliers_3m, const double* alpha_m, const double* alpha_k_m,
                  const int64_t* inlier_idx_or_null, const double v_in[3], const double w_in[3], double k_in, int const_acceleration,
                  int flow_index_mode, double* inliers_out_3m, double v_out[3], double w_out[3], double* k_out, rso_lm_summary* summary,
                  int32_t* guard_out, int32_t* resolves_out);
int rso_refine_rf_listed_max(void); /* test diagnostics: the longest list of clamped inliers the last rso_refine_rf met */

/* main.cc:398-444 / errorMeasure.cpp:66-111 (shrinking variant).  flow image row-major rows x cols x 2.
 * returns the number of kept points. */
int64_t rso_flatten(const double* flow_img, int32_t rows, int32_t cols, double fx, double fy, double cx,
                    double cy, double gamma, double flow_threshold, double* q2n, double* u2n,
                    double* q_px2n, double* flow_px2n);

/* main.cc:466-478: flips z and v when mean z < 0; returns 1 if flipped. */
int rso_canonicalize_sign(double* inliers_3m, int64_t m, double v[3]);

/* main.cc:495-509: x=int(fx*qx+cx+.5), y=int(fy*qy+cy+.5), depth_map(y,x)=z (col-major rows x cols).
 * out-of-image points are skipped (the reference would write out of bounds).  xs/ys may be NULL. */
void rso_scatter_depth(const double* inliers_3m, int64_t m, double fx, double fy, double cx, double cy,
                       int32_t rows, int32_t cols, double* depth_map_colmajor, int32_t* xs, int32_t* ys);

/* rsframe.cc:771-800: per-scanline relative pose, R row-major [rows][9], t [rows][3]. */
void rso_pose_table(const double v[3], const double w[3], double k, double gamma, int32_t rows,
                    double* R_rows9, double* t_rows3);

/* ---- RS -> GS rectifier and the 8-bit depth preview ---------------------------------- */
/* main.cc:480-509: 8-bit depth image.  z_min = +inf, z_max = 0 over the inliers; multiplier = 244/(z_max - z_min);
 * depth_est(y, x) = (uchar)(10 + int((z - z_min) * multiplier)), sequential (last writer wins), row-major
 * rows x cols, zero elsewhere.  Out-of-image points are skipped (the reference would write out of bounds); a
 * non-finite product (z_max == z_min) is defined as 0 here (the reference's int(NaN) is undefined behaviour). */
void rso_depth_preview(const double* inliers_3m, int64_t m, double fx, double fy, double cx, double cy, int32_t rows,
                       int32_t cols, uint8_t* depth_est_rowmajor);

/* RsFrame::backProject (rsframe.cc:803-839; mode 0) / backProjectGs (rsframe.cc:842-878; mode 1):
 * every pixel (x, y) of the BGR rolling-shutter image that is not the marker colour (1,1,1) is lifted with the depth
 * map (planeToSpace rsframe.cc:644-664), moved to the world frame with the relative pose of ITS scanline y (mode 0) or of
 * scanline 0 (mode 1) (cameraToWorldFrame rsframe.cc:712-736), brought back with the pose of scanline 0
 * (worldToCameraFrame rsframe.cc:687-709) and projected (spaceToPlane rsframe.cc:628-641); the pixel is copied to
 * gs(int(py+.5), int(px+.5)) when that lies in the image.  Sequential scan (y outer, x inner): the LAST writer wins.
 * q5_mode 0 = compat (spaceToPlane multiplies the y coordinate by f_x, quirk Q5), 1 = fixed (f_y).
 * coords3d (may be NULL): rows x cols x 3 floats, the world point of every processed pixel, 0 for skipped pixels (the
 * reference leaves those uninitialised).  depth_map is column-major rows x cols; R/t the per-scanline pose table.
 * The 4x4 homogeneous products are evaluated left to right (Eigen's coefficient-based product order, unverifiable
 * here); a non-finite or out-of-int-range projection counts as outside the image (x86 cvttsd2si semantics). */
void rso_back_project(const uint8_t* image_bgr, const double* depth_map_colmajor, const double* R_rows9,
                      const double* t_rows3, double fx, double fy, double cx, double cy, int32_t rows, int32_t cols,
                      int mode, int q5_mode, uint8_t* gs_image_bgr, float* coords3d_or_null);

/* Camera::interpolateCrackyImage (camera.cc:694-774): every black pixel (||bgr||_2 <= 15) in [offset, rows-offset) x
 * [offset, cols-offset) that has a non-black neighbour at distance `offset` (above, below, left, right) becomes the
 * average of its non-black neighbours, rounded to nearest-even and saturated (cv::saturate_cast<uchar>(double)). */
void rso_interpolate_cracky(const uint8_t* image_in_bgr, int32_t rows, int32_t cols, int32_t offset,
                            uint8_t* image_out_bgr);

/* ---- ground-truth flow between two rolling-shutter frames ------------------------------- */
/* Camera::calculateTrueFlow (camera.cc:209-249) with RsFrame::calculateImageCoordinatesRsFrame (rsframe.cc:740-768):
 * every pixel (u, v) of frame 1 has a world point W (unprojection maps, column-major rows x cols each); W is projected
 * into frame 2 with the pose of EVERY scanline i of frame 2 (worldToCameraFrame rsframe.cc:687-709, spaceToPlane
 * :628-641, quirk Q5 as in rso_back_project) and the scanline with the smallest |y_projected - i| wins (first minimum);
 * flow(v, u) = projection with that pose - (u, v).  Void pixels (||W|| == 0, computed as sqrt of the sum of squares
 * like Eigen's norm()) and projections of norm 0 give zero flow.  rows2 = scanlines of frame 2.
 * flow: rows x cols x 2 row-major (cv::Mat_<Point_<double>>); best_row (may be NULL): winning scanline, -1 for void
 * pixels.  If no scanline yields a finite displacement the reference's best_row is uninitialised; here it is 0. */
void rso_true_flow(const double* world_x, const double* world_y, const double* world_z, int32_t rows, int32_t cols,
                   const double* R2_rows9, const double* t2_rows3, int32_t rows2, double fx, double fy, double cx,
                   double cy, int q5_mode, double* flow_rowmajor, int32_t* best_row_or_null);

/* ---- accuracy metrics ------------------------------------------------------------------- */
/* errorMeasure.cpp:178-186: rotation error = || vee( (I + [w_est]x) (I + [w_true]x)^T ) ||, translation error = angle
 * between v_est and v_true (acos of the normalised dot product). */
void rso_velocity_errors(const double w_est[3], const double v_est[3], const double w_true[3], const double v_true[3],
                         double* w_error, double* v_error);

typedef struct rso_reprojection_stats {
    double scale;             /* mean of the accepted per-coordinate ratios estimate / truth (camera.cc:659-667)       */
    double mean_error;        /* sum_error / error_inliers (camera.cc:690); NaN when nothing qualifies                 */
    double sum_error;
    int64_t number_outliers;  /* coordinates with |ratio| > 10 (camera.cc:643-654)                                     */
    int64_t scale_inliers;    /* ratios that are non-zero and not NaN                                                  */
    int64_t error_inliers;    /* points with finite coordinates and error < 50                                         */
} rso_reprojection_stats;

/* Camera::meanReprojectionError (camera.cc:594-691) and Camera::createErrorImage (camera.cc:503-591) share their first
 * two passes: ground-truth world point of every pixel (planeToSpace with the ground-truth depth -- quirk: a ground-
 * truth depth of exactly 0 makes planeToSpace fall back to the ESTIMATED depth map, rsframe.cc:657 -- then
 * cameraToWorldFrame with the ABSOLUTE pose of the pixel's scanline, after relocatePose), stored as float like the
 * reference's cv::Vec3f; per-coordinate float ratios estimate / truth, outliers |ratio| > 10 zeroed, mean of the
 * rest = scale; then the Euclidean error of estimate / scale against the truth, summed where finite and < 50.
 * est_coords: rows x cols x 3 floats (RsFrame::get3dCoordinates); depth maps column-major rows x cols; R/t: absolute
 * per-scanline poses.  error_image (may be NULL): rows x cols bytes, (char)int(error * 255 / max_norm + 0.5); a
 * non-finite value is defined as 0 (undefined behaviour in the reference).  Pixels are visited x outer, y inner. */
void rso_reprojection_error(const float* est_coords, const double* gt_depth_colmajor, const double* est_depth_colmajor,
                            const double* R_abs_rows9, const double* t_abs_rows3, double fx, double fy, double cx,
                            double cy, int32_t rows, int32_t cols, double max_norm, rso_reprojection_stats* stats,
                            uint8_t* error_image_or_null);

/* exposed for direct testing of the restated third-party pieces */
void rso_jacobi_svd9(const double Z_rowmajor[81], double sv[9], double V_rowmajor[81]);
int rso_eigvals_general(const double* A_rowmajor, int n, double* re, double* im);
void rso_eig_sym3(const double S[9], double lam[3], double V_rowmajor[9]);

#ifdef __cplusplus
}
#endif
#endif
