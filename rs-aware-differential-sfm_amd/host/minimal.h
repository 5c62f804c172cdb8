// minimal.h -- drop-in for the reference's src/minimal.h (namespace minimal, structs Velocities / RansacValues):
// same names, argument meaning and return-by-value convention, implemented over the MI355X C ABI
// (include/rsdsfm.h).  Header-only; link librsdsfm_hip.so.
//
// Differences from the reference, all documented in DESIGN.md:
//  * RansacValues carries inlier_idx (the reference cannot express which points are inliers) and rsdsfm_tag (the
//    device-resident copy of the result: nonLinearRefinement on an unmodified RansacValues does not upload it again);
//  * the sampler is the reference's partial Fisher-Yates driven by splitmix64(rsdsfm::ransac_seed()) instead of
//    srand(time(NULL)) / rand()  (quirk Q1) -- set the seed with rsdsfm::set_ransac_seed();
//  * failures throw std::runtime_error instead of producing garbage (the reference has no error path);
//  * show_messages prints what the reference prints (Ceres' BriefReport line, solve time, "Finished i RANSAC trials ...") to
//    std::cout; the T trials of a RANSAC run as one batch on the GPU, so their lines appear after the batch and the
//    per-trial depth-solve reports inside the RANSAC loop (minimal.cc:254) are not printed.
#ifndef RSDSFM_HOST_MINIMAL_H
#define RSDSFM_HOST_MINIMAL_H

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <iostream>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/rsdsfm.h"
#include "rsdsfm_eigen_lite.hpp"

namespace rsdsfm {

// process-wide default context (device 0, private stream): the reference's free functions take no context
inline rsdsfm_ctx*& default_context_slot() {
    static rsdsfm_ctx* ctx = nullptr;
    return ctx;
}
inline rsdsfm_ctx* default_context() {
    rsdsfm_ctx*& ctx = default_context_slot();
    if (!ctx) {
        int rc = rsdsfm_create(&ctx, 0, nullptr);
        if (rc != RSDSFM_OK) throw std::runtime_error("rsdsfm_create failed (" + std::to_string(rc) + "): no usable MI355X device; there is no CPU fallback");
    }
    return ctx;
}
inline void set_default_context(rsdsfm_ctx* ctx) { default_context_slot() = ctx; }
inline uint64_t& ransac_seed() {
    static uint64_t seed = 0x5EED0000ULL;
    return seed;
}
inline void set_ransac_seed(uint64_t s) { ransac_seed() = s; }
inline int& depth_mode() {
    static int mode = RSDSFM_DEPTH_CERES_LM;  // what the reference's Ceres solve produces
    return mode;
}
// the line ceres::Solver::Summary::BriefReport() returns (Ceres 1.14 solver.cc: iterations = successful + unsuccessful steps)
inline std::string brief_report(const rsdsfm_lm_summary& s) {
    const char* term = (s.termination == RSDSFM_TERM_MAX_ITER) ? "NO_CONVERGENCE" : (s.termination == RSDSFM_TERM_FAILURE) ? "FAILURE" : "CONVERGENCE";
    char buf[256];
    std::snprintf(buf, sizeof(buf), "Ceres Solver Report: Iterations: %d, Initial cost: %e, Final cost: %e, Termination: %s",
                  (int)(s.num_successful_steps + s.num_unsuccessful_steps), s.initial_cost, s.final_cost, term);
    return buf;
}
struct StopWatch {
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    double seconds() const { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }
};
inline void check(int rc, const char* what) {
    if (rc != RSDSFM_OK) throw std::runtime_error(std::string(what) + " failed (" + std::to_string(rc) + "): " + rsdsfm_last_error(default_context()));
}

}  // namespace rsdsfm

/** return value of calculateVelocities (reference minimal.h:41-51) */
struct Velocities {
    rsdsfm::lite::Vector3d w;
    rsdsfm::lite::Vector3d v;
    double k;
    Velocities(rsdsfm::lite::Vector3d w_init, rsdsfm::lite::Vector3d v_init) : w(w_init), v(v_init), k(0) {}
    Velocities(rsdsfm::lite::Vector3d w_init, rsdsfm::lite::Vector3d v_init, double k_init) : w(w_init), v(v_init), k(k_init) {}
};

/** return value of ransac (reference minimal.h:57-76) + inlier_idx */
struct RansacValues {
    int num_inliers;
    rsdsfm::lite::Array3Xd inliers;
    rsdsfm::lite::VectorXd beta;  // aliases alpha, as in the reference (quirk Q9)
    rsdsfm::lite::VectorXd alpha;
    rsdsfm::lite::VectorXd alpha_k;
    rsdsfm::lite::Vector3d w;
    rsdsfm::lite::Vector3d v;
    double k;
    std::vector<int64_t> inlier_idx;  // index of each inlier in the arrays handed to ransac()
    // names the device-resident copy of these arrays that minimal::ransac left behind (rsdsfm_last_ransac_tag, include/rsdsfm.h): handed to
    // nonLinearRefinement UNMODIFIED -- what the reference's call sites do -- the refinement starts from that copy instead of uploading the
    // arrays again; set it to 0 after editing inliers / alpha / alpha_k / inlier_idx in place
    uint64_t rsdsfm_tag = 0;

    /** the reference's 5-argument form (minimal.h:67-70): beta, alpha and alpha_k all start as beta_init, k = 0 */
    RansacValues(int num_inliers_init, rsdsfm::lite::Array3Xd inliers_init, rsdsfm::lite::VectorXd beta_init, rsdsfm::lite::Vector3d w_init,
                 rsdsfm::lite::Vector3d v_init)
        : num_inliers(num_inliers_init), inliers(inliers_init), beta(beta_init), alpha(beta_init), alpha_k(beta_init), w(w_init), v(v_init), k(0) {}

    RansacValues(int num_inliers_init, rsdsfm::lite::Array3Xd inliers_init, rsdsfm::lite::VectorXd alpha_init,
                 rsdsfm::lite::VectorXd alpha_k_init, rsdsfm::lite::Vector3d w_init, rsdsfm::lite::Vector3d v_init, double k_init)
        : num_inliers(num_inliers_init), inliers(inliers_init), beta(alpha_init), alpha(alpha_init), alpha_k(alpha_k_init),
          w(w_init), v(v_init), k(k_init) {}
};

namespace minimal {

using rsdsfm::lite::Array2Xd;
using rsdsfm::lite::Array3Xd;
using rsdsfm::lite::ArrayXd;
using rsdsfm::lite::Vector3d;
using rsdsfm::lite::VectorXd;

/** reference minimal.cc:36-177 */
inline Velocities calculateVelocities(const Array2Xd& q, const Array2Xd& u, const ArrayXd& alpha, const ArrayXd& alpha_k,
                                      bool use_alpha_k) {
    if (q.cols() != 9 || u.cols() != 9 || alpha.size() != 9 || alpha_k.size() != 9) throw std::runtime_error("calculateVelocities needs exactly 9 points");
    Vector3d w, v;
    double k = 0;
    rsdsfm::check(rsdsfm_calculate_velocities(rsdsfm::default_context(), q.data(), u.data(), alpha.data(), alpha_k.data(), 1,
                                              use_alpha_k ? 1 : 0, RSDSFM_K_COMPAT, w.data(), v.data(), &k),
                  "rsdsfm_calculate_velocities");
    return Velocities(w, v, k);
}

/** reference minimal.cc:179-186 */
inline ArrayXd getAlpha(const Array2Xd& flow, double h, double gamma) {
    ArrayXd alpha(flow.cols());
    rsdsfm::check(rsdsfm_get_alpha(rsdsfm::default_context(), flow.data(), flow.cols(), h, gamma, alpha.data()), "rsdsfm_get_alpha");
    return alpha;
}

/** reference minimal.cc:188-197 */
inline ArrayXd getAlphaK(const Array2Xd& q, const Array2Xd& flow, double h, double gamma) {
    ArrayXd alpha_k(q.cols());
    rsdsfm::check(rsdsfm_get_alpha_k(rsdsfm::default_context(), q.data(), flow.data(), q.cols(), h, gamma, alpha_k.data()), "rsdsfm_get_alpha_k");
    return alpha_k;
}

/** reference minimal.cc:209-306 */
inline RansacValues ransac(const Array2Xd& q, const Array2Xd& u, const ArrayXd& alpha, const ArrayXd& alpha_k, bool use_alpha_k,
                           int iterations, double tolerance, bool show_messages) {
    const long n = q.cols();
    Array3Xd inl(3, n);
    VectorXd a(n), ak(n);
    std::vector<int64_t> idx((size_t)n);
    rsdsfm_ransac_out out = {};
    out.inliers = inl.data();
    out.alpha = a.data();
    out.alpha_k = ak.data();
    out.inlier_idx = idx.data();
    std::vector<int64_t> trial_count(show_messages && iterations > 0 ? (size_t)iterations : 0);
    std::vector<double> trial_err(trial_count.size());
    if (!trial_count.empty()) {
        out.trial_count = trial_count.data();
        out.trial_err = trial_err.data();
    }
    rsdsfm::check(rsdsfm_ransac(rsdsfm::default_context(), q.data(), u.data(), alpha.data(), alpha_k.data(), n, use_alpha_k ? 1 : 0,
                                iterations, tolerance, nullptr, rsdsfm::ransac_seed(), rsdsfm::depth_mode(), RSDSFM_K_COMPAT, &out),
                  "rsdsfm_ransac");
    if (show_messages) {  // minimal.cc:278-288: best = more inliers, ties by the smaller error sum
        int64_t best = 0;
        double best_err = 0.0;
        for (size_t i = 0; i < trial_count.size(); ++i) {
            if (trial_count[i] > best || (trial_count[i] == best && trial_err[i] < best_err)) {
                best = trial_count[i];
                best_err = trial_err[i];
            }
            std::cout << "Finished " << i + 1 << " RANSAC trials. The current maximum number of inliers is " << best << "." << std::endl;
        }
    }
    const long m = (long)out.num_inliers;
    inl.conservativeResize(3, m);
    a.conservativeResize(m);
    ak.conservativeResize(m);
    idx.resize((size_t)m);
    RansacValues r((int)m, inl, a, ak, Vector3d(out.w[0], out.w[1], out.w[2]), Vector3d(out.v[0], out.v[1], out.v[2]), out.k);
    r.inlier_idx = idx;
    (void)rsdsfm_last_ransac_tag(rsdsfm::default_context(), &r.rsdsfm_tag, nullptr);
    return r;
}

/** reference minimal.cc:199-202 (constant velocity: alpha_k unused) */
inline RansacValues ransac(const Array2Xd& q, const Array2Xd& u, const ArrayXd& alpha, int iterations, double tolerance,
                           bool show_messages) {
    return ransac(q, u, alpha, alpha, false, iterations, tolerance, show_messages);
}

/** reference minimal.cc:204-207 (constant acceleration) */
inline RansacValues ransac(const Array2Xd& q, const Array2Xd& u, const ArrayXd& alpha, const ArrayXd& alpha_k, int iterations,
                           double tolerance, bool show_messages) {
    return ransac(q, u, alpha, alpha_k, true, iterations, tolerance, show_messages);
}

}  // namespace minimal

#endif
