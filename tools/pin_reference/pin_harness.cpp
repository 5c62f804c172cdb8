// pin_harness.cpp -- runs the REFERENCE's own hot-path functions on the committed synthetic inputs and dumps what they return.
//
// Part of the pinning kit (tools/pin_reference/README.md).  It exists because the development image of this repository has neither
// Ceres nor Eigen: the oracle (oracle/rsdsfm_oracle.c) restates what those libraries do from their published algorithms, and this
// harness is how someone WITH the toolchain (Ceres 1.14.0, Eigen 3.3.4: reference README.md:26-45) turns that restatement into a
// pinned one.  Nothing of the reference is copied into this repository: its two source files are compiled / included BY PATH from
// REFERENCE_DIR (CMakeLists.txt next to this file):
//
//   ${REFERENCE_DIR}/src/minimal.cc               compiled as its own translation unit, unmodified
//   ${REFERENCE_DIR}/src/nonlinearRefinement.cc   #included below, unmodified -- behind ONE macro that routes its three
//                                                 `ceres::Solve(options, &problem, &summary)` calls (nonlinearRefinement.cc:95, :162,
//                                                 :227) through ceres::PinnedSolve, which calls the real ceres::Solve with the SAME
//                                                 arguments and keeps a copy of the Summary the reference throws away (iteration
//                                                 counts, per-iteration cost / radius / step records, termination message)
//
// and minimal::ransac's `srand(time(NULL)); rand() % n_temp` (minimal.cc:230-236; quirk Q1: not reproducible as written) draws from
// the rand() defined HERE, which replays the draws that reproduce the committed sample sets through the reference's own partial
// Fisher-Yates permutation (minimal.cc:226-244) -- the reference's code path is untouched, only libc's generator is replaced.
//
// usage: pin_harness <inputs_dir> <outputs_dir> case [case ...]      (inputs: tools/pin_reference/export_inputs.py)
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <limits>
#include <map>
#include <numeric>
#include <stdexcept>
#include <string>
#include <vector>

#include "Eigen/Dense"
#include "Eigen/Eigenvalues"
#include "Eigen/Geometry"
#include <ceres/ceres.h>
#include <ceres/version.h>

#include "minimal.h"              // ${REFERENCE_DIR}/src (include path)
#include "nonlinearRefinement.h"  // ${REFERENCE_DIR}/src

namespace pin {
ceres::Solver::Summary last_summary;  // the Summary of the most recent ceres::Solve the reference issued
int solves = 0;
int max_iterations_override = 0;  // > 0: the next solves stop after that many iterations (the one-step dump below)
}  // namespace pin

namespace ceres {
inline void PinnedSolve(const Solver::Options& options, Problem* problem, Solver::Summary* summary) {
    if (pin::max_iterations_override > 0) {  // (everything else as the reference set it)
        Solver::Options o2 = options;
        o2.max_num_iterations = pin::max_iterations_override;
        Solve(o2, problem, summary);
        pin::last_summary = *summary;
        pin::solves += 1;
        return;
    }
    Solve(options, problem, summary);  // the real one, same arguments
    pin::last_summary = *summary;
    pin::solves += 1;
}
}  // namespace ceres

// every header nonlinearRefinement.cc includes is already included above (include guards), so the macro only meets the three call sites
#define Solve PinnedSolve
#include "nonlinearRefinement.cc"  // ${REFERENCE_DIR}/src, by include path
#undef Solve

// ---- libc's generator replaced by a replay of prepared draws (minimal.cc:230-236) -------------------------------------------------
namespace pin {
std::vector<int> draws;
size_t next_draw = 0;
}  // namespace pin
#ifndef __THROW
#define __THROW
#endif
extern "C" int rand(void) __THROW {
    if (pin::next_draw >= pin::draws.size()) {
        std::fprintf(stderr, "pin_harness: the reference asked for more random draws than were prepared\n");
        std::abort();
    }
    return pin::draws[pin::next_draw++];
}
extern "C" void srand(unsigned) __THROW {}

// ---- the .pin container (tools/pin_reference/pinio.py) ------------------------------------------------------------------------------
namespace pin {
struct Arr {
    uint32_t code = 0;  // 0 f64, 1 i32, 2 i64, 3 u8
    std::vector<uint64_t> dims;
    std::vector<char> bytes;
    size_t count() const {
        size_t n = 1;
        for (uint64_t d : dims) n *= (size_t)d;
        return n;
    }
    const double* f64() const { return reinterpret_cast<const double*>(bytes.data()); }
    const int32_t* i32() const { return reinterpret_cast<const int32_t*>(bytes.data()); }
};
const size_t kItem[4] = {8, 4, 8, 1};

std::map<std::string, Arr> read_pin(const std::string& path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("cannot open " + path);
    char magic[8];
    f.read(magic, 8);
    if (std::memcmp(magic, "RSDPIN01", 8) != 0) throw std::runtime_error("not a .pin file: " + path);
    uint32_t count = 0;
    f.read(reinterpret_cast<char*>(&count), 4);
    std::map<std::string, Arr> out;
    for (uint32_t i = 0; i < count; ++i) {
        uint32_t ln = 0, ndim = 0;
        f.read(reinterpret_cast<char*>(&ln), 4);
        std::string name(ln, '\0');
        f.read(&name[0], ln);
        Arr a;
        f.read(reinterpret_cast<char*>(&a.code), 4);
        f.read(reinterpret_cast<char*>(&ndim), 4);
        a.dims.resize(ndim);
        if (ndim) f.read(reinterpret_cast<char*>(a.dims.data()), 8 * ndim);
        a.bytes.resize(a.count() * kItem[a.code]);
        f.read(a.bytes.data(), (std::streamsize)a.bytes.size());
        out[name] = a;
    }
    return out;
}

struct Writer {
    std::vector<std::pair<std::string, Arr>> items;
    void f64(const std::string& name, const std::vector<uint64_t>& dims, const double* p) {
        Arr a;
        a.code = 0;
        a.dims = dims;
        a.bytes.assign(reinterpret_cast<const char*>(p), reinterpret_cast<const char*>(p) + a.count() * 8);
        items.push_back(std::make_pair(name, a));
    }
    void scalar(const std::string& name, double v) { f64(name, {1}, &v); }
    void write(const std::string& path) const {
        std::ofstream f(path, std::ios::binary);
        f.write("RSDPIN01", 8);
        const uint32_t count = (uint32_t)items.size();
        f.write(reinterpret_cast<const char*>(&count), 4);
        for (const auto& it : items) {
            const uint32_t ln = (uint32_t)it.first.size(), ndim = (uint32_t)it.second.dims.size();
            f.write(reinterpret_cast<const char*>(&ln), 4);
            f.write(it.first.data(), ln);
            f.write(reinterpret_cast<const char*>(&it.second.code), 4);
            f.write(reinterpret_cast<const char*>(&ndim), 4);
            if (ndim) f.write(reinterpret_cast<const char*>(it.second.dims.data()), 8 * ndim);
            f.write(it.second.bytes.data(), (std::streamsize)it.second.bytes.size());
        }
    }
};

// ---- what is kept of a ceres::Solver::Summary ---------------------------------------------------------------------------------------
// summary row: {entries of summary.iterations behind iteration 0, successful among them, unsuccessful among them (invalid steps
// included), termination in the repository's coding (include/rsdsfm.h RSDSFM_TERM_*: 0 gradient, 1 parameter, 2 function tolerance,
// 3 max iterations, 4 failure, 5 min radius), initial cost, final cost, ceres termination_type, trust-region radius of the last entry,
// summary.num_successful_steps, summary.num_unsuccessful_steps (Ceres' own counters, whatever they do with iteration 0)}.
// The counts come from the iterations vector itself, so they do not depend on how Ceres counts iteration 0.  NOTE for the reader of
// the fixture: TrustRegionMinimizer returns from INSIDE the loop on the parameter / function tolerance, so the iteration that met that
// tolerance is not in the vector -- an implementation that counts it reports one iteration more (tests/test_reference_golden.py).
// trace rows (one per entry of summary.iterations, iteration 0 included): {iteration, cost, cost_change, gradient_max_norm, step_norm,
// relative_decrease, trust_region_radius, step_is_successful, step_is_valid}
const int kSummaryCols = 10, kTraceCols = 9, kTraceRows = 64;

int term_code(const ceres::Solver::Summary& s) {
    const std::string& m = s.message;
    if (m.find("Gradient tolerance") != std::string::npos) return 0;
    if (m.find("Parameter tolerance") != std::string::npos) return 1;
    if (m.find("Function tolerance") != std::string::npos) return 2;
    if (m.find("Maximum number of iterations") != std::string::npos) return 3;
    if (m.find("Minimum trust region radius") != std::string::npos || m.find("trust region radius") != std::string::npos) return 5;
    return 4;
}

void keep_summary(const ceres::Solver::Summary& s, double* row, double* trace) {
    int succ = 0, unsucc = 0;
    for (size_t i = 1; i < s.iterations.size(); ++i) (s.iterations[i].step_is_successful ? succ : unsucc) += 1;
    row[0] = s.iterations.empty() ? 0.0 : (double)(s.iterations.size() - 1);
    row[1] = (double)succ;
    row[2] = (double)unsucc;
    row[3] = (double)term_code(s);
    row[4] = s.initial_cost;
    row[5] = s.final_cost;
    row[6] = (double)s.termination_type;
    row[7] = s.iterations.empty() ? 0.0 : s.iterations.back().trust_region_radius;
    row[8] = (double)s.num_successful_steps;
    row[9] = (double)s.num_unsuccessful_steps;
    for (int i = 0; i < kTraceRows * kTraceCols; ++i) trace[i] = std::numeric_limits<double>::quiet_NaN();
    for (size_t i = 0; i < s.iterations.size() && (int)i < kTraceRows; ++i) {
        const ceres::IterationSummary& it = s.iterations[i];
        double* t = trace + i * kTraceCols;
        t[0] = (double)it.iteration;
        t[1] = it.cost;
        t[2] = it.cost_change;
        t[3] = it.gradient_max_norm;
        t[4] = it.step_norm;
        t[5] = it.relative_decrease;
        t[6] = it.trust_region_radius;
        t[7] = it.step_is_successful ? 1.0 : 0.0;
        t[8] = it.step_is_valid ? 1.0 : 0.0;
    }
}
}  // namespace pin

static void run_case(const std::string& in_dir, const std::string& out_dir, const std::string& name) {
    using namespace pin;
    std::map<std::string, Arr> in = read_pin(in_dir + "/" + name + ".pin");
    const int n = (int)in["q"].dims[0];
    const int T = (int)in["samples"].dims[0];
    const bool use_k = in["use_k"].i32()[0] != 0;
    const double tol = in["tolerance"].f64()[0];
    // (n x 2) C-order doubles are the memory of a column-major 2 x n Eigen array
    const Eigen::Array2Xd q = Eigen::Map<const Eigen::Array2Xd>(in["q"].f64(), 2, n);
    const Eigen::Array2Xd u = Eigen::Map<const Eigen::Array2Xd>(in["u"].f64(), 2, n);
    const Eigen::ArrayXd alpha = Eigen::Map<const Eigen::ArrayXd>(in["alpha"].f64(), n);
    const Eigen::ArrayXd alpha_k = Eigen::Map<const Eigen::ArrayXd>(in["alpha_k"].f64(), n);
    const int32_t* samples = in["samples"].i32();
    Writer W;

    // ---- minimal::calculateVelocities (minimal.cc:36-177) on every committed 9-point sample, built as minimal.cc:238-241 builds it
    std::vector<double> hyp_w(3 * T), hyp_v(3 * T), hyp_k(T);
    for (int t = 0; t < T; ++t) {
        Eigen::Array2Xd cq(2, 9), cu(2, 9);
        Eigen::ArrayXd ca(9), cak(9);
        for (int j = 0; j < 9; ++j) {
            const int idx = samples[t * 9 + j];
            cq.col(j) = q.col(idx);
            cu.col(j) = u.col(idx);
            ca(j) = alpha(idx);
            cak(j) = alpha_k(idx);
        }
        const Velocities vel = minimal::calculateVelocities(cq, cu, ca, cak, use_k);
        for (int i = 0; i < 3; ++i) hyp_w[3 * t + i] = vel.w(i), hyp_v[3 * t + i] = vel.v(i);
        hyp_k[t] = vel.k;
    }
    W.f64("hyp_w", {(uint64_t)T, 3}, hyp_w.data());
    W.f64("hyp_v", {(uint64_t)T, 3}, hyp_v.data());
    W.f64("hyp_k", {(uint64_t)T}, hyp_k.data());

    // ---- nonlinear_refinement::estimateInverseDepths (nonlinearRefinement.cc:109-180) for EVERY hypothesis: rho + Ceres' summary
    std::vector<double> rho((size_t)T * n), dsum((size_t)T * kSummaryCols), dtrace((size_t)T * kTraceRows * kTraceCols);
    for (int t = 0; t < T; ++t) {
        const Eigen::Vector3d v(hyp_v[3 * t], hyp_v[3 * t + 1], hyp_v[3 * t + 2]), w(hyp_w[3 * t], hyp_w[3 * t + 1], hyp_w[3 * t + 2]);
        const int before = solves;
        const Eigen::ArrayXd r = nonlinear_refinement::estimateInverseDepths(q, u, v, w, hyp_k[t], alpha, alpha_k, false);
        if (solves != before + 1) throw std::runtime_error("estimateInverseDepths did not go through the pinned Solve");
        for (int i = 0; i < n; ++i) rho[(size_t)t * n + i] = r(i);
        keep_summary(last_summary, &dsum[(size_t)t * kSummaryCols], &dtrace[(size_t)t * kTraceRows * kTraceCols]);
    }
    // ---- the LAST-BIT question of the 1x1 e-blocks: Ceres' Schur eliminator inverts each e-block through InvertPSDMatrix (an LLT solve, then
    // a MULTIPLY by the inverse), the oracle divides (rsdsfm_oracle.c: step = -(gt / (ht + lam))).  The first LM step of real Ceres for up to
    // 100 pixels, each as a problem of its own (one pixel: the global tests are the pixel's) stopped after ONE iteration through the
    // reference's own estimateInverseDepths, with the first finite hypothesis: rho_1 per pixel.  tests/test_reference_golden.py puts the
    // oracle's one-step rho beside it and reports the difference in ulps.
    {
        int t0 = -1;
        for (int t = 0; t < T && t0 < 0; ++t) {
            bool ok = std::isfinite(hyp_k[t]);
            for (int i = 0; i < 3; ++i) ok = ok && std::isfinite(hyp_v[3 * t + i]) && std::isfinite(hyp_w[3 * t + i]);
            if (ok) t0 = t;
        }
        const int m = std::min(n, 100);
        std::vector<double> one(m, 0.0);
        if (t0 >= 0) {
            const Eigen::Vector3d v(hyp_v[3 * t0], hyp_v[3 * t0 + 1], hyp_v[3 * t0 + 2]), w(hyp_w[3 * t0], hyp_w[3 * t0 + 1], hyp_w[3 * t0 + 2]);
            max_iterations_override = 1;
            for (int i = 0; i < m; ++i) {
                const int idx = (int)((int64_t)i * n / m);
                const Eigen::Array2Xd q1 = q.col(idx), u1 = u.col(idx);
                Eigen::ArrayXd a1(1), ak1(1);
                a1(0) = alpha(idx), ak1(0) = alpha_k(idx);
                one[i] = nonlinear_refinement::estimateInverseDepths(q1, u1, v, w, hyp_k[t0], a1, ak1, false)(0);
            }
            max_iterations_override = 0;
        }
        const double which = (double)t0;
        W.f64("one_step_rho", {(uint64_t)m}, one.data());
        W.f64("one_step_hypothesis", {1}, &which);
    }
    W.f64("depth_rho", {(uint64_t)T, (uint64_t)n}, rho.data());
    W.f64("depth_summary", {(uint64_t)T, (uint64_t)kSummaryCols}, dsum.data());
    W.f64("depth_trace", {(uint64_t)T, (uint64_t)kTraceRows, (uint64_t)kTraceCols}, dtrace.data());

    // ---- minimal::ransac (minimal.cc:209-306) with the committed samples injected through rand()
    {
        std::vector<int> perm(n);
        std::iota(perm.begin(), perm.end(), 0);
        draws.clear();
        next_draw = 0;
        for (int t = 0; t < T; ++t) {
            int n_temp = n;
            for (int j = 0; j < 9; ++j) {
                const int want = samples[t * 9 + j];
                int r = -1;
                for (int i = 0; i < n_temp; ++i)
                    if (perm[i] == want) {
                        r = i;
                        break;
                    }
                if (r < 0) throw std::runtime_error("sample set is not reachable by the reference's sampler");
                draws.push_back(r);
                std::swap(perm[n_temp - 1], perm[r]);
                n_temp--;
            }
        }
    }
    const RansacValues rv = minimal::ransac(q, u, alpha, alpha_k, use_k, T, tol, false);
    if (next_draw != draws.size()) throw std::runtime_error("minimal::ransac did not consume the prepared draws");
    const int M = rv.num_inliers;
    {
        std::vector<double> inl((size_t)3 * M), a(M), ak(M);
        for (int i = 0; i < M; ++i) {
            for (int c = 0; c < 3; ++c) inl[(size_t)3 * i + c] = rv.inliers(c, i);
            a[i] = rv.alpha(i);
            ak[i] = rv.alpha_k(i);
        }
        W.scalar("ransac_num_inliers", (double)M);
        W.f64("ransac_inliers", {(uint64_t)M, 3}, inl.data());
        W.f64("ransac_alpha", {(uint64_t)M}, a.data());
        W.f64("ransac_alpha_k", {(uint64_t)M}, ak.data());
        const double wv[7] = {rv.w(0), rv.w(1), rv.w(2), rv.v(0), rv.v(1), rv.v(2), rv.k};
        W.f64("ransac_wvk", {7}, wv);
    }

    // ---- nonlinear_refinement::nonLinearRefinement (nonlinearRefinement.cc:183-252) on the reference's own RANSAC result:
    // "compat" = main.cc:457: the UN-compacted flow (column i for the i-th inlier, quirk Q2); "gather" = the flow of each inlier's own
    // point (the inliers keep the order of the point list, so the j-th inlier is the j-th point whose (x, y) matches)
    for (int mode = 0; mode < 2; ++mode) {
        Eigen::Array2Xd flow = u;
        if (mode == 1) {
            flow = Eigen::Array2Xd::Zero(2, M);
            int j = 0;
            for (int i = 0; i < n && j < M; ++i)
                if (q(0, i) == rv.inliers(0, j) && q(1, i) == rv.inliers(1, j)) {
                    flow.col(j) = u.col(i);
                    ++j;
                }
            if (j != M) throw std::runtime_error("could not match the inliers with their points");
        }
        const int before = solves;
        const RansacValues rf = nonlinear_refinement::nonLinearRefinement(flow, rv, use_k, false);
        if (solves != before + 1) throw std::runtime_error("nonLinearRefinement did not go through the pinned Solve");
        const std::string p = mode == 0 ? "refine_compat_" : "refine_gather_";
        const double wv[7] = {rf.w(0), rf.w(1), rf.w(2), rf.v(0), rf.v(1), rf.v(2), rf.k};
        W.f64(p + "wvk", {7}, wv);
        std::vector<double> z(M);
        for (int i = 0; i < M; ++i) z[i] = rf.inliers(2, i);
        W.f64(p + "z", {(uint64_t)M}, z.data());
        std::vector<double> sum(kSummaryCols), trace((size_t)kTraceRows * kTraceCols);
        keep_summary(last_summary, sum.data(), trace.data());
        W.f64(p + "summary", {(uint64_t)kSummaryCols}, sum.data());
        W.f64(p + "trace", {(uint64_t)kTraceRows, (uint64_t)kTraceCols}, trace.data());
    }
    W.write(out_dir + "/" + name + ".pin");
    std::printf("%s: n = %d, T = %d, inliers = %d, solves = %d\n", name.c_str(), n, T, M, solves);
}

int main(int argc, char** argv) {
    if (argc < 4) {
        std::fprintf(stderr, "usage: pin_harness <inputs_dir> <outputs_dir> case [case ...]\n");
        return 2;
    }
    try {
        for (int i = 3; i < argc; ++i) run_case(argv[1], argv[2], argv[i]);
        std::ofstream v(std::string(argv[2]) + "/versions.txt");
        v << "ceres " << CERES_VERSION_STRING << "\n"
          << "eigen " << EIGEN_WORLD_VERSION << "." << EIGEN_MAJOR_VERSION << "." << EIGEN_MINOR_VERSION << "\n"
#ifdef __VERSION__
          << "compiler " << __VERSION__ << "\n"
#endif
#ifdef __FMA__
          << "fma-contraction possible (built with -mfma / -march): NOT the reference's build (src/CMakeLists.txt:18 has plain -std=c++11)\n"
#else
          << "no-fma target (as the reference's own build)\n"
#endif
            ;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "pin_harness: %s\n", e.what());
        return 1;
    }
    return 0;
}
