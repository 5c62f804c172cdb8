/*
 * rsdsfm_cpu_ref.cpp -- REFERENCE-STRUCTURED CPU BASELINE (TEST / BENCH INFRASTRUCTURE, NOT PRODUCT CODE).
 *
 * BASELINE.md section 3.1 `cpu_ref`.  The oracle (rsdsfm_oracle.c) is a closed-loop port: flat arrays, analytic derivatives, one
 * fused loop per LM iteration.  The reference does not work that way, and its run time is dominated by HOW it works
 * (nonlinearRefinement.cc:109-180 is called once per RANSAC trial, minimal.cc:249):
 *   * it builds a ceres::Problem from scratch per call: per pixel one heap-allocated residual functor (`new RsResidual`, :148-149),
 *     one heap-allocated cost function around it (`new AutoDiffCostFunction<RsResidual, 2, 3, 3, 1, 1>`), one residual block record
 *     and a look-up / insert of its four parameter blocks in the problem's pointer map (`AddResidualBlock`, :150-151);
 *   * Solve() (DENSE_SCHUR, :160-163) orders the parameter blocks for the Schur complement: a graph over all blocks, vertices
 *     sorted by degree, a greedy independent set -- N + 3 vertices;
 *   * every Jacobian evaluation runs the functor on Jet<double, 8> (value + derivatives w.r.t. all 8 scalars of the four parameter
 *     blocks, constant or not: ~9x the arithmetic of the plain residual), through a virtual call per residual block, into a
 *     block-sparse Jacobian; candidate costs use the plain double instantiation;
 *   * the problem and every object in it are deleted at the end of the call.
 * This file restates THAT structure (own classes named after the roles they play; nothing of Ceres is here), driving the same
 * trust-region arithmetic as the oracle, so that its results can be checked against the oracle (tests/test_oracle_cpu_ref.py)
 * and its run time is a structural stand-in for "the reference's CPU path" on the box the GPU is timed on.  It is a cost MODEL, not
 * Ceres: the one published point (report.pdf section 5.5: ~20 s per RANSAC trial + ~30 s refinement at 1920x1080 on a laptop) is
 * quoted beside it.  One thread (Ceres num_threads = 1).  Build: g++ -O2 -ffp-contract=off (oracle/Makefile: cpu_ref).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <unordered_map>
#include <unordered_set>
#include <vector>

extern "C" {
#include "rsdsfm_oracle.h"
}

namespace {

// ---- dual numbers: value + N partial derivatives -----------------------------------------------------------------------
template <int N>
struct Dual {
    double a;
    double v[N];
    Dual() : a(0.0) { memset(v, 0, sizeof(v)); }
    explicit Dual(double x) : a(x) { memset(v, 0, sizeof(v)); }
    Dual(double x, int k) : a(x) {
        memset(v, 0, sizeof(v));
        v[k] = 1.0;
    }
};
template <int N>
inline Dual<N> operator+(const Dual<N>& f, const Dual<N>& g) {
    Dual<N> h;
    h.a = f.a + g.a;
    for (int i = 0; i < N; ++i) h.v[i] = f.v[i] + g.v[i];
    return h;
}
template <int N>
inline Dual<N> operator-(const Dual<N>& f, const Dual<N>& g) {
    Dual<N> h;
    h.a = f.a - g.a;
    for (int i = 0; i < N; ++i) h.v[i] = f.v[i] - g.v[i];
    return h;
}
template <int N>
inline Dual<N> operator*(const Dual<N>& f, const Dual<N>& g) {
    Dual<N> h;
    h.a = f.a * g.a;
    for (int i = 0; i < N; ++i) h.v[i] = f.a * g.v[i] + f.v[i] * g.a;
    return h;
}
template <int N>
inline Dual<N> operator/(const Dual<N>& f, const Dual<N>& g) {
    Dual<N> h;
    const double ig = 1.0 / g.a;
    h.a = f.a * ig;
    for (int i = 0; i < N; ++i) h.v[i] = (f.v[i] - h.a * g.v[i]) * ig;
    return h;
}

// ---- the residual of one pixel (nonlinearRefinement.cc:32-52), templated on the scalar like the reference's functor ---------
struct PixelResidual {
    double x, y, ux, uy, alpha, alpha_k;
    template <class T>
    bool operator()(const T* lin, const T* ang, const T* k, const T* rho, T* r) const {
        const T beta = (T(2.0) / (T(2.0) + *k)) * (T(alpha) + *k * T(alpha_k));
        const T p0 = beta * T(-1.0) * (*rho * (T(x) * lin[2] - lin[0]) + (T(x) * T(y) * ang[0]) - (T(1.0) + T(x) * T(x)) * ang[1] + T(y) * ang[2]);
        const T p1 = beta * T(-1.0) * (*rho * (T(y) * lin[2] - lin[1]) + (T(1.0) + T(y) * T(y)) * ang[0] - T(x) * T(y) * ang[1] - T(x) * ang[2]);
        r[0] = T(ux) - p0;
        r[1] = T(uy) - p1;
        return true;
    }
};

// ---- cost function objects behind a virtual interface ---------------------------------------------------------------------------
struct CostFn {
    virtual ~CostFn() {}
    // params: {v[3], w[3], k[1], rho[1]}; jacobians (may be null): per block a row-major 2 x size array, or null for a block
    virtual bool evaluate(double const* const* params, double* residuals, double** jacobians) const = 0;
};
struct AutoDiffPixelCost : CostFn {
    PixelResidual* f;  // owned
    explicit AutoDiffPixelCost(PixelResidual* fn) : f(fn) {}
    ~AutoDiffPixelCost() override { delete f; }
    bool evaluate(double const* const* p, double* r, double** jac) const override {
        if (!jac) return (*f)(p[0], p[1], p[2], p[3], r);
        typedef Dual<8> D;
        D lin[3] = {D(p[0][0], 0), D(p[0][1], 1), D(p[0][2], 2)};
        D ang[3] = {D(p[1][0], 3), D(p[1][1], 4), D(p[1][2], 5)};
        D k(p[2][0], 6), rho(p[3][0], 7);
        D out[2];
        (*f)(lin, ang, &k, &rho, out);
        r[0] = out[0].a, r[1] = out[1].a;
        static const int first[4] = {0, 3, 6, 7}, size[4] = {3, 3, 1, 1};
        for (int b = 0; b < 4; ++b)
            if (jac[b])
                for (int row = 0; row < 2; ++row)
                    for (int c = 0; c < size[b]; ++c) jac[b][row * size[b] + c] = out[row].v[first[b] + c];
        return true;
    }
};

struct ParamBlock {
    double* state;
    int size;
    bool constant;
    int index;  // position in the reduced program (-1 = constant)
};
struct ResidualBlock {
    CostFn* cost;  // owned
    ParamBlock* blocks[4];
};

struct Problem {
    std::unordered_map<double*, ParamBlock*> params;
    std::vector<ResidualBlock*> residuals;
    ParamBlock* intern(double* p, int size) {
        auto it = params.find(p);
        if (it != params.end()) return it->second;
        ParamBlock* b = new ParamBlock{p, size, false, -1};
        params.emplace(p, b);
        return b;
    }
    void add_residual_block(CostFn* cost, double* v, double* w, double* k, double* rho) {
        ResidualBlock* rb = new ResidualBlock;
        rb->cost = cost;
        rb->blocks[0] = intern(v, 3), rb->blocks[1] = intern(w, 3), rb->blocks[2] = intern(k, 1), rb->blocks[3] = intern(rho, 1);
        residuals.push_back(rb);
    }
    void set_constant(double* p) { params.at(p)->constant = true; }
    ~Problem() {
        for (ResidualBlock* rb : residuals) {
            delete rb->cost;
            delete rb;
        }
        for (auto& kv : params) delete kv.second;
    }
};

// ---- Schur ordering: graph over the variable blocks, degree sort, greedy independent set (the e-blocks) --------------------------
int schur_ordering(const Problem& pb, std::vector<ParamBlock*>* e_blocks, std::vector<ParamBlock*>* f_blocks) {
    std::unordered_map<ParamBlock*, std::unordered_set<ParamBlock*>> graph;
    for (auto& kv : pb.params)
        if (!kv.second->constant) graph[kv.second];
    for (const ResidualBlock* rb : pb.residuals)
        for (int a = 0; a < 4; ++a) {
            if (rb->blocks[a]->constant) continue;
            for (int b = a + 1; b < 4; ++b) {
                if (rb->blocks[b]->constant) continue;
                graph[rb->blocks[a]].insert(rb->blocks[b]);
                graph[rb->blocks[b]].insert(rb->blocks[a]);
            }
        }
    std::vector<ParamBlock*> verts;
    verts.reserve(graph.size());
    for (auto& kv : graph) verts.push_back(kv.first);
    // deterministic: by degree, ties by the address of the user's state (the rho array is contiguous -> pixel order)
    std::sort(verts.begin(), verts.end(), [&](ParamBlock* a, ParamBlock* b) {
        const size_t da = graph[a].size(), db = graph[b].size();
        return da != db ? da < db : a->state < b->state;
    });
    std::unordered_set<ParamBlock*> taken, blocked;
    for (ParamBlock* v : verts) {
        if (blocked.count(v)) continue;
        taken.insert(v);
        e_blocks->push_back(v);
        for (ParamBlock* nb : graph[v]) blocked.insert(nb);
    }
    for (ParamBlock* v : verts)
        if (!taken.count(v)) f_blocks->push_back(v);
    std::sort(f_blocks->begin(), f_blocks->end(), [](ParamBlock* a, ParamBlock* b) { return a->state < b->state; });
    return 0;
}

static inline double clampd(double x, double lo, double hi) { return x < lo ? lo : (x > hi ? hi : x); }
static int chol_solve(double* A, int n, const double* b, double* x) {
    for (int j = 0; j < n; ++j) {
        double d = A[j * n + j];
        for (int t = 0; t < j; ++t) d -= A[j * n + t] * A[j * n + t];
        if (!(d > 0.0)) return -1;
        d = sqrt(d);
        A[j * n + j] = d;
        for (int i = j + 1; i < n; ++i) {
            double s = A[i * n + j];
            for (int t = 0; t < j; ++t) s -= A[i * n + t] * A[j * n + t];
            A[i * n + j] = s / d;
        }
    }
    double y[8];
    for (int i = 0; i < n; ++i) {
        double s = b[i];
        for (int t = 0; t < i; ++t) s -= A[i * n + t] * y[t];
        y[i] = s / A[i * n + i];
    }
    for (int i = n - 1; i >= 0; --i) {
        double s = y[i];
        for (int t = i + 1; t < n; ++t) s -= A[t * n + i] * x[t];
        x[i] = s / A[i * n + i];
    }
    return 0;
}

// ---- the solve: trust-region LM with the Schur complement over the 1x1 e-blocks (Ceres 1.14 defaults; arithmetic as the oracle's
// rso_estimate_inverse_depths / rso_refine, whose comments cite the Ceres sources) -------------------------------------------------
int solve(Problem& pb, rso_lm_summary* summary) {
    rso_lm_summary sm;
    memset(&sm, 0, sizeof(sm));
    std::vector<ParamBlock*> eb, fb;
    schur_ordering(pb, &eb, &fb);
    const int64_t m = (int64_t)pb.residuals.size();
    int np = 0;
    int foff[4] = {0, 0, 0, 0};  // column offset of each f-block in the reduced system
    if (m > 0) {  // the f-blocks in the order the residual blocks name them (v, w, k): the column order of the reduced system
        std::vector<ParamBlock*> named;
        for (int b = 0; b < 3; ++b)
            if (!pb.residuals[0]->blocks[b]->constant) named.push_back(pb.residuals[0]->blocks[b]);
        if (named.size() != fb.size()) return -3;
        for (ParamBlock* nb : named)
            if (std::find(fb.begin(), fb.end(), nb) == fb.end()) return -3;
        fb = named;
    }
    for (size_t i = 0; i < eb.size(); ++i) eb[i]->index = (int)i;
    for (size_t i = 0; i < fb.size(); ++i) {
        fb[i]->index = (int)(eb.size() + i);
        foff[i] = np;
        np += fb[i]->size;
    }
    if (np > 7 || fb.size() > 3) return -1;
    // every residual block of this problem has exactly one e-block (its rho) -- checked, not assumed
    for (const ResidualBlock* rb : pb.residuals)
        if (rb->blocks[3]->constant || rb->blocks[3]->index >= (int)eb.size()) return -2;
    // block-sparse Jacobian: per residual block a 2x1 e-block and a 2xnp f-row
    std::vector<double> r(2 * (size_t)std::max<int64_t>(m, 1)), Je(2 * (size_t)std::max<int64_t>(m, 1)), Jf(2 * 7 * (size_t)std::max<int64_t>(m, 1));
    std::vector<double> srho((size_t)std::max<int64_t>(m, 1)), cand((size_t)std::max<int64_t>(m, 1));
    auto evaluate = [&](bool with_jac, double* cost_out) {
        double cost = 0.0;
        for (int64_t i = 0; i < m; ++i) {
            const ResidualBlock* rb = pb.residuals[(size_t)i];
            const double* p[4] = {rb->blocks[0]->state, rb->blocks[1]->state, rb->blocks[2]->state, rb->blocks[3]->state};
            if (!with_jac) {
                double rr[2];
                rb->cost->evaluate(p, rr, nullptr);
                cost += rr[0] * rr[0] + rr[1] * rr[1];
                continue;
            }
            double j0[6], j1[6], j2[2], j3[2];
            double* jac[4] = {rb->blocks[0]->constant ? nullptr : j0, rb->blocks[1]->constant ? nullptr : j1, rb->blocks[2]->constant ? nullptr : j2, j3};
            rb->cost->evaluate(p, &r[2 * (size_t)i], jac);
            cost += r[2 * i] * r[2 * i] + r[2 * i + 1] * r[2 * i + 1];
            Je[2 * i] = j3[0], Je[2 * i + 1] = j3[1];
            double* row = &Jf[14 * (size_t)i];
            for (size_t b = 0, fi = 0; b < 3; ++b) {
                if (rb->blocks[b]->constant) continue;
                const int sz = rb->blocks[b]->size, off = foff[fi++];
                for (int c = 0; c < sz; ++c) row[off + c] = jac[b][c], row[7 + off + c] = jac[b][sz + c];
            }
        }
        *cost_out = 0.5 * cost;
    };
    auto pvec = [&](double* out) {  // the f-parameters as one vector
        for (size_t b = 0; b < fb.size(); ++b)
            for (int c = 0; c < fb[b]->size; ++c) out[foff[b] + c] = fb[b]->state[c];
    };
    double cost = 0.0;
    evaluate(true, &cost);
    double sp[7], colsq[7] = {0}, gp[7] = {0}, gmax = 0.0, xsq = 0.0, p[7] = {0}, pcur[7];
    pvec(p);
    for (int64_t i = 0; i < m; ++i) {
        const double* row = &Jf[14 * (size_t)i];
        for (int c = 0; c < np; ++c) {
            colsq[c] += row[c] * row[c] + row[7 + c] * row[7 + c];
            gp[c] += row[c] * r[2 * i] + row[7 + c] * r[2 * i + 1];
        }
        srho[(size_t)i] = 1.0 / (1.0 + sqrt(Je[2 * i] * Je[2 * i] + Je[2 * i + 1] * Je[2 * i + 1]));
        const double g = fabs(Je[2 * i] * r[2 * i] + Je[2 * i + 1] * r[2 * i + 1]);
        if (g > gmax) gmax = g;
        const double rho = *pb.residuals[(size_t)i]->blocks[3]->state;
        xsq += rho * rho;
    }
    for (int c = 0; c < np; ++c) {
        sp[c] = 1.0 / (1.0 + sqrt(colsq[c]));
        if (fabs(gp[c]) > gmax) gmax = fabs(gp[c]);
        xsq += p[c] * p[c];
    }
    double x_norm = sqrt(xsq), radius = 1e4, decrease_factor = 2.0;
    int iteration = 0, invalid = 0;
    sm.initial_cost = cost;
    sm.termination = -1;
    if (m == 0 || gmax <= 1e-10) sm.termination = RSO_TERM_GRADIENT;
    while (sm.termination < 0) {
        if (iteration >= 50) {
            sm.termination = RSO_TERM_MAX_ITER;
            break;
        }
        if (radius <= 1e-32) {
            sm.termination = RSO_TERM_MIN_RADIUS;
            break;
        }
        ++iteration;
        const double inv_radius = 1.0 / radius;
        double FtF[49] = {0}, C[49] = {0}, Ftb[7] = {0}, cvec[7] = {0};
        for (int64_t i = 0; i < m; ++i) {  // Schur elimination of the e-blocks, chunk by chunk (one residual block per chunk here)
            const double* row = &Jf[14 * (size_t)i];
            const double E0 = Je[2 * i] * srho[(size_t)i], E1 = Je[2 * i + 1] * srho[(size_t)i];
            const double ht = E0 * E0 + E1 * E1;
            const double lam = clampd(ht, 1e-6, 1e32) * inv_radius;
            const double ete_inv = 1.0 / (ht + lam);
            const double Etb = E0 * r[2 * i] + E1 * r[2 * i + 1];
            double F0[7], F1[7], EtF[7];
            for (int c = 0; c < np; ++c) {
                F0[c] = row[c] * sp[c];
                F1[c] = row[7 + c] * sp[c];
                EtF[c] = E0 * F0[c] + E1 * F1[c];
            }
            for (int a = 0; a < np; ++a) {
                Ftb[a] += F0[a] * r[2 * i] + F1[a] * r[2 * i + 1];
                cvec[a] += EtF[a] * (ete_inv * Etb);
                for (int b = a; b < np; ++b) {
                    FtF[a * 7 + b] += F0[a] * F0[b] + F1[a] * F1[b];
                    C[a * 7 + b] += EtF[a] * (ete_inv * EtF[b]);
                }
            }
        }
        double S[49], rhs[7], yp[7] = {0};
        for (int a = 0; a < np; ++a) {
            const double Dp = clampd(FtF[a * 7 + a], 1e-6, 1e32) * inv_radius;
            rhs[a] = Ftb[a] - cvec[a];
            for (int b = a; b < np; ++b) {
                double sab = FtF[a * 7 + b] - C[a * 7 + b];
                if (a == b) sab += Dp;
                S[a * np + b] = sab;
                S[b * np + a] = sab;
            }
        }
        const bool solve_ok = np == 0 || chol_solve(S, np, rhs, yp) == 0;
        double model_change = 0.0, stepsq = 0.0, ccost = 0.0;
        memcpy(pcur, p, sizeof(p));
        if (solve_ok) {
            double pc[7];
            for (int c = 0; c < np; ++c) {
                pc[c] = p[c] + (-yp[c]) * sp[c];
                const double dx = p[c] - pc[c];
                stepsq += dx * dx;
            }
            for (int64_t i = 0; i < m; ++i) {  // back-substitution
                const double* row = &Jf[14 * (size_t)i];
                const double E0 = Je[2 * i] * srho[(size_t)i], E1 = Je[2 * i + 1] * srho[(size_t)i];
                const double ht = E0 * E0 + E1 * E1;
                const double lam = clampd(ht, 1e-6, 1e32) * inv_radius;
                const double ete_inv = 1.0 / (ht + lam);
                const double Etb = E0 * r[2 * i] + E1 * r[2 * i + 1];
                double Fy0 = 0.0, Fy1 = 0.0;
                for (int c = 0; c < np; ++c) {
                    Fy0 += row[c] * sp[c] * yp[c];
                    Fy1 += row[7 + c] * sp[c] * yp[c];
                }
                const double step_e = -(ete_inv * (Etb - (E0 * Fy0 + E1 * Fy1)));
                const double m0 = -Fy0 + E0 * step_e, m1 = -Fy1 + E1 * step_e;
                model_change -= m0 * (r[2 * i] + m0 / 2.0) + m1 * (r[2 * i + 1] + m1 / 2.0);
                const double rho = *pb.residuals[(size_t)i]->blocks[3]->state;
                cand[(size_t)i] = rho + step_e * srho[(size_t)i];
                const double dx = rho - cand[(size_t)i];
                stepsq += dx * dx;
            }
            // candidate cost: the parameter blocks take the candidate, the residuals are evaluated without Jacobians, and the
            // blocks are restored if the step is not accepted
            for (size_t b = 0; b < fb.size(); ++b)
                for (int c = 0; c < fb[b]->size; ++c) fb[b]->state[c] = pc[foff[b] + c];
            for (int64_t i = 0; i < m; ++i) std::swap(*pb.residuals[(size_t)i]->blocks[3]->state, cand[(size_t)i]);
            evaluate(false, &ccost);
            memcpy(pcur, pc, sizeof(pc));
        }
        auto restore = [&]() {  // undo the candidate
            if (!solve_ok) return;
            for (size_t b = 0; b < fb.size(); ++b)
                for (int c = 0; c < fb[b]->size; ++c) fb[b]->state[c] = p[foff[b] + c];
            for (int64_t i = 0; i < m; ++i) std::swap(*pb.residuals[(size_t)i]->blocks[3]->state, cand[(size_t)i]);
        };
        if (!solve_ok || !(model_change > 0.0)) {
            restore();
            ++sm.num_unsuccessful_steps;
            if (++invalid >= 5) {
                sm.termination = RSO_TERM_FAILURE;
                break;
            }
            radius *= 0.5;
            continue;
        }
        invalid = 0;
        const double step_norm = sqrt(stepsq);
        if (step_norm <= 1e-8 * (x_norm + 1e-8)) {
            restore();
            sm.termination = RSO_TERM_PARAMETER;
            break;
        }
        const double cost_change = cost - ccost;
        if (fabs(cost_change) <= 1e-6 * cost) {
            restore();
            sm.termination = RSO_TERM_FUNCTION;
            break;
        }
        const double rel = cost_change / model_change;
        if (rel > 1e-3) {
            memcpy(p, pcur, sizeof(p));
            evaluate(true, &cost);  // residuals + Jet Jacobians at the accepted point
            xsq = 0.0, gmax = 0.0;
            for (int c = 0; c < np; ++c) gp[c] = 0.0;
            for (int64_t i = 0; i < m; ++i) {
                const double* row = &Jf[14 * (size_t)i];
                const double rho = *pb.residuals[(size_t)i]->blocks[3]->state;
                xsq += rho * rho;
                for (int c = 0; c < np; ++c) gp[c] += row[c] * r[2 * i] + row[7 + c] * r[2 * i + 1];
                const double g = fabs(Je[2 * i] * r[2 * i] + Je[2 * i + 1] * r[2 * i + 1]);
                if (g > gmax) gmax = g;
            }
            for (int c = 0; c < np; ++c) {
                xsq += p[c] * p[c];
                if (fabs(gp[c]) > gmax) gmax = fabs(gp[c]);
            }
            x_norm = sqrt(xsq);
            const double t = 2.0 * rel - 1.0;
            double f = 1.0 - t * t * t;
            if (f < 1.0 / 3.0) f = 1.0 / 3.0;
            radius = std::min(radius / f, 1e16);
            decrease_factor = 2.0;
            ++sm.num_successful_steps;
            if (gmax <= 1e-10) sm.termination = RSO_TERM_GRADIENT;
        } else {
            restore();
            ++sm.num_unsuccessful_steps;
            radius = radius / decrease_factor;
            decrease_factor *= 2.0;
        }
    }
    sm.num_iterations = iteration;
    sm.final_cost = cost;
    sm.final_radius = radius;
    if (summary) *summary = sm;
    return 0;
}

}  // namespace

extern "C" {

/* nonlinear_refinement::estimateInverseDepths (nonlinearRefinement.cc:109-180), structured as the reference */
int rsr_estimate_inverse_depths(const double* q, const double* u, int64_t n, const double v[3], const double w[3], double k,
                                const double* alpha, const double* alpha_k, double* rho_out, rso_lm_summary* summary) {
    if (n < 0) return -1;
    double lin[3] = {v[0], v[1], v[2]}, ang[3] = {w[0], w[1], w[2]}, k_local = k;
    double* inverse_depth = new double[(size_t)std::max<int64_t>(n, 1)];  // :123 "store depth values on the heap"
    int rc;
    {
        Problem problem;
        for (int64_t i = 0; i < n; ++i) {
            inverse_depth[i] = 1.0;  // :140
            CostFn* cost = new AutoDiffPixelCost(new PixelResidual{q[2 * i], q[2 * i + 1], u[2 * i], u[2 * i + 1], alpha[i], alpha_k[i]});
            problem.add_residual_block(cost, lin, ang, &k_local, &inverse_depth[i]);
        }
        if (n > 0) {
            problem.set_constant(&k_local);  // :156-158: only the depths are optimised
            problem.set_constant(lin);
            problem.set_constant(ang);
        }
        rc = solve(problem, summary);
    }  // the problem and every object in it are destroyed here, as at the end of the reference's call
    for (int64_t i = 0; i < n; ++i) rho_out[i] = inverse_depth[i];
    delete[] inverse_depth;
    return rc;
}

/* minimal::ransac (minimal.cc:209-306) over the structured depth solve: one problem build + solve + teardown per trial */
int rsr_ransac(const double* q, const double* u, const double* alpha, const double* alpha_k, int64_t n, int use_alpha_k, int32_t iterations,
               double tol, const int32_t* samples, int k_sign_mode, rso_ransac_out* out) {
    if (n < 9 || !samples || !out) return -1;
    std::vector<double> inv_depth((size_t)n);
    std::vector<uint8_t> mask((size_t)n);
    int64_t best_count = -1;
    double best_err = 0.0;
    out->best_trial = -1;
    for (int32_t t = 0; t < iterations; ++t) {
        double cq[18], cu[18], ca[9], cak[9];
        for (int j = 0; j < 9; ++j) {
            const int64_t idx = samples[t * 9 + j];
            if (idx < 0 || idx >= n) return -2;
            cq[2 * j] = q[2 * idx], cq[2 * j + 1] = q[2 * idx + 1], cu[2 * j] = u[2 * idx], cu[2 * j + 1] = u[2 * idx + 1];
            ca[j] = alpha[idx], cak[j] = alpha_k[idx];
        }
        double w[3], v[3], k;
        rso_calculate_velocities(cq, cu, ca, cak, use_alpha_k, k_sign_mode, w, v, &k);
        rso_lm_summary sm;
        rsr_estimate_inverse_depths(q, u, n, v, w, k, alpha, alpha_k, inv_depth.data(), &sm);
        double err = 0.0;
        const int64_t count = rso_score(q, u, alpha, alpha_k, n, v, w, k, inv_depth.data(), tol, mask.data(), &err);
        if (out->trial_count) out->trial_count[t] = count;
        if (out->trial_err) out->trial_err[t] = err;
        if (out->trial_steps) out->trial_steps[t] = sm.num_successful_steps;
        if (count > best_count || (count == best_count && err < best_err)) {  // minimal.cc:278-285
            best_count = count, best_err = err;
            out->best_trial = t;
            memcpy(out->w, w, sizeof(w)), memcpy(out->v, v, sizeof(v));
            out->k = k;
            if (out->inv_depth) memcpy(out->inv_depth, inv_depth.data(), sizeof(double) * (size_t)n);
            if (out->mask) memcpy(out->mask, mask.data(), (size_t)n);
        }
    }
    out->num_inliers = best_count < 0 ? 0 : best_count;
    out->inlier_error = best_err;
    int64_t o = 0;
    if (out->mask && out->inv_depth)
        for (int64_t j = 0; j < n; ++j) {  // minimal.cc:291-305
            if (!out->mask[j]) continue;
            if (out->inlier_idx) out->inlier_idx[o] = j;
            if (out->inliers) out->inliers[3 * o] = q[2 * j], out->inliers[3 * o + 1] = q[2 * j + 1], out->inliers[3 * o + 2] = 1.0 / out->inv_depth[j];
            if (out->alpha) out->alpha[o] = alpha[j];
            if (out->alpha_k) out->alpha_k[o] = alpha_k[j];
            ++o;
        }
    return 0;
}

/* nonlinear_refinement::nonLinearRefinement (nonlinearRefinement.cc:183-252), structured as the reference */
int rsr_refine(const double* flow, int64_t n_flow, int64_t m, const double* inl, const double* alpha, const double* alpha_k,
               const int64_t* inlier_idx, const double v_in[3], const double w_in[3], double k_in, int const_acceleration, int flow_index_mode,
               double* inl_out, double v_out[3], double w_out[3], double* k_out, rso_lm_summary* summary) {
    if (m < 0 || (flow_index_mode == 1 && !inlier_idx)) return -1;
    double lin[3] = {v_in[0], v_in[1], v_in[2]}, ang[3] = {w_in[0], w_in[1], w_in[2]}, k = k_in;
    double* inverse_depths = new double[(size_t)std::max<int64_t>(m, 1)];
    int rc = 0;
    {
        Problem problem;
        for (int64_t i = 0; i < m; ++i) {
            const int64_t fi = flow_index_mode == 1 ? inlier_idx[i] : i;  // :211-212 (quirk Q2)
            if (fi < 0 || fi >= n_flow) {
                delete[] inverse_depths;
                return -2;
            }
            inverse_depths[i] = 1.0 / inl[3 * i + 2];
            CostFn* cost = new AutoDiffPixelCost(new PixelResidual{inl[3 * i], inl[3 * i + 1], flow[2 * fi], flow[2 * fi + 1], alpha[i], alpha_k[i]});
            problem.add_residual_block(cost, lin, ang, &k, &inverse_depths[i]);
        }
        if (m > 0 && !const_acceleration) problem.set_constant(&k);  // :222-224
        if (m > 0) rc = solve(problem, summary);
        else if (summary) memset(summary, 0, sizeof(*summary));
    }
    for (int64_t i = 0; i < m; ++i) inl_out[3 * i] = inl[3 * i], inl_out[3 * i + 1] = inl[3 * i + 1], inl_out[3 * i + 2] = 1.0 / inverse_depths[i];
    delete[] inverse_depths;
    memcpy(v_out, lin, sizeof(lin)), memcpy(w_out, ang, sizeof(ang));
    *k_out = k;
    return rc;
}

}  // extern "C"
