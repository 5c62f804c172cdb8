// refine_rf_kernels.hip -- the joint refinement (nonlinearRefinement.cc:183-252) with RADIUS-FACTORISED Schur sums: ONE streaming pass per LM
// iteration whatever the step's quality, none after a rejected or invalid step.
//
// Ceres eliminates the inverse depths (1x1 e-blocks) and solves the reduced 6x6 / 7x7 system in the Jacobi-scaled parameters:
//     S = F^T F + D_f^2 - sum_i (E_i^T F_i)^T (E_i^T E_i + D_e,i^2)^-1 (E_i^T F_i),        D^2 = clamp(diag, 1e-6, 1e32) / radius.
// The radius enters an inlier's term only through  ete_inv_i = 1 / (ht_i + clamp(ht_i) / R),  ht_i = |J_rho,i|^2 s_i^2.  For every inlier whose
// clamp is inactive that is  psi(R) / ht_i  with the GLOBAL scalar  psi = 1 / (1 + 1 / R) -- and the Jacobi scale s_i of its column cancels:
//     S(R) = diag(sp) [ JtJ - psi(R) B ] diag(sp) + D_f^2(R),      B = sum_i (J_rho,i^T Jp_i)^T (J_rho,i^T Jp_i) / |J_rho,i|^2,
//     rhs(R) = diag(sp) [ Jtb - psi(R) c ],                         c = sum_i (J_rho,i^T Jp_i)^T (J_rho,i^T r_i) / |J_rho,i|^2,
// with UNSCALED Jacobians in every sum.  So the sums a pass takes at a point serve ANY radius.  A slot's pass back-substitutes iteration i (the
// radius and the reduced step are known by then) and takes JtJ, B, Jtb, c at the candidate it has just formed; the stage behind it decides
// iteration i and -- accepted with whatever quality -- solves iteration i + 1 from those sums at the radius Ceres' rule gives, or -- rejected /
// invalid -- solves again from the sums of the CURRENT point, which it has kept (RfExt), at the reduced radius: no pass.  Slots per solve = LM
// iterations that evaluated a candidate + 1 (the first pass, which is iteration zero and the Schur pass of iteration 1 in one: the
// iteration-zero sums -- column norms, gradient -- are the diagonal of JtJ and Jtb).
//
// Inliers whose clamp is active (|J_rho| s < 1e-3: within ~1 px of the focus of expansion; a handful at most) do not factor: the pass keeps them
// out of B and c and LISTS them (their data, 64 bytes each); the stage adds their exact terms for the radius in question; the pass's own
// back-substitution takes the exact branch for them.  More than kRfListCap of them: guard.
//
// Arithmetic: the library's own (fused multiply-adds, residual / Jacobian from the bilinear form of the model, the reduced system built from
// other sums than the reference's) -- oracle/rsdsfm_oracle.c rso_refine mode 2 restates it.  Every DECISION of the trust-region loop
// (invalid step, parameter / function / gradient tolerance, step quality against 1e-3, Cholesky pivots, non-finite sums) is checked against
// a relative band around its threshold; inside a band the solve ends with termination = kTermRestartExact and the host runs it again on the
// iterate-by-iterate slot kernels (refine_kernels.hip: the reference's arithmetic; rsdsfm_set_lm_arithmetic(1) selects them outright).
//
// Shape: workgroups of 8 waves, one per CU (two waves per SIMD): the 59 / 75 accumulators of a lane are 118 / 150 vector registers, which rules
// out four waves per SIMD (128 registers) for ANY body; the memory latency is covered by the next two inliers' loads in flight instead.
#include "refine_common.hpp"

// RF_LOADS_ONLY=1 (tools/build_rfproxy.sh -> tools/refine_rf_phases.py): a TIMING build whose NP = 6 pass performs the loop's loads and its one store
// but none of its arithmetic -- wrong results (every solve is sent back by a guard), the loop phase's duration is what is read: 10.1 us against
// 11.1 us with the arithmetic, i.e. the loop phase is bound by its memory accesses (51.6 MB per pass), not by instruction issue.
#ifndef RF_LOADS_ONLY
#define RF_LOADS_ONLY 0
#endif
#ifndef RF_NP7_PREFETCH
#define RF_NP7_PREFETCH 0
#endif

namespace rsdsfm {

namespace {

constexpr int kRfListCap = 64;  // listed (clamped) inliers per point, all ranks together
constexpr int kRfEntry = 8;     // doubles per list entry: x, y, u_x, u_y, beta | alpha, alpha_k, rho at the point, inlier index

// one row of a pass: the radius-free Schur sums at the point the pass formed, the back-substitution sums beside them
template <int NP>
struct RfRow {
    static constexpr int TRI = NP * (NP + 1) / 2;
    static constexpr int OFF_JTJ = 0, OFF_B = TRI, OFF_JTB = 2 * TRI, OFF_C = 2 * TRI + NP;
    static constexpr int NSCHUR = 2 * TRI + 2 * NP;
    static constexpr int MODEL = NSCHUR, STEPSQ = NSCHUR + 1, COST2 = NSCHUR + 2, XSQ = NSCHUR + 3, ZSUM = NSCHUR + 4, GMAX = NSCHUR + 5;
    static constexpr int NW = NSCHUR + 6;  // 60 / 76: even (16-byte loads in the reductions)
    static constexpr int NF = TRI + NP;    // what a listed inlier contributes to: B and c
    static_assert(NW % 2 == 0, "rows are read as double2");
};

// what the stage keeps of the CURRENT point (two copies alternate like the chunk states: in -> out): its Schur sums and its list
struct RfExt {
    double sums[RfRow<7>::NSCHUR];
    double nlist;
    double list[kRfListCap * kRfEntry];
};
constexpr int kRfExtDoubles = (int)((sizeof(RfExt) + 63) / 64 * 8);
// a list as the stage reads it: [count (int32 in the first word) | pad | entries]; the pass appends with an atomic counter
constexpr int kRfListDoubles = 2 + kRfListCap * kRfEntry;

enum RfGuard : int {
    kRfGuardNonFinite = 1, kRfGuardGradient = 2, kRfGuardModel = 3, kRfGuardParameter = 4, kRfGuardFunction = 5, kRfGuardQuality = 6,
    kRfGuardPivot = 7, kRfGuardList = 8, kRfGuardRadius = 9
};
// relative half-widths of the bands (what the two arithmetics can differ by at each test)
constexpr double kBandGradient = 1e-4, kBandParameter = 1e-2, kBandFunction = 1e-4, kBandQuality = 1e-4, kBandModel = 1e-12, kBandPivot = 1e-9;

// uniform parameters of one point of the model, as the per-inlier code reads them
struct RfPoint {
    double v0, v1, v2, w0, w1, w2, k, c1, c2;  // c1 = 2 / (2 + k), c2 = 2 / (2 + k)^2
};
__device__ __forceinline__ RfPoint rf_point(const double* p) {
    RfPoint P;
    P.v0 = p[0], P.v1 = p[1], P.v2 = p[2], P.w0 = p[3], P.w1 = p[4], P.w2 = p[5], P.k = p[6];
    const double t = 2.0 + p[6];
    P.c1 = 2.0 / t;
    P.c2 = 2.0 / (t * t);
    return P;
}
__device__ __forceinline__ RfPoint rf_point_uniform(const RfPoint& Q) {
    RfPoint P;
    P.v0 = uniform_d(Q.v0), P.v1 = uniform_d(Q.v1), P.v2 = uniform_d(Q.v2), P.w0 = uniform_d(Q.w0), P.w1 = uniform_d(Q.w1), P.w2 = uniform_d(Q.w2);
    P.k = uniform_d(Q.k), P.c1 = uniform_d(Q.c1), P.c2 = uniform_d(Q.c2);
    return P;
}

// the hypothesis-independent products of an inlier
struct RfGeom {
    double x, y, xy, xx1, yy1;
};
__device__ __forceinline__ RfGeom rf_geom(double x, double y) {
    RfGeom g;
    g.x = x, g.y = y, g.xy = x * y, g.xx1 = __builtin_fma(x, x, 1.0), g.yy1 = __builtin_fma(y, y, 1.0);
    return g;
}
// the model is bilinear: pred = beta ( rho A(x, y) v + B(x, y) w );  a = A v, bw = B w  (nonlinearRefinement.cc:36-49, signs folded)
__device__ __forceinline__ void rf_av(const RfGeom& g, double v0, double v1, double v2, double& a0, double& a1) {
    a0 = __builtin_fma(g.x, v2, -v0);
    a1 = __builtin_fma(g.y, v2, -v1);
}
__device__ __forceinline__ void rf_bw(const RfGeom& g, double w0, double w1, double w2, double& b0, double& b1) {
    b0 = __builtin_fma(g.y, w2, __builtin_fma(-g.xx1, w1, g.xy * w0));
    b1 = __builtin_fma(-g.x, w2, __builtin_fma(-g.xy, w1, g.yy1 * w0));
}
// beta and d beta / d k of an inlier at a point (nonlinearRefinement.cc:35): NP == 6 reads beta precomputed
template <int NP>
__device__ __forceinline__ void rf_beta(double ab, double ak, const RfPoint& P, double& be, double& dbe) {
    if (NP == 6) {
        be = ab, dbe = 0.0;
    } else {
        be = P.c1 * __builtin_fma(P.k, ak, ab);
        dbe = P.c2 * __builtin_fma(2.0, ak, -ab);
    }
}
// whether an inlier's clamp may be active at a point, from h = |J_rho|^2 there and h0 = |J_rho|^2 at the start parameters:
// ht = h / (1 + sqrt(h0))^2 >= 1e-6 is implied by h >= 2.02e-6 (1 + h0)  ((1 + t)^2 <= 2 (1 + t^2)); everything else (NaN included) is listed
__device__ __forceinline__ bool rf_flagged(double h, double h0) { return !(h >= __builtin_fma(2.02e-6, h0, 2.02e-6) && h <= 1e30); }

// residual, J_rho = beta a and the two rows of the UNSCALED parameter Jacobian at (P, rho)
template <int NP>
struct RfEval {
    double r0, r1, J0, J1, h, in0, in1;
    double P0[NP], P1[NP];
};
template <int NP>
__device__ __forceinline__ void rf_resid(const RfGeom& g, double ux, double uy, double be, const RfPoint& P, double rho, RfEval<NP>& o) {
    double a0, a1, b0, b1;
    rf_av(g, P.v0, P.v1, P.v2, a0, a1);
    rf_bw(g, P.w0, P.w1, P.w2, b0, b1);
    o.in0 = __builtin_fma(rho, a0, b0);
    o.in1 = __builtin_fma(rho, a1, b1);
    o.r0 = __builtin_fma(be, o.in0, ux);
    o.r1 = __builtin_fma(be, o.in1, uy);
    o.J0 = be * a0;
    o.J1 = be * a1;
    o.h = __builtin_fma(o.J0, o.J0, o.J1 * o.J1);
}
template <int NP>
__device__ __forceinline__ void rf_jac(const RfGeom& g, double be, double dbe, double rho, RfEval<NP>& o) {
    const double br = be * rho;
    o.P0[0] = -br, o.P1[0] = 0.0;
    o.P0[1] = 0.0, o.P1[1] = -br;
    o.P0[2] = br * g.x, o.P1[2] = br * g.y;
    o.P0[3] = be * g.xy, o.P1[3] = be * g.yy1;
    o.P0[4] = -(be * g.xx1), o.P1[4] = -(be * g.xy);
    o.P0[5] = be * g.y, o.P1[5] = -(be * g.x);
    if (NP == 7) {
        o.P0[NP - 1] = dbe * o.in0;
        o.P1[NP - 1] = dbe * o.in1;
    }
}
// (row 0 of column 1 and row 1 of column 0 are structural zeros: their products are left out, not multiplied)
__device__ __forceinline__ constexpr bool rf_z0(int c) { return c == 1; }
__device__ __forceinline__ constexpr bool rf_z1(int c) { return c == 0; }

// JtJ and Jtb of one inlier (every inlier), B and c (mask = 1 / h for an unlisted inlier, 0 for a listed one: straight-line code)
// lds (NP == 7 passes): 3 NP of the accumulators live in LDS, [slot][thread] -- c, and the k column of JtJ and of B: slots [0, NP) c_a,
// [NP, 2 NP) JtJ(a, k), [2 NP, 3 NP) B(a, k).  75 accumulators + the body's temporaries do not fit 256 registers, and what the compiler spills goes
// to SCRATCH memory: a kernel with a private segment that runs on several streams at once (the sequence solve's lanes) made the runtime's scratch
// provisioning hang the queue and corrupt spilled values (reproduced with tools/seq_determinism_probe.py, gone with the private segment).
template <int NP>
__device__ __forceinline__ void rf_schur_accumulate(const RfEval<NP>& o, double ih_mask, double (&acc)[RfRow<NP>::NW], double* lds = nullptr) {
    using RR = RfRow<NP>;
    double EJ[NP], W[NP];
#pragma unroll
    for (int c = 0; c < NP; ++c) {
        if (rf_z1(c))
            EJ[c] = o.J0 * o.P0[c];
        else if (rf_z0(c))
            EJ[c] = o.J1 * o.P1[c];
        else
            EJ[c] = __builtin_fma(o.J0, o.P0[c], o.J1 * o.P1[c]);
        W[c] = EJ[c] * ih_mask;
    }
    const double gr = __builtin_fma(o.J0, o.r0, o.J1 * o.r1);
    int tri = 0;
#pragma unroll
    for (int a = 0; a < NP; ++a) {
        if (rf_z1(a))
            acc[RR::OFF_JTB + a] = __builtin_fma(o.P0[a], o.r0, acc[RR::OFF_JTB + a]);
        else if (rf_z0(a))
            acc[RR::OFF_JTB + a] = __builtin_fma(o.P1[a], o.r1, acc[RR::OFF_JTB + a]);
        else
            acc[RR::OFF_JTB + a] = __builtin_fma(o.P0[a], o.r0, __builtin_fma(o.P1[a], o.r1, acc[RR::OFF_JTB + a]));
        if (lds)
            lds[a * kFB] = __builtin_fma(W[a], gr, lds[a * kFB]);
        else
            acc[RR::OFF_C + a] = __builtin_fma(W[a], gr, acc[RR::OFF_C + a]);
#pragma unroll
        for (int b = a; b < NP; ++b) {
            const bool t0 = !(rf_z0(a) || rf_z0(b)), t1 = !(rf_z1(a) || rf_z1(b));
            const bool in_lds = lds != nullptr && b == NP - 1;
            double jj = in_lds ? lds[(NP + a) * kFB] : acc[RR::OFF_JTJ + tri];
            if (t0 && t1)
                jj = __builtin_fma(o.P0[a], o.P0[b], __builtin_fma(o.P1[a], o.P1[b], jj));
            else if (t0)
                jj = __builtin_fma(o.P0[a], o.P0[b], jj);
            else if (t1)
                jj = __builtin_fma(o.P1[a], o.P1[b], jj);
            if (in_lds) {
                lds[(NP + a) * kFB] = jj;
                lds[(2 * NP + a) * kFB] = __builtin_fma(EJ[a], W[b], lds[(2 * NP + a) * kFB]);
            } else {
                acc[RR::OFF_JTJ + tri] = jj;
                acc[RR::OFF_B + tri] = __builtin_fma(EJ[a], W[b], acc[RR::OFF_B + tri]);
            }
            ++tri;
        }
    }
}

// the exact e-block inverse of a listed inlier at radius R: 1 / (ht + clamp(ht) / R), ht = |J_rho s|^2, s = 1 / (1 + |J_rho(x0)|); also returns s
__device__ __forceinline__ double rf_ete_inv_exact(double J0, double J1, double h0, double inv_radius, double& sr, double& E0, double& E1) {
    sr = 1.0 / (1.0 + sqrt(h0));
    E0 = J0 * sr, E1 = J1 * sr;
    const double ht = __builtin_fma(E0, E0, E1 * E1);
    const double lam = clampd(ht, kMinLmDiag, kMaxLmDiag) * inv_radius;
    return 1.0 / (ht + lam);
}

__device__ __forceinline__ void rf_list_append(int* __restrict__ count, double* __restrict__ entries, double x, double y, double ux, double uy, double ab,
                                               double ak, double rho, int64_t idx) {
    const int k = atomicAdd(count, 1);
    if (k < kRfListCap) {
        double* e = entries + (size_t)k * kRfEntry;
        e[0] = x, e[1] = y, e[2] = ux, e[3] = uy, e[4] = ab, e[5] = ak, e[6] = rho, e[7] = (double)idx;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// the single-workgroup stage, run redundantly in every workgroup's prologue (and by refine_rf_apply_kernel behind the last pass of a chunk)
// ---------------------------------------------------------------------------------------------------------------------------------------
struct RfLists {  // the lists of the pass whose rows are applied, one per rank (a single context: one): counters and entries
    const int* counts;
    const double* entries;
    int nlists, count_stride, entry_stride;  // (strides in ints / doubles)
};

// LDS of the stage
template <int NP>
struct RfStageLds {
    using RR = RfRow<NP>;
    double grp[ReduceShape<RR::NW>::G][RR::NW];
    double s[RR::NW];                       // the reduced row
    double cur[RR::NSCHUR];                 // Schur sums of the current point
    double list[kRfListCap * kRfEntry];     // its listed inliers, sorted by (rank, index)
    double fe[kRfListCap][RR::NF];          // their terms at the radius in question
    double F[RR::NF];
    int nlist, action, loop, bad;
    double radius;
};

// Fixed-order reduction of rows[nrows][stride] (the first NW doubles of each; the LAST one a maximum of absolute values, the others sums) into
// group sums grp[G][NW]: thread (group g, slot pair sp) adds the rows g, g + G, g + 2 G, ... of its two slots in order, 16 loads in flight
// (reduce_partials_groups' scheme with the one max slot in a fixed place: no per-element select); reduce_partials_slots finishes it.
template <int NW>
__device__ __forceinline__ void rf_reduce_rows_groups(const double* __restrict__ rows, int nrows, int stride, double (*grp)[NW], int tid) {
    constexpr int NH = NW / 2, G = kFB / NH, U = 16;
    static_assert(G == ReduceShape<NW>::G, "grp is sized by ReduceShape");
    const int g = tid / NH, sp = tid - g * NH;
    if (g < G) {
        const bool last = sp == NH - 1;
        double a0 = 0.0, a1 = 0.0, m1 = 0.0;
        const double2* __restrict__ base = reinterpret_cast<const double2*>(rows) + sp;
        const int stride2 = stride / 2;
        for (int b = g; b < nrows; b += U * G) {
            double2 x[U];
#pragma unroll
            for (int j = 0; j < U; ++j) {
                const int bj = b + j * G;
                x[j] = bj < nrows ? base[(uint32_t)bj * (uint32_t)stride2] : make_double2(0.0, 0.0);
            }
#pragma unroll
            for (int j = 0; j < U; ++j) {
                a0 += x[j].x;
                a1 += x[j].y;
                m1 = fmax(m1, x[j].y);
            }
        }
        grp[g][2 * sp] = a0;
        grp[g][2 * sp + 1] = last ? m1 : a1;
    }
}

// one listed inlier's exact terms of B and c at (P, R)
template <int NP>
__device__ __forceinline__ void rf_listed_terms(const double* __restrict__ e, const RfPoint& P, const RfPoint& P0, double inv_radius, double* __restrict__ out) {
    using RR = RfRow<NP>;
    const RfGeom g = rf_geom(e[0], e[1]);
    double be, dbe, be0, dbe0;
    rf_beta<NP>(e[4], e[5], P, be, dbe);
    rf_beta<NP>(e[4], e[5], P0, be0, dbe0);
    RfEval<NP> o;
    rf_resid<NP>(g, e[2], e[3], be, P, e[6], o);
    rf_jac<NP>(g, be, dbe, e[6], o);
    double a00, a10;
    rf_av(g, P0.v0, P0.v1, P0.v2, a00, a10);
    const double h0 = (be0 * be0) * __builtin_fma(a00, a00, a10 * a10);
    double sr, E0, E1;
    const double ete_inv = rf_ete_inv_exact(o.J0, o.J1, h0, inv_radius, sr, E0, E1);
    const double Etb = __builtin_fma(E0, o.r0, E1 * o.r1);
    double EJ[NP], W[NP];
#pragma unroll
    for (int c = 0; c < NP; ++c) {
        EJ[c] = __builtin_fma(E0, o.P0[c], E1 * o.P1[c]);
        W[c] = ete_inv * EJ[c];
    }
    int tri = 0;
#pragma unroll
    for (int a = 0; a < NP; ++a) {
        out[RR::TRI + a] = W[a] * Etb;
#pragma unroll
        for (int b = a; b < NP; ++b) out[tri++] = EJ[a] * W[b];
    }
}

// the reduced system at radius R from the sums of the current point (+ the listed inliers' terms F), Cholesky with pivots checked against
// the band; returns 1 solved, 0 not positive definite (an invalid step, as the reference's failed factorisation), -1 inside the band
template <int NP>
__device__ __forceinline__ int rf_solve(RefineState* st, const double* cur, const double* F, bool have_F, double radius) {
    using RR = RfRow<NP>;
    const double inv_radius = 1.0 / radius;
    const double psi = 1.0 / (1.0 + inv_radius);
    double S[NP][NP], rhs[NP], sp[NP];
#pragma unroll
    for (int c = 0; c < NP; ++c) sp[c] = st->sp[c];
    {
        int tri = 0;
#pragma unroll
        for (int a = 0; a < NP; ++a) {
            double ra = __builtin_fma(-psi, cur[RR::OFF_C + a], cur[RR::OFF_JTB + a]);
            if (have_F) ra -= F[RR::TRI + a];
            rhs[a] = ra * sp[a];
#pragma unroll
            for (int b = a; b < NP; ++b) {
                double mab = __builtin_fma(-psi, cur[RR::OFF_B + tri], cur[RR::OFF_JTJ + tri]);
                if (have_F) mab -= F[tri];
                double sab = (mab * sp[a]) * sp[b];
                if (a == b) sab = __builtin_fma(clampd((cur[RR::OFF_JTJ + tri] * sp[a]) * sp[a], kMinLmDiag, kMaxLmDiag), inv_radius, sab);  // + D_f^2
                S[a][b] = sab;
                S[b][a] = sab;
                ++tri;
            }
        }
    }
    int status = 1;
    double inv_d[NP];
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        if (status == 1) {
            const double sjj = S[j][j];
            double d = sjj;
#pragma unroll
            for (int t = 0; t < j; ++t) d = __builtin_fma(-S[j][t], S[j][t], d);
            if (!(fabs(d) > kBandPivot * fabs(sjj)) || !(fabs(d) > 1e-200 && fabs(d) < 1e200)) {
                status = -1;  // (NaN lands here too; and a pivot outside the range of the function cores below)
            } else if (d < 0.0) {
                status = 0;
            } else {
                const double id = rcp_core(sqrt_core(d));  // = 1.0 / sqrt(d), correctly rounded twice (device_math.hpp)
                inv_d[j] = id;
#pragma unroll
                for (int i = j + 1; i < NP; ++i) {
                    double sacc = S[i][j];
#pragma unroll
                    for (int t = 0; t < j; ++t) sacc = __builtin_fma(-S[i][t], S[j][t], sacc);
                    S[i][j] = sacc * id;
                }
            }
        }
    }
    if (status != 1) return status;
    double yv[NP], yp[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        double sacc = rhs[i];
#pragma unroll
        for (int t = 0; t < i; ++t) sacc = __builtin_fma(-S[i][t], yv[t], sacc);
        yv[i] = sacc * inv_d[i];
    }
#pragma unroll
    for (int i = NP - 1; i >= 0; --i) {
        double sacc = yv[i];
#pragma unroll
        for (int t = i + 1; t < NP; ++t) sacc = __builtin_fma(-S[t][i], yp[t], sacc);
        yp[i] = sacc * inv_d[i];
    }
    double stepsq = 0.0;
#pragma unroll
    for (int c = 0; c < 7; ++c) st->pc[c] = st->p[c], st->dp[c] = 0.0;
#pragma unroll
    for (int c = 0; c < NP; ++c) {
        const double d = -(yp[c] * sp[c]);
        st->yp[c] = yp[c];
        st->dp[c] = d;
        st->pc[c] = st->p[c] + d;
        stepsq = __builtin_fma(d, d, stepsq);
    }
    st->stepsq_p = stepsq;
    return 1;
}

__device__ __forceinline__ bool rf_in_band(double value, double threshold, double band) { return fabs(value - threshold) <= band * fabs(threshold); }

__device__ __forceinline__ void rf_restart(RefineState* st, int guard) {
    st->termination = kTermRestartExact;
    st->rf_guard = guard;
}

// `first`: the rows are the first pass's (iteration zero + the Schur sums of iteration 1).  ext_out: written by the publishing workgroup only.
template <int NP>
__device__ __forceinline__ void rf_apply_body(RefineState* st, RfStageLds<NP>& L, const double* __restrict__ rows, int nrows, int row_stride, RfLists lists,
                              const RfExt* __restrict__ ext_in, RfExt* __restrict__ ext_out, int64_t m_total, double* __restrict__ trace, int trace_rows,
                              unsigned long long* tks = nullptr, unsigned long long* beat = nullptr) {
    using RR = RfRow<NP>;
    const int tid = threadIdx.x;
#define RF_BEAT(code) do { if (beat && tid == 0) beat[1] = (beat[1] & ~255ull) | (code); } while (0)
    RF_BEAT(10);
    // the reduced row (the same order for workgroup partials and for gathered rank rows)
    if (tid == 0) L.bad = 0;
    rf_reduce_rows_groups<RR::NW>(rows, nrows, row_stride, L.grp, tid);
    // the lists' counts beside it
    int ntot = 0;
    bool overflow = false;
    for (int r = 0; r < lists.nlists; ++r) {
        const int n = lists.counts[(size_t)r * lists.count_stride];
        overflow = overflow || n > kRfListCap || n < 0;
        ntot += n;
    }
    overflow = overflow || ntot > kRfListCap;
    __syncthreads();
    reduce_partials_slots<RR::NW>(L.grp, RR::GMAX, L.s, tid);
    if (tid < RR::NW && !(fabs(L.s[tid]) < 1e300)) L.bad = 1;  // (a non-finite sum, NaN included: every thread looks at the slot it has just written)
    __syncthreads();
    if (tks) tks[0] = wall_clock64();
    RF_BEAT(11);
    const bool first = st->iteration == 0 && st->num_unsuccessful == 0 && st->slots == 0;
    // ---- the decision of the iteration whose back-substitution the pass carried (lane 0) ----
    if (tid == 0) {
        const double* s = L.s;
        int action = 0;  // 0: the current point stays (rejected / invalid), 1: the pass's point becomes the current one, 2: over
        st->slots += 1;
        double* tr = (trace && st->iteration >= 1 && st->iteration <= trace_rows) ? trace + (int64_t)(st->iteration - 1) * kRefineTraceCols : nullptr;
        const bool finite = L.bad == 0;
        if (!finite || overflow) {
            rf_restart(st, overflow ? kRfGuardList : kRfGuardNonFinite);
            action = 2;
        } else if (first) {
            // iteration zero (rso_refine :1641-1671): cost, gradient, Jacobi scales of the parameter columns from the diagonal of JtJ
            double gmax = s[RR::GMAX], xsq = s[RR::XSQ];
            int tri = 0;
#pragma unroll
            for (int c = 0; c < NP; ++c) {
                st->sp[c] = 1.0 / (1.0 + sqrt(s[RR::OFF_JTJ + tri]));
                tri += NP - c;
                gmax = fmax(gmax, fabs(s[RR::OFF_JTB + c]));
                xsq = __builtin_fma(st->p[c], st->p[c], xsq);
            }
            st->cost = 0.5 * s[RR::COST2];
            st->initial_cost = st->cost;
            st->zsum = s[RR::ZSUM];
            st->gmax = gmax;
            st->x_norm = sqrt(xsq);
            st->radius = kInitialRadius;
            st->decrease_factor = 2.0;
            action = 1;
            if (m_total == 0 || gmax <= kGradientTol) st->termination = RSDSFM_TERM_GRADIENT, action = 2;
            if (m_total != 0 && rf_in_band(gmax, kGradientTol, kBandGradient)) rf_restart(st, kRfGuardGradient), action = 2;
        } else {
            const double model_change = s[RR::MODEL];
            if (tr) {
                tr[0] = (double)st->iteration, tr[1] = st->cost, tr[3] = model_change, tr[5] = st->radius;
                tr[2] = tr[4] = tr[6] = __builtin_nan("");
            }
            if (fabs(model_change) <= kBandModel * st->cost) {
                rf_restart(st, kRfGuardModel);
                action = 2;
            } else if (!(model_change > 0.0)) {  // HandleInvalidStep
                if (tr) tr[7] = RSDSFM_TRACE_INVALID;
                st->num_unsuccessful += 1;
                st->invalid_run += 1;
                if (st->invalid_run >= kMaxInvalid)
                    st->termination = RSDSFM_TERM_FAILURE, action = 2;
                else
                    st->radius *= 0.5;
            } else {
                st->invalid_run = 0;
                const double step_norm = sqrt(st->stepsq_p + s[RR::STEPSQ]);
                const double ccost = 0.5 * s[RR::COST2];
                const double ptol = kParameterTol * (st->x_norm + kParameterTol);
                const double cost_change = st->cost - ccost;
                const double ftol = kFunctionTol * st->cost;
                const double rel = cost_change / model_change;
                if (tr) tr[2] = ccost, tr[6] = step_norm;
                if (rf_in_band(step_norm, ptol, kBandParameter)) {
                    rf_restart(st, kRfGuardParameter), action = 2;
                } else if (step_norm <= ptol) {
                    if (tr) tr[7] = RSDSFM_TRACE_PARAMETER_TOL;
                    st->termination = RSDSFM_TERM_PARAMETER, action = 2;
                } else if (rf_in_band(fabs(cost_change), ftol, kBandFunction)) {
                    rf_restart(st, kRfGuardFunction), action = 2;
                } else if (fabs(cost_change) <= ftol) {
                    if (tr) tr[7] = RSDSFM_TRACE_FUNCTION_TOL;
                    st->termination = RSDSFM_TERM_FUNCTION, action = 2;
                } else if (rf_in_band(rel, kMinRelDecrease, kBandQuality)) {
                    rf_restart(st, kRfGuardQuality), action = 2;
                } else if (rel > kMinRelDecrease) {  // HandleSuccessfulStep
                    if (tr) tr[4] = rel;
                    double gmax = s[RR::GMAX], xsq = s[RR::XSQ];
#pragma unroll
                    for (int c = 0; c < 7; ++c) st->p[c] = st->pc[c];
#pragma unroll
                    for (int c = 0; c < NP; ++c) {
                        xsq = __builtin_fma(st->p[c], st->p[c], xsq);
                        gmax = fmax(gmax, fabs(s[RR::OFF_JTB + c]));
                    }
                    st->cur ^= 1;
                    st->zsum = s[RR::ZSUM];
                    st->cost = ccost;
                    st->gmax = gmax;
                    st->x_norm = sqrt(xsq);
                    st->radius = radius_accept(st->radius, rel);
                    st->decrease_factor = 2.0;
                    st->num_successful += 1;
                    action = 1;
                    if (rf_in_band(gmax, kGradientTol, kBandGradient)) {
                        rf_restart(st, kRfGuardGradient), action = 2;
                    } else if (gmax <= kGradientTol) {
                        st->termination = RSDSFM_TERM_GRADIENT, action = 2;
                    }
                    if (tr) tr[7] = (st->termination == RSDSFM_TERM_GRADIENT) ? RSDSFM_TRACE_ACCEPTED_GRADIENT_TOL : RSDSFM_TRACE_ACCEPTED;
                } else {  // HandleUnsuccessfulStep
                    if (tr) tr[4] = rel, tr[7] = RSDSFM_TRACE_REJECTED;
                    st->num_unsuccessful += 1;
                    st->radius = st->radius / st->decrease_factor;
                    st->decrease_factor *= 2.0;
                }
            }
        }
        L.action = action;
    }
    __syncthreads();
    const int action = L.action;
    // ---- the current point's sums and list: the pass's (its point was taken) or the kept ones ----
    if (action == 1) {
        if (tid < RR::NSCHUR) L.cur[tid] = L.s[tid];
        // gather the lists in (rank, index) order
        int base = 0;
        for (int r = 0; r < lists.nlists; ++r) {
            const double* lr = lists.entries + (size_t)r * lists.entry_stride;
            const int n = lists.counts[(size_t)r * lists.count_stride];
            if (tid < n) {
                const double* e = lr + (size_t)tid * kRfEntry;
                const double key = e[7];
                int rank = 0;
                for (int j = 0; j < n; ++j) rank += (lr[(size_t)j * kRfEntry + 7] < key) ? 1 : 0;
#pragma unroll
                for (int q = 0; q < kRfEntry; ++q) L.list[(size_t)(base + rank) * kRfEntry + q] = e[q];
            }
            base += n;
        }
        if (tid == 0) L.nlist = base;
    } else if (action == 0) {
        if (tid < RR::NSCHUR) L.cur[tid] = ext_in->sums[tid];
        const int n = (int)ext_in->nlist;
        for (int q = tid; q < n * kRfEntry; q += kFB) L.list[q] = ext_in->list[q];
        if (tid == 0) L.nlist = n;
    }
    __syncthreads();
    if (tks) tks[1] = wall_clock64();
    RF_BEAT(12 + action);
    if (action == 2) return;
    // ---- the reduced solve of the next iteration, again at half the radius while the system does not factor (no pass needed) ----
    const RfPoint P0 = rf_point(st->p0);
    bool resolve = action == 0;
    for (;;) {
        if (tid == 0) {
            int go = 1;
            if (st->iteration >= kMaxIter) st->termination = RSDSFM_TERM_MAX_ITER, go = 0;  // top-of-loop checks of TrustRegionMinimizer
            else if (rf_in_band(st->radius, kMinRadius, 1e-6)) rf_restart(st, kRfGuardRadius), go = 0;
            else if (st->radius <= kMinRadius) st->termination = RSDSFM_TERM_MIN_RADIUS, go = 0;
            if (go) st->iteration += 1;
            L.loop = go;
            L.radius = st->radius;
        }
        __syncthreads();
        RF_BEAT(20);
        const int go = L.loop, n = L.nlist;
        const double radius = L.radius;
        __syncthreads();  // (read by everyone before lane 0 writes L.loop again)
        if (!go) break;
        if (n > 0) {
            const RfPoint P = rf_point(st->p);
            if (tid < n) rf_listed_terms<NP>(L.list + (size_t)tid * kRfEntry, P, P0, 1.0 / radius, L.fe[tid]);
            __syncthreads();
            if (tid < RR::NF) {
                double f = L.fe[0][tid];
                for (int e = 1; e < n; ++e) f += L.fe[e][tid];
                L.F[tid] = f;
            }
            __syncthreads();
        }
        if (tid == 0) {
            const int rc = rf_solve<NP>(st, L.cur, L.F, n > 0, radius);
            int again = 0;
            if (rc < 0) {
                rf_restart(st, kRfGuardPivot);
            } else if (rc == 0) {  // the reference's failed factorisation: an invalid step (no back-substitution, no candidate)
                double* tr = (trace && st->iteration >= 1 && st->iteration <= trace_rows) ? trace + (int64_t)(st->iteration - 1) * kRefineTraceCols : nullptr;
                if (tr) {
                    tr[0] = (double)st->iteration, tr[1] = st->cost, tr[3] = 0.0, tr[5] = st->radius, tr[7] = RSDSFM_TRACE_INVALID;
                    tr[2] = tr[4] = tr[6] = __builtin_nan("");
                }
                st->num_unsuccessful += 1;
                st->invalid_run += 1;
                if (st->invalid_run >= kMaxInvalid)
                    st->termination = RSDSFM_TERM_FAILURE;
                else
                    st->radius *= 0.5, again = 1;
            } else if (resolve) {
                st->rf_resolves += 1;
            }
            st->solve_ok = rc == 1 ? 1 : 0;
            L.loop = again;
        }
        __syncthreads();
        RF_BEAT(21);
        const int again = L.loop;
        __syncthreads();
        if (!again) break;
        resolve = true;
    }
    RF_BEAT(30);
    // ---- what the next stage may need of the current point ----
    if (ext_out) {
        if (tid < RR::NSCHUR) ext_out->sums[tid] = L.cur[tid];
        const int n = L.nlist;
        for (int q = tid; q < n * kRfEntry; q += kFB) ext_out->list[q] = L.list[q];
        if (tid == 0) ext_out->nlist = (double)n;
    }
    RF_BEAT(31);
#undef RF_BEAT
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// the streaming pass
// ---------------------------------------------------------------------------------------------------------------------------------------
struct RfPassArgs {
    int64_t m;                  // < 0: the count and the logical grid are the state's
    const double4* xyuv_in;     // later passes read the records the first pass wrote
    double4* xyuv_out;
    double* beta;               // NP == 6: written by the first pass, read by the others
    const double* alpha;
    const double* alpha_k;
    double* rho_a;
    double* rho_b;
    const RefineState* st_in;
    RefineState* st_out;
    const double* rows_prev;    // the previous pass's rows ([nrows_prev][row_stride]; nrows_prev < 0: the state's logical grid)
    int nrows_prev, row_stride;
    RfLists lists_prev;
    int* list_count;            // this pass appends here ...
    double* list_entries;
    int* list_count_zero;       // ... and leaves the NEXT pass's counter at zero (three counters rotate: the one before is still being read)
    const RfExt* ext_in;
    RfExt* ext_out;
    double* partials;           // [grid][NW]
    double* trace;
    int trace_rows;
    int64_t m_total;            // inliers of ALL ranks (what iteration zero's m == 0 test looks at); < 0: this context's count
    const int64_t* m_total_dev;
    // first pass only
    const double2* flow;
    int64_t n_flow;
    const double* inl;
    const int64_t* inlier_idx;
    int flow_index_mode;
    int* bad_index;
    unsigned long long* beat;    // opt-in (RSDSFM_SYNC_WATCHDOG_S): host-mapped heartbeat words of this context, see rf_beat
    int slot;
    unsigned long long* stamps;  // opt-in (RSDSFM_RF_STAMPS=1): workgroup 0 adds the 100 MHz ticks of its phases here (tools/refine_rf_probe.py)
};

// per-inlier inputs of a later pass, as loaded
struct RfLoad {
    double4 c4;
    double rho, ab, ak;
};
template <int NP>
__device__ __forceinline__ RfLoad rf_load(const RfPassArgs& A, const double* __restrict__ rho, int64_t i, bool live) {
    RfLoad l;
    l.c4 = make_double4(0.0, 0.0, 0.0, 0.0), l.rho = 0.0, l.ab = 0.0, l.ak = 0.0;
    if (live) {
        l.c4 = A.xyuv_in[i];
        l.rho = rho[i];
        if (NP == 6) {
            l.ab = A.beta[i];
        } else {
            l.ab = A.alpha[i], l.ak = A.alpha_k[i];
        }
    }
    return l;
}

}  // namespace

// (the kernels themselves have external names: rocprofv3 and the profile summaries list them by name)
template <int NP, bool FIRST, bool ZSUM>
__global__ __launch_bounds__(kFB) void refine_rf_pass_kernel(const RfPassArgs A) {
    using RR = RfRow<NP>;
    __shared__ RfStageLds<NP> s_stage;
    __shared__ double s_red[kFB / 64][RR::NW];
    __shared__ RefineState s_state;
    const bool stamp = A.stamps && blockIdx.x == 0 && threadIdx.x == 0;
    if (A.beat && threadIdx.x == 0) {
        if (A.beat[7]) atomicAdd(&A.beat[2], 1ull);  // ([7]: count every workgroup of this context's passes that started ... -- a system-scope atomic per workgroup: slow)
        if (blockIdx.x == 0) A.beat[0] = ((unsigned long long)A.slot << 8) | 1;
        if (blockIdx.x < 512) A.beat[8 + blockIdx.x] = ((unsigned long long)A.slot << 8) | 1;
    }
    __shared__ unsigned long long tk[8];  // (LDS, not a private array: a private array indexed through a pointer lives in scratch memory, and a kernel that needs scratch is a kernel the runtime has to provision for on every queue)
    if (stamp) {
#pragma unroll
        for (int i = 0; i < 8; ++i) tk[i] = 0;
    }
    if (stamp) tk[0] = wall_clock64();
    state_to_lds(&s_state, A.st_in);
    RefineState* st = &s_state;
    if (stamp) tk[1] = wall_clock64();
    if (FIRST) {
        if (threadIdx.x < 7) st->p0[threadIdx.x] = st->p[threadIdx.x];
        if (threadIdx.x == 0) st->rf = 1, st->rf_guard = 0, st->rf_resolves = 0, st->pending_apply = 0;
        __syncthreads();
    } else if (st->termination < 0 && st->pending_apply) {
        int64_t mt = A.m_total_dev ? *A.m_total_dev : (A.m_total >= 0 ? A.m_total : (A.m >= 0 ? A.m : st->m));
        rf_apply_body<NP>(st, s_stage, A.rows_prev, A.nrows_prev >= 0 ? A.nrows_prev : st->grid, A.row_stride, A.lists_prev, A.ext_in,
                          blockIdx.x == 0 ? A.ext_out : nullptr, mt, blockIdx.x == 0 ? A.trace : nullptr, A.trace_rows, stamp ? tk + 4 : nullptr);
        __syncthreads();
    }
    if (threadIdx.x == 0) st->pending_apply = st->termination < 0 ? 1 : 0;  // (this pass runs: its rows wait for a stage)
    __syncthreads();
    if (stamp) tk[2] = wall_clock64();
    if (A.beat && threadIdx.x == 0) {
        if (blockIdx.x == 0) A.beat[0] = ((unsigned long long)A.slot << 8) | 2;
        if (blockIdx.x < 512) A.beat[8 + blockIdx.x] = ((unsigned long long)A.slot << 8) | ((st->termination >= 0 || (A.m < 0 && (int)blockIdx.x >= st->grid)) ? 9 : 2);
        if (A.beat[7] && (st->termination >= 0 || (A.m < 0 && (int)blockIdx.x >= st->grid))) atomicAdd(&A.beat[3], 1ull);  // ... and that left (here, or at the end)
    }
    if (blockIdx.x == 0) {
        state_from_lds(A.st_out, st);
        if (threadIdx.x == 0) *A.list_count_zero = 0;
    }
    if (st->termination >= 0) return;
    const PassShape ps = pass_shape(st, A.m);
    if (!ps.live) return;
    const int64_t m = ps.m;
    double acc[RR::NW];
#pragma unroll
    for (int s = 0; s < RR::NW; ++s) acc[s] = 0.0;
    constexpr bool kLdsC = NP == 7 && !FIRST;  // (see rf_schur_accumulate)
    __shared__ double s_cacc[kLdsC ? 3 * NP : 1][kLdsC ? kFB : 1];
    double* const lds_c = kLdsC ? &s_cacc[0][threadIdx.x] : nullptr;
    if (kLdsC) {
#pragma unroll
        for (int a = 0; a < 3 * NP; ++a) lds_c[a * kFB] = 0.0;
    }
    const int64_t stride = (int64_t)ps.grid * kFB;
    const int64_t i0 = (int64_t)blockIdx.x * kFB + threadIdx.x;
    RfPoint P0q = rf_point(st->p0);
    const RfPoint P0 = rf_point_uniform(P0q);
    if (FIRST) {
        RfPoint Pq = rf_point(st->p);
        const RfPoint P = rf_point_uniform(Pq);
        for (int64_t i = i0; i < m; i += stride) {
            int64_t fi = (A.flow_index_mode == RSDSFM_FLOW_GATHERED) ? A.inlier_idx[i] : i;
            if (fi < 0 || fi >= A.n_flow) {
                *A.bad_index = 1;
                fi = 0;
            }
            const double2 f = A.flow[fi];
            const double x = A.inl[3 * i], y = A.inl[3 * i + 1];
            const double rho = 1.0 / A.inl[3 * i + 2];  // nonlinearRefinement.cc:213
            const double al = A.alpha[i], ak = A.alpha_k[i];
            A.xyuv_out[i] = make_double4(x, y, f.x, f.y);
            const double ab = NP == 6 ? P.c1 * __builtin_fma(P.k, ak, al) : al;
            if (NP == 6) A.beta[i] = ab;
            A.rho_a[i] = rho;
            const RfGeom g = rf_geom(x, y);
            double be, dbe;
            rf_beta<NP>(ab, ak, P, be, dbe);
            RfEval<NP> o;
            rf_resid<NP>(g, f.x, f.y, be, P, rho, o);
            rf_jac<NP>(g, be, dbe, rho, o);
            acc[RR::COST2] = __builtin_fma(o.r0, o.r0, __builtin_fma(o.r1, o.r1, acc[RR::COST2]));
            acc[RR::GMAX] = fmax(acc[RR::GMAX], fabs(__builtin_fma(o.J0, o.r0, o.J1 * o.r1)));
            acc[RR::XSQ] = __builtin_fma(rho, rho, acc[RR::XSQ]);
            if (ZSUM) acc[RR::ZSUM] += 1.0 / rho;
            const bool flagged = rf_flagged(o.h, o.h);  // (the start point: h0 = h)
            const double ihm = flagged ? 0.0 : rcp_core(flagged ? 1.0 : o.h);
            rf_schur_accumulate<NP>(o, ihm, acc);
            if (flagged) rf_list_append(A.list_count, A.list_entries, x, y, f.x, f.y, ab, ak, rho, i);
        }
    } else {
        RfPoint Pq = rf_point(st->p), Pcq = rf_point(st->pc);
        const RfPoint P = rf_point_uniform(Pq), Pc = rf_point_uniform(Pcq);
        double dp[7];
#pragma unroll
        for (int c = 0; c < 7; ++c) dp[c] = uniform_d(st->dp[c]);
        const double inv_radius = uniform_d(1.0 / st->radius);
        const double psi = uniform_d(1.0 / (1.0 + 1.0 / st->radius));
        const double* __restrict__ rho = st->cur ? A.rho_b : A.rho_a;
        double* __restrict__ cand = st->cur ? A.rho_a : A.rho_b;
        // one inlier: back-substitution at the current point, then everything at the candidate
        auto body = [&](const RfLoad& l0, int64_t i) {
            const double x = l0.c4.x, y = l0.c4.y, ux = l0.c4.z, uy = l0.c4.w, rh = l0.rho;
            const RfGeom g = rf_geom(x, y);
            double be, dbe, bec, dbec, be0, dbe0;
            rf_beta<NP>(l0.ab, l0.ak, P, be, dbe);
            rf_beta<NP>(l0.ab, l0.ak, Pc, bec, dbec);
            rf_beta<NP>(l0.ab, l0.ak, P0, be0, dbe0);
            // the current point: residual, J_rho, the linearised step of the parameters t = Jp dp
            RfEval<NP> o;
            rf_resid<NP>(g, ux, uy, be, P, rh, o);
            double da0, da1, db0, db1;
            rf_av(g, dp[0], dp[1], dp[2], da0, da1);
            rf_bw(g, dp[3], dp[4], dp[5], db0, db1);
            double t0 = be * __builtin_fma(rh, da0, db0), t1 = be * __builtin_fma(rh, da1, db1);
            if (NP == 7) {
                const double dk = dbe * dp[6];
                t0 = __builtin_fma(dk, o.in0, t0);
                t1 = __builtin_fma(dk, o.in1, t1);
            }
            double a00, a10;
            rf_av(g, P0.v0, P0.v1, P0.v2, a00, a10);
            const double h0 = (be0 * be0) * __builtin_fma(a00, a00, a10 * a10);
            const double gJ = __builtin_fma(o.J0, o.r0, o.J1 * o.r1), tJ = __builtin_fma(o.J0, t0, o.J1 * t1);
            const bool flagged = rf_flagged(o.h, h0);
            // back-substitution: d rho = -ete_inv E^T (r + t) s = -psi J^T (r + t) / |J|^2 while the clamp is inactive
            double drho = -((psi * (gJ + tJ)) * rcp_core(flagged ? 1.0 : o.h));
            if (__builtin_amdgcn_ballot_w64(flagged) != 0) {  // (a wave-uniform branch: left as straight-line code the compiler turns the exact form -- two divisions, a square root -- into selects every inlier pays for)
                if (flagged) {
                    asm volatile("; listed inlier: exact e-block inverse" ::: "memory");  // (keeps the branch a branch)
                    double sr, E0, E1;
                    const double ete_inv = rf_ete_inv_exact(o.J0, o.J1, h0, inv_radius, sr, E0, E1);
                    drho = -((ete_inv * (sr * (gJ + tJ))) * sr);
                }
            }
            const double m0 = __builtin_fma(o.J0, drho, t0), m1 = __builtin_fma(o.J1, drho, t1);
            acc[RR::MODEL] -= __builtin_fma(m0, __builtin_fma(0.5, m0, o.r0), m1 * __builtin_fma(0.5, m1, o.r1));
            const double cd = rh + drho;
            cand[i] = cd;
            acc[RR::STEPSQ] = __builtin_fma(drho, drho, acc[RR::STEPSQ]);
            // the candidate: cost, gradient, norms (HandleSuccessfulStep's, used when the step is accepted) and the Schur sums of the next iteration
            RfEval<NP> oc;
            rf_resid<NP>(g, ux, uy, bec, Pc, cd, oc);
            rf_jac<NP>(g, bec, dbec, cd, oc);
            acc[RR::COST2] = __builtin_fma(oc.r0, oc.r0, __builtin_fma(oc.r1, oc.r1, acc[RR::COST2]));
            acc[RR::GMAX] = fmax(acc[RR::GMAX], fabs(__builtin_fma(oc.J0, oc.r0, oc.J1 * oc.r1)));
            acc[RR::XSQ] = __builtin_fma(cd, cd, acc[RR::XSQ]);
            if (ZSUM) acc[RR::ZSUM] += 1.0 / cd;
            const bool flagged_c = rf_flagged(oc.h, h0);
            const double ihm = flagged_c ? 0.0 : rcp_core(flagged_c ? 1.0 : oc.h);
            rf_schur_accumulate<NP>(oc, ihm, acc, lds_c);
            if (flagged_c) rf_list_append(A.list_count, A.list_entries, x, y, ux, uy, l0.ab, l0.ak, cd, i);
        };
        // Two inliers' loads in flight beyond the one in the arithmetic: two waves per SIMD, so the latency is covered here and not by occupancy.
        // (Measured and dropped: three register sets taking turns instead of the two copies per iteration -- the third set spills, 10.4 instead
        // of 8.3 us per loop of the older wave; the first two sets requested ahead of the stage -- vmcnt counts in order, so the stage's own
        // row loads then wait for them: stage + 1.4 us, loop - 1.4 us.)
        const int64_t last = m - 1;
        if (NP == 6) {
            RfLoad l0 = rf_load<NP>(A, rho, i0, i0 <= last);
            RfLoad l1 = rf_load<NP>(A, rho, i0 + stride, i0 + stride <= last);
            for (int64_t i = i0; i < m; i += stride) {
                const int64_t i2 = i + 2 * stride;
                const RfLoad l2 = rf_load<NP>(A, rho, i2, i2 <= last);
#if RF_LOADS_ONLY
                acc[0] += l0.c4.x + l0.c4.y + l0.c4.z + l0.c4.w + l0.rho + l0.ab;  // (diagnostic build: the loop's memory accesses without its arithmetic)
                cand[i] = l0.rho;
#else
                body(l0, i);
#endif
                l0 = l1;
                l1 = l2;
            }
        } else {
            // (k refined: 75 accumulators.  No register set for a load in flight: whatever the compiler cannot keep goes to scratch, and a kernel
            // with a private segment must not run on several streams at once -- see rf_schur_accumulate.  RF_NP7_PREFETCH: experiments)
#if RF_NP7_PREFETCH == 1
            RfLoad l0 = rf_load<NP>(A, rho, i0, i0 <= last);
            for (int64_t i = i0; i < m; i += stride) {
                const RfLoad l1 = rf_load<NP>(A, rho, i + stride, i + stride <= last);
                body(l0, i);
                l0 = l1;
            }
#else
            for (int64_t i = i0; i < m; i += stride) body(rf_load<NP>(A, rho, i, true), i);
#endif
        }
    }
    if (kLdsC) {  // back to the registers for the reduction (the body's temporaries are dead by now)
        int tri = 0;
#pragma unroll
        for (int a = 0; a < NP; ++a) {
            acc[RR::OFF_C + a] = lds_c[a * kFB];
            tri += NP - 1 - a;  // index of (a, NP - 1) in the upper triangle
            acc[RR::OFF_JTJ + tri] = lds_c[(NP + a) * kFB];
            acc[RR::OFF_B + tri] = lds_c[(2 * NP + a) * kFB];
            tri += 1;
        }
    }
    if (A.beat && threadIdx.x == 0 && blockIdx.x < 512) A.beat[8 + blockIdx.x] = ((unsigned long long)A.slot << 8) | 3;
    if (A.beat && (threadIdx.x & 63) == 0 && blockIdx.x < 64) A.beat[8 + 256 + blockIdx.x * 4 + 0] |= 1ull << (threadIdx.x >> 6);  // (waves of the first 64 workgroups that left the loop; [.. + 1]: that passed the reduction)
    if (A.beat && blockIdx.x == 0 && threadIdx.x == 0) A.beat[0] = ((unsigned long long)A.slot << 8) | 3;
    if (stamp) tk[3] = wall_clock64();
    if (A.stamps && blockIdx.x == 0 && (threadIdx.x & 63) == 0) A.stamps[8 + (threadIdx.x >> 6)] += wall_clock64();  // (per wave: when its loop ended; sums over passes)
    static_assert(RR::GMAX == RR::NW - 1, "the max slot is the row's last");
    block_reduce_store_halving<RR::NW>(acc, s_red, A.partials + (int64_t)blockIdx.x * RR::NW);
    if (A.beat && threadIdx.x == 0) {
        if (A.beat[7]) atomicAdd(&A.beat[3], 1ull);
        if (blockIdx.x == 0) A.beat[0] = ((unsigned long long)A.slot << 8) | 4;
        if (blockIdx.x < 512) A.beat[8 + blockIdx.x] = ((unsigned long long)A.slot << 8) | 4;
    }
    if (stamp) {
        tk[4] = wall_clock64();
        A.stamps[0] += tk[1] - tk[0], A.stamps[1] += tk[2] - tk[1], A.stamps[2] += tk[3] - tk[2], A.stamps[3] += tk[4] - tk[3], A.stamps[4] += 1;
        if (tk[4 + 0]) A.stamps[5] += tk[4 + 0] - tk[1], A.stamps[6] += tk[4 + 1] - tk[4 + 0], A.stamps[7] += tk[2] - tk[4 + 1];  // stage: rows reduced | decided | solved
    }
}

// the stage on its own: behind the LAST pass of a chunk (-> the published state), behind every pass while several solves share the GPU,
// and behind every exchange of the column-tiled solve's last slot
template <int NP>
__global__ __launch_bounds__(kFB) void refine_rf_apply_kernel(const double* __restrict__ rows, int nrows, int row_stride, RfLists lists,
                                                             const RefineState* __restrict__ st_in, RefineState* __restrict__ st_out,
                                                             const RfExt* __restrict__ ext_in, RfExt* __restrict__ ext_out, int64_t m_total,
                                                             const int64_t* __restrict__ m_total_dev, double* __restrict__ trace, int trace_rows,
                                                             unsigned long long* __restrict__ beat, int slot) {
    __shared__ RfStageLds<NP> s_stage;
    __shared__ RefineState s_state;
    if (beat && threadIdx.x == 0) beat[1] = ((unsigned long long)slot << 8) | 1;
    state_to_lds(&s_state, st_in);
    RefineState* st = &s_state;
    if (st->termination < 0 && st->pending_apply) {
        const int64_t mt = m_total_dev ? *m_total_dev : (m_total >= 0 ? m_total : st->m);
        rf_apply_body<NP>(st, s_stage, rows, nrows >= 0 ? nrows : st->grid, row_stride, lists, ext_in, ext_out, mt, trace, trace_rows, nullptr, beat);
    }
    __syncthreads();
    if (beat && threadIdx.x == 0) beat[1] = ((unsigned long long)slot << 8) | 40;
    if (threadIdx.x == 0) st->pending_apply = 0;
    __syncthreads();
    state_from_lds(st_out, st);
}

// column-tiled solve: the shard's partials [workgroups][NW] reduced to ONE row [NW | list] (the all-gather payload; the list sorted by index)
template <int NP>
__global__ __launch_bounds__(kFB) void refine_rf_row_kernel(const double* __restrict__ partials, int nblocks, const RefineState* __restrict__ st,
                                                           const int* __restrict__ list_count, const double* __restrict__ list_entries,
                                                           double* __restrict__ row) {
    using RR = RfRow<NP>;
    __shared__ double s_grp[ReduceShape<RR::NW>::G][RR::NW];
    __shared__ double s[RR::NW];
    if (nblocks < 0) nblocks = st->grid;
    reduce_partials_groups<RR::NW>(partials, nblocks, RR::GMAX, s_grp, threadIdx.x, RR::NW, 0);
    __syncthreads();
    reduce_partials_slots<RR::NW>(s_grp, RR::GMAX, s, threadIdx.x);
    __syncthreads();
    if (threadIdx.x < RR::NW) row[threadIdx.x] = s[threadIdx.x];
    const int n = *list_count;
    double* lr = row + RR::NW;
    if (threadIdx.x == 0) {  // [count, 0 | pad | entries]
        reinterpret_cast<int*>(lr)[0] = n;
        reinterpret_cast<int*>(lr)[1] = 0;
        lr[1] = 0.0;
    }
    if ((int)threadIdx.x < n && n <= kRfListCap) {
        const double* e = list_entries + (size_t)threadIdx.x * kRfEntry;
        int rank = 0;
        for (int j = 0; j < n; ++j) rank += (list_entries[(size_t)j * kRfEntry + 7] < e[7]) ? 1 : 0;
#pragma unroll
        for (int q = 0; q < kRfEntry; ++q) lr[2 + (size_t)rank * kRfEntry + q] = e[q];
    }
}


// ---------------------------------------------------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------------------------------------------------
// Layout of the extra doubles behind the rows / chunk states of RefineBuffers::partials (refine_partials_doubles_cap reserves them):
//   [3 lists of kRfListCap entries][2 RfExt]; the lists' counters live behind the state (refine_rf_counters)
int refine_rf_extra_doubles() { return 3 * kRfListCap * kRfEntry + 2 * kRfExtDoubles + 8; }
int refine_rf_row_doubles(int np) { return (np == 7 ? RfRow<7>::NW : RfRow<6>::NW) + kRfListDoubles; }

namespace {
inline int rf_grid_cap(const Ctx* c) { return c->num_cus; }
inline int rf_grid(const Ctx* c, int64_t m) {
    int64_t b = (m + kFB - 1) / kFB;
    const int64_t cap = rf_grid_cap(c);
    if (b < 1) b = 1;
    if (b > cap) {
        const int64_t iters = (b + cap - 1) / cap;
        b = (b + iters - 1) / iters;
    }
    return (int)b;
}
struct RfLayout {
    double* rows[2];
    RefineState* cs[2];
    double* list[3];
    RfExt* ext[2];
};
inline RfLayout rf_layout(const Ctx* c, const RefineBuffers& B) {
    RfLayout Lo;
    double* base = B.partials;
    const size_t half = (size_t)refine_partials_half_doubles(c);
    Lo.rows[0] = base, Lo.rows[1] = base + half;
    double* p = base + 2 * half;
    const size_t sd = (size_t)refine_state_doubles();
    Lo.cs[0] = reinterpret_cast<RefineState*>(p), Lo.cs[1] = reinterpret_cast<RefineState*>(p + sd);
    p += 2 * sd + 8;
    for (int i = 0; i < 3; ++i) Lo.list[i] = p + (size_t)i * kRfListCap * kRfEntry;
    p += 3 * kRfListCap * kRfEntry;
    Lo.ext[0] = reinterpret_cast<RfExt*>(p), Lo.ext[1] = reinterpret_cast<RfExt*>(p + kRfExtDoubles);
    return Lo;
}
inline int rf_mod3(int g) { return ((g % 3) + 3) % 3; }
inline int* rf_list_count(const RefineBuffers& B, int g) { return refine_rf_counters(B) + rf_mod3(g); }
inline double* rf_list_entries(const RfLayout& Lo, int g) { return Lo.list[rf_mod3(g)]; }
inline RfLists rf_lists_of(const RfLayout& Lo, const RefineBuffers& B, int g, const double* rows_all, int nranks, int nw) {
    RfLists l;
    if (rows_all) {  // gathered rows [nranks][nw | count | pad | entries]
        l.counts = reinterpret_cast<const int*>(rows_all + nw), l.count_stride = 2 * (nw + kRfListDoubles);
        l.entries = rows_all + nw + 2, l.entry_stride = nw + kRfListDoubles, l.nlists = nranks;
    } else {
        l.counts = rf_list_count(B, g), l.count_stride = 0, l.entries = rf_list_entries(Lo, g), l.entry_stride = 0, l.nlists = 1;
    }
    return l;
}

// opt-in heartbeat (environment RSDSFM_SYNC_WATCHDOG_S): 8 host-mapped words per context -- [0] (slot << 8 | phase) of the last pass's
// workgroup 0 (1 started, 2 stage done, 3 loop done, 4 row written), [2] / [3] workgroups of passes that started / left
struct RfBeat {
    const Ctx* c;
    unsigned long long* words;
};
RfBeat g_beats[32];
int g_nbeats = 0;
unsigned long long* rf_beat(Ctx* c) {
    static const bool on = getenv("RSDSFM_SYNC_WATCHDOG_S") != nullptr && getenv("RSDSFM_NO_BEAT") == nullptr;  // (NO_BEAT: the watchdog without the kernels' heartbeat)
    if (!on) return nullptr;
    for (int i = 0; i < g_nbeats; ++i)
        if (g_beats[i].c == c) return g_beats[i].words;
    if (g_nbeats >= 32) return nullptr;
    unsigned long long* w = nullptr;
    if (hipHostMalloc((void**)&w, 64 + 8 * 512, hipHostMallocDefault) != hipSuccess) return nullptr;
    for (int i = 0; i < 8 + 512; ++i) w[i] = 0;  // ([8 + b]: (slot << 8 | phase) of workgroup b of the last pass)
    w[7] = getenv("RSDSFM_BEAT_COUNT_WORKGROUPS") ? 1 : 0;
    g_beats[g_nbeats].c = c, g_beats[g_nbeats].words = w;
    ++g_nbeats;
    return w;
}
// opt-in phase stamps (environment RSDSFM_RF_STAMPS=1, read once): a small device buffer per process, never freed
unsigned long long* rf_stamps(Ctx* c) {
    static int on = -1;
    static unsigned long long* buf = nullptr;
    if (on < 0) {
        const char* e = getenv("RSDSFM_RF_STAMPS");
        on = (e && e[0] == '1') ? 1 : 0;
        if (on && (hipMalloc(&buf, 128) != hipSuccess || hipMemset(buf, 0, 128) != hipSuccess)) buf = nullptr;
    }
    (void)c;
    return buf;
}
template <int NP, bool ZSUM>
void rf_pass_launch_t(Ctx* c, const RfPassArgs& A, bool first, int grid) {
    if (first)
        hipLaunchKernelGGL((refine_rf_pass_kernel<NP, true, ZSUM>), dim3(grid), dim3(kFB), 0, c->stream, A);
    else
        hipLaunchKernelGGL((refine_rf_pass_kernel<NP, false, ZSUM>), dim3(grid), dim3(kFB), 0, c->stream, A);
}
}  // namespace

// Slot g (GLOBAL index within the solve: 0 = the first pass) of a chunk that started at slot g_first.  rows_all_prev / nranks: the column-tiled
// solve's gathered rows of slot g - 1 (null: a single context -- the previous pass's partials and its list).  State in: the published state for
// the first slot of a chunk, else what slot g - 1 left.
int refine_rf_pass_launch(Ctx* c, const RefineBuffers& B, int np, int g, int g_first, const double* rows_all_prev, int nranks, int64_t m_total,
                          const int64_t* m_total_dev, bool publish) {
    const RfLayout Lo = rf_layout(c, B);
    const int grid = B.m_on_device ? rf_grid_cap(c) : rf_grid(c, B.m);
    const int nw = np == 7 ? RfRow<7>::NW : RfRow<6>::NW;
    RfPassArgs A = {};
    A.m = B.m_on_device ? -1 : B.m;
    A.xyuv_in = reinterpret_cast<const double4*>(B.uu);
    A.xyuv_out = reinterpret_cast<double4*>(B.uu);
    A.beta = B.beta;
    A.alpha = B.alpha;
    A.alpha_k = B.alpha_k;
    A.rho_a = B.rho_a;
    A.rho_b = B.rho_b;
    A.st_in = g == g_first ? B.state : Lo.cs[(g - 1) & 1];
    A.st_out = publish ? B.state : Lo.cs[g & 1];  // (publish: the CLOSING pass of a chunk -- see refine_enqueue_chunk)
    if (rows_all_prev)
        A.rows_prev = rows_all_prev, A.nrows_prev = nranks, A.row_stride = nw + kRfListDoubles;
    else
        A.rows_prev = Lo.rows[(g - 1) & 1], A.nrows_prev = B.m_on_device ? -1 : grid, A.row_stride = nw;
    A.lists_prev = rf_lists_of(Lo, B, g - 1, rows_all_prev, nranks, nw);
    A.list_count = rf_list_count(B, g);
    A.list_entries = rf_list_entries(Lo, g);
    A.list_count_zero = rf_list_count(B, g + 1);
    A.ext_in = Lo.ext[(g - 1) & 1];
    A.ext_out = Lo.ext[g & 1];
    A.partials = Lo.rows[g & 1];
    A.trace = c->d_refine_trace;
    A.trace_rows = c->refine_trace_rows;
    A.m_total = m_total;
    A.m_total_dev = m_total_dev;
    A.flow = reinterpret_cast<const double2*>(B.flow);
    A.n_flow = B.n_flow;
    A.inl = B.inl;
    A.inlier_idx = B.inlier_idx;
    A.flow_index_mode = B.flow_index_mode;
    A.bad_index = B.bad_index;
    A.stamps = rf_stamps(c);
    A.beat = rf_beat(c);
    A.slot = g;
    const bool first = g == 0;
    if (np == 7) {
        if (B.want_zsum) rf_pass_launch_t<7, true>(c, A, first, grid); else rf_pass_launch_t<7, false>(c, A, first, grid);
    } else {
        if (B.want_zsum) rf_pass_launch_t<6, true>(c, A, first, grid); else rf_pass_launch_t<6, false>(c, A, first, grid);
    }
    RSDSFM_HIP_CHECK(c, hipGetLastError());
    return RSDSFM_OK;
}

// the stage behind slot g on its own: in place (to_published = false: the next pass's prologue finds nothing pending) or into the published
// state (the last slot of a chunk)
int refine_rf_apply_launch(Ctx* c, const RefineBuffers& B, int np, int g, bool to_published, const double* rows_all, int nranks, int64_t m_total,
                           const int64_t* m_total_dev) {
    const RfLayout Lo = rf_layout(c, B);
    const int grid = B.m_on_device ? rf_grid_cap(c) : rf_grid(c, B.m);
    const int nw = np == 7 ? RfRow<7>::NW : RfRow<6>::NW;
    const double* rows = rows_all ? rows_all : Lo.rows[g & 1];
    const int nrows = rows_all ? nranks : (B.m_on_device ? -1 : grid);
    const int stride = rows_all ? nw + kRfListDoubles : nw;
    const RfLists lists = rf_lists_of(Lo, B, g, rows_all, nranks, nw);
    RefineState* st_io = Lo.cs[g & 1];
    // (the stage of slot g is "the prologue of slot g + 1": ext in = what slot g's prologue left, out = the other copy)
    if (np == 7)
        hipLaunchKernelGGL(refine_rf_apply_kernel<7>, dim3(1), dim3(kFB), 0, c->stream, rows, nrows, stride, lists, st_io, to_published ? B.state : st_io,
                           Lo.ext[g & 1], Lo.ext[(g + 1) & 1], m_total, m_total_dev, c->d_refine_trace, c->refine_trace_rows, rf_beat(c), g);
    else
        hipLaunchKernelGGL(refine_rf_apply_kernel<6>, dim3(1), dim3(kFB), 0, c->stream, rows, nrows, stride, lists, st_io, to_published ? B.state : st_io,
                           Lo.ext[g & 1], Lo.ext[(g + 1) & 1], m_total, m_total_dev, c->d_refine_trace, c->refine_trace_rows, rf_beat(c), g);
    RSDSFM_HIP_CHECK(c, hipGetLastError());
    return RSDSFM_OK;
}

// column-tiled solve: the shard's row of slot g (behind its pass)
int refine_rf_row_launch(Ctx* c, const RefineBuffers& B, int np, int g, double* row) {
    const RfLayout Lo = rf_layout(c, B);
    const int grid = B.m_on_device ? rf_grid_cap(c) : rf_grid(c, B.m);
    if (np == 7)
        hipLaunchKernelGGL(refine_rf_row_kernel<7>, dim3(1), dim3(kFB), 0, c->stream, Lo.rows[g & 1], B.m_on_device ? -1 : grid, static_cast<const RefineState*>(Lo.cs[g & 1]),
                           rf_list_count(B, g), rf_list_entries(Lo, g), row);
    else
        hipLaunchKernelGGL(refine_rf_row_kernel<6>, dim3(1), dim3(kFB), 0, c->stream, Lo.rows[g & 1], B.m_on_device ? -1 : grid, static_cast<const RefineState*>(Lo.cs[g & 1]),
                           rf_list_count(B, g), rf_list_entries(Lo, g), row);
    RSDSFM_HIP_CHECK(c, hipGetLastError());
    return RSDSFM_OK;
}

// diagnostics (sync_stream's watchdog, capi.hip): the published state, the two chunk states and the list counters of a refinement, read through
// a stream of their own while the context's stream is stuck
void refine_rf_debug_dump(Ctx* c, const RefineBuffers& B) {
    const RfLayout Lo = rf_layout(c, B);
    hipStream_t s2 = nullptr;
    if (hipStreamCreateWithFlags(&s2, hipStreamNonBlocking) != hipSuccess) return;
    RefineState h[3];
    int cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const RefineState* src[3] = {B.state, Lo.cs[0], Lo.cs[1]};
    for (int i = 0; i < 3; ++i) (void)hipMemcpyAsync(&h[i], src[i], sizeof(RefineState), hipMemcpyDeviceToHost, s2);
    (void)hipMemcpyAsync(cnt, B.bad_index, sizeof(int) * 8, hipMemcpyDeviceToHost, s2);
    if (hipStreamSynchronize(s2) != hipSuccess) {
        fprintf(stderr, "[rsdsfm] refinement dump: the copies did not complete either\n");
        return;
    }
    const char* names[3] = {"published", "chunk state 0", "chunk state 1"};
    for (int i = 0; i < 3; ++i)
        fprintf(stderr, "[rsdsfm] %s: np %d termination %d iteration %d slots %d pending_apply %d rf %d guard %d cur %d solve_ok %d m %lld grid %d radius %g cost %g successful %d unsuccessful %d invalid_run %d\n",
                names[i], h[i].np, h[i].termination, h[i].iteration, h[i].slots, h[i].pending_apply, h[i].rf, h[i].rf_guard, h[i].cur, h[i].solve_ok, (long long)h[i].m, h[i].grid, h[i].radius,
                h[i].cost, h[i].num_successful, h[i].num_unsuccessful, h[i].invalid_run);
    fprintf(stderr, "[rsdsfm] bad_index %d, list counters %d %d %d (m_on_device %d, m %lld)\n", cnt[0], cnt[4], cnt[5], cnt[6], (int)B.m_on_device, (long long)B.m);
    for (int i = 0; i < g_nbeats; ++i)
        fprintf(stderr, "[rsdsfm] context %p%s: last pass slot %llu phase %llu; pass workgroups started %llu, left %llu; last stage kernel slot %llu phase %llu\n", (const void*)g_beats[i].c, g_beats[i].c == c ? " (the waiting one)" : "",
                g_beats[i].words[0] >> 8, g_beats[i].words[0] & 255, g_beats[i].words[2], g_beats[i].words[3], g_beats[i].words[1] >> 8, g_beats[i].words[1] & 255);
    for (int i = 0; i < g_nbeats; ++i) {
        if (g_beats[i].c != c) continue;
        int hist[16] = {0};
        const unsigned long long last = g_beats[i].words[0] >> 8;
        fprintf(stderr, "[rsdsfm] workgroups of the waiting context's pass (slot %llu) not at phase 4 / 9:", last);
        for (int b = 0; b < 256; ++b) {
            const unsigned long long w = g_beats[i].words[8 + b];
            hist[w & 15] += 1;
            if ((w & 255) != 4 && (w & 255) != 9) fprintf(stderr, " [%d: slot %llu phase %llu]", b, w >> 8, w & 255);
        }
        fprintf(stderr, "\n[rsdsfm] phase histogram:");
        for (int p = 0; p < 10; ++p) fprintf(stderr, " %d:%d", p, hist[p]);
        fprintf(stderr, "\n");
    }
}

// the accumulated phase stamps {state load, stage, publish + loop, row reduction, passes} in 100 MHz ticks; zeroes them (profiling tools only)
int refine_rf_read_stamps(Ctx* c, unsigned long long out[16]) {
    unsigned long long* b = rf_stamps(c);
    if (!b) return RSDSFM_ERR_INVALID;
    RSDSFM_HIP_CHECK(c, hipStreamSynchronize(c->stream));
    RSDSFM_HIP_CHECK(c, hipMemcpy(out, b, 128, hipMemcpyDeviceToHost));
    RSDSFM_HIP_CHECK(c, hipMemset(b, 0, 128));
    return RSDSFM_OK;
}

}  // namespace rsdsfm
