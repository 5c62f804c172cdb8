// minimal9_kernels.hip -- batched 9-point minimal solver on MI355X (gfx950).
//
// Replaces minimal::calculateVelocities (reference minimal.cc:36-177): one wavefront LANE per hypothesis when there are many, one WAVE
// per hypothesis when there are few (the RANSAC case).  Each lane owns a private 9x9 work matrix W and a 9x9 V (two-sided Jacobi SVD,
// Eigen 3.3.4 JacobiSVD restated) in LDS with a lane-interleaved layout  element e of lane l  at  lds[e * 64 + l]  -- every dynamic index
// is the same across lanes of an instruction, so ds_read_b64/ds_write_b64 are conflict-free and nothing spills to scratch; the optional
// k estimation (3x3 and 6x6 LU inverses + Hessenberg / Francis-QR eigenvalues) runs in registers with compile-time indices.  One wave per
// workgroup (the per-lane state is 1.4 KB, i.e. 92 KB of the CU's 160 KB LDS).  T is at most a few thousand: this kernel is
// latency-bound and tiny; correctness and agreement with the CPU oracle (same algorithms, same operation order) matter, not FLOPs.
#include <float.h>

#include "device_math.hpp"
#include "rsdsfm_internal.hpp"

namespace rsdsfm {

namespace {

constexpr double kPi = 3.14159265358979323846;

// lane-interleaved LDS vector: v[i] is element i of this lane
struct LVec {
    double* p;
    __device__ __forceinline__ double& operator[](int i) const { return p[i * 64]; }
    __device__ __forceinline__ LVec at(int off) const { return LVec{p + off * 64}; }
};

struct Rot {
    double c, s;
};

// Jacobi.h JacobiRotation::makeJacobi(x, y, z)
__device__ __forceinline__ Rot make_jacobi(double x, double y, double z) {
    Rot j;
    double deno = 2.0 * fabs(y);
    if (deno < DBL_MIN) {
        j.c = 1.0;
        j.s = 0.0;
    } else {
        double tau = (x - z) / deno;
        double w = sqrt(tau * tau + 1.0);
        double t = (tau > 0.0) ? 1.0 / (tau + w) : 1.0 / (tau - w);
        double sign_t = t > 0.0 ? 1.0 : -1.0;
        double n = 1.0 / sqrt(t * t + 1.0);
        j.s = -sign_t * copysign(1.0, y) * fabs(t) * n;  // y / |y| (y != 0 here) without the division
        j.c = n;
    }
    return j;
}

// x' = c x + s y ; y' = -s x + c y over 9 strided elements of M.  All 18 LDS reads are issued before the first use
// (a single wave has nobody to hide the ~64-cycle ds_read latency behind, so back-to-back dependent reads dominate).
__device__ __forceinline__ void rot_plane(LVec M, int x0, int incx, int y0, int incy, int n, Rot j) {
    (void)n;
    if (j.c == 1.0 && j.s == 0.0) return;
    double xv[9], yv[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        xv[i] = M[x0 + i * incx];
        yv[i] = M[y0 + i * incy];
    }
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        M[x0 + i * incx] = j.c * xv[i] + j.s * yv[i];
        M[y0 + i * incy] = -j.s * xv[i] + j.c * yv[i];
    }
}

// JacobiSVD.h real_2x2_jacobi_svd on W(p,q)
__device__ __forceinline__ void real_2x2_jacobi_svd(LVec W, int p, int q, Rot& jl, Rot& jr) {
    double m0 = W[p * 9 + p], m1 = W[p * 9 + q], m2 = W[q * 9 + p], m3 = W[q * 9 + q];
    Rot rot1;
    double t = m0 + m3;
    double d = m2 - m1;
    if (fabs(d) < DBL_MIN) {
        rot1.s = 0.0;
        rot1.c = 1.0;
    } else {
        double u = t / d;
        double tmp = sqrt(1.0 + u * u);
        rot1.s = 1.0 / tmp;
        rot1.c = u / tmp;
    }
    if (!(rot1.c == 1.0 && rot1.s == 0.0)) {  // m.applyOnTheLeft(0,1,rot1)
        double a0 = rot1.c * m0 + rot1.s * m2, a2 = -rot1.s * m0 + rot1.c * m2;
        double a1 = rot1.c * m1 + rot1.s * m3, a3 = -rot1.s * m1 + rot1.c * m3;
        m0 = a0, m1 = a1, m2 = a2, m3 = a3;
    }
    jr = make_jacobi(m0, m1, m3);
    Rot jt = {jr.c, -jr.s};
    jl.c = rot1.c * jt.c - rot1.s * jt.s;
    jl.s = rot1.c * jt.s + rot1.s * jt.c;
}

// Two-sided Jacobi SVD of the 9x9 in W (destroyed); V accumulates the right rotations; sv/col: 9 slots each.
// On return e[i] = V(i, column of the smallest singular value) is written to `e_out`.
__device__ void jacobi_svd9_nullvec(LVec W, LVec V, LVec sv, LVec col, double e_out[9]) {
    const double precision = 2.0 * DBL_EPSILON;
    double scale = 0.0;
    for (int i = 0; i < 81; ++i) scale = fmax(scale, fabs(W[i]));
    if (scale == 0.0) scale = 1.0;
    for (int i = 0; i < 81; ++i) W[i] = W[i] / scale;
    for (int i = 0; i < 9; ++i)
        for (int j = 0; j < 9; ++j) V[i * 9 + j] = (i == j) ? 1.0 : 0.0;
    double max_diag = 0.0;
    for (int i = 0; i < 9; ++i) max_diag = fmax(max_diag, fabs(W[i * 9 + i]));
    bool finished = false;
    int sweeps = 0;
    while (!finished && sweeps < 1000) {
        finished = true;
        ++sweeps;
        for (int p = 1; p < 9; ++p) {
            for (int q = 0; q < p; ++q) {
                double threshold = precision * max_diag;
                if (threshold < DBL_MIN) threshold = DBL_MIN;
                if (fabs(W[p * 9 + q]) > threshold || fabs(W[q * 9 + p]) > threshold) {
                    finished = false;
                    Rot jl, jr;
                    real_2x2_jacobi_svd(W, p, q, jl, jr);
                    rot_plane(W, p * 9, 1, q * 9, 1, 9, jl);  // applyOnTheLeft: rows p, q
                    Rot jrt = {jr.c, -jr.s};
                    rot_plane(W, p, 9, q, 9, 9, jrt);  // applyOnTheRight: cols p, q
                    rot_plane(V, p, 9, q, 9, 9, jrt);
                    max_diag = fmax(max_diag, fmax(fabs(W[p * 9 + p]), fabs(W[q * 9 + q])));
                }
            }
        }
    }
    for (int i = 0; i < 9; ++i) {
        sv[i] = fabs(W[i * 9 + i]) * scale;
        col[i] = (double)i;
    }
    // selection sort, descending (JacobiSVD step 4); track the column permutation instead of moving V
    for (int i = 0; i < 9; ++i) {
        int pos = i;
        double best = sv[i];
        for (int j = i + 1; j < 9; ++j) {
            double sj = sv[j];
            if (sj > best) {
                best = sj;
                pos = j;
            }
        }
        if (best == 0.0) break;
        if (pos != i) {
            double t = sv[i];
            sv[i] = sv[pos];
            sv[pos] = t;
            double tc = col[i];
            col[i] = col[pos];
            col[pos] = tc;
        }
    }
    const int c8 = (int)col[8];
#pragma unroll
    for (int i = 0; i < 9; ++i) e_out[i] = V[i * 9 + c8];
}

// makeJacobi through the in-range cores of division, reciprocal and square root (device_math.hpp): the same bits for operands in
// range; wd / ws track the range tests of the operands that are not in range by construction
__device__ __forceinline__ Rot make_jacobi_core(double x, double y, double z, uint32_t& wd, uint32_t& ws) {
    Rot j;
    double deno = 2.0 * fabs(y);
    if (deno < DBL_MIN) {
        j.c = 1.0;
        j.s = 0.0;
    } else {
        const double xz = x - z;
        div_range_track(wd, xz);
        div_range_track(wd, deno);
        double tau = div_core(xz, deno);
        const double ww = tau * tau + 1.0;
        sqrt_range_track(ws, ww);  // (>= 1: only an overflow of tau * tau takes it out of range)
        double w = sqrt_core(ww);
        const double den = (tau > 0.0) ? tau + w : tau - w;  // |den| >= 1, < 2^513 while ww is in range
        double t = rcp_core(den);
        double sign_t = t > 0.0 ? 1.0 : -1.0;
        double n = rcp_core(sqrt_core(t * t + 1.0));  // |t| <= 1: the argument lies in [1, 2], its root in [1, 1.42]
        j.s = -sign_t * copysign(1.0, y) * fabs(t) * n;
        j.c = n;
    }
    return j;
}

// ---------------------------------------------------------------------------------------------------
// wave-cooperative variant of jacobi_svd9_nullvec: ONE hypothesis per wave (used when there are few hypotheses, the
// RANSAC case: T = 5 ... a few hundred, where the one-lane-per-hypothesis kernel leaves the machine idle and serialises
// 108 LDS accesses + 108 multiply-adds per rotation on a single lane).  The matrix lives once in LDS (the lane-0
// copy, element stride 64 doubles); all lanes evaluate the (uniform) 2x2 rotation, lanes 0..8 apply it to the nine
// columns (left) / rows (right) of W and lanes 16..24 to the rows of V.  A wave's LDS operations execute in order, so
// no barriers are needed.  Every element goes through exactly the operations of the scalar version: results are
// bit-identical.
// ---------------------------------------------------------------------------------------------------
// COMPACT copies (element e at M[e]): with the per-lane layout's stride of 64 doubles every element of one lane's copy sits in the same
// LDS bank pair, and the 9 (W) + 9 (V) lanes of a rotation serialised on it (SQ_LDS_BANK_CONFLICT: 0.97 M cycles per launch)
__device__ __forceinline__ double& sh(double* M, int e) { return M[e]; }
constexpr int kCoopV = 98;      // V behind W, two doubles off a multiple of 32: the row accesses of the two halves (stride 9) then use disjoint banks
constexpr int kCoopSlots = 3;   // LDS slots (of 64 doubles) behind the per-lane block that hold the compact W and V

// CORE: the rotations through the in-range function cores; *outside is set when an operand was out of range (the caller has the
// hypotheses computed again with CORE = false)
template <bool CORE>
__device__ __forceinline__ void jacobi_svd9_nullvec_coop(double* W, double* V, int lane, LVec sv, LVec col, double (&e_out)[9], bool& outside, int& sweeps_out, int& rotations_out) {
    uint32_t wd = 0, ws = 0;
    int rotations = 0;
    const double precision = 2.0 * DBL_EPSILON;
    // scale = max |W| (exact in any order)
    double m = fabs(sh(W, lane));
    if (lane < 17) m = fmax(m, fabs(sh(W, 64 + lane)));
    double scale = wave_max(m);
    if (scale == 0.0) scale = 1.0;
    sh(W, lane) = sh(W, lane) / scale;
    if (lane < 17) sh(W, 64 + lane) = sh(W, 64 + lane) / scale;
    {
        const int e0 = lane, e1 = 64 + lane;
        sh(V, e0) = (e0 / 9 == e0 % 9) ? 1.0 : 0.0;
        if (lane < 17) sh(V, e1) = (e1 / 9 == e1 % 9) ? 1.0 : 0.0;
    }
    double max_diag = wave_max(lane < 9 ? fabs(sh(W, lane * 10)) : 0.0);
    // lanes 0..8 work on W, lanes 16..24 on V during the right rotation
    const int li = lane & 15;
    const bool rowlane = li < 9 && lane < 32;
    double* RM = lane < 16 ? W : V;
    bool finished = false;
    int sweeps = 0;
    while (!finished && sweeps < 1000) {
        finished = true;
        ++sweeps;
        for (int p = 1; p < 9; ++p) {
            for (int q = 0; q < p; ++q) {
                double threshold = precision * max_diag;
                if (threshold < DBL_MIN) threshold = DBL_MIN;
                // the 2x2 block, broadcast to every lane
                double m0 = sh(W, p * 9 + p), m1 = sh(W, p * 9 + q), m2 = sh(W, q * 9 + p), m3 = sh(W, q * 9 + q);
                if (fabs(m1) > threshold || fabs(m2) > threshold) {
                    finished = false;
                    ++rotations;
                    // real_2x2_jacobi_svd (JacobiSVD.h), identical to the scalar version
                    Rot rot1, jl, jr;
                    const double t = m0 + m3;
                    const double d = m2 - m1;
                    if (fabs(d) < DBL_MIN) {
                        rot1.s = 0.0;
                        rot1.c = 1.0;
                    } else {
                        if (CORE) {
                            div_range_track(wd, t);
                            div_range_track(wd, d);
                            const double uu = div_core(t, d);
                            const double a1p = 1.0 + uu * uu;
                            sqrt_range_track(ws, a1p);
                            const double tmp = sqrt_core(a1p);  // in [1, 2^512) while a1p is in range
                            rot1.s = rcp_core(tmp);
                            div_range_track(wd, uu);
                            div_range_track(wd, tmp);
                            rot1.c = div_core(uu, tmp);
                        } else {
                            const double uu = t / d;
                            const double tmp = sqrt(1.0 + uu * uu);
                            rot1.s = 1.0 / tmp;
                            rot1.c = uu / tmp;
                        }
                    }
                    if (!(rot1.c == 1.0 && rot1.s == 0.0)) {
                        const double a0 = rot1.c * m0 + rot1.s * m2, a2 = -rot1.s * m0 + rot1.c * m2;
                        const double a1 = rot1.c * m1 + rot1.s * m3, a3 = -rot1.s * m1 + rot1.c * m3;
                        m0 = a0, m1 = a1, m2 = a2, m3 = a3;
                    }
                    jr = CORE ? make_jacobi_core(m0, m1, m3, wd, ws) : make_jacobi(m0, m1, m3);
                    const Rot jt = {jr.c, -jr.s};
                    jl.c = rot1.c * jt.c - rot1.s * jt.s;
                    jl.s = rot1.c * jt.s + rot1.s * jt.c;
                    // applyOnTheLeft(p, q, jl): rows p, q of W, one column per lane
                    if (lane < 9 && !(jl.c == 1.0 && jl.s == 0.0)) {
                        const double x = sh(W, p * 9 + lane), y = sh(W, q * 9 + lane);
                        sh(W, p * 9 + lane) = jl.c * x + jl.s * y;
                        sh(W, q * 9 + lane) = -jl.s * x + jl.c * y;
                    }
                    // applyOnTheRight(p, q, jr) on W and V: columns p, q, one row per lane
                    const Rot jrt = {jr.c, -jr.s};
                    if (rowlane && !(jrt.c == 1.0 && jrt.s == 0.0)) {
                        const double x = sh(RM, li * 9 + p), y = sh(RM, li * 9 + q);
                        sh(RM, li * 9 + p) = jrt.c * x + jrt.s * y;
                        sh(RM, li * 9 + q) = -jrt.s * x + jrt.c * y;
                    }
                    max_diag = fmax(max_diag, fmax(fabs(sh(W, p * 9 + p)), fabs(sh(W, q * 9 + q))));
                }
            }
        }
    }
    if (CORE) outside = wd >= kDivRangeKeys || ws >= kSqrtRangeKeys;
    sweeps_out = sweeps, rotations_out = rotations;
    for (int i = 0; i < 9; ++i) {
        sv[i] = fabs(sh(W, i * 9 + i)) * scale;
        col[i] = (double)i;
    }
    for (int i = 0; i < 9; ++i) {  // selection sort, descending (JacobiSVD step 4), as in the scalar version
        int pos = i;
        double best = sv[i];
        for (int j = i + 1; j < 9; ++j) {
            const double sj = sv[j];
            if (sj > best) {
                best = sj;
                pos = j;
            }
        }
        if (best == 0.0) break;
        if (pos != i) {
            const double tt = sv[i];
            sv[i] = sv[pos];
            sv[pos] = tt;
            const double tc = col[i];
            col[i] = col[pos];
            col[pos] = tc;
        }
    }
    const int c8 = (int)col[8];
#pragma unroll
    for (int i = 0; i < 9; ++i) e_out[i] = sh(V, i * 9 + c8);
}

// ---- the k estimation's linear algebra (minimal.cc:58-80), in REGISTERS ----------------------------------------------------------------
// The k estimation runs on a wave of its own (one hypothesis per wave when T is small): what it costs is its instruction count at ~5 clocks
// each, and with its matrices in LDS (rounds 1-5) most instructions were address arithmetic and waits for ds_read (tools/k_sections.py:
// 262 000 clocks per hypothesis, 86 % of them the eigenvalues; in registers 93 000).  Every index below is a compile-time constant.
//
// PartialPivLU inverse (MatrixXd::inverse()) of the row-major NN x NN matrix in lu (destroyed): the pivot row is applied by uniform
// compare-and-swap instances; every element goes through the operations of Eigen's loop nest in the same order.
template <int NN>
__device__ __forceinline__ int inverse_lu_reg(double (&lu)[NN * NN], double (&Ainv)[NN * NN]) {
    int piv[NN];
#pragma unroll
    for (int i = 0; i < NN; ++i) piv[i] = i;
    bool singular = false;
#pragma unroll
    for (int k = 0; k < NN; ++k) {
        int pr = k;
        double best = fabs(lu[k * NN + k]);
#pragma unroll
        for (int i = k + 1; i < NN; ++i) {
            const double c = fabs(lu[i * NN + k]);
            if (c > best) {
                best = c;
                pr = i;
            }
        }
        if (best == 0.0) singular = true;  // (inverse_lu returns here; what follows is then never looked at)
#pragma unroll
        for (int P = k + 1; P < NN; ++P) {
            if (pr == P) {
#pragma unroll
                for (int j = 0; j < NN; ++j) {
                    const double t = lu[k * NN + j];
                    lu[k * NN + j] = lu[P * NN + j];
                    lu[P * NN + j] = t;
                }
                const int ti = piv[k];
                piv[k] = piv[P];
                piv[P] = ti;
            }
        }
#pragma unroll
        for (int i = k + 1; i < NN; ++i) {
            const double f = lu[i * NN + k] / lu[k * NN + k];
            lu[i * NN + k] = f;
#pragma unroll
            for (int j = k + 1; j < NN; ++j) lu[i * NN + j] -= f * lu[k * NN + j];
        }
    }
    if (singular) return -1;
#pragma unroll
    for (int c = 0; c < NN; ++c) {
        double y[NN];
#pragma unroll
        for (int i = 0; i < NN; ++i) {
            double s = (piv[i] == c) ? 1.0 : 0.0;
#pragma unroll
            for (int j = 0; j < i; ++j) s -= lu[i * NN + j] * y[j];
            y[i] = s;
        }
#pragma unroll
        for (int i = NN - 1; i >= 0; --i) {
            double s = y[i];
#pragma unroll
            for (int j = i + 1; j < NN; ++j) s -= lu[i * NN + j] * Ainv[j * NN + c];
            Ainv[i * NN + c] = s / lu[i * NN + i];
        }
    }
    return 0;
}

// ---------------------------------------------------------------------------------------------------
// Eigenvalues of the general real 6x6 (EigenSolver: Householder Hessenberg + Francis double-shift QR, hqr), matrix in registers: the active
// block's end n is a template parameter (a wave-uniform switch picks the instance), the loops over l, m, k are unrolled with uniform guards.
// The operations on an element and their order are the textbook loop nest's (the oracle's rso_eigvals_general: same bits), with one restriction that
// cannot change an eigenvalue: the row part of a Francis step stops at column n -- the entries in rows <= n of columns > n are only ever
// read by later row parts, which write the same entries (n never grows) -- while the column part keeps all its rows (the active block's
// start l may move up again).
// ---------------------------------------------------------------------------------------------------
#define H_(i, j) H[(i) * 6 + (j)]

__device__ __forceinline__ void hessenberg6_reg(double (&H)[36]) {
#pragma unroll
    for (int m = 1; m < 5; ++m) {
        double scale = 0.0;
#pragma unroll
        for (int i = m; i < 6; ++i) scale += fabs(H_(i, m - 1));
        if (scale != 0.0) {
            double ort[6] = {0, 0, 0, 0, 0, 0};
            double hh = 0.0;
#pragma unroll
            for (int i = 5; i >= m; --i) {
                const double o = H_(i, m - 1) / scale;
                ort[i] = o;
                hh += o * o;
            }
            double g = sqrt(hh);
            if (ort[m] > 0) g = -g;
            hh -= ort[m] * g;
            ort[m] = ort[m] - g;
#pragma unroll
            for (int j = m; j < 6; ++j) {
                double f = 0.0;
#pragma unroll
                for (int i = 5; i >= m; --i) f += ort[i] * H_(i, j);
                f /= hh;
#pragma unroll
                for (int i = m; i < 6; ++i) H_(i, j) -= f * ort[i];
            }
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                double f = 0.0;
#pragma unroll
                for (int j = 5; j >= m; --j) f += ort[j] * H_(i, j);
                f /= hh;
#pragma unroll
                for (int j = m; j < 6; ++j) H_(i, j) -= f * ort[j];
            }
            H_(m, m - 1) = scale * g;
#pragma unroll
            for (int i = m + 1; i < 6; ++i) H_(i, m - 1) = 0.0;
        }
    }
}

// one pass of hqr's `while (n >= low)` loop with n = N; returns the new n
template <int N>
__device__ __forceinline__ int qr_pass_reg(double (&H)[36], LVec ev, double norm, double& exshift, int& iter) {
    const double eps = DBL_EPSILON;
    // l: the first small subdiagonal entry from the bottom (0 when there is none)
    int l = 0;
    {
        bool searching = true;
#pragma unroll
        for (int L = N; L >= 1; --L) {
            if (searching) {
                double s = fabs(H_(L - 1, L - 1)) + fabs(H_(L, L));
                if (s == 0.0) s = norm;
                if (fabs(H_(L, L - 1)) < eps * s) {
                    l = L;
                    searching = false;
                }
            }
        }
    }
    if (l == N) {  // one root
        const double hnn = H_(N, N) + exshift;
        H_(N, N) = hnn;
        ev[N] = hnn;
        ev[6 + N] = 0.0;
        iter = 0;
        return N - 1;
    }
    if constexpr (N >= 1) {
        if (l == N - 1) {  // two roots
            const double w = H_(N, N - 1) * H_(N - 1, N);
            const double p = (H_(N - 1, N - 1) - H_(N, N)) / 2.0;
            const double q = p * p + w;
            double z = sqrt(fabs(q));
            H_(N, N) = H_(N, N) + exshift;
            H_(N - 1, N - 1) = H_(N - 1, N - 1) + exshift;
            const double x = H_(N, N);
            double re_lo, re_hi, im_lo, im_hi;
            if (q >= 0) {
                z = (p >= 0) ? p + z : p - z;
                re_lo = x + z;
                re_hi = re_lo;
                if (z != 0.0) re_hi = x - w / z;
                im_lo = 0.0;
                im_hi = 0.0;
            } else {
                re_lo = x + p;
                re_hi = x + p;
                im_lo = z;
                im_hi = -z;
            }
            ev[N - 1] = re_lo;
            ev[N] = re_hi;
            ev[6 + N - 1] = im_lo;
            ev[6 + N] = im_hi;
            iter = 0;
            return N - 2;
        }
    }
    if constexpr (N >= 2) {
        double p = 0, q = 0, r = 0, s = 0, z = 0;
        double x = H_(N, N);
        double y = H_(N - 1, N - 1);
        double w = H_(N, N - 1) * H_(N - 1, N);
        if (iter == 10) {
            exshift += x;
#pragma unroll
            for (int i = 0; i <= N; ++i) H_(i, i) -= x;
            s = fabs(H_(N, N - 1)) + fabs(H_(N - 1, N - 2));
            x = y = 0.75 * s;
            w = -0.4375 * s * s;
        }
        if (iter == 30) {
            s = (y - x) / 2.0;
            s = s * s + w;
            if (s > 0) {
                s = sqrt(s);
                if (y < x) s = -s;
                s = x - w / ((y - x) / 2.0 + s);
#pragma unroll
                for (int i = 0; i <= N; ++i) H_(i, i) -= s;
                exshift += s;
                x = y = w = 0.964;
            }
        }
        iter++;
        // m: where the double-shift sweep starts
        int m = l;
        {
            bool searching = true;
#pragma unroll
            for (int M = N - 2; M >= 0; --M) {
                if (searching && M >= l) {
                    z = H_(M, M);
                    r = x - z;
                    s = y - z;
                    p = (r * s - w) / H_(M + 1, M) + H_(M, M + 1);
                    q = H_(M + 1, M + 1) - z - r - s;
                    r = H_(M + 2, M + 1);
                    s = fabs(p) + fabs(q) + fabs(r);
                    p /= s;
                    q /= s;
                    r /= s;
                    if (M == l) {
                        m = M;
                        searching = false;
                    }
                    if (M > 0) {  // (M == 0 is l: the search has ended above; Mm keeps the dead instance's indices inside the array)
                        const int Mm = M > 0 ? M - 1 : 0;
                        if (searching && fabs(H_(M, Mm)) * (fabs(q) + fabs(r)) < eps * (fabs(p) * (fabs(H_(Mm, Mm)) + fabs(z) + fabs(H_(M + 1, M + 1))))) {
                            m = M;
                            searching = false;
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int i = 2; i <= N; ++i) {
            if (i >= m + 2) {
                H_(i, i - 2) = 0.0;
                if (i >= 3 && i > m + 2) H_(i, i >= 3 ? i - 3 : 0) = 0.0;
            }
        }
        bool live = true;
#pragma unroll
        for (int K = 0; K <= N - 1; ++K) {
            if (live && K >= m) {
                const bool notlast = (K != N - 1);  // (a constant of the unrolled instance; K2 / Km keep a dead instance's indices inside the array)
                const int K2 = notlast ? K + 2 : K, Km = K > 0 ? K - 1 : 0;
                if (K > 0) {
                    if (K != m) {
                        p = H_(K, Km);
                        q = H_(K + 1, Km);
                        r = notlast ? H_(K2, Km) : 0.0;
                        x = fabs(p) + fabs(q) + fabs(r);
                        if (x != 0.0) {
                            p /= x;
                            q /= x;
                            r /= x;
                        }
                    }
                }
                if (x == 0.0) {
                    live = false;
                } else {
                    s = sqrt(p * p + q * q + r * r);
                    if (p < 0) s = -s;
                    if (s != 0) {
                        if (K > 0) {
                            if (K != m)
                                H_(K, Km) = -s * x;
                            else if (l != m)
                                H_(K, Km) = -H_(K, Km);
                        }
                        p += s;
                        x = p / s;
                        y = q / s;
                        z = r / s;
                        q /= p;
                        r /= p;
#pragma unroll
                        for (int j = K; j <= N; ++j) {
                            p = H_(K, j) + q * H_(K + 1, j);
                            if (notlast) {
                                p += r * H_(K2, j);
                                H_(K2, j) -= p * z;
                            }
                            H_(K, j) -= p * x;
                            H_(K + 1, j) -= p * y;
                        }
                        const int imax = (N < K + 3) ? N : K + 3;
#pragma unroll
                        for (int i = 0; i <= imax; ++i) {
                            p = x * H_(i, K) + y * H_(i, K + 1);
                            if (notlast) {
                                p += z * H_(i, K2);
                                H_(i, K2) -= p * r;
                            }
                            H_(i, K) -= p;
                            H_(i, K + 1) -= p * q;
                        }
                    }
                }
            }
        }
    }
    return N;
}

// eigenvalues of the general real 6x6 in H (destroyed): re = ev[0..6), im = ev[6..12), by index (minimal.cc:75-80 scans them in index order).
// (ev lives in LDS: the instances' stores to it get merged into one store with a run-time index, which would put a register array into scratch.)
__device__ __forceinline__ int eig6_values_reg(double (&H)[36], LVec ev) {
    hessenberg6_reg(H);
    double norm = 0.0;
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = (i - 1 > 0 ? i - 1 : 0); j < 6; ++j) norm += fabs(H_(i, j));
    double exshift = 0.0;
    int n = 5, iter = 0, total = 0;
    while (n >= 0) {
        if (++total > 10000) return -2;
        switch (n) {
            case 5: n = qr_pass_reg<5>(H, ev, norm, exshift, iter); break;
            case 4: n = qr_pass_reg<4>(H, ev, norm, exshift, iter); break;
            case 3: n = qr_pass_reg<3>(H, ev, norm, exshift, iter); break;
            case 2: n = qr_pass_reg<2>(H, ev, norm, exshift, iter); break;
            case 1: n = qr_pass_reg<1>(H, ev, norm, exshift, iter); break;
            default: n = qr_pass_reg<0>(H, ev, norm, exshift, iter); break;
        }
    }
    return 0;
}
#undef H_

// ---- 3x3 helpers in registers (fully unrolled, statically indexed) ----
__device__ __forceinline__ void mm3(const double (&A)[9], const double (&B)[9], double (&C)[9]) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            double s = 0.0;
#pragma unroll
            for (int t = 0; t < 3; ++t) s += A[i * 3 + t] * B[t * 3 + j];
            C[i * 3 + j] = s;
        }
}
__device__ __forceinline__ void tr3(const double (&A)[9], double (&At)[9]) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) At[j * 3 + i] = A[i * 3 + j];
}
// Eigen AngleAxisd(angle, axis).toRotationMatrix()
__device__ __forceinline__ void angle_axis_R(double angle, double ax0, double ax1, double ax2, double (&R)[9]) {
    double sn = sin(angle), c = cos(angle);
    double sa0 = sn * ax0, sa1 = sn * ax1, sa2 = sn * ax2;
    double c0 = (1.0 - c) * ax0, c1 = (1.0 - c) * ax1, c2 = (1.0 - c) * ax2;
    double tmp;
    tmp = c0 * ax1;
    R[1] = tmp - sa2;
    R[3] = tmp + sa2;
    tmp = c0 * ax2;
    R[2] = tmp + sa1;
    R[6] = tmp - sa1;
    tmp = c1 * ax2;
    R[5] = tmp - sa0;
    R[7] = tmp + sa0;
    R[0] = c0 * ax0 + c;
    R[4] = c1 * ax1 + c;
    R[8] = c2 * ax2 + c;
}

// cyclic Jacobi eigen-solver of a symmetric 3x3 (ascending eigenvalues); mirrors rso_eig_sym3
__device__ void eig_sym3(const double (&S)[9], double (&lam)[3], double (&V)[9]) {
    double a[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        a[i] = S[i];
        V[i] = (i % 4 == 0) ? 1.0 : 0.0;
    }
    for (int sweep = 0; sweep < 64; ++sweep) {
        bool rotated = false;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
#pragma unroll
            for (int q = p + 1; q < 3; ++q) {
                double apq = a[p * 3 + q];
                if (apq != 0.0) {
                    if (fabs(apq) <= 1e-20 * (fabs(a[p * 3 + p]) + fabs(a[q * 3 + q]))) {
                        a[p * 3 + q] = a[q * 3 + p] = 0.0;
                    } else {
                        rotated = true;
                        double theta = (a[q * 3 + q] - a[p * 3 + p]) / (2.0 * apq);
                        double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                        double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
#pragma unroll
                        for (int k = 0; k < 3; ++k) {
                            double akp = a[k * 3 + p], akq = a[k * 3 + q];
                            a[k * 3 + p] = c * akp - s * akq;
                            a[k * 3 + q] = s * akp + c * akq;
                        }
#pragma unroll
                        for (int k = 0; k < 3; ++k) {
                            double apk = a[p * 3 + k], aqk = a[q * 3 + k];
                            a[p * 3 + k] = c * apk - s * aqk;
                            a[q * 3 + k] = s * apk + c * aqk;
                        }
                        a[p * 3 + q] = a[q * 3 + p] = 0.0;
#pragma unroll
                        for (int k = 0; k < 3; ++k) {
                            double vkp = V[k * 3 + p], vkq = V[k * 3 + q];
                            V[k * 3 + p] = c * vkp - s * vkq;
                            V[k * 3 + q] = s * vkp + c * vkq;
                        }
                    }
                }
            }
        }
        if (!rotated) break;
    }
    lam[0] = a[0];
    lam[1] = a[4];
    lam[2] = a[8];
    // bubble sort ascending with the oracle's comparison sequence (0,1),(1,2),(0,1)
#define RSDSFM_CSWAP(j)                               \
    if (lam[j] > lam[j + 1]) {                        \
        double t_ = lam[j];                           \
        lam[j] = lam[j + 1];                          \
        lam[j + 1] = t_;                              \
        _Pragma("unroll") for (int r_ = 0; r_ < 3; ++r_) { \
            double tv_ = V[r_ * 3 + j];               \
            V[r_ * 3 + j] = V[r_ * 3 + j + 1];        \
            V[r_ * 3 + j + 1] = tv_;                  \
        }                                             \
    }
    RSDSFM_CSWAP(0)
    RSDSFM_CSWAP(1)
    RSDSFM_CSWAP(0)
#undef RSDSFM_CSWAP
}

constexpr int kSlotsNoK = 81 + 81 + 18;           // W, V, sv/col
constexpr int kSlotsK = kSlotsNoK;                // (the k estimation's two LDS arrays -- P, the eigenvalues -- lie in the not yet used V slots)

}  // namespace

// hyp_out: [T][8] = w(3), v(3), k, status (0 ok, -1 no real k, -2 singular system)
// COOP = false: one hypothesis per LANE (throughput mode, many hypotheses); COOP = true: one hypothesis per WAVE -- every lane
// evaluates the (cheap, register / per-lane-LDS) scalar parts redundantly and the 9x9 SVD is shared (latency mode)
// the solver proper: the body of minimal9_kernel and of the solver workgroups of minimal9_flatten_kernel (one wave; `nblocks` = number
// of workgroups that run this body, for the grid-strided clearing loop)
#ifndef RSDSFM_K_SECTIONS
#define RSDSFM_K_SECTIONS 0
#endif
template <bool COOP>
__device__ __forceinline__ void minimal9_body(double* lds, int block, int nblocks, int lane, const double* __restrict__ q,
                                              const double* __restrict__ u, const double* __restrict__ alpha,
                                              const double* __restrict__ alpha_k, const int32_t* __restrict__ samples, int T,
                                              int use_alpha_k, int k_sign_mode, double* __restrict__ hyp_out,
                                              uint64_t* __restrict__ zero_words, int64_t n_zero_words, const Minimal9Direct& direct) {
    // RANSAC: the per-hypothesis LM states, score marks and flag words must be zero before the depth solves start; the workgroups of
    // this launch clear them on the way (8-byte words, grid-strided) instead of a fill launch in front of it
    if (zero_words)
        for (int64_t i = (int64_t)block * 64 + lane; i < n_zero_words; i += (int64_t)nblocks * 64) zero_words[i] = 0ull;
    const int t = COOP ? block : block * 64 + lane;
    if (t >= T) return;  // per-lane independent work, no workgroup barriers below
    const unsigned long long body_clk0 = (COOP && direct.probe) ? __builtin_amdgcn_s_memtime() : 0ull;
    LVec base{lds + lane};
    LVec Z = base.at(0), V = base.at(81), sv = base.at(162), col = base.at(171);

    // ---- gather the 9 points (minimal.cc:240-243) and build Z (minimal.cc:45-54) ----
    double al[9], alk[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const int64_t idx = samples ? (int64_t)samples[t * 9 + i] : (int64_t)t * 9 + i;
        double x, y, ux, uy;
        if (direct.img) {
            // dense flow: point idx of the column-major scan IS pixel (column idx / rows, row idx % rows); same expressions, same bits
            const int pi = (int)(idx / direct.rows), pj = (int)(idx % direct.rows);
            const double2 f = reinterpret_cast<const double2*>(direct.img)[(int64_t)pj * direct.cols + pi];
            const FlatPoint fp = flatten_point(f, pi, pj, direct.fx, direct.fy, direct.cx, direct.cy, direct.gamma, (double)direct.rows);
            x = fp.qx, y = fp.qy, ux = fp.ux, uy = fp.uy;
            al[i] = direct.alpha_ones ? fp.alpha * 0.0 + 1.0 : fp.alpha;
            alk[i] = fp.alpha_k;
        } else {
            x = q[2 * idx], y = q[2 * idx + 1], ux = u[2 * idx], uy = u[2 * idx + 1];
            al[i] = alpha[idx];
            alk[i] = alpha_k[idx];
        }
        Z[i * 9 + 0] = -uy;
        Z[i * 9 + 1] = ux;
        Z[i * 9 + 2] = uy * x - ux * y;
        Z[i * 9 + 3] = x * x;
        Z[i * 9 + 4] = 2.0 * x * y;
        Z[i * 9 + 5] = 2.0 * x;
        Z[i * 9 + 6] = y * y;
        Z[i * 9 + 7] = 2 * y;
        Z[i * 9 + 8] = 1.0;
    }
    int rc = 0;
    double k = 0.0;
    double beta[9];
#if RSDSFM_K_SECTIONS
    unsigned long long ksec[7] = {0, 0, 0, 0, 0, 0, 0};
#define KSEC(i) ksec[i] = __builtin_amdgcn_s_memtime()
#else
#define KSEC(i)
#endif
    KSEC(0);
    if (use_alpha_k) {
        // minimal.cc:58-80 in registers (inverse_lu_reg, eig6_values_reg); LDS holds Z (built above) and, in the not yet used V slots, P between
        // its construction and the product P PK^-1, and the eigenvalues
        LVec P = base.at(81), ev = base.at(117);
        double a3[9], a_inv[9];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) a3[i * 3 + j] = Z[i * 9 + j];
        if (inverse_lu_reg<3>(a3, a_inv) != 0) {
            k = INFINITY;
            rc = -2;
        } else {
            // dga = dg * a_inv  (dg = Z[3:9, 0:3])
            double dga[18];
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                const double z0 = Z[(3 + i) * 9 + 0], z1 = Z[(3 + i) * 9 + 1], z2 = Z[(3 + i) * 9 + 2];
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    double s = 0.0;
                    s += z0 * a_inv[0 * 3 + j];
                    s += z1 * a_inv[1 * 3 + j];
                    s += z2 * a_inv[2 * 3 + j];
                    dga[i * 3 + j] = s;
                }
            }
            KSEC(1);
            // P (-> LDS) and PK (registers):  alpha_f6 .* efhj - (dga * diag(alpha_f3)) * bc ,  bc = Z[0:3, 3:9], efhj = Z[3:9, 3:9]
            double bc[18], PKr[36];
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 6; ++j) bc[i * 6 + j] = Z[i * 9 + 3 + j];
#pragma unroll
            for (int i = 0; i < 6; ++i) {
#pragma unroll
                for (int j = 0; j < 6; ++j) {
                    const double zz = Z[(3 + i) * 9 + 3 + j];
                    double s = 0.0;
                    s += (dga[i * 3 + 0] * al[0]) * bc[0 * 6 + j];
                    s += (dga[i * 3 + 1] * al[1]) * bc[1 * 6 + j];
                    s += (dga[i * 3 + 2] * al[2]) * bc[2 * 6 + j];
                    P[i * 6 + j] = al[3 + i] * zz - s;
                    double sk = 0.0;
                    sk += (dga[i * 3 + 0] * alk[0]) * bc[0 * 6 + j];
                    sk += (dga[i * 3 + 1] * alk[1]) * bc[1 * 6 + j];
                    sk += (dga[i * 3 + 2] * alk[2]) * bc[2 * 6 + j];
                    PKr[i * 6 + j] = alk[3 + i] * zz - sk;
                }
            }
            KSEC(2);
            double PKinv[36];
            if (inverse_lu_reg<6>(PKr, PKinv) != 0) {
                k = INFINITY;
                rc = -2;
            } else {
                KSEC(3);
                double Hr[36];  // P PK^-1
#pragma unroll
                for (int i = 0; i < 6; ++i) {
                    double pr[6];
#pragma unroll
                    for (int tt = 0; tt < 6; ++tt) pr[tt] = P[i * 6 + tt];
#pragma unroll
                    for (int j = 0; j < 6; ++j) {
                        double s = 0.0;
#pragma unroll
                        for (int tt = 0; tt < 6; ++tt) s += pr[tt] * PKinv[tt * 6 + j];
                        Hr[i * 6 + j] = s;
                    }
                }
                KSEC(4);
                if (eig6_values_reg(Hr, ev) != 0) {
                    k = INFINITY;
                    rc = -2;
                } else {
                    k = INFINITY;  // real eigenvalue of smallest magnitude (minimal.cc:75-80)
                    for (int i = 0; i < 6; ++i)
                        if (fabs(ev[6 + i]) < 0.00001 && fabs(ev[i]) < fabs(k)) k = ev[i];
                    if (isinf(k)) rc = -1;
                    if (k_sign_mode == RSDSFM_K_FIXED && !isinf(k)) k = -k;
                }
            }
        }
        KSEC(5);
#pragma unroll
        for (int i = 0; i < 9; ++i) beta[i] = (al[i] + k * alk[i]) * (2.0 / (2.0 + k));
    } else {
#pragma unroll
        for (int i = 0; i < 9; ++i) beta[i] = al[i];
    }
#pragma unroll
    for (int i = 0; i < 9; ++i)
        for (int j = 3; j < 9; ++j) Z[i * 9 + j] = Z[i * 9 + j] * beta[i];

    // ---- null vector (minimal.cc:98-103) ----
    double e[9];
    if (COOP) {
        // ONE copy of Z (lane 0's: every lane built the same), compact, behind the per-lane block; V beside it
        double* Wc = lds + (size_t)(use_alpha_k ? kSlotsK : kSlotsNoK) * 64;
        __builtin_amdgcn_wave_barrier();
        Wc[lane] = lds[lane * 64];
        if (lane < 17) Wc[64 + lane] = lds[(64 + lane) * 64];
        __builtin_amdgcn_wave_barrier();
        int n_sweeps = 0, n_rotations = 0;
        bool outside = false;
        const unsigned long long svd_clk0 = direct.probe ? __builtin_amdgcn_s_memtime() : 0ull;
        if (direct.core_flag) {
            jacobi_svd9_nullvec_coop<true>(Wc, Wc + kCoopV, lane, sv, col, e, outside, n_sweeps, n_rotations);
            if (outside && lane == 0) *direct.core_flag = direct.core_epoch;  // (every writer of this launch stores the same value)
        } else {
            jacobi_svd9_nullvec_coop<false>(Wc, Wc + kCoopV, lane, sv, col, e, outside, n_sweeps, n_rotations);
        }
        if (direct.probe && lane == 0) {
            direct.probe[4 * t] = (double)n_sweeps;
            direct.probe[4 * t + 1] = (double)n_rotations;
            direct.probe[4 * t + 2] = (double)(__builtin_amdgcn_s_memtime() - svd_clk0);
        }
    } else {
        jacobi_svd9_nullvec(Z, V, sv, col, e);
    }
    const double norm_v0 = sqrt(e[0] * e[0] + e[1] * e[1] + e[2] * e[2]);
#pragma unroll
    for (int i = 0; i < 9; ++i) e[i] = e[i] / norm_v0;

    // ---- Ma et al. recovery of omega (minimal.cc:105-174) ----
    const double S[9] = {e[3], e[4], e[5], e[4], e[6], e[7], e[5], e[7], e[8]};
    double lamb[3], v1[9];
    eig_sym3(S, lamb, v1);
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        double tt = v1[r * 3 + 0];
        v1[r * 3 + 0] = v1[r * 3 + 2];
        v1[r * 3 + 2] = tt;
    }
    const double sigma0 = (2 * lamb[2] + lamb[1] - lamb[0]) / 3;
    const double sigma1 = (lamb[2] + 2 * lamb[1] + lamb[0]) / 3;
    const double sigma2 = (-lamb[2] + lamb[1] + 2 * lamb[0]) / 3;
    const double lambda = sigma0 - sigma2;
    double theta = 0;
    if (!(lambda < 0.000001)) theta = acos(-sigma1 / lambda);
    double r_v[9], r_u[9], r_vt[9], v_[9], u_[9], r_z1[9], r_z2[9], negv[9];
    angle_axis_R((theta - kPi) / 2, 0, 1, 0, r_v);
    angle_axis_R(theta, 0, 1, 0, r_u);
    tr3(r_v, r_vt);
    mm3(v1, r_vt, v_);
#pragma unroll
    for (int i = 0; i < 9; ++i) negv[i] = -v_[i];
    mm3(negv, r_u, u_);
    angle_axis_R(kPi / 2, 0, 0, 1, r_z1);
    angle_axis_R(-kPi / 2, 0, 0, 1, r_z2);
    const double sig1[9] = {1, 0, 0, 0, 1, 0, 0, 0, 0};
    double sig_lamb[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) sig_lamb[i] = lambda * sig1[i];
    double dots[4];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
#pragma unroll
        for (int zz = 0; zz < 2; ++zz) {
            double t1[9], t2[9], bt[9], mm[9];
            if (b == 0) {
                if (zz == 0) mm3(v_, r_z1, t1); else mm3(v_, r_z2, t1);
                mm3(t1, sig1, t2);
                tr3(v_, bt);
            } else {
                if (zz == 0) mm3(u_, r_z1, t1); else mm3(u_, r_z2, t1);
                mm3(t1, sig1, t2);
                tr3(u_, bt);
            }
            mm3(t2, bt, mm);
            dots[b * 2 + zz] = mm[7] * e[0] + mm[2] * e[1] + mm[3] * e[2];  // (M(2,1), M(0,2), M(1,0)) . v0
        }
    }
    int index_max = 0;
    double best = dots[0];
#pragma unroll
    for (int c = 1; c < 4; ++c)
        if (dots[c] > best) {
            best = dots[c];
            index_max = c;
        }
    double wb[9], wz[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        wb[i] = (index_max < 2) ? u_[i] : v_[i];  // omega from the OTHER basis (minimal.cc:159-173)
        wz[i] = (index_max % 2 == 0) ? r_z1[i] : r_z2[i];
    }
    double t1[9], t2[9], bt[9], w_hat[9];
    mm3(wb, wz, t1);
    mm3(t1, sig_lamb, t2);
    tr3(wb, bt);
    mm3(t2, bt, w_hat);
    if (COOP && lane != 0) return;
    if (COOP && direct.probe) direct.probe[4 * t + 3] = (double)(__builtin_amdgcn_s_memtime() - body_clk0);
#if RSDSFM_K_SECTIONS
    if (COOP && direct.probe && use_alpha_k) {  // (diagnostic build: tools/k_sections.py) sections of the k estimation in shader clocks
        direct.probe[4 * t + 0] = (double)(ksec[2] - ksec[0]);  // Z, 3x3 inverse, dga, P, PK
        direct.probe[4 * t + 1] = (double)(ksec[3] - ksec[2]);  // 6x6 inverse
        direct.probe[4 * t + 2] = (double)(ksec[4] - ksec[3]);  // P PK^-1
        direct.probe[4 * t + 3] = (double)(ksec[5] - ksec[4]);  // eigenvalues
    }
#endif
    double* o = hyp_out + (int64_t)t * 8;
    o[0] = w_hat[7];
    o[1] = w_hat[2];
    o[2] = w_hat[3];
    o[3] = e[0];
    o[4] = e[1];
    o[5] = e[2];
    o[6] = k;
    o[7] = (double)rc;
}

template <bool COOP>
__global__ __launch_bounds__(64) void minimal9_kernel(const double* __restrict__ q, const double* __restrict__ u,
                                                     const double* __restrict__ alpha,
                                                     const double* __restrict__ alpha_k,
                                                     const int32_t* __restrict__ samples, int T, int use_alpha_k,
                                                     int k_sign_mode, double* __restrict__ hyp_out, uint64_t* __restrict__ zero_words,
                                                     int64_t n_zero_words, Minimal9Direct direct) {
    extern __shared__ double lds[];
    minimal9_body<COOP>(lds, (int)blockIdx.x, (int)gridDim.x, (int)threadIdx.x, q, u, alpha, alpha_k, samples, T, use_alpha_k, k_sign_mode, hyp_out,
                        zero_words, n_zero_words, direct);
}

// ---------------------------------------------------------------------------------------------------
// The minimal solver and the flatten of a DENSE frame in ONE launch.  The solver is latency-bound on T waves (one hypothesis per
// wave, ~186 us of a serial division / square-root chain) while 950+ SIMDs idle; the flatten (main.cc:398-444: 16 B read + 48 B
// written per pixel) needs ~10 us of the whole GPU.  Two streams could overlap them, but the cross-stream join costs more than the
// flatten (measured).  So the workgroups of one launch take two roles: the first T run the solver on wave 0 (the
// other three waves leave at once), forming their sampled points straight from the flow image (Minimal9Direct); the rest flatten.
// A dense frame needs no scan: when every pixel carries flow, point index = column * rows + row, so every tile knows where it
// writes.  Pixels that ARE dropped are counted (integer atomics); the workgroup that finishes last publishes rows * cols - dropped
// to host-mapped memory and resets the counters, and the host -- which set the RANSAC up for n = rows * cols -- runs the general
// flatten and everything behind it again if the count says the frame was not dense (frame_host.hip).  Same expressions
// (flatten_point), same output order: the arrays of a dense frame are bit-identical to flatten_tile_kernel's.
// ---------------------------------------------------------------------------------------------------
constexpr int kDF_W = 8, kDF_H = 64;  // tile columns (one 128-byte line per image row) x tile rows (one wave per column)

__global__ __launch_bounds__(256) void minimal9_flatten_kernel(const int32_t* __restrict__ samples, int T, int use_alpha_k, int k_sign_mode,
                                                              double* __restrict__ hyp_out, uint64_t* __restrict__ zero_words,
                                                              int64_t n_zero_words, Minimal9Direct direct, double thr,
                                                              double2* __restrict__ q, double2* __restrict__ u,
                                                              double* __restrict__ alpha, double* __restrict__ alpha_k,
                                                              unsigned long long* __restrict__ counters, int64_t* __restrict__ total_out,
                                                              int tiles_x, int nflat) {
    extern __shared__ double lds[];
    const int tid = threadIdx.x;
    if ((int)blockIdx.x < T) {
        if (tid >= 64) return;
        minimal9_body<true>(lds, (int)blockIdx.x, T, tid, nullptr, nullptr, nullptr, nullptr, samples, T, use_alpha_k, k_sign_mode, hyp_out, zero_words,
                            n_zero_words, direct);
        return;
    }
    // ---- dense flatten of one 8 x 64 tile (the tile lives at the start of the dynamic LDS block) ----
    double2(*s_tile)[kDF_W + 1] = reinterpret_cast<double2(*)[kDF_W + 1]>(lds);
    const int b = (int)blockIdx.x - T;
    const int rows = direct.rows, cols = direct.cols;
    const int c0 = (b % tiles_x) * kDF_W, r0 = (b / tiles_x) * kDF_H;
    const double2* img = reinterpret_cast<const double2*>(direct.img);
#pragma unroll
    for (int k = 0; k < kDF_H * kDF_W / 256; ++k) {
        const int lr = tid / kDF_W + k * (256 / kDF_W), lc = tid % kDF_W;
        const int r = r0 + lr, c = c0 + lc;
        s_tile[lr][lc] = (r < rows && c < cols) ? img[(int64_t)r * cols + c] : make_double2(0.0, 0.0);
    }
    __syncthreads();
    const int lane = tid & 63, wv = tid >> 6;
    const double h = (double)rows;
    unsigned dropped = 0;
    for (int lc = wv; lc < kDF_W; lc += 4) {
        const int i = c0 + lc, j = r0 + lane;  // column i, row j
        if (i >= cols) break;
        const double2 f = s_tile[lane][lc];
        const bool inside = j < rows;
        const bool keep = inside && f.x * f.x + f.y * f.y > thr;
        dropped += (unsigned)__popcll(__ballot(inside && !keep));
        if (keep) {
            const int64_t o = (int64_t)i * rows + j;  // every pixel in front of this one is kept: a dense frame
            const FlatPoint fp = flatten_point(f, i, j, direct.fx, direct.fy, direct.cx, direct.cy, direct.gamma, h);
            q[o] = make_double2(fp.qx, fp.qy);
            u[o] = make_double2(fp.ux, fp.uy);
            alpha[o] = direct.alpha_ones ? fp.alpha * 0.0 + 1.0 : fp.alpha;
            alpha_k[o] = fp.alpha_k;
        }
    }
    // counters[0] = dropped pixels, counters[1] = flatten workgroups done; the last one publishes the count and resets both.
    // No fence (a device-scope fence writes back the XCD's whole L2: measured ruinous in round 2): the only data another workgroup
    // reads are the two counters, which are device-scope atomics performed at the memory side.  What has to be ordered is "this
    // workgroup's dropped-count is added BEFORE its ticket is drawn": the add RETURNS its old value, the value travels through LDS
    // into the operand of the ticket atomic (>> 63: always 0, but a real data dependency), so the ticket cannot issue earlier.
    __shared__ unsigned long long s_prev[4];
    if (lane == 0) s_prev[wv] = dropped ? atomicAdd(&counters[0], (unsigned long long)dropped) : 0ull;
    __syncthreads();
    if (tid == 0) {
        const unsigned long long seen = s_prev[0] | s_prev[1] | s_prev[2] | s_prev[3];
        const unsigned long long ticket = atomicAdd(&counters[1], 1ull + (seen >> 63));
        if (ticket == (unsigned long long)(nflat - 1)) {
            const unsigned long long d = atomicAdd(&counters[0], ticket >> 63);  // (reads the sum; ordered behind the ticket the same way)
            *total_out = (int64_t)rows * cols - (int64_t)d;
            counters[0] = 0ull;
            counters[1] = 0ull;
        }
    }
}

int minimal9_flatten_launch(Ctx* c, const int32_t* samples, int T, int use_alpha_k, int k_sign_mode, double* hyp_out, void* zero_begin, size_t zero_bytes,
                            const Minimal9Direct& direct, double thr, double* d_q, double* d_u, double* d_alpha, double* d_alpha_k,
                            unsigned long long* d_counters, int64_t* total_out) {
    const size_t lds_bytes = (size_t)((use_alpha_k ? kSlotsK : kSlotsNoK) + kCoopSlots) * 64 * sizeof(double);
    static_assert(sizeof(double2) * kDF_H * (kDF_W + 1) <= (size_t)kSlotsNoK * 64 * sizeof(double), "the flatten tile fits the solver's LDS block");
    static bool attr_set_dev[64] = {false};
    bool& attr_set = attr_set_dev[c->device & 63];
    if (!attr_set) {
        RSDSFM_HIP_CHECK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(minimal9_flatten_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                (int)((size_t)(kSlotsK + kCoopSlots) * 64 * sizeof(double))));
        attr_set = true;
    }
    const int tiles_x = (direct.cols + kDF_W - 1) / kDF_W, tiles_y = (direct.rows + kDF_H - 1) / kDF_H;
    const int nflat = tiles_x * tiles_y;
    hipLaunchKernelGGL(minimal9_flatten_kernel, dim3(T + nflat), dim3(256), lds_bytes, c->stream, samples, T, use_alpha_k, k_sign_mode, hyp_out,
                       reinterpret_cast<uint64_t*>(zero_begin), (int64_t)(zero_bytes / 8), direct, thr, reinterpret_cast<double2*>(d_q),
                       reinterpret_cast<double2*>(d_u), d_alpha, d_alpha_k, d_counters, total_out, tiles_x, nflat);
    RSDSFM_HIP_CHECK(c, hipGetLastError());
    return RSDSFM_OK;
}

int minimal9_launch(Ctx* c, const double* q, const double* u, const double* alpha, const double* alpha_k,
                    const int32_t* samples, int T, int use_alpha_k, int k_sign_mode, double* hyp_out, void* zero_begin, size_t zero_bytes,
                    const Minimal9Direct* direct_or_null) {
    if (T <= 0) return RSDSFM_OK;
    const Minimal9Direct direct = direct_or_null ? *direct_or_null : Minimal9Direct();
    uint64_t* zero_words = reinterpret_cast<uint64_t*>(zero_begin);
    const int64_t n_zero_words = (int64_t)(zero_bytes / 8);
    const size_t lds_bytes = (size_t)((use_alpha_k ? kSlotsK : kSlotsNoK) + kCoopSlots) * 64 * sizeof(double);
    static bool attr_set_dev[64] = {false};  // (per device: a function attribute belongs to the device that is current)
    bool& attr_set = attr_set_dev[c->device & 63];
    if (!attr_set) {
        RSDSFM_HIP_CHECK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(minimal9_kernel<false>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize,
                                                (int)((size_t)(kSlotsK + kCoopSlots) * 64 * sizeof(double))));
        RSDSFM_HIP_CHECK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(minimal9_kernel<true>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize,
                                                (int)((size_t)(kSlotsK + kCoopSlots) * 64 * sizeof(double))));
        attr_set = true;
    }
    // few hypotheses (RANSAC: T = 5 ... a few hundred): one wave per hypothesis, 9x9 SVD shared by the wave; many: one lane each
    if (T <= c->num_cus * 2)
        hipLaunchKernelGGL(minimal9_kernel<true>, dim3(T), dim3(64), lds_bytes, c->stream, q, u, alpha, alpha_k, samples, T, use_alpha_k,
                           k_sign_mode, hyp_out, zero_words, n_zero_words, direct);
    else
        hipLaunchKernelGGL(minimal9_kernel<false>, dim3((T + 63) / 64), dim3(64), lds_bytes, c->stream, q, u, alpha, alpha_k, samples, T,
                           use_alpha_k, k_sign_mode, hyp_out, zero_words, n_zero_words, direct);
    RSDSFM_HIP_CHECK(c, hipGetLastError());
    return RSDSFM_OK;
}

}  // namespace rsdsfm
