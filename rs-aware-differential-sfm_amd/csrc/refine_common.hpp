// refine_common.hpp -- shared pieces of the joint refinement's kernels (refine_kernels.hip: the iterate-by-iterate slot kernels in the
// reference's arithmetic; refine_rf_kernels.hip: the radius-factorised pass): workgroup shape, residual / Jacobian of one inlier, the
// fixed-order reductions, row layouts.  Everything here has internal linkage per translation unit.
#pragma once
#include "device_math.hpp"
#include "rsdsfm_internal.hpp"

namespace rsdsfm {
namespace {


// Workgroups of 8 waves, at most one per CU: the streaming passes run at two waves per SIMD either way (254 registers), and half as many
// partial rows leave half as much for the single-workgroup stages to reduce -- those stages pull their rows through ONE CU (measured with
// s_memtime stamps at 1280x720, 512 rows of 4 waves each: 3.4 us for the 14 back-substitution columns, 4.8 us for the 54 Schur columns,
// against 1.2 us for the decision and 2.4 us for the Cholesky solve).
constexpr int kFB = 512;
constexpr int kWorkgroupsPerCu = 1;

// Streaming passes: `m_arg >= 0` is the inlier count and the launch grid is the logical grid (host knows both); `m_arg < 0` means
// both are device-resident (RefineState::m / ::grid) and the launch grid is an upper bound: workgroups beyond the logical grid leave,
// the others stride by the LOGICAL grid, so every partial row holds exactly the sums it holds on the host-sized launch.
struct PassShape {
    int64_t m;
    int grid;
    bool live;
};
__device__ __forceinline__ PassShape pass_shape(const RefineState* __restrict__ st, int64_t m_arg) {
    PassShape ps;
    if (m_arg >= 0) {
        ps.m = m_arg, ps.grid = (int)gridDim.x, ps.live = true;
    } else {
        ps.m = st->m, ps.grid = st->grid, ps.live = (int)blockIdx.x < ps.grid;
    }
    return ps;
}

template <int NP>
struct RJ {
    double r[2];
    double Jp[2][NP];
    double Jr[2];
};

// residual and analytic Jacobian at (p, rho); p = (v0,v1,v2,w0,w1,w2,k)
// beta of nonlinearRefinement.cc:35 and (NP == 7) its derivative with respect to k
__device__ __forceinline__ double beta_of(double alpha, double alpha_k, double k) { return (2.0 / (2.0 + k)) * (alpha + k * alpha_k); }
__device__ __forceinline__ double dbeta_of(double alpha, double alpha_k, double k) { return 2.0 * (2.0 * alpha_k - alpha) / ((2.0 + k) * (2.0 + k)); }

template <int NP>
__device__ __forceinline__ void resid_jac_beta(double x, double y, double ux, double uy, double beta, double dbeta,
                                               const double (&p)[7], double rho, RJ<NP>& o);

template <int NP>
__device__ __forceinline__ void resid_jac(double x, double y, double ux, double uy, double alpha, double alpha_k,
                                          const double (&p)[7], double rho, RJ<NP>& o) {
    const double k = p[6];
    resid_jac_beta<NP>(x, y, ux, uy, beta_of(alpha, alpha_k, k), NP == 7 ? dbeta_of(alpha, alpha_k, k) : 0.0, p, rho, o);
}

// residual and Jacobian for a given beta (and d beta / d k): the streaming passes read beta precomputed when k is fixed (NP == 6)
template <int NP>
__device__ __forceinline__ void resid_jac_beta(double x, double y, double ux, double uy, double beta, double dbeta,
                                               const double (&p)[7], double rho, RJ<NP>& o) {
    const double a0 = x * p[2] - p[0], a1 = y * p[2] - p[1];
    const double in0 = rho * a0 + (x * y * p[3]) - (1.0 + x * x) * p[4] + y * p[5];
    const double in1 = rho * a1 + (1.0 + y * y) * p[3] - x * y * p[4] - x * p[5];
    o.r[0] = ux - beta * -1.0 * in0;
    o.r[1] = uy - beta * -1.0 * in1;
    const double br = beta * rho;
    o.Jp[0][0] = -br;
    o.Jp[1][0] = 0.0;
    o.Jp[0][1] = 0.0;
    o.Jp[1][1] = -br;
    o.Jp[0][2] = br * x;
    o.Jp[1][2] = br * y;
    o.Jp[0][3] = beta * (x * y);
    o.Jp[1][3] = beta * (1.0 + y * y);
    o.Jp[0][4] = -(beta * (1.0 + x * x));
    o.Jp[1][4] = -(beta * (x * y));
    o.Jp[0][5] = beta * y;
    o.Jp[1][5] = -(beta * x);
    if (NP == 7) {
        o.Jp[0][NP - 1] = dbeta * in0;
        o.Jp[1][NP - 1] = dbeta * in1;
    }
    o.Jr[0] = beta * a0;
    o.Jr[1] = beta * a1;
}

// Wave reduction of NV per-lane values into red_row[NV] (one row per wave; valid for other threads after a workgroup
// barrier).  NV is up to 70 here, and a thread only handles a handful of inliers, so NV DPP butterflies (18 instructions
// each) were almost half of the streaming kernels' instruction count.  The sums go through a per-wave LDS transpose
// instead, 8 slots per round (row stride 65: conflict-free): every lane stores its 8 values, lane (slot, eighth) adds 8
// consecutive lanes' values, lane `slot` adds the 8 eighths in order -- a fixed order, ~5 instructions per slot.  The
// (at most one) max slot is reduced with the DPP butterfly afterwards.  A wave's LDS operations execute in order; the
// wave_barrier only pins the compiler's ordering.
template <int NV>
__device__ __forceinline__ void wave_reduce_to_row(const double (&v)[NV], int max_slot, double* red_row) {
    __shared__ double s_T[kFB / 64][8 * 65];
    __shared__ double s_part[kFB / 64][8][8];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    double* Tw = s_T[wv];
    constexpr int R = (NV + 7) / 8;
#pragma unroll
    for (int r = 0; r < R; ++r) {
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (r * 8 + k < NV) Tw[k * 65 + lane] = v[r * 8 + k];
        __builtin_amdgcn_wave_barrier();
        const int sl = lane & 7, pt = lane >> 3;
        const double* row = Tw + sl * 65 + pt * 8;
        double part = row[0];
#pragma unroll
        for (int j = 1; j < 8; ++j) part += row[j];
        s_part[wv][sl][pt] = part;
        __builtin_amdgcn_wave_barrier();
        if (lane < 8 && r * 8 + lane < NV) {
            double t = s_part[wv][lane][0];
#pragma unroll
            for (int p2 = 1; p2 < 8; ++p2) t += s_part[wv][lane][p2];
            red_row[r * 8 + lane] = t;
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (max_slot >= 0) {
#pragma unroll
        for (int s = 0; s < NV; ++s)
            if (s == max_slot) {  // uniform branch; only this slot pays for a butterfly
                const double r = wave_max(v[s]);
                if (lane == 0) red_row[s] = r;
            }
    }
}

// Wave reduction of the first N (<= NV) per-lane values by HALVING: at step s (partner = lane ^ (1 << s)) a lane keeps one half of the values
// it still holds and hands the other half to its partner, which keeps exactly that half -- N / 2 + N / 4 + ... ~ N exchanged doubles per lane
// where a butterfly per value moves 6 N, and no LDS until every lane is down to ceil(N / 64) values (wave_reduce_to_row above pushes every
// value of every lane through LDS twice: ~3 us for the 60 sums of a refinement pass, the LDS pipe being the limit).  Lane l ends with the wave
// sums of the logical indices [offset, offset + count) it returns; the combination tree is fixed (deterministic), another one than
// wave_reduce_to_row's.  Steps 0 / 1 exchange within quads (DPP), the others through ds_bpermute.
template <int S>
__device__ __forceinline__ double halving_exchange(double x) {
    if (S == 0) return dpp_move<0xb1, 0xf>(x);  // quad_perm:[1,0,3,2]
    if (S == 1) return dpp_move<0x4e, 0xf>(x);  // quad_perm:[2,3,0,1]
    return __shfl_xor(x, 1 << S, 64);
}
template <int NV, int N, int S>
struct HalvingStep {
    __device__ static __forceinline__ void run(double (&v)[NV], int lane, int& offset, int& count) {
        constexpr int H = (N + 1) / 2;
        const bool bit = (lane >> S) & 1;
#pragma unroll
        for (int j = 0; j < H; ++j) {
            const double lo = v[j];
            const double hi = (H + j < N) ? v[H + j] : 0.0;
            const double send = bit ? lo : hi, keep = bit ? hi : lo;
            v[j] = keep + halving_exchange<S>(send);
        }
        offset += bit ? H : 0;
        count = bit ? count - min(count, H) : min(count, H);
        HalvingStep<NV, H, S + 1>::run(v, lane, offset, count);
    }
};
template <int NV, int N>
struct HalvingStep<NV, N, 6> {
    __device__ static __forceinline__ void run(double (&)[NV], int, int&, int&) {}
};
__host__ __device__ constexpr int halving_final(int n) {  // values a lane still holds after the six steps
    for (int s = 0; s < 6; ++s) n = (n + 1) / 2;
    return n;
}
// reduces v[0 .. N) over the wave; afterwards v[j], j < count, is the wave sum of logical index offset + j (count <= halving_final(N))
template <int NV, int N>
__device__ __forceinline__ void wave_reduce_halving(double (&v)[NV], int& offset, int& count) {
    offset = 0, count = N;
    HalvingStep<NV, N, 0>::run(v, (int)(threadIdx.x & 63), offset, count);
}
// workgroup reduction on top of it: sums of v[0 .. NV - 1) and the maximum of v[NV - 1] (the rows of refine_rf_kernels.hip keep their one
// max slot last) -> out_row[NV]
template <int NV>
__device__ __forceinline__ void block_reduce_store_halving(double (&v)[NV], double (*s_red)[NV], double* __restrict__ out_row) {
    const int tid = threadIdx.x, wv = tid >> 6;
    const double mx = wave_max(v[NV - 1]);
    int offset, count;
    wave_reduce_halving<NV, NV - 1>(v, offset, count);
    constexpr int NF = halving_final(NV - 1);
#pragma unroll
    for (int j = 0; j < NF; ++j)
        if (j < count) s_red[wv][offset + j] = v[j];
    if ((tid & 63) == 0) s_red[wv][NV - 1] = mx;
    __syncthreads();
    if (tid < NV) {
        double r = s_red[0][tid];
        for (int w2 = 1; w2 < kFB / 64; ++w2) r = (tid == NV - 1) ? fmax(r, s_red[w2][tid]) : r + s_red[w2][tid];
        out_row[tid] = r;
    }
}

// generic fixed-order workgroup reduction of NV per-thread values; kinds: slot s is a max slot iff s == max_slot
template <int NV>
__device__ __forceinline__ void block_reduce_store(const double (&v)[NV], int max_slot, double (*s_red)[NV],
                                                   double* __restrict__ out_row) {
    const int tid = threadIdx.x, wv = tid >> 6;
    wave_reduce_to_row<NV>(v, max_slot, s_red[wv]);
    __syncthreads();
    if (tid < NV) {
        double r = s_red[0][tid];
        for (int w2 = 1; w2 < kFB / 64; ++w2) r = (tid == max_slot) ? fmax(r, s_red[w2][tid]) : r + s_red[w2][tid];
        out_row[tid] = r;
    }
}

// single-workgroup fixed-order reduction of partials[nblocks][NV] into s_out[NV].  Thread (group g, slot pair sp) adds the rows
// g, g + G, g + 2G, ... of its two slots in order (adjacent lanes read adjacent 16-byte pieces of a row: coalesced, 16 loads in
// flight), then thread s adds the G group sums of slot s in order.  (The first version gave every thread whole rows and reduced the NV per-thread sums with NV / 8 rounds of
// LDS transposes: 7 us of the 10 us refine_solve_kernel, measured by returning right after the reduction.)
// stride / offset (in doubles; both even when NV is): the NV slots are a column range of wider rows (the slot rows of the column-tiled solve)
template <int NV>
struct ReduceShape {
    static constexpr int W = (NV % 2 == 0) ? 2 : 1;  // slots per lane: pairs as double2 when the rows are 16-byte aligned (NV even)
    static constexpr int NH = NV / W;                // lanes per row
    static constexpr int G = kFB / NH;               // row groups (9 for the 54 Schur sums of NP = 6)
};
// part 1 (thread `tid` of kFB): the group sums into s_grp[G][NV]; part 2, behind a workgroup barrier: thread s < NV adds the G group sums of slot s
template <int NV>
__device__ __forceinline__ void reduce_partials_groups(const double* __restrict__ partials, int nblocks, int max_slot, double (*s_grp)[NV], int tid,
                                                       int stride, int offset) {
    constexpr int W = ReduceShape<NV>::W, NH = ReduceShape<NV>::NH, G = ReduceShape<NV>::G;
    constexpr int U = 16;  // independent loads in flight per thread: the reduction is bound by load latency
    const int g = tid / NH, sp = tid - g * NH;
    if (g < G) {
        const bool mx0 = (W * sp == max_slot), mx1 = (W * sp + 1 == max_slot);
        double a0 = 0.0, a1 = 0.0;
        for (int b = g; b < nblocks; b += U * G) {
            double v0[U], v1[U];
#pragma unroll
            for (int j = 0; j < U; ++j) {  // rows past the end contribute the identity (sums: + 0.0; the max slot holds absolute values)
                const int bj = b + j * G;
                const int64_t row = bj < nblocks ? bj : g;
                if (W == 2) {
                    const double2 x = reinterpret_cast<const double2*>(partials + row * stride + offset)[sp];
                    v0[j] = bj < nblocks ? x.x : 0.0;
                    v1[j] = bj < nblocks ? x.y : 0.0;
                } else {
                    const double x = partials[row * stride + offset + sp];
                    v0[j] = bj < nblocks ? x : 0.0;
                    v1[j] = 0.0;
                }
            }
#pragma unroll
            for (int j = 0; j < U; ++j) {
                a0 = mx0 ? fmax(a0, v0[j]) : a0 + v0[j];
                if (W == 2) a1 = mx1 ? fmax(a1, v1[j]) : a1 + v1[j];
            }
        }
        s_grp[g][W * sp] = a0;
        if (W == 2) s_grp[g][W * sp + 1] = a1;
    }
}
template <int NV>
__device__ __forceinline__ void reduce_partials_slots(const double (*s_grp)[NV], int max_slot, double* s_out, int tid) {
    constexpr int G = ReduceShape<NV>::G;
    if (tid < NV) {
        double r = s_grp[0][tid];
#pragma unroll
        for (int g2 = 1; g2 < G; ++g2) r = (tid == max_slot) ? fmax(r, s_grp[g2][tid]) : r + s_grp[g2][tid];
        s_out[tid] = r;
    }
}
template <int NV>
__device__ __forceinline__ void reduce_partials(const double* __restrict__ partials, int nblocks, int max_slot,
                                                double (*s_red)[NV], double* s_out, int stride = NV, int offset = 0) {
    __shared__ double s_grp[ReduceShape<NV>::G][NV];
    reduce_partials_groups<NV>(partials, nblocks, max_slot, s_grp, threadIdx.x, stride, offset);
    __syncthreads();
    reduce_partials_slots<NV>(s_grp, max_slot, s_out, threadIdx.x);
    __syncthreads();
    (void)s_red;
}

// The group sums of TWO column ranges of the same rows with every load of both issued before the first addition (one round trip to the rows
// instead of two): thread tid is lane (gA, spA) of range A and lane (gB, spB) of range B, each adds its rows in reduce_partials_groups' order.
// Only for row counts one batch covers (nblocks <= 16 G of both ranges: 288 rows for the 54 Schur sums); returns false otherwise.
template <int NA, int NB>
__device__ __forceinline__ bool reduce_two_ranges_groups(const double* __restrict__ rows, int nblocks, int stride, int offA, int maxA, bool doA,
                                                         double (*s_grpA)[NA], int offB, int maxB, bool doB, double (*s_grpB)[NB], int tid) {
    using SA = ReduceShape<NA>;
    using SB = ReduceShape<NB>;
    static_assert(SA::W == 2 && SB::W == 2, "slot pairs");
    constexpr int U = 16;
    if (nblocks > U * SA::G || nblocks > U * SB::G) return false;
    const int gA = tid / SA::NH, spA = tid - gA * SA::NH, gB = tid / SB::NH, spB = tid - gB * SB::NH;
    const bool onA = doA && gA < SA::G, onB = doB && gB < SB::G;
    double2 vA[U], vB[U];
#pragma unroll
    for (int j = 0; j < U; ++j) {
        const int bj = gA + j * SA::G;
        vA[j] = (onA && bj < nblocks) ? reinterpret_cast<const double2*>(rows + (int64_t)bj * stride + offA)[spA] : make_double2(0.0, 0.0);
    }
#pragma unroll
    for (int j = 0; j < U; ++j) {
        const int bj = gB + j * SB::G;
        vB[j] = (onB && bj < nblocks) ? reinterpret_cast<const double2*>(rows + (int64_t)bj * stride + offB)[spB] : make_double2(0.0, 0.0);
    }
    if (onA) {
        const bool mx0 = (2 * spA == maxA), mx1 = (2 * spA + 1 == maxA);
        double a0 = 0.0, a1 = 0.0;
#pragma unroll
        for (int j = 0; j < U; ++j) {
            a0 = mx0 ? fmax(a0, vA[j].x) : a0 + vA[j].x;
            a1 = mx1 ? fmax(a1, vA[j].y) : a1 + vA[j].y;
        }
        s_grpA[gA][2 * spA] = a0;
        s_grpA[gA][2 * spA + 1] = a1;
    }
    if (onB) {
        const bool mx0 = (2 * spB == maxB), mx1 = (2 * spB + 1 == maxB);
        double a0 = 0.0, a1 = 0.0;
#pragma unroll
        for (int j = 0; j < U; ++j) {
            a0 = mx0 ? fmax(a0, vB[j].x) : a0 + vB[j].x;
            a1 = mx1 ? fmax(a1, vB[j].y) : a1 + vB[j].y;
        }
        s_grpB[gB][2 * spB] = a0;
        s_grpB[gB][2 * spB + 1] = a1;
    }
    return true;
}

template <int NP>
struct Counts {
    static constexpr int TRI = NP * (NP + 1) / 2;
    static constexpr int NINIT = 1 + NP + NP + 1 + 1 + 1;   // cost2, colsq[NP], gp[NP], gmax_rho (max), xsq_rho, sum of 1 / rho (want_zsum)
    static constexpr int INIT_MAX = 1 + 2 * NP;
    static constexpr int NSCHUR = 2 * TRI + 2 * NP;         // FtF tri, C tri, Ftb, cvec
    static constexpr int NBACK = 3 + 1 + NP + 1 + 1 + 1 + 1;  // model, stepsq_rho, ccost2 | cost2@cand (= ccost2), gp[NP], gmax_rho, xsq_rho, sum of 1 / rho@cand (want_zsum), one unused slot (keeps the rows of NP = 6 an even number of doubles: 16-byte loads in reduce_partials)
    static constexpr int BACK_MAX = 3 + 1 + NP;
};

// one row of a SLOT (see the slot kernels below): the Schur sums and the back-substitution sums side by side
template <int NP>
struct SlotRow {
    using CT = Counts<NP>;
    static constexpr int OFF_BACK = CT::NSCHUR;                                   // (even: NBACK rows of NP = 6 stay 16-byte aligned)
    static constexpr int NW = CT::NSCHUR + CT::NBACK + ((CT::NBACK & 1) ? 1 : 0);  // row width (even)
    // Whether the back-substitution pass speculates at all.  With k refined (NP = 7) it does not: the steps of that problem are rejected or
    // accepted with qualities of 0.65 .. 0.93 most of the time (tools/refine_slots.py with SLOTS_ACCEL=1: 15 of 21 at worst), and a pass
    // that carries both evaluations needs more than 256 registers there (one wave per SIMD).  Its slots alternate Schur / back-substitution.
    static constexpr bool SPECULATES = NP == 6;
};


// a wave-uniform double read from LDS, moved to scalar registers (the slot pass used to read its state through the scalar data path: 26 doubles
// of pose, scales and step that would otherwise occupy 52 vector registers of a kernel that has none to spare)
__device__ __forceinline__ double uniform_d(double x) {
    const int lo = __builtin_amdgcn_readfirstlane(__double2loint(x)), hi = __builtin_amdgcn_readfirstlane(__double2hiint(x));
    return __hiloint2double(hi, lo);
}

// the state of a slot kernel: copied to LDS, one 8-byte word per thread
__device__ __forceinline__ void state_to_lds(RefineState* s_st, const RefineState* __restrict__ st_in) {
    static_assert(sizeof(RefineState) % 8 == 0 && sizeof(RefineState) / 8 <= kFB, "copied as 8-byte words, one per thread");
    if (threadIdx.x < sizeof(RefineState) / 8) reinterpret_cast<double*>(s_st)[threadIdx.x] = reinterpret_cast<const double*>(st_in)[threadIdx.x];
    __syncthreads();
}
__device__ __forceinline__ void state_from_lds(RefineState* __restrict__ st_out, const RefineState* s_st) {
    if (threadIdx.x < sizeof(RefineState) / 8) reinterpret_cast<double*>(st_out)[threadIdx.x] = reinterpret_cast<const double*>(s_st)[threadIdx.x];
}

}  // namespace
}  // namespace rsdsfm
