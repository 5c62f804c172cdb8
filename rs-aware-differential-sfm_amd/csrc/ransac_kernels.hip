// ransac_kernels.hip -- hypothesis-batched RANSAC on MI355X (gfx950).
//
// Replaces the trial loop of minimal::ransac (reference minimal.cc:230-289) and its inlier compaction
// (minimal.cc:291-305).  The reference runs, per trial, a Ceres depth solve of ALL points and then scores them;
// here all T hypotheses are processed together:
//
//   ransac_lm_kernel      every workgroup keeps a tile of pixels in registers (48 B/pixel read ONCE per round for
//                         all hypotheses), loops over the hypotheses and block-reduces the NS speculative-LM sums
//                         of each (DPP wave reductions), accumulating them in LDS -> partials[block][T][NS]
//   ransac_decide_kernel  one workgroup per hypothesis: fixed-order reduction of the partials + the Ceres
//                         trust-region state machine (lm_advance); counts the hypotheses that need another round
//   ransac_score_kernel   per hypothesis: replay the accepted LM steps per pixel (or closed form), residual norm,
//                         inlier test (minimal.cc:255-275) -> partial {count, sum err}
//   ransac_select_kernel  fixed-order reduction, then the reference's sequential best-trial rule (minimal.cc:278)
//   ransac_final_kernel / ransac_scan_kernel / ransac_scatter_kernel
//                         dense rho + mask of the best trial, and the order-preserving compaction of
//                         (x, y, 1/rho), alpha, alpha_k, index (block counts -> exclusive scan -> scatter)
//
// Pixels are streamed (HBM-bound when T is small, VALU-bound for large T: ~250 fp64 ops per pixel-hypothesis).
#include <algorithm>

#include "device_math.hpp"
#include "lm_common.hpp"
#include "rsdsfm_internal.hpp"

namespace rsdsfm {

namespace {

constexpr int kRB = 256;  // workgroup size of the pixel kernels
constexpr int kRP = RSDSFM_FUSED ? 5 : 6;  // (the fused build's point_error does not share terms with the pixel model: 6 would take 256 VGPRs and one wave per SIMD there)
                                        // pixels per thread per tile, register-resident across the hypothesis loop: every hypothesis pays one workgroup reduction of its 22 sums per tile, so more pixels per thread amortise it (round 1, fused model: 4 / 5: 451 / 419 us, 6 dropped to one wave per SIMD; round 2, reference arithmetic with the scores taken from the pixel model: 5 / 6: 0.476 / 0.471 ms at 246 VGPRs, 6.60 / 6.42 ms for the 4K tiled solve; 7 / 8 fit 256 VGPRs without scratch but run at 0.59 / 0.58 ms)
// LM iterations speculated in round 0 (template parameter K0 of ransac_lm_kernel<true, K0>): KMAX = 3 (default) decides every
// hypothesis that ends with <= 2 accepted steps in one pass; K0 = 2 (rsdsfm_set_ransac_speculation) speculates two iterations,
// cheaper when every hypothesis stops after one accepted step (outlier-dominated costs).  Round 0 also fuses the inlier score
// {count, sum err} of ONE of the speculated iterates, the one after `fused_base` accepted steps: hypotheses that end there are
// scored without another pass over the pixels, all others by ransac_score_kernel.  DeepFlow-like data (0.3 px noise + 10 %
// outliers) ends after TWO accepted steps for ~96 % of the hypotheses, noise-free data (ground-truth flow) after THREE for all of
// them, outlier-dominated data after ONE: the base follows the context's previous solve (Ctx::ransac_fused_base).  That is a
// scheduling decision only: both kernels add a hypothesis' inlier errors in the same order (wave halves in lane order, waves,
// tiles, row groups), so the error sum has the same bits whichever of them forms it -- a solve's result never depends on the
// context's history (tests/test_gpu_frame.py::test_solve_does_not_depend_on_the_contexts_history).
constexpr int kFused = 1;
constexpr int kTiledFusedBase = 2;  // the column-tiled (multi-GPU) solve: a fixed base, every rank must take the same decisions
constexpr int NSR = NS + 2 * kFused;
constexpr int kNSum = NSR - (KMAX + 1);  // sum slots (the KMAX + 1 gradient-max slots are reduced with fmax)
constexpr int kTStride = 65;             // row stride of the per-wave transpose buffer (bank-conflict-free)

// per-K0 shape of round 0: LM slots used, fused scores, sum slots that go through the transpose
template <int K0>
struct R0Shape {
    static constexpr int NSk = 3 + 5 * K0;
    static constexpr int F = kFused;
    static constexpr int nsum = NSk - (K0 + 1) + 2 * F;
};
static_assert(R0Shape<KMAX>::nsum == kNSum, "full shape uses every sum slot");

// BIDX: the speculated iterate that is scored (fused_base - 1), -1: none.  CORE (reference arithmetic only): the error's square root
// through its in-range core, *worst tracking the range test of its argument (see lm_pixel_t)
template <int BIDX, bool CORE = false>
struct ScoreHook {
    double x, y, ux, uy, al, ak, two_over, tol;
    const Pose* pose;
    double* sc;  // {count, err} of the fused state
    uint32_t* worst = nullptr;
    __device__ __forceinline__ void operator()(int j, double rho, const PixelModel& m) const {
        if (j == BIDX) {
#if RSDSFM_FUSED
            (void)m;
            const double e = point_error(x, y, ux, uy, al, ak, *pose, two_over, rho);
#else
            const double e = CORE ? point_error_from_model_core(m, rho, *worst) : point_error_from_model(m, rho);  // bit-identical to point_error(...): device_math.hpp
#endif
            if (e < tol) {
                sc[0] += 1.0;
                sc[1] += e;
            }
        }
    }
};

struct Tile {
    double x[kRP], y[kRP], ux[kRP], uy[kRP], al[kRP], ak[kRP];
    bool ok[kRP];
};

__device__ __forceinline__ void load_tile(Tile& t, const double2* __restrict__ q, const double2* __restrict__ u,
                                          const double* __restrict__ alpha, const double* __restrict__ alpha_k,
                                          int64_t base, int64_t n) {
#pragma unroll
    for (int j = 0; j < kRP; ++j) {
        const int64_t i = base + threadIdx.x + (int64_t)j * kRB;
        t.ok[j] = i < n;
        const int64_t ii = t.ok[j] ? i : n - 1;  // out-of-range lanes load a valid point and are masked by ok[]
        const double2 qq = q[ii], uu = u[ii];
        t.x[j] = qq.x;
        t.y[j] = qq.y;
        t.ux[j] = uu.x;
        t.uy[j] = uu.y;
        t.al[j] = alpha[ii];
        t.ak[j] = alpha_k[ii];
    }
}

__device__ __forceinline__ bool is_nan_bits(double x) {  // (integer test: the pose is wave-uniform, this stays on the scalar unit)
    const uint32_t h = (uint32_t)__double2hiint(x) & 0x7FFFFFFFu;
    return h > 0x7FF00000u || (h == 0x7FF00000u && __double2loint(x) != 0);
}
__device__ __forceinline__ bool pose_has_nan(const Pose& p) {
    return ((int)is_nan_bits(p.w[0]) | (int)is_nan_bits(p.w[1]) | (int)is_nan_bits(p.w[2]) | (int)is_nan_bits(p.v[0]) | (int)is_nan_bits(p.v[1]) |
            (int)is_nan_bits(p.v[2]) | (int)is_nan_bits(p.k)) != 0;
}

// hypothesis pose from the (wave-uniform) hypothesis table; written field by field so that it stays in registers
#define RSDSFM_LOAD_POSE(pose, hyp, t)                       \
    Pose pose;                                               \
    {                                                        \
        const double* h_ = (hyp) + (int64_t)(t) * 8;          \
        pose.w[0] = h_[0], pose.w[1] = h_[1], pose.w[2] = h_[2]; \
        pose.v[0] = h_[3], pose.v[1] = h_[4], pose.v[2] = h_[5]; \
        pose.k = h_[6];                                      \
    }

}  // namespace

// ---------------------------------------------------------------------------------------------------
// per-hypothesis speculative LM sums
// ---------------------------------------------------------------------------------------------------
// round 0: every hypothesis starts from the built-in plan; round r > 0: only hypotheses whose state machine is
// still running (status 0) and expects launch r take part.  partials: [T][gridDim.x][NSR] (hypothesis-major).
// CORE (round 0 of the reference-arithmetic build, with flag words): the two square roots and the reciprocal of a pixel-hypothesis go
// through their in-range cores (device_math.hpp: 16 instructions less, the same bits for arguments in range); flags[3] is raised when
// an argument was outside -- a zero Jacobian, a zero or non-finite error -- and the host runs the RANSAC again with CORE = false.
template <bool R0, int K0, int BASE, bool CORE>
__global__ __launch_bounds__(kRB) void ransac_lm_kernel(const double2* __restrict__ q, const double2* __restrict__ u,
                                                       const double* __restrict__ alpha,
                                                       const double* __restrict__ alpha_k, int64_t n,
                                                       const double* __restrict__ hyp, int T,
                                                       const LmState* __restrict__ states,
                                                       double* __restrict__ partials, int round, double tol,
                                                       int* __restrict__ running_flag, int ntile_blocks, int ngroups,
                                                       const int* __restrict__ m9_core_flag, int m9_core_epoch,
                                                       unsigned long long* __restrict__ clk_probe, int clk_bid) {
    extern __shared__ double s_acc[];  // [T][NSR]
    // the decide kernel of this round counts the still-running hypotheses into *running_flag; it runs after this kernel
    // (stream order), so the counter is cleared here instead of by a separate memset
    if (running_flag && blockIdx.x == 0 && threadIdx.x == 0) {
        *running_flag = 0;
        // the minimal solver of this run met an SVD operand outside the range of its function cores (minimal9_kernels.hip): reported
        // through the same flag word as this kernel's own cores -- the host starts the run over with the standard functions
        if (m9_core_flag && *m9_core_flag == m9_core_epoch) running_flag[3] = 1;
    }
    // XCD-aware block -> (tile block, hypothesis group) mapping.  The hypotheses are split over `ngroups` workgroups per tile
    // block (short workgroups: a full last round of the chip), and all of them read the same pixels.  Workgroups are dispatched
    // round-robin over the 8 XCDs, each with its own L2: the FULL groups (0 .. ngroups - 2, ceil(T / ngroups) hypotheses each) of
    // one tile block are placed 8 linear ids apart (same XCD) inside one window of 8 * (ngroups - 1) consecutive ids (dispatched
    // together), so the tile comes from HBM once and the other groups hit that XCD's L2.  The LAST group holds the remainder
    // (fewer hypotheses: shorter workgroups) and is dispatched behind all windows, where it fills the tail of the launch.
    // Measured at 1280x720, T = 50 (6 groups: 5 x 9 + 5 hypotheses), same box: groups on gridDim.y 0.491 ms with 252 MB of
    // L2-miss traffic per launch (the 44 MB of inputs fetched 5.7 times); all groups in windows 0.534 ms (51 MB: the short
    // workgroups no longer fill the tail); windows with balanced groups 0.499 ms; this mapping 0.487 ms.
    const int bid = (int)blockIdx.x;
    const int tiles8 = ((ntile_blocks + 7) / 8) * 8;
    const int nfull = ngroups - 1;
    int tb, grp;
    if (nfull > 0 && bid < tiles8 * nfull) {
        const int window = 8 * nfull;
        tb = (bid / window) * 8 + (bid % 8);  // tile block: plays the role of blockIdx.x of a (tile blocks, groups) grid
        grp = (bid % window) / 8;
    } else {
        tb = bid - tiles8 * nfull;
        grp = nfull;
    }
    if (tb >= ntile_blocks) return;
    // rsdsfm_set_profiling: one lane of one workgroup in the middle of the launch (clk_bid; -1 = off) stamps the shader clock counter
    // and the 100 MHz counter at both ends of its life -- the clock this kernel actually runs at (the chip lowers it under fp64 load)
    if (bid == clk_bid && threadIdx.x == 0) {
        clk_probe[0] = __builtin_amdgcn_s_memtime();
        clk_probe[1] = __builtin_amdgcn_s_memrealtime();
    }
    __shared__ LmPlanLds plan;
    __shared__ double s_red[2][kRB / 64][NSR];
    __shared__ double s_T[kRB / 64][kNSum * kTStride];  // per-wave transpose buffer of the sum slots
    __shared__ double s_half[kRB / 64][2][kNSum];
    __shared__ int s_active;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    // hypothesis groups: with one tile per workgroup and 2 workgroups resident per CU, 900 tiles (1280x720) are 1.76 rounds of the
    // chip; splitting the hypotheses over `ngroups` workgroups per tile makes them short enough that the last round is full
    const int per_group = (T + ngroups - 1) / ngroups;
    const int t_begin = grp * per_group, t_end = min(T, t_begin + per_group);
    using Shape = R0Shape<R0 ? K0 : KMAX>;
    constexpr int NSk = Shape::NSk, F = Shape::F, nsum = Shape::nsum;
    for (int i = tid; i < T * NSR; i += kRB) s_acc[i] = 0.0;
    for (int i = tid; i < 2 * (kRB / 64) * NSR; i += kRB) (&s_red[0][0][0])[i] = 0.0;  // slots a shape does not use stay zero
    if (R0 && tid == 0) {
        plan.n_hist = 0;
        plan.K = K0;
        plan.write_which = 0;
        double r = kInitialRadius;
        for (int j = 0; j < KMAX; ++j) {
            plan.inv_cand[j] = 1.0 / r;
            r = radius_accept(r, 1.0);
        }
    }
    __syncthreads();
    LmPlanFirstK<K0> pf;  // round 0: plan shape known at compile time (K0 speculated iterations, nothing accepted yet)
    pf.write_which = 0;
#pragma unroll
    for (int j = 0; j < KMAX; ++j) pf.inv_cand[j] = R0 ? plan.inv_cand[j] : 0.0;

    const int64_t tile_pixels = (int64_t)kRB * kRP;
    const int64_t ntiles = (n + tile_pixels - 1) / tile_pixels;
    uint32_t worst_key = 0;  // CORE: the range tests of this thread's function-core arguments (sqrt_range_track)
    for (int64_t tile = tb; tile < ntiles; tile += ntile_blocks) {
        Tile px;
        load_tile(px, q, u, alpha, alpha_k, tile * tile_pixels, n);
        const bool full_tile = (tile + 1) * tile_pixels <= n;
        for (int t = t_begin; t < t_end; ++t) {
            if (!R0) {
                __syncthreads();  // previous hypothesis is done with `plan`
                if (tid == 0) {
                    const LmState& st = states[t];
                    const int act = (st.status == 0 && st.next_launch == round) ? 1 : 0;
                    s_active = act;
                    if (act) {
                        plan.n_hist = st.n_hist;
                        plan.K = st.K;
                        plan.write_which = 0;
                    }
                }
                __syncthreads();
                if (!s_active) continue;
                if (tid < kMaxIter) plan.inv_hist[tid] = 1.0 / states[t].hist[tid];
                if (tid < KMAX) plan.inv_cand[tid] = 1.0 / states[t].cand[tid];
                __syncthreads();
            }
            RSDSFM_LOAD_POSE(pose, hyp, t)
            const double two_over = 2.0 / (2.0 + pose.k);
            double acc[NS];
#pragma unroll
            for (int s = 0; s < NS; ++s) acc[s] = 0.0;
            double sc[2 * kFused];
#pragma unroll
            for (int s = 0; s < 2 * kFused; ++s) sc[s] = 0.0;
            static_assert(!CORE || (R0 && !RSDSFM_FUSED), "the cores stand for the reference arithmetic's functions in round 0");
            if (R0 && full_tile) {  // round 0, no ragged lanes: one straight-line block, the kRP pixel chains interleave
                uint32_t wk = 0;
#pragma unroll
                for (int j = 0; j < kRP; ++j) {
                    const ScoreHook<(R0 ? BASE - 1 : -1), CORE> hook{px.x[j], px.y[j], px.ux[j], px.uy[j], px.al[j], px.ak[j], two_over, tol, &pose, sc, &wk};
                    (void)lm_pixel_t<CORE>(px.x[j], px.y[j], px.ux[j], px.uy[j], px.al[j], px.ak[j], pose, two_over, pf, acc, hook, nullptr, &wk);
                }
                // A hypothesis with a NaN in its pose (a degenerate sample: one trial in a few hundred) does not count: a NaN argument
                // goes through the same instructions in the cores and in the standard functions, and with a NaN pose component every
                // sum that depends on the Jacobi scaling is NaN either way (the residual is) -- such a hypothesis must not cost a restart.
                if (CORE && !pose_has_nan(pose)) worst_key = max(worst_key, wk);
            } else {
#pragma unroll
                for (int j = 0; j < kRP; ++j)
                    if (px.ok[j]) {
                        const ScoreHook<(R0 ? BASE - 1 : -1)> hook{px.x[j], px.y[j], px.ux[j], px.uy[j], px.al[j], px.ak[j], two_over, tol, &pose, sc};
                        (void)lm_pixel(px.x[j], px.y[j], px.ux[j], px.uy[j], px.al[j], px.ak[j], pose, two_over, plan, acc, hook);
                    }
            }
            double(*red)[NSR] = s_red[t & 1];
            // ---- wave reduction of the NSR values ----
            // The 4 max slots go through DPP butterflies.  The 18 SUM slots are transposed through LDS instead of 18 x 6
            // DPP steps: every lane stores its 18 values (conflict-free, row stride 65), then lane (slot, half) adds the 32
            // values of its half in lane order and the two halves are added -- a fixed order, 18 stores + 32 loads + 32
            // adds per lane instead of ~320 DPP/VALU instructions per hypothesis.  A wave's LDS operations execute in
            // order, so no barrier is needed inside the wave.
            // The max slots: max |J r| is only ever compared with the gradient tolerance (lm_advance: `<= kGradientTol`), and "the maximum
            // is at most tol" is "no value is above tol" -- so what travels from here on is 1.0 if a pixel of this wave is above the
            // tolerance and 0.0 otherwise (one compare and a ballot per slot instead of a six-step DPP butterfly of 64-bit maxima:
            // ~68 instructions per hypothesis and thread); every later stage combines the slots with fmax as before and the test decides
            // exactly as it did on the maxima (NaNs never entered them: fmax drops a NaN operand, and so does the comparison).
#pragma unroll
            for (int s = 0; s < NSk; ++s)
                if (is_max_slot(s)) {
                    const bool above = __builtin_amdgcn_ballot_w64(acc[s] > kGradientTol) != 0;  // (wave-uniform)
                    if (lane == 0) red[wv][s] = above ? 1.0 : 0.0;
                }
            double* Tw = s_T[wv];
            {
                int kk = 0;
#pragma unroll
                for (int s = 0; s < NSk; ++s)
                    if (!is_max_slot(s)) {
                        Tw[kk * kTStride + lane] = acc[s];
                        ++kk;
                    }
#pragma unroll
                for (int s = 0; s < 2 * F; ++s) Tw[(nsum - 2 * F + s) * kTStride + lane] = sc[s];
            }
            __builtin_amdgcn_wave_barrier();  // compiler ordering only: the wave's stores precede its loads below
            {
                const int sl = lane & 31, half = lane >> 5;
                if (sl < nsum) {
                    const double* row = Tw + sl * kTStride + half * 32;
                    double part = row[0];
#pragma unroll
                    for (int j = 1; j < 32; ++j) part += row[j];
                    s_half[wv][half][sl] = part;
                }
                __builtin_amdgcn_wave_barrier();
                if (lane < nsum) {
                    // slot id of sum index `lane`: the used sum slots in increasing slot order, then the fused scores
                    int slot = NS + (lane - (nsum - 2 * F));
                    int kk = 0;
#pragma unroll
                    for (int s = 0; s < NSk; ++s)
                        if (!is_max_slot(s)) {
                            if (kk == lane) slot = s;
                            ++kk;
                        }
                    red[wv][slot] = s_half[wv][0][lane] + s_half[wv][1][lane];
                }
            }
            __syncthreads();
            if (tid < NSR) {
                double r = red[0][tid];
                for (int w2 = 1; w2 < kRB / 64; ++w2) r = is_max_slot(tid) ? fmax(r, red[w2][tid]) : r + red[w2][tid];
                double& a = s_acc[t * NSR + tid];
                a = is_max_slot(tid) ? fmax(a, r) : a + r;
            }
        }
        __syncthreads();
    }
    __syncthreads();
    if (CORE && worst_key >= kSqrtRangeKeys) running_flag[3] = 1;  // (benign race: every writer stores the same word)
    // hypothesis-major rows [T][gridDim.x][NSR]: the per-hypothesis reduction that follows reads one contiguous block
    for (int i = t_begin * NSR + tid; i < t_end * NSR; i += kRB) {
        const int t = i / NSR, sl = i - t * NSR;
        partials[((int64_t)t * ntile_blocks + tb) * NSR + sl] = s_acc[i];
    }
    if (bid == clk_bid && tid == 0) {
        clk_probe[2] = __builtin_amdgcn_s_memtime();
        clk_probe[3] = __builtin_amdgcn_s_memrealtime();
    }
}

// fixed-order reduction of the rows of hypothesis t into s_sums[NSR] (256 threads).  hyp_major: rows are
// partials[T][nblocks][NSR] (what ransac_lm_kernel writes: contiguous per hypothesis); otherwise [nblocks][T][NSR] (the gathered
// per-rank rows of the column-tiled solve, ranks in place of workgroups).  Thread (group g, slot pair sp) adds the rows g, g + G,
// ... of its two slots in order with 16 independent 16-byte loads in flight (the reduction is bound by load latency), then thread
// s adds the G group sums of slot s in order -- no wave butterflies (the first version reduced 22 per-thread sums with 22 DPP
// butterflies: 12.1 -> 9.6 (hypothesis-major rows) -> see DESIGN for this version).
// rank_stride2 (rows of the column-tiled solve only): distance between two ranks' rows in 16-byte units; T * NSR / 2 when 0 (the rows follow
// each other), one more when every rank's rows are followed by a trailer (ransac_lm_rows_kernel)
__device__ __forceinline__ void reduce_hyp_sums(const double* __restrict__ partials, int nblocks, int T, int t, bool hyp_major,
                                                double (*s_red)[NSR], double* s_sums, int rank_stride2 = 0) {
    static_assert(NSR % 2 == 0, "slot pairs are read as double2");
    constexpr int NH = NSR / 2, G = 256 / NH, U = 16;
    __shared__ double s_grp[G][NSR];
    const int tid = threadIdx.x;
    const int g = tid / NH, sp = tid - g * NH;
    const double2* __restrict__ p2 = reinterpret_cast<const double2*>(partials);
    if (g < G) {
        const bool mx0 = is_max_slot(2 * sp), mx1 = is_max_slot(2 * sp + 1);
        double a0 = 0.0, a1 = 0.0;
        for (int b = g; b < nblocks; b += U * G) {
            double2 v[U];
#pragma unroll
            for (int j = 0; j < U; ++j) {  // rows past the end contribute the identity (sums: + 0.0; max slots hold absolute values)
                const int bj = b + j * G;
                const int64_t br = bj < nblocks ? bj : g;
                const double2 x = p2[hyp_major ? ((int64_t)t * nblocks + br) * NH + sp : br * (int64_t)(rank_stride2 ? rank_stride2 : T * NH) + (int64_t)t * NH + sp];
                v[j] = bj < nblocks ? x : make_double2(0.0, 0.0);
            }
#pragma unroll
            for (int j = 0; j < U; ++j) {
                a0 = mx0 ? fmax(a0, v[j].x) : a0 + v[j].x;
                a1 = mx1 ? fmax(a1, v[j].y) : a1 + v[j].y;
            }
        }
        s_grp[g][2 * sp] = a0;
        s_grp[g][2 * sp + 1] = a1;
    }
    __syncthreads();
    if (tid < NSR) {
        double r = s_grp[0][tid];
#pragma unroll
        for (int g2 = 1; g2 < G; ++g2) r = is_max_slot(tid) ? fmax(r, s_grp[g2][tid]) : r + s_grp[g2][tid];
        s_sums[tid] = r;
    }
    __syncthreads();
    (void)s_red;
}

// row-tiled solve: the shard's partials of every hypothesis reduced to one row [T][NSR] (the all-gather payload;
// the gathered [ranks][T][NSR] array is then what ransac_decide_kernel reduces, ranks in place of workgroups)
// core_flags (may be null): the flag words of a launch that ran the in-range function cores (ransac_lm_kernel CORE); the rows are then
// followed by a 16-byte trailer whose first double is 1.0 when this shard's launch met an argument out of range -- the flag travels
// with the rows, the decide stage raises it on every rank and all ranks start over together
__global__ __launch_bounds__(256) void ransac_lm_rows_kernel(const double* __restrict__ partials, int nblocks, int T,
                                                            const LmState* __restrict__ states, int round,
                                                            double* __restrict__ rows, const int* __restrict__ core_flags) {
    __shared__ double s_red[4][NSR];
    __shared__ double s_sums[NSR];
    const int t = blockIdx.x;
    if (core_flags && t == 0 && threadIdx.x < 2) rows[(int64_t)T * NSR + threadIdx.x] = (threadIdx.x == 0 && core_flags[3] != 0) ? 1.0 : 0.0;
    const LmState* state = states + t;
    if (round > 0 && (state->status != 0 || state->next_launch != round)) {
        if (threadIdx.x < NSR) rows[(int64_t)t * NSR + threadIdx.x] = 0.0;
        return;
    }
    reduce_hyp_sums(partials, nblocks, T, t, true, s_red, s_sums);
    if (threadIdx.x < NSR) rows[(int64_t)t * NSR + threadIdx.x] = s_sums[threadIdx.x];
}

// one workgroup per hypothesis
// flags[0]: hypotheses still running after this round; flags[1]: hypotheses that finished WITHOUT a fused score
// (accepted-step count != 1) and need the separate score pass.  scored[t] = 1 when trial_count/err were filled here.
// pred_flag (may be null): counts the hypotheses that round 0 did not finish with at most one accepted step -- the predictor input
// of Ctx::ransac_k0.  k0: LM iterations round 0 speculated (its fused scores: kFused for k0 = KMAX, one otherwise).
__global__ __launch_bounds__(256) void ransac_decide_kernel(const double* __restrict__ partials, int nblocks, int T, int hyp_major,
                                                           LmState* states, int64_t n, int round, int k0, int* flags, int* pred_flag,
                                                           int* __restrict__ scored, double* __restrict__ trial_count,
                                                           double* __restrict__ trial_err, int fused_base, int* steps_hist,
                                                           int* __restrict__ unscored_list, int rank_stride2 = 0,
                                                           double* __restrict__ cnt_rt = nullptr, int cnt_stride = 0) {
    __shared__ double s_red[4][NSR];
    __shared__ double s_sums[NSR];
    __shared__ int s_fused_score;
    const int t = blockIdx.x;
    const int tid = threadIdx.x;
    LmState* state = states + t;
    if (tid == 0) s_fused_score = 0;
    if (rank_stride2 && t == 0 && tid < nblocks) {  // the ranks' trailers (ransac_lm_rows_kernel): some shard left the range of the function cores
        if (partials[(int64_t)tid * rank_stride2 * 2 + (int64_t)T * NSR] != 0.0) flags[3] = 1;  // (benign race: every writer stores the same word)
    }
    if (round > 0 && (state->status != 0 || state->next_launch != round)) return;
    reduce_hyp_sums(partials, nblocks, T, t, hyp_major != 0, s_red, s_sums, rank_stride2);
    if (tid == 0) {
        LmScal st = *static_cast<const LmScal*>(state);
        // (a state without launches in a later round: a hypothesis the analytic pass handed over -- ransac_lma_kernels.hip lma_publish --, whose
        // first iterate-by-iterate launch this is)
        const bool first = round == 0 || st.launches == 0;
        const int used_K = (round == 0) ? k0 : st.K;
        lm_advance(st, state->hist, s_sums, n, first, used_K, 0, round);
        if (pred_flag && round == 0 && (st.status == 0 || st.n_hist >= 2)) atomicAdd(pred_flag, 1);
        // where the hypotheses of this solve end (1, 2, >= 3 accepted steps or still running after round 0): the next solve's fused_base
        if (steps_hist && round == 0) atomicAdd(&steps_hist[st.status == 0 ? 3 : (st.n_hist < 1 ? 0 : (st.n_hist > 3 ? 3 : st.n_hist))], 1);
        if (st.status == 0) {
            atomicAdd(&flags[0], 1);
        } else if (round == 0 && st.n_hist == fused_base) {  // the fused iterate is the final one
            trial_count[t] = s_sums[NS];
            trial_err[t] = s_sums[NS + 1];
            scored[t] = 1;
            s_fused_score = 1;
        } else {
            // finished where round 0 did not fuse the score: counted, and (where the caller keeps a list) appended for the scoring pass,
            // which then only looks at these (the order of the entries is the order of arrival: it decides which workgroup scores which
            // hypothesis, not what the sums are)
            const int pos = atomicAdd(&flags[1], 1);
            if (unscored_list) unscored_list[pos] = t;
        }
        *static_cast<LmScal*>(state) = st;
    }
    // column-tiled solve: the ranks' shares of this hypothesis' inlier count (cnt_rt[rank][hypothesis]: exact integers in doubles).  The winner's
    // shares ARE the slabs' inlier counts -- what the compaction will find -- so the counts need no exchange of their own (ransac_pick_kernel).
    if (cnt_rt) {
        __syncthreads();
        if (s_fused_score && tid < nblocks)
            cnt_rt[(int64_t)tid * cnt_stride + t] = partials[((int64_t)tid * (rank_stride2 ? rank_stride2 : T * (NSR / 2)) + (int64_t)t * (NSR / 2)) * 2 + NS];
    }
}

// ---------------------------------------------------------------------------------------------------
// scoring
// ---------------------------------------------------------------------------------------------------
// rho of one pixel under hypothesis state `st` (accepted LM steps replayed) or closed form
__device__ __forceinline__ double hyp_rho(double x, double y, double ux, double uy, double al, double ak, const Pose& pose,
                                          double two_over, int depth_mode, const LmPlanLds& plan) {
    if (depth_mode == RSDSFM_DEPTH_CLOSED_FORM) return closed_form_rho(x, y, ux, uy, al, ak, pose, two_over);
    double dummy[NS];
    return lm_pixel(x, y, ux, uy, al, ak, pose, two_over, plan, dummy);  // plan.K == 0: replay only
}

// the error of one pixel at rho: the reference build takes it from the pixel model the replay built (bit-identical, device_math.hpp)
__device__ __forceinline__ double score_error(const PixelModel& m, double x, double y, double ux, double uy, double al, double ak,
                                              const Pose& pose, double two_over, double rho) {
#if RSDSFM_FUSED
    (void)m;
    return point_error(x, y, ux, uy, al, ak, pose, two_over, rho);
#else
    (void)x, (void)y, (void)ux, (void)uy, (void)al, (void)ak, (void)pose, (void)two_over;
    return point_error_from_model(m, rho);
#endif
}

// the kRP pixels of a lane under one hypothesis with NH accepted steps (compile time): one straight-line block
template <int NH>
__device__ __forceinline__ void score_pixels(const Tile& px, bool full_tile, const Pose& pose, double two_over, const LmState& st, double tol,
                                             double& cnt, double& es) {
    LmPlanReplay<NH> plan;
#pragma unroll
    for (int h = 0; h < NH; ++h) plan.ih[h] = 1.0 / st.hist[h];  // (uniform: scalar loads, one division per accepted step)
    double dummy[NS];
#pragma unroll
    for (int j = 0; j < kRP; ++j) {
        if (full_tile || px.ok[j]) {
            PixelModel m;
            const double rho = lm_pixel(px.x[j], px.y[j], px.ux[j], px.uy[j], px.al[j], px.ak[j], pose, two_over, plan, dummy, NoHook(), &m);
            const double err = score_error(m, px.x[j], px.y[j], px.ux[j], px.uy[j], px.al[j], px.ak[j], pose, two_over, rho);
            if (err < tol) {
                cnt += 1.0;
                es += err;
            }
        }
    }
}

// XCD-aware (tile block, hypothesis group) of a linear workgroup id: see ransac_lm_kernel
__device__ __forceinline__ void tile_group_of_block(int bid, int ntile_blocks, int ngroups, int& tb, int& grp) {
    const int tiles8 = ((ntile_blocks + 7) / 8) * 8;
    const int nfull = ngroups - 1;
    if (nfull > 0 && bid < tiles8 * nfull) {
        const int window = 8 * nfull;
        tb = (bid / window) * 8 + (bid % 8);
        grp = (bid % window) / 8;
    } else {
        tb = bid - tiles8 * nfull;
        grp = nfull;
    }
}

// score partials: [ntile_blocks][T][2] = {count, sum of inlier errors}.  Like ransac_lm_kernel the hypotheses of a tile are split
// over `ngroups` short workgroups (one long workgroup per tile left the second round of the chip mostly empty: 600 tiles on 512
// resident slots), and hypotheses with up to four accepted steps -- all of them on ordinary data -- run a replay whose depth is a
// template parameter: no plan in LDS, no workgroup barriers around it, the six pixel chains of a lane interleaved.  The per-lane,
// per-wave and per-workgroup order of the sums is unchanged.
__global__ __launch_bounds__(kRB) void ransac_score_kernel(const double2* __restrict__ q, const double2* __restrict__ u,
                                                          const double* __restrict__ alpha,
                                                          const double* __restrict__ alpha_k, int64_t n,
                                                          const double* __restrict__ hyp, int T,
                                                          const LmState* __restrict__ states, int depth_mode, double tol,
                                                          const int* __restrict__ scored, double* __restrict__ partials, int ntile_blocks,
                                                          int ngroups, const int* __restrict__ list, const int* __restrict__ list_count) {
    extern __shared__ double s_acc[];  // [T][2]
    __shared__ LmPlanLds plan;
    __shared__ double s_red[2][kRB / 64][2];
    __shared__ double s_T2[kRB / 64][2 * kTStride];
    __shared__ double s_half2[kRB / 64][2][2];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    int tb, grp;
    tile_group_of_block((int)blockIdx.x, ntile_blocks, ngroups, tb, grp);
    if (tb >= ntile_blocks) return;
    // The hypotheses of this workgroup: a slice of the hypothesis range (and of those the ones not scored yet), or -- with `list`, the
    // indices ransac_decide_kernel appended for the hypotheses that finished without a fused score, *list_count of them -- a slice of
    // that list: the few hypotheses left over by round 0 (~2 of 50 on DeepFlow-like data) then share their tiles in ONE group of
    // workgroups instead of one group each, and the other groups leave before touching a tile.  Which workgroup scores a hypothesis
    // never changes its sums (per hypothesis: lanes, waves, tile blocks in order).
    int t_begin, t_end;
    if (list) {
        const int U = min(*list_count, T);
        const int per = max(2, (U + ngroups - 1) / ngroups);
        t_begin = grp * per, t_end = min(U, t_begin + per);  // (positions in the list)
        if (t_begin >= t_end) return;
    } else {
        const int per_group = (T + ngroups - 1) / ngroups;
        t_begin = grp * per_group, t_end = min(T, t_begin + per_group);
        if (scored) {  // nothing to score in this group (the usual case when round 0 fused the right state): leave before touching a tile
            bool any = false;
            for (int t = t_begin; t < t_end; ++t) any = any || !scored[t];
            if (!any) return;
        }
    }
    for (int i = tid; i < T * 2; i += kRB) s_acc[i] = 0.0;
    __syncthreads();
    const int64_t tile_pixels = (int64_t)kRB * kRP;
    const int64_t ntiles = (n + tile_pixels - 1) / tile_pixels;
    for (int64_t tile = tb; tile < ntiles; tile += ntile_blocks) {
        Tile px;
        load_tile(px, q, u, alpha, alpha_k, tile * tile_pixels, n);
        const bool full_tile = (tile + 1) * tile_pixels <= n;
        int done = 0;  // hypotheses this workgroup has scored on this tile (selects the reduction's double buffer)
        for (int ti = t_begin; ti < t_end; ++ti) {
            const int t = list ? list[ti] : ti;
            if (!list && scored && scored[t]) continue;  // already scored by the fused LM pass (uniform branch)
            RSDSFM_LOAD_POSE(pose, hyp, t)
            const double two_over = 2.0 / (2.0 + pose.k);
            double cnt = 0.0, es = 0.0;
            const int nh = depth_mode == RSDSFM_DEPTH_CERES_LM ? states[t].n_hist : -1;  // uniform
            if (nh == 0) score_pixels<0>(px, full_tile, pose, two_over, states[t], tol, cnt, es);
            else if (nh == 1) score_pixels<1>(px, full_tile, pose, two_over, states[t], tol, cnt, es);
            else if (nh == 2) score_pixels<2>(px, full_tile, pose, two_over, states[t], tol, cnt, es);
            else if (nh == 3) score_pixels<3>(px, full_tile, pose, two_over, states[t], tol, cnt, es);
            else if (nh == 4) score_pixels<4>(px, full_tile, pose, two_over, states[t], tol, cnt, es);
            else {  // closed form, or a long trajectory: the plan goes through LDS
                if (depth_mode == RSDSFM_DEPTH_CERES_LM) {
                    __syncthreads();
                    if (tid == 0) {
                        plan.n_hist = states[t].n_hist;
                        plan.K = 0;
                        plan.write_which = 0;
                    }
                    if (tid < kMaxIter) plan.inv_hist[tid] = 1.0 / states[t].hist[tid];
                    __syncthreads();
                }
#pragma unroll
                for (int j = 0; j < kRP; ++j) {
                    if (px.ok[j]) {
                        const double rho = hyp_rho(px.x[j], px.y[j], px.ux[j], px.uy[j], px.al[j], px.ak[j], pose, two_over, depth_mode, plan);
                        const double err = point_error(px.x[j], px.y[j], px.ux[j], px.uy[j], px.al[j], px.ak[j], pose, two_over, rho);
                        if (err < tol) {
                            cnt += 1.0;
                            es += err;
                        }
                    }
                }
            }
            double(*red)[2] = s_red[done & 1];
            ++done;
            // the wave's two sums in the order ransac_lm_kernel adds its fused scores (transposed through LDS: each half of the wave
            // in lane order, then the halves) -- the inlier-error sum of a hypothesis has the same bits whichever kernel forms it
            {
                double* Tw = s_T2[wv];
                Tw[lane] = cnt;
                Tw[kTStride + lane] = es;
                __builtin_amdgcn_wave_barrier();
                const int sl = lane & 31, half = lane >> 5;
                if (sl < 2) {
                    const double* row = Tw + sl * kTStride + half * 32;
                    double part = row[0];
#pragma unroll
                    for (int j = 1; j < 32; ++j) part += row[j];
                    s_half2[wv][half][sl] = part;
                }
                __builtin_amdgcn_wave_barrier();
                if (lane < 2) red[wv][lane] = s_half2[wv][0][lane] + s_half2[wv][1][lane];
                __builtin_amdgcn_wave_barrier();
            }
            __syncthreads();
            if (tid < 2) {
                double r = red[0][tid];
                for (int w2 = 1; w2 < kRB / 64; ++w2) r += red[w2][tid];
                s_acc[t * 2 + tid] += r;
            }
        }
        __syncthreads();
    }
    __syncthreads();
    double* out = partials + (int64_t)tb * T * 2;
    if (list) {
        for (int i = t_begin * 2 + tid; i < t_end * 2; i += kRB) {
            const int t = list[i >> 1];
            out[t * 2 + (i & 1)] = s_acc[t * 2 + (i & 1)];
        }
    } else {
        for (int i = t_begin * 2 + tid; i < t_end * 2; i += kRB) out[i] = s_acc[i];
    }
}

// the two score sums of hypothesis t over the tile blocks in the order reduce_hyp_sums (ransac_decide_kernel) adds a slot of the
// fused rows: row group g adds the rows g, g + G, g + 2 G, ... in order, then the G group sums are added in order
__device__ __forceinline__ void reduce_score_rows(const double* __restrict__ partials, int nblocks, int T, int t, double& c_out, double& e_out) {
    constexpr int G = 256 / (NSR / 2);
    __shared__ double s_grp[G][2];
    const int tid = threadIdx.x;
    if (tid < G) {
        // (16 independent 16-byte loads in flight, then the ordered additions: the loop is bound by load latency, like reduce_hyp_sums)
        constexpr int U = 16;
        const double2* __restrict__ p2 = reinterpret_cast<const double2*>(partials);
        double c = 0.0, e = 0.0;
        for (int b = tid; b < nblocks; b += U * G) {
            double2 v[U];
#pragma unroll
            for (int j = 0; j < U; ++j) {
                const int bj = b + j * G;
                const double2 x = p2[(int64_t)(bj < nblocks ? bj : tid) * T + t];
                v[j] = bj < nblocks ? x : make_double2(0.0, 0.0);  // (rows past the end add +0.0 to a non-negative sum: the identity)
            }
#pragma unroll
            for (int j = 0; j < U; ++j) {
                c += v[j].x;
                e += v[j].y;
            }
        }
        s_grp[tid][0] = c;
        s_grp[tid][1] = e;
    }
    __syncthreads();
    double c = s_grp[0][0], e = s_grp[0][1];
#pragma unroll
    for (int g2 = 1; g2 < G; ++g2) c += s_grp[g2][0], e += s_grp[g2][1];
    c_out = c, e_out = e;
}

// fixed-order reduction of one hypothesis batch's score partials into trial_count / trial_err (already offset):
// one workgroup per hypothesis, threads stride over the pixel workgroups (independent loads in flight), then a
// DPP wave reduction and the 4 waves in order
__global__ __launch_bounds__(256) void ransac_reduce_scores_kernel(const double* __restrict__ partials, int nblocks, int T,
                                                                  const int* __restrict__ scored,
                                                                  double* __restrict__ trial_count,
                                                                  double* __restrict__ trial_err, double* __restrict__ cnt_rt = nullptr,
                                                                  int cnt_stride = 0) {
    const int t = blockIdx.x, tid = threadIdx.x;
    if (scored && scored[t]) return;
    double c, e;
    reduce_score_rows(partials, nblocks, T, t, c, e);
    if (tid == 0) {
        trial_count[t] = c;
        trial_err[t] = e;
    }
    if (cnt_rt && tid < nblocks) cnt_rt[(int64_t)tid * cnt_stride + t] = partials[((int64_t)tid * T + t) * 2];  // (the ranks' shares: see ransac_decide_kernel)
}

// row-tiled solve: the shard's score partials reduced to rows[T][2] (zeros for hypotheses already scored)
__global__ __launch_bounds__(256) void ransac_score_rows_kernel(const double* __restrict__ partials, int nblocks, int T,
                                                               const int* __restrict__ scored, double* __restrict__ rows) {
    __shared__ double s_red[4][2];
    const int t = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (scored && scored[t]) {
        if (tid < 2) rows[(int64_t)t * 2 + tid] = 0.0;
        return;
    }
    double c = 0.0, e = 0.0;
    for (int b = tid; b < nblocks; b += 256) {
        c += partials[((int64_t)b * T + t) * 2 + 0];
        e += partials[((int64_t)b * T + t) * 2 + 1];
    }
    c = wave_sum(c);
    e = wave_sum(e);
    if (lane == 0) {
        s_red[wv][0] = c;
        s_red[wv][1] = e;
    }
    __syncthreads();
    if (tid == 0) {
        rows[(int64_t)t * 2 + 0] = ((s_red[0][0] + s_red[1][0]) + s_red[2][0]) + s_red[3][0];
        rows[(int64_t)t * 2 + 1] = ((s_red[0][1] + s_red[1][1]) + s_red[2][1]) + s_red[3][1];
    }
}

// the reference's best-trial rule (minimal.cc:278-285) over all trials in order: strictly more inliers, or
// equally many with a strictly smaller error sum; the earlier trial wins ties.
// (one wave: the trial scores are fetched 64 at a time by the lanes, then walked in trial order through readlane -- the
// sequential rule is kept literally, only the 2 T dependent global loads of a single-thread loop are gone)
// the reference's sequential best-trial rule (minimal.cc:278-285) evaluated by one wave: every lane walks the trials in order
__device__ __forceinline__ void pick_best_trial(const double* __restrict__ trial_count, const double* __restrict__ trial_err, int T,
                                                int lane, int& bi, double& best_count, double& best_err) {
    best_count = -1.0, best_err = 0.0;
    bi = -1;
    for (int t0 = 0; t0 < T; t0 += 64) {
        const int t = t0 + lane;
        const double cl = t < T ? trial_count[t] : 0.0, el = t < T ? trial_err[t] : 0.0;
        const int nt = T - t0 < 64 ? T - t0 : 64;
        for (int j = 0; j < nt; ++j) {
            const double c = lane_value(cl, j), e = lane_value(el, j);  // (v_readlane: j is uniform; __shfl went through four ds_bpermute per trial)
            if (c > best_count || (c == best_count && e < best_err)) {
                best_count = c;
                best_err = e;
                bi = t0 + j;
            }
        }
    }
}

// best_host (optional): host-mapped pinned copy of the result, written by the kernel itself (no copy kernel behind the stage)
__global__ __launch_bounds__(64) void ransac_pick_kernel(const double* __restrict__ trial_count, const double* __restrict__ trial_err, int T,
                                                        const double* __restrict__ hyp, RansacBest* best, RansacBest* best_host,
                                                        const int* __restrict__ flags, int* __restrict__ flags_host, int scored_ahead,
                                                        const double* __restrict__ cnt_rt = nullptr, int cnt_stride = 0, int nranks = 0,
                                                        int64_t* __restrict__ m_all = nullptr, double tie_margin = 0.0, PickLazy lazy = PickLazy()) {
    if (blockIdx.x != 0) return;
    const int lane = threadIdx.x;
    // With flags this pick runs on speculation (behind round 0, before the host has seen them).  If hypotheses are still running, or
    // ended where round 0 did not score them and no scoring pass was enqueued ahead, the trial scores are incomplete -- the host's own
    // test in ransac_advance -- and the final stage and the caller's work behind it (the refinement) leave at once instead of running
    // on a winner that does not count: ~250 us of kernels the host would otherwise wait for before it can enqueue the next LM round.
    // (flags[3]: round 0 met an argument outside the range of its in-range function cores -- ransac_lm_kernel's CORE --: the host runs the
    // whole RANSAC again)
    int undecided = (flags && (flags[0] != 0 || (flags[1] > 0 && !scored_ahead) || flags[3] != 0)) ? 1 : 0;
    double best_count, best_err;
    int bi;
    pick_best_trial(trial_count, trial_err, T, lane, bi, best_count, best_err);
    // analytic LM trajectory, guard (d) (lma_common.hpp): minimal.cc:278-285 breaks a tie in the inlier count by the SUM of the inlier errors.
    // Where another trial has the winner's count and an error sum within tie_margin x count of the winner's (noise-free data: every good
    // hypothesis explains every pixel and the sums are rounding noise) the reference's winner is decided by the last bits of ITS arithmetic:
    // the result does not count and the host runs the RANSAC again on the iterate-by-iterate kernels.
    // (tie_margin < 0: only reported in RansacBest::lma_tie -- an iterate-by-iterate run telling the context what kind of data it is on)
    // (only where every trial has its score: with flags that say otherwise the scores of some trials are whatever the buffers hold)
    int tie = 0;
    if (tie_margin != 0.0 && bi >= 0 && best_count > 0.0 && !(flags && (flags[0] != 0 || (flags[1] > 0 && !scored_ahead)))) {
        const double tm = fabs(tie_margin);
        for (int t0 = 0; t0 < T; t0 += 64) {
            const int t = t0 + lane;
            const bool hit = t < T && t != bi && trial_count[t] == best_count && fabs(trial_err[t] - best_err) <= tm * best_count;
            if (__builtin_amdgcn_ballot_w64(hit) != 0) tie = 1;
        }
    }
    if (tie_margin > 0.0) undecided |= tie;
    // Lazy error sums (the analytic pixel pass in its count-only form, ransac_lma_kernel ERR = false): minimal.cc:278-285 reads a trial's error
    // sum only to break a tie in the inlier count, so the winner is the earliest trial with the smallest error sum among the trials S that share
    // the best count -- whatever the sums of the others.  One member: it has won.  More: every member needs its error sum in the REFERENCE's
    // arithmetic (the tie is broken as the reference breaks it, down to the last bit); the members that only have a count (scored == 2) go to the
    // scoring pass's list, the result is undecided, and the host runs ransac_score_kernel on the list and this kernel again.
    int pending = 0, list_n = 0, ns = 0;
    if (bi >= 0 && best_count > 0.0 && !(flags && (flags[0] != 0 || (flags[1] > 0 && !scored_ahead) || flags[3] != 0))) {
        int nco = 0;
        for (int t0 = 0; t0 < T; t0 += 64) {
            const int t = t0 + lane;
            const bool hit = t < T && trial_count[t] == best_count;
            ns += __builtin_popcountll(__builtin_amdgcn_ballot_w64(hit));
            nco += __builtin_popcountll(__builtin_amdgcn_ballot_w64(hit && lazy.scored && lazy.scored[t] == 2));
        }
        if (ns > 1 && nco > 0) {
            int base = 0;
            for (int t0 = 0; t0 < T; t0 += 64) {
                const int t = t0 + lane;
                const bool co = t < T && trial_count[t] == best_count && lazy.scored[t] == 2;
                const unsigned long long mk = __builtin_amdgcn_ballot_w64(co);
                if (co) {
                    lazy.list[base + __builtin_popcountll(mk & ((1ull << lane) - 1ull))] = t;
                    lazy.scored[t] = 0;  // (ransac_score_kernel / ransac_reduce_scores_kernel: not scored yet)
                }
                base += __builtin_popcountll(mk);
            }
            if (lane == 0) *lazy.list_count = nco;
            pending = 1;
            list_n = nco;
            undecided = 1;
        }
    }
    // the round's flag words (final since ransac_decide_kernel) travel to host-mapped memory with this launch: no copy behind the stage
    // ([3] bit 2: the tie; [1]: the scoring pass's list, which the lazy error sums above may have written)
    if (flags_host && lane < 8) flags_host[lane] = (lane == 1 && pending) ? list_n : (flags[lane] | ((lane == 3 && tie && tie_margin > 0.0) ? 4 : 0));
    // column-tiled solve: the inlier counts of ALL slabs for the winner, from the shares the decide / merge stages kept
    if (m_all)
        for (int r = lane; r < nranks; r += 64) m_all[r] = bi >= 0 ? (int64_t)cnt_rt[(int64_t)r * cnt_stride + bi] : 0;
    const double h = (lane < 8 && bi >= 0) ? hyp[(int64_t)bi * 8 + lane] : 0.0;
    if (lane < 8) best->hyp[lane] = h;
    if (lane < 8 && best_host) best_host->hyp[lane] = h;
    if (lane == 0) {
        best->best_trial = bi;
        best->undecided = undecided;
        best->lma_tie = tie;
        best->lazy_pending = pending;
        best->shared_best = ns;
        best->num_inliers = bi >= 0 ? (int64_t)best_count : 0;
        best->inlier_error = best_err;
        if (best_host) {
            best_host->best_trial = bi;
            best_host->undecided = undecided;
            best_host->lma_tie = tie;
            best_host->lazy_pending = pending;
            best_host->shared_best = ns;
            best_host->num_inliers = bi >= 0 ? (int64_t)best_count : 0;
            best_host->inlier_error = best_err;
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// best trial: dense rho + mask, then order-preserving compaction
// ---------------------------------------------------------------------------------------------------
// Block b owns the contiguous pixel range [b*chunk, (b+1)*chunk), chunk a multiple of kRB.
__global__ __launch_bounds__(kRB) void ransac_final_kernel(const double2* __restrict__ q, const double2* __restrict__ u,
                                                          const double* __restrict__ alpha,
                                                          const double* __restrict__ alpha_k, int64_t n, int64_t chunk,
                                                          const RansacBest* __restrict__ best,
                                                          const LmState* __restrict__ states, int depth_mode, double tol,
                                                          double* __restrict__ rho_out, uint8_t* __restrict__ mask_out,
                                                          int64_t* __restrict__ block_counts) {
    __shared__ LmPlanLds plan;
    __shared__ int s_cnt[kRB / 64];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (best->undecided) return;  // (see ransac_pick_kernel)
    const int bt = best->best_trial;
    Pose pose;
    pose.w[0] = best->hyp[0], pose.w[1] = best->hyp[1], pose.w[2] = best->hyp[2];
    pose.v[0] = best->hyp[3], pose.v[1] = best->hyp[4], pose.v[2] = best->hyp[5];
    pose.k = best->hyp[6];
    if (depth_mode == RSDSFM_DEPTH_CERES_LM && bt >= 0) {
        if (tid == 0) {
            plan.n_hist = states[bt].n_hist;
            plan.K = 0;
            plan.write_which = 0;
        }
        if (tid < kMaxIter) plan.inv_hist[tid] = 1.0 / states[bt].hist[tid];
    }
    __syncthreads();
    const double two_over = 2.0 / (2.0 + pose.k);
    const int64_t i0 = (int64_t)blockIdx.x * chunk;
    const int64_t i1 = (i0 + chunk < n) ? i0 + chunk : n;
    int count = 0;
    for (int64_t i = i0 + tid; i < i1; i += kRB) {
        double2 qq = q[i], uu = u[i];
        const double al = alpha[i], ak = alpha_k[i];
        bool in = false;
        double rho = 0.0;
        if (bt >= 0) {
            rho = hyp_rho(qq.x, qq.y, uu.x, uu.y, al, ak, pose, two_over, depth_mode, plan);
            in = point_error(qq.x, qq.y, uu.x, uu.y, al, ak, pose, two_over, rho) < tol;
        }
        rho_out[i] = rho;
        mask_out[i] = in ? 1 : 0;
        count += in ? 1 : 0;
    }
    // block count (integers: order irrelevant)
    for (int off = 32; off >= 1; off >>= 1) count += __shfl_xor(count, off, 64);
    if (lane == 0) s_cnt[wv] = count;
    __syncthreads();
    if (tid == 0) {
        int c = 0;
        for (int w2 = 0; w2 < kRB / 64; ++w2) c += s_cnt[w2];
        block_counts[blockIdx.x] = c;
    }
}

// exclusive scan of the block counts (single workgroup; nblocks <= a few thousand)
__global__ __launch_bounds__(256) void ransac_scan_kernel(const int64_t* __restrict__ block_counts, int nblocks,
                                                         int64_t* __restrict__ block_offsets, RansacBest* best, RansacBest* best_host) {
    __shared__ int64_t s_part[256];
    const int tid = threadIdx.x;
    const int per = (nblocks + 255) / 256;
    int64_t sum = 0;
    for (int j = 0; j < per; ++j) {
        const int b = tid * per + j;
        if (b < nblocks) sum += block_counts[b];
    }
    s_part[tid] = sum;
    __syncthreads();
    if (tid == 0) {
        int64_t run = 0;
        for (int i = 0; i < 256; ++i) {
            int64_t v = s_part[i];
            s_part[i] = run;
            run += v;
        }
        best->num_inliers_scan = run;
        if (best_host) best_host->num_inliers_scan = run;
    }
    __syncthreads();
    int64_t run = s_part[tid];
    for (int j = 0; j < per; ++j) {
        const int b = tid * per + j;
        if (b < nblocks) {
            block_offsets[b] = run;
            run += block_counts[b];
        }
    }
}

__global__ __launch_bounds__(kRB) void ransac_scatter_kernel(const double2* __restrict__ q, const double* __restrict__ alpha,
                                                            const double* __restrict__ alpha_k, int64_t n, int64_t chunk,
                                                            const double* __restrict__ rho, const uint8_t* __restrict__ mask,
                                                            const int64_t* __restrict__ block_counts, RansacBest* best,
                                                            RansacBest* best_host, int64_t* __restrict__ inlier_idx, double* __restrict__ inliers,
                                                            double* __restrict__ out_alpha, double* __restrict__ out_alpha_k) {
    __shared__ int s_wave[kRB / 64];
    __shared__ int64_t s_base;
    __shared__ int64_t s_pre[kRB / 64], s_all[kRB / 64];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    // behind a final stage that left at once (RansacBest::undecided, see ransac_pick_kernel) the counts and the mask are stale, possibly
    // of a solve of another size: nothing may be written.  (The word is loaded here and tested behind the scan of the counts, whose loads
    // it travels with: a test up here would put a dependent load in front of every workgroup.)
    const int undecided = best->undecided;
    const int64_t i0 = (int64_t)blockIdx.x * chunk;
    const int64_t i1 = (i0 + chunk < n) ? i0 + chunk : n;
    // exclusive scan of the per-workgroup inlier counts, done by every workgroup for itself (<= 2048 integers: exact in any
    // order) instead of a single-workgroup scan kernel between ransac_final_kernel and this one; workgroup 0 records the total
    {
        int64_t pre = 0, all = 0;
        for (int b = tid; b < (int)gridDim.x; b += kRB) {
            const int64_t cnt = block_counts[b];
            all += cnt;
            if (b < (int)blockIdx.x) pre += cnt;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            pre += __shfl_down(pre, off);
            all += __shfl_down(all, off);
        }
        if (lane == 0) s_pre[wv] = pre, s_all[wv] = all;
        __syncthreads();
        if (tid == 0) {
            int64_t p2 = 0, a2 = 0;
            for (int w2 = 0; w2 < kRB / 64; ++w2) p2 += s_pre[w2], a2 += s_all[w2];
            s_base = p2;
            if (blockIdx.x == 0 && !undecided) {
                best->num_inliers_scan = a2;
                if (best_host) best_host->num_inliers_scan = a2;
            }
        }
    }
    if (undecided) return;  // (uniform)
    __syncthreads();
    for (int64_t start = i0; start < i1; start += kRB) {
        const int64_t i = start + tid;
        const bool in = (i < i1) && mask[i] != 0;
        const unsigned long long bal = __ballot(in);
        const int prefix = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) s_wave[wv] = __popcll(bal);
        __syncthreads();
        int woff = 0, total = 0;
#pragma unroll
        for (int w2 = 0; w2 < kRB / 64; ++w2) {
            const int c = s_wave[w2];
            if (w2 < wv) woff += c;
            total += c;
        }
        if (in) {
            const int64_t o = s_base + woff + prefix;
            double2 qq = q[i];
            if (inlier_idx) inlier_idx[o] = i;
            if (inliers) {
                inliers[3 * o + 0] = qq.x;
                inliers[3 * o + 1] = qq.y;
                inliers[3 * o + 2] = 1.0 / rho[i];  // minimal.cc:299
            }
            if (out_alpha) out_alpha[o] = alpha[i];
            if (out_alpha_k) out_alpha_k[o] = alpha_k[i];
        }
        __syncthreads();
        if (tid == 0) s_base += total;
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------------
int ransac_pixel_grid(const Ctx* c, int64_t n) {
    const int64_t tiles = (n + (int64_t)kRB * kRP - 1) / ((int64_t)kRB * kRP);
    int64_t g = tiles < 1 ? 1 : tiles;
    const int64_t cap = (int64_t)c->num_cus * 4;
    if (g > cap) {
        const int64_t iters = (g + cap - 1) / cap;
        g = (g + iters - 1) / iters;
    }
    return (int)g;
}

// hypothesis groups (gridDim.y) of ransac_lm_kernel: enough workgroups for >= 8 rounds of the chip's resident slots (2 per CU at
// this kernel's register count), but at least 8 hypotheses per workgroup so that the tile load and the prologue stay amortised.
// Measured at 1280x720, T = 50 (900 tiles): 631 / 575 / 560 / 556 us for 1 / 3 / 4 / 5 groups.
static int ransac_lm_groups(const Ctx* c, int grid, int T) {
    const int slots = c->num_cus * 2;
    int g = (8 * slots + grid - 1) / grid;
    g = std::min(g, T / 8);
    return std::max(g, 1);
}

int ransac_lm_partials_doubles(const Ctx* c, int64_t n, int batch) { return ransac_pixel_grid(c, n) * batch * NSR; }

static int lm_launch(Ctx* c, const dim3& g2, int k0, const double* q, const double* u, const double* a, const double* ak, int64_t n,
                     const double* hyp, int T, const LmState* states, double* partials, int round, double tol, int* flags, int fused_base,
                     bool core_math, const int* m9_flag, int m9_epoch, unsigned long long* clk = nullptr) {
    const double2* q2 = reinterpret_cast<const double2*>(q);
    const double2* u2 = reinterpret_cast<const double2*>(u);
    const size_t lds = sizeof(double) * T * NSR;
    // g2 = (tile blocks, hypothesis groups) is flattened into a 1-D grid of windows of 8 tile blocks x groups (see the kernel)
    const int tiles = (int)g2.x, groups = (int)g2.y;
    const dim3 g1((unsigned)(((tiles + 7) / 8) * 8 * groups));
    // the workgroup that stamps the clocks (see the kernel): group 0 of the middle window's first tile
    const int clk_bid = !clk ? -1 : groups > 1 ? (((tiles + 7) / 8) / 2) * 8 * (groups - 1) : tiles / 2;
#define RSDSFM_LM_LAUNCH(R0, K0, BASE, CORE) \
    hipLaunchKernelGGL((ransac_lm_kernel<R0, K0, BASE, CORE>), g1, dim3(kRB), lds, c->stream, q2, u2, a, ak, n, hyp, T, states, partials, round, tol, flags, tiles, groups, m9_flag, m9_epoch, clk, clk_bid)
    // (core: see the kernel's CORE -- needs the flag words, and stands for the reference arithmetic's functions only)
    const bool core = core_math && flags != nullptr && !RSDSFM_FUSED;
    (void)core;
#if RSDSFM_FUSED
#define RSDSFM_LM_LAUNCH2(R0, K0, BASE) RSDSFM_LM_LAUNCH(R0, K0, BASE, false)
#else
#define RSDSFM_LM_LAUNCH2(R0, K0, BASE)                 \
    do {                                                \
        if (core) RSDSFM_LM_LAUNCH(R0, K0, BASE, true); \
        else RSDSFM_LM_LAUNCH(R0, K0, BASE, false);     \
    } while (0)
#endif
    if (round != 0)
        RSDSFM_LM_LAUNCH(false, KMAX, 0, false);  // continuation rounds score nothing
    else if (k0 == 2)
        RSDSFM_LM_LAUNCH2(true, 2, 1);
    else if (fused_base == 1)
        RSDSFM_LM_LAUNCH2(true, KMAX, 1);
    else if (fused_base == 3)
        RSDSFM_LM_LAUNCH2(true, KMAX, 3);
    else
        RSDSFM_LM_LAUNCH2(true, KMAX, 2);
#undef RSDSFM_LM_LAUNCH2
#undef RSDSFM_LM_LAUNCH
    RSDSFM_HIP_CHECK(c, hipGetLastError());
    return RSDSFM_OK;
}

// flags: device int[8] = {running, unscored, not-done-with-one-step, -, hypotheses that ended after 0 / 1 / 2 / >= 3 (or not yet)
// accepted steps}; flags[0] is cleared here, the others by the caller once per batch.  k0 (2 or KMAX): LM iterations speculated by
// round 0; fused_base (1 .. k0): the speculated iterate whose score round 0 fuses (the same values for every round of a batch).
int ransac_lm_round_launch(Ctx* c, const double* q, const double* u, const double* a, const double* ak, int64_t n,
                           const double* hyp, int T, LmState* states, double* partials, int* flags, int* scored,
                           double* trial_count, double* trial_err, int round, double tol, int k0, int fused_base, bool core_math,
                           const int* m9_core_flag, int m9_core_epoch, int* unscored_list) {
    const int grid = ransac_pixel_grid(c, n);
    const dim3 g2(grid, ransac_lm_groups(c, grid, T));
    if (k0 != 2) k0 = KMAX;
    if (fused_base < 1 || fused_base > k0) fused_base = std::min(2, k0);
    const bool prof = c->profile && round == 0 && c->ev_prof[0] && c->ev_prof[1];
    if (prof) RSDSFM_HIP_CHECK(c, hipEventRecord(c->ev_prof[0], c->stream));
    int rc = lm_launch(c, g2, k0, q, u, a, ak, n, hyp, T, states, partials, round, tol, flags, fused_base, core_math, round == 0 ? m9_core_flag : nullptr, m9_core_epoch,
                       prof ? c->d_clk_probe : nullptr);
    if (rc != RSDSFM_OK) return rc;
    if (prof) {
        RSDSFM_HIP_CHECK(c, hipEventRecord(c->ev_prof[1], c->stream));
        c->prof_pending = true;
        c->prof_what = 0;
    }
    hipLaunchKernelGGL(ransac_decide_kernel, dim3(T), dim3(256), 0, c->stream, partials, grid, T, 1, states, n, round, k0, flags, flags + 2, scored,
                       trial_count, trial_err, fused_base, flags + 4, unscored_list);
    RSDSFM_HIP_CHECK(c, hipGetLastError());
    return RSDSFM_OK;
}

// scores the hypotheses of the batch [0, T) that are not yet scored; trial_count / trial_err point at the batch's slots
int ransac_score_launch(Ctx* c, const double* q, const double* u, const double* a, const double* ak, int64_t n,
                        const double* hyp, int T, const LmState* states, int depth_mode, double tol, const int* scored,
                        double* partials, double* trial_count, double* trial_err, const int* unscored_list, const int* unscored_count) {
    const int grid = ransac_pixel_grid(c, n);
    const int groups = ransac_lm_groups(c, grid, T);
    hipLaunchKernelGGL(ransac_score_kernel, dim3(((grid + 7) / 8) * 8 * groups), dim3(kRB), sizeof(double) * T * 2, c->stream,
                       reinterpret_cast<const double2*>(q), reinterpret_cast<const double2*>(u), a, ak, n, hyp, T, states,
                       depth_mode, tol, scored, partials, grid, groups, unscored_list, unscored_list ? unscored_count : nullptr);
    RSDSFM_HIP_CHECK(c, hipGetLastError());
    hipLaunchKernelGGL(ransac_reduce_scores_kernel, dim3(T), dim3(256), 0, c->stream, partials, grid, T, scored, trial_count,
                       trial_err);
    RSDSFM_HIP_CHECK(c, hipGetLastError());
    return RSDSFM_OK;
}

// ---- row-tiled stages (one shard of the points per rank; see dist.py TiledFrameSolve) ----
// core_flags (round 0 of the native tiled driver, may be null): 8 zeroed device words; round 0 then runs the in-range function cores, folds
// the minimal solver's flag (m9_flag / m9_epoch, may be null / 0) in, and the rows carry a 16-byte trailer (ransac_lm_rows_kernel): the
// all-gather payload is ransac_rows_payload_doubles(T, true) doubles per rank
int ransac_lm_rows_launch(Ctx* c, const double* q, const double* u, const double* a, const double* ak, int64_t n,
                          const double* hyp, int T, const LmState* states, double* partials, int round, double tol, double* rows,
                          int* core_flags, const int* m9_flag, int m9_epoch) {
    const int grid = ransac_pixel_grid(c, n);
    const dim3 g2(grid, ransac_lm_groups(c, grid, T));
    const bool core = core_flags != nullptr && round == 0;
    int rc = lm_launch(c, g2, KMAX, q, u, a, ak, n, hyp, T, states, partials, round, tol, core ? core_flags : nullptr, kTiledFusedBase, core,
                       core ? m9_flag : nullptr, m9_epoch);
    if (rc != RSDSFM_OK) return rc;
    hipLaunchKernelGGL(ransac_lm_rows_kernel, dim3(T), dim3(256), 0, c->stream, partials, grid, T, states, round, rows,
                       static_cast<const int*>(core ? core_flags : nullptr));
    RSDSFM_HIP_CHECK(c, hipGetLastError());
    return RSDSFM_OK;
}

int ransac_rows_payload_doubles(int T, bool core_trailer) { return T * NSR + (core_trailer ? 2 : 0); }

int ransac_decide_rows_launch(Ctx* c, const double* rows_all, int nranks, int T, LmState* states, int64_t n_total, int round,
                              int* flags, int* scored, double* trial_count, double* trial_err, bool core_trailer, double* cnt_rt, int cnt_stride) {
    if (cnt_rt && nranks > 256) return fail(c, RSDSFM_ERR_INVALID, "more than 256 ranks");
    RSDSFM_HIP_CHECK(c, hipMemsetAsync(flags, 0, sizeof(int), c->stream));
    hipLaunchKernelGGL(ransac_decide_kernel, dim3(T), dim3(256), 0, c->stream, rows_all, nranks, T, 0, states, n_total, round, (int)KMAX, flags,
                       static_cast<int*>(nullptr), scored, trial_count, trial_err, kTiledFusedBase, static_cast<int*>(nullptr),
                       static_cast<int*>(nullptr), core_trailer ? T * NSR / 2 + 1 : 0, cnt_rt, cnt_stride);
    RSDSFM_HIP_CHECK(c, hipGetLastError());
    return RSDSFM_OK;
}

int ransac_score_rows_launch(Ctx* c, const double* q, const double* u, const double* a, const double* ak, int64_t n,
                             const double* hyp, int T, const LmState* states, int depth_mode, double tol, const int* scored,
                             double* partials, double* rows) {
    const int grid = ransac_pixel_grid(c, n);
    const int groups = ransac_lm_groups(c, grid, T);
    hipLaunchKernelGGL(ransac_score_kernel, dim3(((grid + 7) / 8) * 8 * groups), dim3(kRB), sizeof(double) * T * 2, c->stream,
                       reinterpret_cast<const double2*>(q), reinterpret_cast<const double2*>(u), a, ak, n, hyp, T, states,
                       depth_mode, tol, scored, partials, grid, groups, static_cast<const int*>(nullptr), static_cast<const int*>(nullptr));
    RSDSFM_HIP_CHECK(c, hipGetLastError());
    hipLaunchKernelGGL(ransac_score_rows_kernel, dim3(T), dim3(256), 0, c->stream, partials, grid, T, scored, rows);
    RSDSFM_HIP_CHECK(c, hipGetLastError());
    return RSDSFM_OK;
}

int ransac_score_merge_launch(Ctx* c, const double* rows_all, int nranks, int T, const int* scored, double* trial_count,
                              double* trial_err, double* cnt_rt, int cnt_stride) {
    if (cnt_rt && nranks > 256) return fail(c, RSDSFM_ERR_INVALID, "more than 256 ranks");
    hipLaunchKernelGGL(ransac_reduce_scores_kernel, dim3(T), dim3(256), 0, c->stream, rows_all, nranks, T, scored, trial_count,
                       trial_err, cnt_rt, cnt_stride);
    RSDSFM_HIP_CHECK(c, hipGetLastError());
    return RSDSFM_OK;
}

int ransac_rows_doubles() { return NSR; }

int ransac_pick_launch(Ctx* c, const double* trial_count, const double* trial_err, int T, const double* hyp, RansacBest* best,
                       RansacBest* best_host, const int* d_flags, int* h_flags, int scored_ahead, const double* cnt_rt, int cnt_stride,
                       int nranks, int64_t* m_all, double tie_margin, PickLazy lazy) {
    hipLaunchKernelGGL(ransac_pick_kernel, dim3(1), dim3(64), 0, c->stream, trial_count, trial_err, T, hyp, best, best_host, d_flags, h_flags, scored_ahead,
                       cnt_rt, cnt_stride, nranks, m_all, tie_margin, lazy);
    RSDSFM_HIP_CHECK(c, hipGetLastError());
    return RSDSFM_OK;
}

int ransac_final_launch(Ctx* c, const double* q, const double* u, const double* a, const double* ak, int64_t n,
                        RansacBest* best, const LmState* states, int depth_mode, double tol, double* rho, uint8_t* mask,
                        int64_t* block_counts, int64_t* block_offsets, int64_t* inlier_idx, double* inliers,
                        double* out_alpha, double* out_alpha_k, RansacBest* best_host) {
    int64_t blocks = (n + kRB - 1) / kRB;
    if (blocks < 1) blocks = 1;
    const int64_t cap = 2048;
    int64_t chunk = kRB;
    if (blocks > cap) {
        chunk = ((blocks + cap - 1) / cap) * kRB;
        blocks = (n + chunk - 1) / chunk;
    }
    hipLaunchKernelGGL(ransac_final_kernel, dim3((int)blocks), dim3(kRB), 0, c->stream, reinterpret_cast<const double2*>(q),
                       reinterpret_cast<const double2*>(u), a, ak, n, chunk, best, states, depth_mode, tol, rho, mask,
                       block_counts);
    RSDSFM_HIP_CHECK(c, hipGetLastError());
    if (inlier_idx || inliers || out_alpha || out_alpha_k) {  // the compaction scans the workgroup counts itself
        hipLaunchKernelGGL(ransac_scatter_kernel, dim3((int)blocks), dim3(kRB), 0, c->stream, reinterpret_cast<const double2*>(q),
                           a, ak, n, chunk, rho, mask, block_counts, best, best_host, inlier_idx, inliers, out_alpha, out_alpha_k);
    } else {  // no compacted outputs requested: only the total is needed
        hipLaunchKernelGGL(ransac_scan_kernel, dim3(1), dim3(256), 0, c->stream, block_counts, (int)blocks, block_offsets, best, best_host);
    }
    RSDSFM_HIP_CHECK(c, hipGetLastError());
    return RSDSFM_OK;
}

}  // namespace rsdsfm
