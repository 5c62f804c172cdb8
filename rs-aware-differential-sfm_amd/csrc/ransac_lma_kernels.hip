// ransac_lma_kernels.hip -- the T depth solves of a RANSAC (reference minimal.cc:230-289: per trial a Ceres solve of ALL points, then
// the inlier score) on the ANALYTIC LM TRAJECTORY (lma_common.hpp), MI355X (gfx950).
//
//   ransac_lma_kernel         ONE pass over the pixels for all T hypotheses.  A LANE owns a HYPOTHESIS (and one of P = 256 / T pixel
//                             slots of its workgroup) for the whole launch: the pose and the hypothesis' five sums, its gradient maximum and
//                             the {count, error sum} of up to three fused iterates live in the lane's registers across the workgroup's
//                             pixel range -- no cross-lane reduction, no LDS, no barrier inside the pixel loop (the iterate-by-iterate
//                             kernel reduces 22 sums across the workgroup per hypothesis and tile).  The pixels of guards (a) / (b) go
//                             to the hypothesis' list.
//   ransac_lma_rows_kernel    one workgroup per hypothesis: fixed-order reduction of the workgroups' partial rows, the listed pixels on the
//                             reference's exact recurrence (sorted by pixel index: deterministic sums) -> the row the decide stage consumes
//                             (the all-gather payload of the column-tiled solve)
//   ransac_lma_decide_kernel  Ceres' trust-region loop on the closed forms (one lane per hypothesis), guard (c), the score of the final
//                             iterate -> LmState (accepted radii: what ransac_final_kernel / ransac_score_kernel replay), trial scores
// The single-context solve runs rows + decide as ONE launch (ransac_lma_rows_decide_kernel).
#include <stdlib.h>

#include <algorithm>

#include "lma_common.hpp"
#include "lma_stages.hpp"
#include "rsdsfm_internal.hpp"

namespace rsdsfm {

namespace {

__device__ __forceinline__ bool is_nan_bits2(double x) {
    const uint32_t h = (uint32_t)__double2hiint(x) & 0x7FFFFFFFu;
    return h > 0x7FF00000u || (h == 0x7FF00000u && __double2loint(x) != 0);
}
__device__ __forceinline__ Pose load_pose(const double* __restrict__ hyp, int t) {
    const double* h_ = hyp + (int64_t)t * 8;
    Pose p;
    p.w[0] = h_[0], p.w[1] = h_[1], p.w[2] = h_[2];
    p.v[0] = h_[3], p.v[1] = h_[4], p.v[2] = h_[5];
    p.k = h_[6];
    return p;
}
__device__ __forceinline__ bool pose_nan(const Pose& p) {
    return is_nan_bits2(p.w[0]) || is_nan_bits2(p.w[1]) || is_nan_bits2(p.w[2]) || is_nan_bits2(p.v[0]) || is_nan_bits2(p.v[1]) || is_nan_bits2(p.v[2]) ||
           is_nan_bits2(p.k);
}

}  // namespace

// ---------------------------------------------------------------------------------------------------
// the pixel pass
// ---------------------------------------------------------------------------------------------------
// Workgroup b owns the pixels [b chunk, (b + 1) chunk).  Thread g = slot T + t owns hypothesis t and pixel slot `slot` < P = 256 / T: it walks
// the pixels b chunk + slot, + P, + 2 P, ...  partials: [T][gridDim.x][kLmaSlots] (hypothesis-major, like ransac_lm_kernel's).
// lists: irr_count[T] (zeroed per solve), irr_list[T][kLmaListCap] pixel indices in order of arrival (the rows stage sorts them).
// ERR = false (the frame solve's RANSAC: `LmaCand::count_only`): inlier COUNTS only.  minimal.cc:278-285 looks at a trial's error sum only to
// break a tie in the count, so the pass leaves the two square roots per pixel-hypothesis out and ransac_pick_kernel asks for the exact error
// sums (ransac_score_kernel: the reference's arithmetic) of the trials that share the best count, when there is more than one.
template <int NC, bool ERR>
__global__ __launch_bounds__(kLB) __attribute__((amdgpu_waves_per_eu(4, 5))) void ransac_lma_kernel(const double2* __restrict__ q, const double2* __restrict__ u, const double* __restrict__ alpha,
                                                        const double* __restrict__ alpha_k, int64_t n, const double* __restrict__ hyp, int T,
                                                        const LmaCand cd, double* __restrict__ partials, int* __restrict__ irr_count,
                                                        int* __restrict__ irr_list, int64_t chunk, unsigned long long* __restrict__ clk_probe, int clk_bid,
                                                        int Tg, int nwg) {
    __shared__ double s_part[kLB][kLmaSlots + 1];
    const int g = threadIdx.x;
    // rsdsfm_set_profiling: one lane of one workgroup in the middle of the launch (clk_bid; -1 = off) stamps the shader clock counter and the
    // 100 MHz counter at both ends of its life -- the clock this kernel actually runs at
    if ((int)blockIdx.x == clk_bid && g == 0) {
        clk_probe[0] = __builtin_amdgcn_s_memtime();
        clk_probe[1] = __builtin_amdgcn_s_memrealtime();
    }
    // Hypothesis groups (lma_pass_launch): the T hypotheses in groups of Tg <= 256, `nwg` workgroups per group, each group's workgroups covering
    // every pixel.  With ONE group the P = 256 / T pixel slots leave 256 mod T lanes idle -- 2 % at T = 50, 22 % at T = 100, 49 % at T = 130 --;
    // groups of 50 / 43 / 26 bring that under 10 % for every T at the price of reading the pixels once per group (a VALU-bound pass).
    const int grp = (int)blockIdx.x / nwg, bid = (int)blockIdx.x - grp * nwg;
    const int t0 = grp * Tg, Tl = min(Tg, T - t0);  // this group's hypotheses [t0, t0 + Tl)
    const int P = kLB / Tg;
    const int slot = g / Tg, tl = g - slot * Tg, t = t0 + tl;
    const bool active = slot < P && tl < Tl;
    const int64_t p0 = (int64_t)bid * chunk;
    const int len = (int)min(chunk, n - p0);  // pixels of this workgroup (32-bit indices below: the bases are uniform)
    const double2* __restrict__ qb = q + p0;
    const double2* __restrict__ ub = u + p0;
    const double* __restrict__ ab = alpha + p0;
    const double* __restrict__ akb = alpha_k + p0;
    double A = 0.0, B = 0.0, C = 0.0, D = 0.0, E = 0.0, G = 0.0;
    int cnt[NC];
    double es[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) cnt[c] = 0, es[c] = 0.0;
    if (active) {
        const Pose pose = load_pose(hyp, t);
        const double two_over = 2.0 / (2.0 + pose.k);
        // one pixel under this lane's hypothesis; returns whether it goes to the list.  Straight-line on purpose (no branch around the square
        // roots: the error of a pixel that is not an inlier is multiplied by a 0.0 mask instead of being skipped -- the argument is floored into
        // the core's range first, so the product is exact and the sum has the same bits), so that the two pixels of an iteration form ONE basic
        // block and their chains interleave
        auto pixel = [&](const PixIn& px) -> bool {
            const LmaPx v = lma_pixel(px.x, px.y, px.ux, px.uy, px.al, px.ak, pose, two_over);
            A += v.a;
            B += v.ge;
            C = __builtin_fma(v.e0, v.e0, C);
            D = __builtin_fma(v.rhos, v.rhos, D);
            E = __builtin_fma(v.rhos, v.e0, E);
            G = lma_max_abs(G, v.g);
            const double m = lma_margin(v, cd);
            bool listed = v.clamped;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const double e2 = __builtin_fma(v.ge, cd.phi2[c], v.a);
                const bool in = e2 < cd.tol2;
                listed = listed || (fabs(e2 - cd.tol2) <= m);
                cnt[c] += in ? 1 : 0;
                if (ERR) {
                    const double err = sqrt_core(lma_max_pos(e2, kLmaSqrtMin));  // (finite for every e2, NaN included: v_max_f64 drops it)
                    es[c] = __builtin_fma(err, __hiloint2double((in && e2 >= kLmaSqrtMin) ? 0x3FF00000 : 0, 0), es[c]);
                }
            }
            return listed;
        };
        auto push = [&](unsigned i) {  // guards (a) / (b): rare (a handful of pixels per hypothesis)
            const int pos = atomicAdd(&irr_count[t], 1);
            if (pos < kLmaListCap) irr_list[(int64_t)t * kLmaListCap + pos] = (int)(p0 + (int64_t)i);
        };
        // whole rounds of the P slots: a uniform trip count (every workgroup but the last owns a multiple of P pixels), the next pixel's
        // loads in flight under this pixel's arithmetic; then the ragged round of the last workgroup
        const int rounds = len / P;
        unsigned i = (unsigned)slot;
        const unsigned uP = (unsigned)P;
        // (two register sets: the loads of one pixel fly under the arithmetic of the other, and no copies between them)
        PixIn pa, pb;
        int r = 0;
        if (rounds > 0) pa = load_pix(qb, ub, ab, akb, i);
        for (; r + 1 < rounds; r += 2, i += 2 * uP) {
            pb = load_pix(qb, ub, ab, akb, i + uP);
            const bool la = pixel(pa);
            pa = load_pix(qb, ub, ab, akb, r + 2 < rounds ? i + 2 * uP : i);  // (no branch between the two pixels: past the end, a pixel that is not used)
            const bool lb = pixel(pb);
            if (la || lb) {
                if (la) push(i);
                if (lb) push(i + uP);
            }
        }
        if (r < rounds) {
            if (pixel(pa)) push(i);
            i += uP;
        }
        if (i < (unsigned)len && pixel(load_pix(qb, ub, ab, akb, i))) push(i);
    }
    // the P slots of a hypothesis, in slot order
    s_part[g][0] = A, s_part[g][1] = B, s_part[g][2] = C, s_part[g][3] = D, s_part[g][4] = E, s_part[g][kLmaG] = G;
#pragma unroll
    for (int c = 0; c < kLmaNC; ++c) {
        s_part[g][6 + 2 * c] = c < NC ? (double)cnt[c < NC ? c : 0] : 0.0;
        s_part[g][7 + 2 * c] = c < NC ? es[c < NC ? c : 0] : 0.0;
    }
    __syncthreads();
    for (int o = g; o < Tl * kLmaSlots; o += kLB) {
        const int tt = o / kLmaSlots, sl = o - tt * kLmaSlots;
        double r = s_part[tt][sl];
        for (int s = 1; s < P; ++s) r = sl == kLmaG ? fmax(r, s_part[s * Tg + tt][sl]) : r + s_part[s * Tg + tt][sl];
        partials[((int64_t)(t0 + tt) * nwg + bid) * kLmaSlots + sl] = r;
    }
    if ((int)blockIdx.x == clk_bid && g == 0) {
        clk_probe[2] = __builtin_amdgcn_s_memtime();
        clk_probe[3] = __builtin_amdgcn_s_memrealtime();
    }
}

// flags: the RANSAC's flag words (ransac_host.hip): [1] += hypotheses that ended where no score was fused (+ their list), [3] |= 2 a guard
// tripped (the run starts over on the iterate-by-iterate kernels), [4 + min(steps, 3)] histogram of the accepted steps
__device__ __forceinline__ void lma_publish(int t, const LmScal& st, const double* hist_l, int fallback, bool scored, double count, double err,
                                            LmState* states, int* flags, int* __restrict__ scored_out, double* __restrict__ trial_count,
                                            double* __restrict__ trial_err, int* steps_hist, int* __restrict__ unscored_list, int* guard_word, bool count_only) {
    LmState* state = states + t;
    if (fallback) {
        // a guard of THIS hypothesis tripped (a decision inside the undecided band, a list that overflows -- e.g. a hypothesis whose k puts beta
        // near zero: every pixel clamped --, listed pixels off the tabulated plan): the hypothesis goes on ITERATE BY ITERATE, alone -- its state
        // is the one ransac_lm_kernel's continuation rounds start a solve from (nothing accepted, KMAX planned iterations, launch 1), the run
        // counts it as still running, and the host enqueues round 1 on the iterate-by-iterate kernels, where only such hypotheses take part.
        // flags[3] bit 4 + bits 8..: diagnostics (which guards)
        LmScal f = {};
        f.status = 0;
        f.K = KMAX;
        f.termination = -1;
        f.rho_holds = -1;
        f.launches = 0;  // (ransac_decide_kernel: a state without launches takes the sums of its first launch as iteration zero)
        f.next_launch = 1;
        f.radius = kInitialRadius;
        f.decrease_factor = 2.0;
        double r = kInitialRadius;
        for (int j = 0; j < KMAX; ++j) {
            f.cand[j] = r;
            r = radius_accept(r, 1.0);
        }
        *static_cast<LmScal*>(state) = f;
        atomicAdd(&flags[0], 1);
        atomicOr(&flags[3], 16 | (1 << (8 + fallback)));
        (void)guard_word;
        return;
    }
    *static_cast<LmScal*>(state) = st;
    for (int j = 0; j < st.n_hist && j < kMaxIter; ++j) state->hist[j] = hist_l[j];
    if (steps_hist) atomicAdd(&steps_hist[st.n_hist < 1 ? 0 : (st.n_hist > 3 ? 3 : st.n_hist)], 1);
    if (scored) {
        trial_count[t] = count;
        trial_err[t] = err;
        scored_out[t] = count_only ? 2 : 1;  // (2: the count alone -- ransac_pick_kernel asks for the error sum where the tie rule needs it)
    } else {
        const int pos = atomicAdd(&flags[1], 1);
        if (unscored_list) unscored_list[pos] = t;
    }
}

// single context: rows + decide in one launch (one workgroup per hypothesis)
__global__ __launch_bounds__(kLB) void ransac_lma_rows_decide_kernel(const double* __restrict__ partials, int nblocks, int T, const double2* __restrict__ q,
                                                                    const double2* __restrict__ u, const double* __restrict__ alpha,
                                                                    const double* __restrict__ alpha_k, int64_t n, const double* __restrict__ hyp,
                                                                    const LmaCand cd, const int* __restrict__ irr_count, const int* __restrict__ irr_list,
                                                                    LmState* states, int* flags, int* __restrict__ scored_out, double* __restrict__ trial_count,
                                                                    double* __restrict__ trial_err, int* __restrict__ unscored_list, int* guard_word,
                                                                    const int* __restrict__ m9_core_flag, int m9_core_epoch) {
    __shared__ double s_row[kLmaRow];
    const int t = blockIdx.x;
    // the minimal solver of this run met an SVD operand outside the range of its function cores (minimal9_kernels.hip): reported through the
    // run's flag word like ransac_lm_kernel does -- the host starts the run over with the standard functions
    if (t == 0 && threadIdx.x == 0 && m9_core_flag && *m9_core_flag == m9_core_epoch) atomicOr(&flags[3], 1);
    lma_rows_stage(partials, nblocks, T, t, q, u, alpha, alpha_k, load_pose(hyp, t), cd, irr_count, irr_list, s_row);
    __syncthreads();
    if (threadIdx.x == 0) {
        LmScal st;
        double hist_l[kMaxIter];
        bool scored;
        double count, err;
        const Pose pose = load_pose(hyp, t);
        const int fb = lma_decide(s_row, n, cd, cd.plan, pose_nan(pose), st, hist_l, scored, count, err);
        lma_publish(t, st, hist_l, fb, scored, count, err, states, flags, scored_out, trial_count, trial_err, flags + 4, unscored_list, guard_word, cd.count_only != 0);
    }
}

// column-tiled solve: the rank's rows [T][kLmaRow] ...
__global__ __launch_bounds__(kLB) void ransac_lma_rows_kernel(const double* __restrict__ partials, int nblocks, int T, const double2* __restrict__ q,
                                                             const double2* __restrict__ u, const double* __restrict__ alpha,
                                                             const double* __restrict__ alpha_k, const double* __restrict__ hyp, const LmaCand cd,
                                                             const int* __restrict__ irr_count, const int* __restrict__ irr_list, double* __restrict__ rows) {
    __shared__ double s_row[kLmaRow];
    const int t = blockIdx.x;
    lma_rows_stage(partials, nblocks, T, t, q, u, alpha, alpha_k, load_pose(hyp, t), cd, irr_count, irr_list, s_row);
    __syncthreads();
    if (threadIdx.x < kLmaRow) rows[(int64_t)t * kLmaRow + threadIdx.x] = s_row[threadIdx.x];
}

// ... and the decide stage on the gathered rows [nranks][T][kLmaRow] (rank_stride doubles between two ranks' blocks), ranks in rank order.
// cnt_rt (optional): the ranks' shares of the fused inlier count of the final iterate, [rank][hypothesis] (see ransac_decide_kernel)
__global__ __launch_bounds__(64) void ransac_lma_decide_kernel(const double* __restrict__ rows_all, int nranks, int64_t rank_stride, int T, int64_t n_total,
                                                              const double* __restrict__ hyp, const LmaCand cd, LmState* states, int* flags,
                                                              int* __restrict__ scored_out, double* __restrict__ trial_count,
                                                              double* __restrict__ trial_err, int* __restrict__ unscored_list, int* guard_word,
                                                              double* __restrict__ cnt_rt, int cnt_stride, const int* __restrict__ m9_core_flag, int m9_core_epoch) {
    const int t = blockIdx.x * 64 + threadIdx.x;
    // (the minimal solver's range flag: replicated -- every rank ran the same solver on the same points -- and read locally)
    if (t == 0 && m9_core_flag && *m9_core_flag == m9_core_epoch) atomicOr(&flags[3], 1);
    if (t >= T) return;
    double row[kLmaRow];
    for (int j = 0; j < kLmaRow; ++j) {
        const bool is_max = j == kLmaG || j == kLmaRowX0 + 1 || (j >= kLmaRowXk && j < kLmaRowScore && ((j - kLmaRowXk) % 5) == 4) || j == 7;
        double r = rows_all[(int64_t)t * kLmaRow + j];
        for (int rk = 1; rk < nranks; ++rk) {
            const double x = rows_all[rk * rank_stride + (int64_t)t * kLmaRow + j];
            r = is_max ? fmax(r, x) : r + x;
        }
        row[j] = r;
    }
    LmScal st;
    double hist_l[kMaxIter];
    bool scored;
    double count, err;
    const Pose pose = load_pose(hyp, t);
    const int fb = lma_decide(row, n_total, cd, cd.plan, pose_nan(pose), st, hist_l, scored, count, err);
    lma_publish(t, st, hist_l, fb, scored, count, err, states, flags, scored_out, trial_count, trial_err, nullptr, unscored_list, guard_word, cd.count_only != 0);
    if (cnt_rt && scored && !fb) {
        int cc = -1;
        for (int c = 0; c < cd.nc; ++c)
            if (cd.steps[c] == st.n_hist) cc = c;
        for (int rk = 0; rk < nranks; ++rk) cnt_rt[(int64_t)rk * cnt_stride + t] = cc >= 0 ? rows_all[rk * rank_stride + (int64_t)t * kLmaRow + kLmaRowScore + 2 * cc] : 0.0;
    }
}

// ---------------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------------
// fused iterates of a pixel pass: the accepted-step counts in `steps` (each 1 .. kLmaKP - 1, distinct), at most kLmaNC
LmaCand lma_candidates(const int* steps, int nc, double tol, bool count_only = false) {
    LmaCand cd;
    const LmaPlan plan = lma_plan();
    cd.plan = plan;
    cd.nc = std::min(std::max(nc, 1), (int)kLmaNC);
    for (int c = 0; c < kLmaNC; ++c) {
        const int s = c < cd.nc ? std::min(std::max(steps[c], 1), kLmaKP - 1) : 1;
        cd.steps[c] = s;
        cd.phi2[c] = plan.phi[s] * plan.phi[s];
    }
    cd.tol = tol;
    cd.tol2 = tol * tol;
    cd.c1 = std::max(0.5 * kLmaEta * tol, kLmaMarginFloor);
    cd.c1x2 = 2.0 * cd.c1;
    cd.count_only = count_only ? 1 : 0;
    return cd;
}

// workgroups of the pixel pass and the pixels each owns: ONE round of the chip's resident workgroups (every workgroup does the same work: they
// start and end together; 2048 workgroups on 1280 resident slots ran 1.6 rounds, the second one 60 % empty), fewer for small inputs
// (>= 8 pixels per slot and workgroup)
// hypotheses per group for a batch of T: the fewest groups that keep >= 90 % of the lanes busy (P Tg of 256 with P = 256 / Tg pixel slots;
// all groups but the last hold Tg hypotheses), else the best of up to 8 groups
int ransac_lma_group_size(int T, int* groups_out) {
    int best_ng = 1;
    double best_u = 0.0;
    for (int ng = 1; ng <= 8 && ng <= T; ++ng) {
        const int tg = (T + ng - 1) / ng, ngu = (T + tg - 1) / tg;  // (groups actually used)
        const double u = (double)T * (kLB / tg) / ((double)ngu * kLB);
        if (u > best_u + 1e-9) best_u = u, best_ng = ng;
        if (u >= 0.9) {
            best_ng = ng;
            break;
        }
    }
    const int tg = (T + best_ng - 1) / best_ng;
    if (groups_out) *groups_out = (T + tg - 1) / tg;
    return tg;
}

// (T here: the hypotheses of ONE group; the launch has `groups` times as many workgroups)
int ransac_lma_grid(const Ctx* c, int64_t n, int T, int64_t* chunk_out, int blocks_per_cu) {
    const int P = std::max(1, kLB / std::max(T, 1));
    if (n <= 0) {  // (an empty slab of the column-tiled solve: one workgroup that finds nothing to do)
        *chunk_out = P;
        return 1;
    }
    int64_t g = (n + (int64_t)P * 8 - 1) / ((int64_t)P * 8);
    g = std::max<int64_t>(1, std::min<int64_t>(g, (int64_t)c->num_cus * std::max(1, std::min(blocks_per_cu, 8))));
    int64_t chunk = (n + g - 1) / g;
    chunk = ((chunk + P - 1) / P) * P;  // whole rounds of the P slots
    g = std::max<int64_t>(1, (n + chunk - 1) / chunk);
    *chunk_out = chunk;
    return (int)g;
}
int64_t ransac_lma_partials_doubles(const Ctx* c, int64_t n, int batch) {
    // (the grid depends on T through P = 256 / T: sized for the largest grid any T <= batch can ask for)
    const int64_t gmax = std::min<int64_t>((int64_t)c->num_cus * 8, std::max<int64_t>(1, (n + 7) / 8));
    return gmax * batch * kLmaSlots;
}
size_t ransac_lma_list_ints(int batch) { return (size_t)batch * kLmaListCap; }

static int lma_pass_launch(Ctx* c, const double* q, const double* u, const double* a, const double* ak, int64_t n, const double* hyp, int T,
                           const LmaCand& cd, double* partials, int* irr_count, int* irr_list, int* grid_out, unsigned long long* clk = nullptr) {
    // resident workgroups per CU of the kernel that is about to run (registers set it: 5 at 94 VGPRs)
    static int occ_tab[2][kLmaNC + 1] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
    const int nc = std::min(std::max(cd.nc, 1), (int)kLmaNC);
    int* occ = occ_tab[cd.count_only ? 1 : 0];
    if (occ[nc] == 0) {
        int nb = 0;
        const void* fn = cd.count_only ? (nc == 1 ? reinterpret_cast<const void*>(&ransac_lma_kernel<1, false>) : nc == 2 ? reinterpret_cast<const void*>(&ransac_lma_kernel<2, false>) : reinterpret_cast<const void*>(&ransac_lma_kernel<3, false>))
                                       : (nc == 1 ? reinterpret_cast<const void*>(&ransac_lma_kernel<1, true>) : nc == 2 ? reinterpret_cast<const void*>(&ransac_lma_kernel<2, true>) : reinterpret_cast<const void*>(&ransac_lma_kernel<3, true>));
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, kLB, 0) != hipSuccess || nb < 1) nb = 4;
        occ[nc] = nb;
    }
    int64_t chunk;
    static const int grid_override = getenv("RSDSFM_LMA_BLOCKS_PER_CU") ? atoi(getenv("RSDSFM_LMA_BLOCKS_PER_CU")) : 0;  // (experiments)
    int groups = 1;
    const int Tg = ransac_lma_group_size(T, &groups);
    // (the resident workgroups are shared by the groups: every group gets 1 / groups of one round of the chip)
    const int per_cu = std::max(1, (grid_override > 0 ? grid_override : occ[nc]));
    int nwg = ransac_lma_grid(c, n, Tg, &chunk, per_cu);
    if (groups > 1) {
        const int64_t cap = std::max<int64_t>(1, (int64_t)c->num_cus * std::min(per_cu, 8) / groups);
        if (nwg > cap) {
            const int P = std::max(1, kLB / Tg);
            chunk = (n + cap - 1) / cap;
            chunk = ((chunk + P - 1) / P) * P;
            nwg = (int)std::max<int64_t>(1, (n + chunk - 1) / chunk);
        }
    }
    const int grid = nwg * groups;
    *grid_out = nwg;  // (workgroups PER HYPOTHESIS: what the rows stages index the partials with)
    const double2* q2 = reinterpret_cast<const double2*>(q);
    const double2* u2 = reinterpret_cast<const double2*>(u);
    const int clk_bid = clk ? grid / 2 : -1;
#define RSDSFM_LMA_LAUNCH(NC, ERR) \
    hipLaunchKernelGGL((ransac_lma_kernel<NC, ERR>), dim3(grid), dim3(kLB), 0, c->stream, q2, u2, a, ak, n, hyp, T, cd, partials, irr_count, irr_list, chunk, clk, clk_bid, Tg, nwg)
    if (cd.count_only) {
        if (cd.nc == 1) RSDSFM_LMA_LAUNCH(1, false);
        else if (cd.nc == 2) RSDSFM_LMA_LAUNCH(2, false);
        else RSDSFM_LMA_LAUNCH(3, false);
    } else if (cd.nc == 1) RSDSFM_LMA_LAUNCH(1, true);
    else if (cd.nc == 2) RSDSFM_LMA_LAUNCH(2, true);
    else RSDSFM_LMA_LAUNCH(3, true);
#undef RSDSFM_LMA_LAUNCH
    RSDSFM_HIP_CHECK(c, hipGetLastError());
    return RSDSFM_OK;
}

// the T (<= 256) depth solves of one hypothesis batch: pixel pass + rows / decide.  irr_count must be zero on entry.
int ransac_lma_launch(Ctx* c, const double* q, const double* u, const double* a, const double* ak, int64_t n, const double* hyp, int T,
                      LmState* states, double* partials, int* flags, int* scored, double* trial_count, double* trial_err, double tol,
                      const int* cand_steps, int ncand, int* irr_count, int* irr_list, int* unscored_list, int* guard_word, const int* m9_core_flag,
                      int m9_core_epoch, bool count_only) {
    if (T < 1 || T > kLB) return fail(c, RSDSFM_ERR_INVALID, "hypothesis batch of the analytic pass exceeds 256");
    const LmaCand cd = lma_candidates(cand_steps, ncand, tol, count_only);
    int grid = 0;
    const bool prof = c->profile && c->ev_prof[0] && c->ev_prof[1];
    if (prof) RSDSFM_HIP_CHECK(c, hipEventRecord(c->ev_prof[0], c->stream));
    int rc = lma_pass_launch(c, q, u, a, ak, n, hyp, T, cd, partials, irr_count, irr_list, &grid, prof ? c->d_clk_probe : nullptr);
    if (rc != RSDSFM_OK) return rc;
    if (prof) {
        RSDSFM_HIP_CHECK(c, hipEventRecord(c->ev_prof[1], c->stream));
        c->prof_pending = true;
        c->prof_what = 0;
    }
    hipLaunchKernelGGL(ransac_lma_rows_decide_kernel, dim3(T), dim3(kLB), 0, c->stream, partials, grid, T, reinterpret_cast<const double2*>(q),
                       reinterpret_cast<const double2*>(u), a, ak, n, hyp, cd, irr_count, irr_list, states, flags, scored, trial_count, trial_err,
                       unscored_list, guard_word, m9_core_flag, m9_core_epoch);
    RSDSFM_HIP_CHECK(c, hipGetLastError());
    return RSDSFM_OK;
}

// column-tiled solve: pixel pass + the rank's rows (the all-gather payload: T * kLmaRow doubles) ...
int ransac_lma_rows_doubles() { return kLmaRow; }
int ransac_lma_rows_launch(Ctx* c, const double* q, const double* u, const double* a, const double* ak, int64_t n, const double* hyp, int T,
                           double* partials, double tol, const int* cand_steps, int ncand, int* irr_count, int* irr_list, double* rows) {
    if (T < 1 || T > kLB) return fail(c, RSDSFM_ERR_INVALID, "hypothesis batch of the analytic pass exceeds 256");
    const LmaCand cd = lma_candidates(cand_steps, ncand, tol);
    int grid = 0;
    int rc = lma_pass_launch(c, q, u, a, ak, n, hyp, T, cd, partials, irr_count, irr_list, &grid);
    if (rc != RSDSFM_OK) return rc;
    hipLaunchKernelGGL(ransac_lma_rows_kernel, dim3(T), dim3(kLB), 0, c->stream, partials, grid, T, reinterpret_cast<const double2*>(q),
                       reinterpret_cast<const double2*>(u), a, ak, hyp, cd, irr_count, irr_list, rows);
    RSDSFM_HIP_CHECK(c, hipGetLastError());
    return RSDSFM_OK;
}
// ... and the decide stage on the gathered rows
int ransac_lma_decide_rows_launch(Ctx* c, const double* rows_all, int nranks, int64_t rank_stride, int T, int64_t n_total, const double* hyp,
                                  LmState* states, int* flags, int* scored, double* trial_count, double* trial_err, double tol, const int* cand_steps,
                                  int ncand, int* unscored_list, int* guard_word, double* cnt_rt, int cnt_stride, const int* m9_core_flag, int m9_core_epoch) {
    const LmaCand cd = lma_candidates(cand_steps, ncand, tol);
    hipLaunchKernelGGL(ransac_lma_decide_kernel, dim3((T + 63) / 64), dim3(64), 0, c->stream, rows_all, nranks, rank_stride, T, n_total, hyp, cd, states,
                       flags, scored, trial_count, trial_err, unscored_list, guard_word, cnt_rt, cnt_stride, m9_core_flag, m9_core_epoch);
    RSDSFM_HIP_CHECK(c, hipGetLastError());
    return RSDSFM_OK;
}

}  // namespace rsdsfm
