// ransac_host.hip -- host orchestration of minimal::calculateVelocities / minimal::ransac over the HIP kernels
// (reference minimal.cc:36-177, :209-306) and their C-ABI entry points.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "rsdsfm_internal.hpp"

namespace rsdsfm {

namespace {


inline uint64_t splitmix64(uint64_t& state) {
    uint64_t z = (state += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

// The reference sampler (minimal.cc:226-244): a persistent index permutation, 9 partial Fisher-Yates draws per
// trial (swap the drawn slot with the last live slot).  rand() is replaced by splitmix64(seed) (quirk Q1: the
// reference reseeds with time(NULL) inside the loop and is not reproducible).  The permutation is kept sparse:
// only 9 T slots ever differ from the identity.
// The permutation the reference shuffles in place (an index vector of n entries) is kept sparse: only touched positions are
// stored, in a flat open-addressing table (at most 18 T entries; a node-based map cost ~40 us per solve on the host, in the
// middle of the frame pipeline with the GPU idle).
void sample_indices(int64_t n, int T, uint64_t seed, std::vector<int32_t>& out) {
    // (the table outlives the call -- one per host thread -- and its slots carry the number of the call that wrote them: no clearing
    // pass; allocating and filling 48 KB per solve was a fifth of this function's 4.7 us, which the GPU spends idle)
    struct Slot {
        int64_t key;
        int32_t val;
        uint32_t gen;
    };
    thread_local std::vector<Slot> table;
    thread_local uint32_t gen = 0;
    size_t cap = 64;
    while (cap < (size_t)T * 18 * 4) cap <<= 1;
    if (table.size() != cap || gen == 0xFFFFFFFFu) {
        table.assign(cap, Slot{-1, 0, 0});
        gen = 0;
    }
    gen += 1;
    Slot* tb = table.data();
    const uint32_t g = gen;
    auto slot = [&](int64_t i) -> size_t {
        size_t h = (size_t)(((uint64_t)i * 0x9E3779B97F4A7C15ull) >> 20) & (cap - 1);
        while (tb[h].gen == g && tb[h].key != i) h = (h + 1) & (cap - 1);
        return h;
    };
    auto get = [&](int64_t i) -> int32_t {
        const size_t h = slot(i);
        return tb[h].gen == g ? tb[h].val : (int32_t)i;
    };
    auto put = [&](int64_t i, int32_t v) {
        const size_t h = slot(i);
        tb[h].key = i;
        tb[h].val = v;
        tb[h].gen = g;
    };
    out.resize((size_t)T * 9);
    uint64_t st = seed;
    for (int t = 0; t < T; ++t) {
        int64_t n_temp = n;
        for (int j = 0; j < 9; ++j) {
            const int64_t r = (int64_t)(splitmix64(st) % (uint64_t)n_temp);
            const int32_t a = get(n_temp - 1), b = get(r);
            put(n_temp - 1, b);
            put(r, a);
            out[(size_t)t * 9 + j] = b;
            n_temp--;
        }
    }
}

}  // namespace

void sample_indices(int64_t n, int T, uint64_t seed, int32_t* out) {
    std::vector<int32_t> v;
    sample_indices(n, T, seed, v);
    if (!v.empty()) memcpy(out, v.data(), sizeof(int32_t) * v.size());
}

// pinned host memory a RANSAC with T trials uses (the end of the block, kPinnedTail bytes, belongs to the frame solve)
size_t ransac_pinned_bytes(int T) {
    const size_t Tn = (size_t)std::max(T, 1);
    return sizeof(RansacBest) + 8 * sizeof(int) + sizeof(double) * 2 * Tn + sizeof(double) * 8 * Tn + sizeof(LmState) * Tn + 64 +
           sizeof(int32_t) * Tn * 9 + kPinnedTail;
}

// Device-resident RANSAC.  d_* inputs and the arrays of `out` are device pointers (any of the out arrays may be
// NULL); scalars of `out` and its trial_* arrays (host) are filled after one final synchronisation.
// spec_tail (optional): work of the CALLER that only needs the RANSAC's device-resident result (RansacBest + the compacted inlier
// arrays).  It is enqueued behind the SPECULATED final stage -- i.e. before the host has read the round-0 flags -- and
// *spec_tail_held tells the caller afterwards whether that final stage was the one that counts; if not (more LM rounds or a scoring
// pass were needed) what the tail computed is garbage and the caller starts over from the host-side result.  The frame solve hands in
// the start of its refinement (refine_begin), which removes the host round trip between the two stages.
//
// The run is a small resumable state machine (RansacRun, rsdsfm_internal.hpp): ransac_begin enqueues everything up to the FIRST point
// where the host has to wait (round 0 of the first hypothesis batch with its speculated final stage and the caller's tail; the whole
// run in closed-form mode) and returns without waiting; ransac_finish waits and drives the rest.  ransac_device = both.  The sequence
// solve (frame_host.hip) begins the next frame pair on another stream between the two halves.
namespace {

enum RansacPc { kPcStart = 0, kPcBatchBegin, kPcRoundEnqueue, kPcRoundWait, kPcAfterRounds, kPcScore, kPcFinal, kPcFinalWait, kPcDone };

// a guard of the analytic pass tripped: the depth solves of the run start over on the iterate-by-iterate kernels, from the hypotheses
int lma_start_over(Ctx* c, RansacRun& R, int guards) {
    R.analytic = false;
    R.lma_restarted = true;
    R.lma_guard |= guards;
    R.tie_margin = -kLmaTie;  // (from here on a tie is only reported)
    R.not_one_step = 0;
    R.final_done = R.spec_scored = R.tail_enqueued = R.spec_final = false;
    RSDSFM_HIP_CHECK(c, hipMemsetAsync(R.zero_begin, 0, R.zero_bytes, c->stream));  // states, scored, flags, list counters
    R.b0 = 0;
    R.pc = kPcBatchBegin;
    return RSDSFM_OK;
}

// runs the state machine; with yield_at_wait it returns (RSDSFM_OK, R.pc != kPcDone) in front of the first host wait
int ransac_advance(Ctx* c, RansacRun& R, bool yield_at_wait) {
    rsdsfm_ransac_out* out = R.out;
    const int T = R.T, Tn = R.Tn, batch = R.batch, depth_mode = R.depth_mode;
    const int64_t n = R.n;
    const double tol = R.tol;
    int rc = RSDSFM_OK;
    for (;;) {
        switch (R.pc) {
            case kPcStart: {
                // (with T > 0 the region is cleared by the workgroups of minimal9_kernel: one launch less in front of the solver)
                const bool zero_in_minimal9 = T > 0 && R.zero_bytes % 8 == 0 && (reinterpret_cast<uintptr_t>(R.zero_begin) & 7) == 0;
                if (!zero_in_minimal9) RSDSFM_HIP_CHECK(c, hipMemsetAsync(R.zero_begin, 0, R.zero_bytes, c->stream));
                if (T > 0) {
                    memcpy(R.h_samples_pinned, R.samples.data(), sizeof(int32_t) * (size_t)T * 9);
                    // the wave-per-hypothesis SVD through the in-range function cores (minimal9_kernels.hip), where round 0 of the LM depth
                    // solves is there to pass its flag on
                    Minimal9Direct dir = R.direct ? *R.direct : Minimal9Direct();
                    R.core_epoch = 0;
                    if (R.core_math && depth_mode == RSDSFM_DEPTH_CERES_LM && T <= c->num_cus * 2 && c->d_core_flag) {
                        c->core_epoch = c->core_epoch >= 0x3fffffff ? 1 : c->core_epoch + 1;
                        dir.core_flag = c->d_core_flag;
                        dir.core_epoch = R.core_epoch = c->core_epoch;
                    }
                    if (R.direct && R.dense && T <= c->num_cus * 2)  // one launch: T solver workgroups + the dense flatten's
                        rc = minimal9_flatten_launch(c, R.h_samples_pinned, T, R.use_alpha_k, R.k_sign_mode, R.d_hyp, zero_in_minimal9 ? R.zero_begin : nullptr,
                                                     zero_in_minimal9 ? R.zero_bytes : 0, dir, R.dense->thr, R.dense->d_q, R.dense->d_u, R.dense->d_alpha,
                                                     R.dense->d_alpha_k, R.dense->d_counters, R.dense->total_out);
                    else
                        rc = minimal9_launch(c, R.d_q, R.d_u, R.d_a, R.d_ak, R.h_samples_pinned, T, R.use_alpha_k, R.k_sign_mode, R.d_hyp,
                                             zero_in_minimal9 ? R.zero_begin : nullptr, zero_in_minimal9 ? R.zero_bytes : 0, &dir);
                    if (rc != RSDSFM_OK) return rc;
                    if (R.after_minimal9) {  // (the frame solve: join the stream that ran the flatten beside the minimal solver)
                        rc = (*R.after_minimal9)();
                        if (rc != RSDSFM_OK) return rc;
                    }
                }
                R.b0 = 0;
                R.pc = kPcBatchBegin;
                break;
            }
            case kPcBatchBegin: {
                if (R.b0 >= T) {
                    R.pc = kPcFinal;
                    break;
                }
                R.B = std::min(batch, T - R.b0);
                R.need_score = true;
                if (depth_mode == RSDSFM_DEPTH_CERES_LM) {
                    if (R.b0 > 0)  // batch 0: cleared above (the flag words and the analytic pass's list counters are adjacent)
                        RSDSFM_HIP_CHECK(c, hipMemsetAsync(R.d_flags, 0, (size_t)(reinterpret_cast<char*>(R.d_irr_count + R.batch) - reinterpret_cast<char*>(R.d_flags)), c->stream));
                    R.round = 0;
                    R.pc = kPcRoundEnqueue;
                } else {
                    R.pc = kPcScore;
                }
                break;
            }
            case kPcRoundEnqueue: {
                const int b0 = R.b0, B = R.B, round = R.round;
                bool flags_via_pick = false;  // this round's flag words reach the host with the speculated pick kernel
                if (round > 4 * kMaxIter) return fail(c, RSDSFM_ERR_NUMERIC, "LM state machines did not terminate");
                if (R.analytic && round == 0) {
                    // the B depth solves on the analytic LM trajectory (ransac_lma_kernels.hip): one pixel pass + one decide launch; hypotheses
                    // that end where no score was fused go to the scoring pass like round 0's, and a hypothesis whose own guards tripped comes
                    // back as "still running": rounds 1, 2, ... below are the iterate-by-iterate kernels', where only such hypotheses take part
                    rc = ransac_lma_launch(c, R.d_q, R.d_u, R.d_a, R.d_ak, n, R.d_hyp + (size_t)b0 * 8, B, R.d_states + b0, R.d_partials, R.d_flags,
                                           R.d_scored + b0, R.d_tcount + b0, R.d_terr + b0, tol, R.lma_cand, 2, R.d_irr_count, R.d_irr_list, R.d_unscored,
                                           nullptr, R.core_epoch ? c->d_core_flag : nullptr, R.core_epoch, R.count_only);
                } else {
                    rc = ransac_lm_round_launch(c, R.d_q, R.d_u, R.d_a, R.d_ak, n, R.d_hyp + (size_t)b0 * 8, B, R.d_states + b0, R.d_partials, R.d_flags,
                                                R.d_scored + b0, R.d_tcount + b0, R.d_terr + b0, round, tol, R.k0, R.fused_base, R.core_math,
                                                R.core_epoch ? c->d_core_flag : nullptr, R.core_epoch, R.d_unscored);
                }
                if (rc != RSDSFM_OK) return rc;
                if (round == 0 && B == T) {
                    // one batch, and on typical data every hypothesis is decided and scored by round 0: the final stage
                    // (best trial, its rho + mask, compaction) is enqueued BEFORE the host reads the flags, which saves a
                    // host round trip with an idle GPU; if the flags say otherwise its output is simply recomputed below.
                    // Where the context's previous solve needed the separate scoring pass (noise-free data: every hypothesis
                    // ends after three accepted steps, which round 0 does not score), that pass is enqueued ahead of the final
                    // stage as well: it only touches hypotheses round 0 left unscored (a no-op on other data) and saves the
                    // round trip plus a discarded final stage (~57 us).  What is enqueued when never changes a result.
                    // Count-only analytic pass (the frame solve): the trials that SHARE the best count get their error sums from that same
                    // pass, but which they are is the pick kernel's to say -- so a pick goes ahead of it too (it leaves the list alone where
                    // the decide stage has put hypotheses on it, and writes it where the counts are complete and tie).
                    const PickLazy lazy = R.count_only ? PickLazy{R.d_scored, R.d_unscored, R.d_flags + 1} : PickLazy();
                    if (c->ransac_score_idle < kScoreIdleLimit) {
                        if (R.count_only) {
                            rc = ransac_pick_launch(c, R.d_tcount, R.d_terr, T, R.d_hyp, R.d_best, nullptr, R.d_flags, nullptr, 0, nullptr, 0, 0, nullptr, 0.0, lazy);
                            if (rc != RSDSFM_OK) return rc;
                        }
                        rc = ransac_score_launch(c, R.d_q, R.d_u, R.d_a, R.d_ak, n, R.d_hyp, T, R.d_states, depth_mode, tol, R.d_scored, R.d_partials,
                                                 R.d_tcount, R.d_terr, R.d_unscored, R.d_flags + 1);
                        if (rc != RSDSFM_OK) return rc;
                        R.spec_scored = true;
                    }
                    rc = ransac_pick_launch(c, R.d_tcount, R.d_terr, T, R.d_hyp, R.d_best, R.h_best, R.d_flags, R.h_running, R.spec_scored ? 1 : 0,  // (+ the flag words)
                                            nullptr, 0, 0, nullptr, R.tie_margin, lazy);
                    if (rc != RSDSFM_OK) return rc;
                    flags_via_pick = true;
                    rc = ransac_final_launch(c, R.d_q, R.d_u, R.d_a, R.d_ak, n, R.d_best, R.d_states, depth_mode, tol, R.d_rho, R.d_mask, R.d_bcounts,
                                             R.d_boffs, out->inlier_idx, out->inliers, out->alpha, out->alpha_k, R.h_best);
                    if (rc != RSDSFM_OK) return rc;
                    R.final_done = true;
                    R.spec_final = true;
                    if (R.spec_tail && R.tail_ahead) {
                        rc = (*R.spec_tail)(R.d_best);
                        if (rc != RSDSFM_OK) return rc;
                        R.tail_enqueued = true;
                    }
                }
                if (!flags_via_pick) RSDSFM_HIP_CHECK(c, hipMemcpyAsync(R.h_running, R.d_flags, sizeof(int) * 8, hipMemcpyDeviceToHost, c->stream));
                R.pc = kPcRoundWait;
                if (yield_at_wait) return RSDSFM_OK;
                break;
            }
            case kPcRoundWait: {
                yield_at_wait = false;  // (a run yields once: after its first wait the caller is blocked in it anyway)
                if (int rcs = sync_stream(c, "ransac_finish line 244")) return rcs;
                const int* h_running = R.h_running;
                if (R.round == 0 && R.analytic && (h_running[3] & 16) != 0) R.lma_guard |= h_running[3] >> 8;  // (hypotheses handed over: diagnostics)
                if (R.round == 0 && R.analytic && (h_running[3] & 6) != 0 && !((h_running[3] & 1) && R.core_math)) {
                    // the analytic pass: a guard tripped (bit 1; which: bits 8..) or the speculated pick met a tie it must not break (bit 2):
                    // the depth solves start over on the iterate-by-iterate kernels, from the hypotheses (which are fine).  What was
                    // enqueued behind on speculation has left at once (the pick marks such a run undecided).
                    rc = lma_start_over(c, R, (h_running[3] >> 8) | ((h_running[3] & 4) ? (1 << 7) : 0));
                    if (rc != RSDSFM_OK) return rc;
                    break;
                }
                if (R.round == 0 && R.core_math && (h_running[3] & 1) != 0) {
                    // round 0 met an argument outside the range of its in-range function cores (ransac_lm_kernel CORE: a zero Jacobian,
                    // a zero or non-finite error -- not on real data): its sums may differ from the standard functions', so the LM rounds
                    // start over from the hypotheses with the standard functions.  What was enqueued behind on speculation has left at
                    // once (the pick kernel marks such a run undecided).
                    R.core_math = false;
                    R.restarted = true;
                    R.not_one_step = 0;
                    R.final_done = R.spec_scored = R.tail_enqueued = R.spec_final = false;
                    R.dense = nullptr;  // (the flatten that rode in the minimal solver's launch has run)
                    R.after_minimal9 = nullptr;  // (and so has a flatten the caller enqueued behind / beside the solver: q, u, alpha are valid)
                    R.pc = kPcStart;    // the hypotheses too: the minimal solver's SVD may have been the one (it clears the states again)
                    break;
                }
                if (R.round == 0) {
                    R.not_one_step += h_running[2];
                    if (R.b0 == 0) {  // where this solve's hypotheses ended: the state the context's next solve fuses the score of
                        int best = 0;
                        for (int a2 = 1; a2 <= 3; ++a2)
                            if (h_running[4 + a2] > (best ? h_running[4 + best] : 0)) best = a2;
                        if (best) R.fused_base_next = best;
                        // (analytic pass: the two iterates where most hypotheses ended; the second defaults to the neighbour below / above)
                        int second = 0;
                        for (int a2 = 1; a2 <= 3; ++a2)
                            if (a2 != best && h_running[4 + a2] > (second ? h_running[4 + second] : 0)) second = a2;
                        if (best) {
                            R.lma_cand_next[0] = best;
                            R.lma_cand_next[1] = second ? second : (best > 1 ? best - 1 : 2);
                        }
                    }
                }
                if (h_running[0] == 0) {
                    R.pc = kPcAfterRounds;
                } else {
                    R.final_done = false;  // more LM rounds: the speculated final stage saw incomplete trials
                    R.round += 1;
                    R.pc = kPcRoundEnqueue;
                }
                break;
            }
            case kPcAfterRounds: {
                R.need_score = R.h_running[1] > 0;  // hypotheses whose final iterate is not the fused one-step state
                if (R.B == T) R.score_hint_next = R.need_score ? 1 : 0;
                // round 0 decided everything and the pass already ran (unless the pick behind it found trials sharing the best count without their error sums)
                if (R.need_score && R.final_done && R.spec_scored && !R.h_best->lazy_pending) R.need_score = false;
                if (R.need_score) R.final_done = false;
                R.pc = kPcScore;
                break;
            }
            case kPcScore: {
                if (R.need_score) {
                    rc = ransac_score_launch(c, R.d_q, R.d_u, R.d_a, R.d_ak, n, R.d_hyp + (size_t)R.b0 * 8, R.B, R.d_states + R.b0, depth_mode, tol,
                                             depth_mode == RSDSFM_DEPTH_CERES_LM ? R.d_scored + R.b0 : nullptr, R.d_partials, R.d_tcount + R.b0,
                                             R.d_terr + R.b0, depth_mode == RSDSFM_DEPTH_CERES_LM ? R.d_unscored : nullptr, R.d_flags + 1);
                    if (rc != RSDSFM_OK) return rc;
                }
                R.b0 += batch;
                R.pc = kPcBatchBegin;
                break;
            }
            case kPcFinal: {
                // best trial, its dense rho + mask, order-preserving compaction
                if (!R.final_done) {
                    R.tail_enqueued = false;  // the speculated final stage (and whatever was enqueued behind it) saw incomplete trials
                    R.spec_final = false;
                    rc = ransac_pick_launch(c, R.d_tcount, R.d_terr, T, R.d_hyp, R.d_best, R.h_best, nullptr, nullptr, 0, nullptr, 0, 0, nullptr, R.tie_margin,  // h_best: host-mapped, written by the kernels
                                            R.count_only ? PickLazy{R.d_scored, R.d_unscored, R.d_flags + 1} : PickLazy());
                    if (rc != RSDSFM_OK) return rc;
                    rc = ransac_final_launch(c, R.d_q, R.d_u, R.d_a, R.d_ak, n, R.d_best, R.d_states, depth_mode, tol, R.d_rho, R.d_mask, R.d_bcounts,
                                             R.d_boffs, out->inlier_idx, out->inliers, out->alpha, out->alpha_k, R.h_best);
                    if (rc != RSDSFM_OK) return rc;
                }
                // the caller's tail behind the DEFINITIVE final stage, where it is not already behind a speculated one that held: it
                // starts from the device-resident result without a host round trip in between
                if (R.spec_tail && !R.tail_enqueued) {
                    rc = (*R.spec_tail)(R.d_best);
                    if (rc != RSDSFM_OK) return rc;
                    R.tail_enqueued = true;
                }
                // per-trial diagnostics are copied back only when the caller asked for them (the frame solve does not)
                if (T > 0 && out->trial_count) RSDSFM_HIP_CHECK(c, hipMemcpyAsync(R.h_tcount, R.d_tcount, sizeof(double) * T, hipMemcpyDeviceToHost, c->stream));
                if (T > 0 && out->trial_err) RSDSFM_HIP_CHECK(c, hipMemcpyAsync(R.h_terr, R.d_terr, sizeof(double) * T, hipMemcpyDeviceToHost, c->stream));
                if (T > 0 && out->trial_vel) RSDSFM_HIP_CHECK(c, hipMemcpyAsync(R.h_hyp, R.d_hyp, sizeof(double) * 8 * T, hipMemcpyDeviceToHost, c->stream));
                if (T > 0 && out->trial_steps && depth_mode == RSDSFM_DEPTH_CERES_LM)
                    RSDSFM_HIP_CHECK(c, hipMemcpyAsync(R.h_states, R.d_states, sizeof(LmState) * T, hipMemcpyDeviceToHost, c->stream));
                R.pc = kPcFinalWait;
                if (yield_at_wait) return RSDSFM_OK;
                break;
            }
            case kPcFinalWait: {
                if (int rcs = sync_stream(c, "ransac_finish line 345")) return rcs;
                const RansacBest* h_best = R.h_best;
#ifdef RSDSFM_DEBUG_HOOKS  // (debug builds only: -DRSDSFM_DEBUG_HOOKS)
                if (getenv("RSDSFM_RANSAC_DEBUG")) fprintf(stderr, "[ransac] final wait: lazy_pending %d undecided %d flags %d %d %d %d hist %d %d %d %d spec_scored %d spec_final %d tail %d idle %d\n", h_best->lazy_pending, h_best->undecided, R.h_running[0], R.h_running[1], R.h_running[2], R.h_running[3], R.h_running[4], R.h_running[5], R.h_running[6], R.h_running[7], (int)R.spec_scored, (int)R.spec_final, (int)R.tail_enqueued, c->ransac_score_idle);
#endif
                if (R.count_only && h_best->lazy_pending) {
                    // the definitive pick found several trials sharing the best count, some without their error sum (everything behind it has
                    // left at once): the scoring pass on the list the pick wrote, then the final stage again
                    if (++R.lazy_rounds > T + 2) return fail(c, RSDSFM_ERR_NUMERIC, "lazy error sums did not settle");
                    R.final_done = R.tail_enqueued = R.spec_final = false;
                    R.need_score = true;
                    R.score_hint_next = 1;
                    R.b0 = 0;
                    R.B = T;
                    R.pc = kPcScore;
                    break;
                }
                if (R.analytic && h_best->lma_tie) {  // guard (d) at the definitive pick (everything behind it has left at once)
                    rc = lma_start_over(c, R, 1 << 7);
                    if (rc != RSDSFM_OK) return rc;
                    break;
                }
                R.lma_tie_seen = h_best->lma_tie != 0;
                R.shared_best = h_best->shared_best;
                if (h_best->num_inliers != h_best->num_inliers_scan) {
                    // analytic pass: the winner's mask comes from the reference's exact replay (ransac_final_kernel) and must count what the
                    // closed form counted -- the guards' claim, checked on every run.  A difference is not an error of the data: start over.
                    if (R.analytic) {
                        rc = lma_start_over(c, R, 1 << 10);
                        if (rc != RSDSFM_OK) return rc;
                        break;
                    }
                    return fail(c, RSDSFM_ERR_NUMERIC, "inlier count mismatch between scoring and compaction");
                }
                out->num_inliers = h_best->num_inliers;
                out->best_trial = h_best->best_trial;
                memcpy(out->w, &h_best->hyp[0], 3 * sizeof(double));
                memcpy(out->v, &h_best->hyp[3], 3 * sizeof(double));
                out->k = h_best->hyp[6];
                out->inlier_error = h_best->inlier_error;
                for (int t = 0; t < T; ++t) {
                    if (out->trial_count) out->trial_count[t] = (int64_t)R.h_tcount[t];
                    if (out->trial_err) out->trial_err[t] = R.h_terr[t];
                    if (out->trial_vel) memcpy(out->trial_vel + (size_t)7 * t, R.h_hyp + (size_t)8 * t, 7 * sizeof(double));
                    if (out->trial_steps) out->trial_steps[t] = depth_mode == RSDSFM_DEPTH_CERES_LM ? R.h_states[t].num_successful : 1;
                }
                if (R.spec_tail_held) *R.spec_tail_held = R.tail_enqueued;
                // the scheduling hints of the context's NEXT solve (never a result).  A run the caller discards -- the frame solve's
                // speculation on a dense flow that turned out to have dropped pixels ran on garbage -- leaves them alone.
                R.hints_ready = true;
                R.pc = kPcDone;
                (void)Tn;
                return RSDSFM_OK;
            }
            default:
                return RSDSFM_OK;
        }
    }
}

}  // namespace

void ransac_commit_hints(Ctx* c, const RansacRun& R) {
    if (!R.hints_ready || R.depth_mode != RSDSFM_DEPTH_CERES_LM || R.T <= 0) return;
    if (R.fused_base_next) c->ransac_fused_base = R.fused_base_next;
    if (R.score_hint_next >= 0) c->ransac_score_idle = R.score_hint_next ? 0 : std::min(c->ransac_score_idle + 1, kScoreIdleLimit);
    c->ransac_not_one_step = R.not_one_step;
    c->ransac_spec_miss = R.spec_final ? 0 : std::min(c->ransac_spec_miss + 1, 2);
    // (the analytic pass's fused iterates stay {2, 1} -- Ctx::lma_cand -- whatever the previous solve's step histogram was: which iterates are fused
    // decides whose error sums come from the closed form and whose from the exact scoring pass, and those differ in their last bits; a
    // result must not depend on what the context solved before.  R.lma_cand_next is kept as a diagnostic only.)
    if (R.count_only) c->lma_count_only_runs += 1, c->lma_lazy_runs += (R.lazy_rounds > 0 || R.shared_best > 1) ? 1 : 0;
    if (R.lazy_rounds > 0 || R.shared_best > 1) c->lma_unique_run = 0;
    else if (R.shared_best == 1) c->lma_unique_run = std::min(c->lma_unique_run + 1, 2);
    if (R.lma_guard) c->lma_last_guard = R.lma_guard, c->lma_handed_over += 1;
    if (R.lma_restarted) {  // a global guard tripped (a tie, the count check): this kind of data stays on the iterate-by-iterate kernels for a while
        c->lma_hold = 16;
        c->lma_restarts += 1;
        c->lma_last_guard = R.lma_guard;
    } else if (c->lma_hold > 0 && !R.analytic) {
        // an iterate-by-iterate run inside a hold: it goes on while the data keeps showing ties the analytic arithmetic cannot break
        c->lma_hold = R.lma_tie_seen ? 16 : c->lma_hold - 1;
    }
    if (R.restarted) {
        c->ransac_standard_math = 16;
        c->ransac_restarts += 1;
    } else if (c->ransac_standard_math > 0) {
        c->ransac_standard_math -= 1;
    }
}

int ransac_begin(Ctx* c, const double* d_q, const double* d_u, const double* d_a, const double* d_ak, int64_t n, int use_alpha_k, int T,
                 double tol, const int32_t* h_samples, uint64_t seed, int depth_mode, int k_sign_mode, rsdsfm_ransac_out* out,
                 const RansacSpecTail* spec_tail, bool* spec_tail_held, RansacRun* run, const Minimal9Direct* direct,
                 const std::function<int()>* after_minimal9, const DenseFlatten* dense, bool tail_ahead, bool count_only) {
    RansacRun& R = *run;
    R = RansacRun();
    R.tail_ahead = tail_ahead;
    R.core_math = c->ransac_math_mode == 0 && c->ransac_standard_math == 0;
    R.analytic = depth_mode == RSDSFM_DEPTH_CERES_LM && T > 0 && c->lm_arithmetic == 0 && c->lma_hold == 0;
    // guard (d): the analytic pass must not break a tie (+); an iterate-by-iterate run of a context that may go back to the analytic pass
    // reports such ties (-: they renew the hold)
    R.tie_margin = depth_mode != RSDSFM_DEPTH_CERES_LM || c->lm_arithmetic != 0 ? 0.0 : (R.analytic ? kLmaTie : -kLmaTie);
    R.lma_cand[0] = c->lma_cand[0], R.lma_cand[1] = c->lma_cand[1];
    // count-only analytic pass + lazy error sums (ransac_lma_kernel ERR = false, ransac_pick_kernel): for callers that do not read the trials' error
    // sums (the frame solve); one hypothesis batch.  No tie guard there: where a tie has to be broken the sums are the reference arithmetic's.
    // Only behind two solves whose best count was unique (a selective tolerance): with a permissive one -- BASELINE's 0.05 admits every
    // pixel under any good hypothesis -- the error sums decide every solve and the fused ones are far cheaper than the scoring pass.
    R.count_only = count_only && R.analytic && T <= kRansacBatch && (c->lma_unique_run >= 2 || c->lma_count_only_force);
    if (R.count_only) R.tie_margin = 0.0;
    if (spec_tail_held) *spec_tail_held = false;
    if (!out) return fail(c, RSDSFM_ERR_INVALID, "null out");
    if (n < 9) return fail(c, RSDSFM_ERR_INVALID, "ransac needs at least 9 points (the reference would compute rand() % 0)");
    if (T < 0) return fail(c, RSDSFM_ERR_INVALID, "negative iterations");
    if (depth_mode != RSDSFM_DEPTH_CLOSED_FORM && depth_mode != RSDSFM_DEPTH_CERES_LM) return fail(c, RSDSFM_ERR_INVALID, "unknown depth_mode");
    if (n > (int64_t)INT32_MAX) return fail(c, RSDSFM_ERR_INVALID, "n exceeds the int32 sample index range");
    if (T > 0) {
        if (h_samples) {
            R.samples.assign(h_samples, h_samples + (size_t)T * 9);
            for (int32_t s : R.samples)
                if (s < 0 || s >= n) return fail(c, RSDSFM_ERR_INVALID, "sample index out of range");
        } else {
            sample_indices(n, T, seed, R.samples);
        }
    }
    const int Tn = std::max(T, 1);
    const int batch = std::min(Tn, kRansacBatch);
    const size_t partials_doubles = std::max<size_t>((size_t)ransac_lm_partials_doubles(c, n, batch), (size_t)ransac_lma_partials_doubles(c, n, batch));
    size_t need = Arena::need(sizeof(double) * Tn * 8) + Arena::need(sizeof(int) * batch) + Arena::need(sizeof(int) * ransac_lma_list_ints(batch)) +
                  Arena::need(sizeof(LmState) * Tn) + Arena::need(sizeof(double) * partials_doubles) +
                  Arena::need(sizeof(int) * 8) + 2 * Arena::need(sizeof(int) * Tn) + 2 * Arena::need(sizeof(double) * Tn) + Arena::need(sizeof(RansacBest)) +
                  2 * Arena::need(sizeof(int64_t) * 2048) + Arena::need(sizeof(double) * (size_t)n) + Arena::need((size_t)n) + 4096;
    int rc = ensure_ws(c, need);
    if (rc != RSDSFM_OK) return rc;
    Arena ws(c->d_ws);
    R.d_hyp = ws.take<double>((size_t)Tn * 8);
    // states, scored and flags are adjacent so that ONE memset clears them
    R.zero_begin = ws.base + ws.off;
    R.d_states = ws.take<LmState>(Tn);
    R.d_scored = ws.take<int>(Tn);
    R.d_flags = ws.take<int>(8);  // {running, unscored, not finished with <= 1 accepted step, restart bits, ended after 0 / 1 / 2 / >= 3 accepted steps}
    R.d_irr_count = ws.take<int>(batch);  // analytic pass: listed pixels per hypothesis of the batch
    R.zero_bytes = (size_t)((ws.base + ws.off) - R.zero_begin);
    R.d_irr_list = ws.take<int>(ransac_lma_list_ints(batch));
    R.d_unscored = ws.take<int>(batch);
    R.d_partials = ws.take<double>(partials_doubles);
    R.d_tcount = ws.take<double>(Tn);
    R.d_terr = ws.take<double>(Tn);
    R.d_best = ws.take<RansacBest>(1);
    R.d_bcounts = ws.take<int64_t>(2048);
    R.d_boffs = ws.take<int64_t>(2048);
    R.d_rho = out->inv_depth ? out->inv_depth : ws.take<double>((size_t)n);
    R.d_mask = out->mask ? out->mask : ws.take<uint8_t>((size_t)n);

    rc = ensure_pinned(c, ransac_pinned_bytes(T));
    if (rc != RSDSFM_OK) return rc;
    if (!c->d_core_flag) {
        RSDSFM_HIP_CHECK(c, hipMalloc(reinterpret_cast<void**>(&c->d_core_flag), 64));
        RSDSFM_HIP_CHECK(c, hipMemsetAsync(c->d_core_flag, 0, 64, c->stream));
    }
    char* hp = static_cast<char*>(c->h_pinned);
    R.h_best = reinterpret_cast<RansacBest*>(hp);
    R.h_running = reinterpret_cast<int*>(hp + sizeof(RansacBest));
    R.h_tcount = reinterpret_cast<double*>(hp + sizeof(RansacBest) + 32);
    R.h_terr = R.h_tcount + Tn;
    R.h_hyp = R.h_terr + Tn;
    R.h_states = reinterpret_cast<LmState*>(R.h_hyp + (size_t)8 * Tn);
    // the sample table stays in host-mapped pinned memory: minimal9_kernel reads its 9 indices per hypothesis from there
    R.h_samples_pinned = reinterpret_cast<int32_t*>(R.h_states + Tn);

    R.d_q = d_q, R.d_u = d_u, R.d_a = d_a, R.d_ak = d_ak;
    R.n = n, R.T = T, R.Tn = Tn, R.batch = batch, R.tol = tol, R.depth_mode = depth_mode, R.use_alpha_k = use_alpha_k, R.k_sign_mode = k_sign_mode;
    R.out = out, R.spec_tail = spec_tail, R.spec_tail_held = spec_tail_held;
    R.direct = direct, R.after_minimal9 = after_minimal9, R.dense = dense;
    // speculation depth of round 0 (see ransac_kernels.hip): explicit, or two iterations behind a solve none of whose hypotheses
    // went beyond one accepted step (outlier-dominated costs), three otherwise.  Like fused_base below: scheduling only.
    R.k0 = c->ransac_k0 != 0 ? c->ransac_k0 : (c->ransac_not_one_step == 0 ? 2 : (int)KMAX);
    // the speculated iterate whose inlier score round 0 fuses: where most hypotheses of the context's previous solve ended (two
    // iterations can only confirm an end after ONE accepted step).  A scheduling decision: the results do not depend on it.
    R.fused_base = R.k0 == 2 ? 1 : std::min(std::max(c->ransac_fused_base, 1), (int)KMAX);
    R.pc = kPcStart;
    return ransac_advance(c, R, true);
}

int ransac_finish(Ctx* c, RansacRun* run) { return ransac_advance(c, *run, false); }

int ransac_device(Ctx* c, const double* d_q, const double* d_u, const double* d_a, const double* d_ak, int64_t n,
                  int use_alpha_k, int T, double tol, const int32_t* h_samples, uint64_t seed, int depth_mode,
                  int k_sign_mode, rsdsfm_ransac_out* out, const RansacSpecTail* spec_tail, bool* spec_tail_held) {
    RansacRun run;
    int rc = ransac_begin(c, d_q, d_u, d_a, d_ak, n, use_alpha_k, T, tol, h_samples, seed, depth_mode, k_sign_mode, out, spec_tail, spec_tail_held,
                          &run, nullptr, nullptr);
    if (rc == RSDSFM_OK) rc = ransac_finish(c, &run);
    if (rc == RSDSFM_OK) ransac_commit_hints(c, run);
    return rc;
}

}  // namespace rsdsfm

using namespace rsdsfm;

extern "C" {

int rsdsfm_calculate_velocities(rsdsfm_ctx* ctx, const double* q, const double* u, const double* alpha, const double* alpha_k,
                                int32_t count, int use_alpha_k, int k_sign_mode, double* w, double* v, double* k) {
    if (!ctx) return RSDSFM_ERR_INVALID;
    Ctx* c = &ctx->c;
    DeviceGuard device_guard_(c);
    if (count < 0 || (count > 0 && (!q || !u || !alpha || !alpha_k || !w || !v || !k))) return fail(c, RSDSFM_ERR_INVALID, "bad arguments");
    if (count == 0) return RSDSFM_OK;
    const size_t T = (size_t)count;
    int rc = ensure_stage(c, 2 * Arena::need(144 * T) + 2 * Arena::need(72 * T) + Arena::need(64 * T));
    if (rc != RSDSFM_OK) return rc;
    Arena sa(c->d_stage);
    double* d_q = sa.take<double>(18 * T);
    double* d_u = sa.take<double>(18 * T);
    double* d_a = sa.take<double>(9 * T);
    double* d_ak = sa.take<double>(9 * T);
    double* d_h = sa.take<double>(8 * T);
    RSDSFM_HIP_CHECK(c, hipMemcpyAsync(d_q, q, 144 * T, hipMemcpyHostToDevice, c->stream));
    RSDSFM_HIP_CHECK(c, hipMemcpyAsync(d_u, u, 144 * T, hipMemcpyHostToDevice, c->stream));
    RSDSFM_HIP_CHECK(c, hipMemcpyAsync(d_a, alpha, 72 * T, hipMemcpyHostToDevice, c->stream));
    RSDSFM_HIP_CHECK(c, hipMemcpyAsync(d_ak, alpha_k, 72 * T, hipMemcpyHostToDevice, c->stream));
    rc = minimal9_launch(c, d_q, d_u, d_a, d_ak, nullptr, count, use_alpha_k, k_sign_mode, d_h);
    if (rc != RSDSFM_OK) return rc;
    std::vector<double> h(8 * T);
    RSDSFM_HIP_CHECK(c, hipMemcpyAsync(h.data(), d_h, 64 * T, hipMemcpyDeviceToHost, c->stream));
    RSDSFM_HIP_CHECK(c, hipStreamSynchronize(c->stream));
    int worst = 0;
    for (size_t t = 0; t < T; ++t) {
        memcpy(w + 3 * t, &h[8 * t], 3 * sizeof(double));
        memcpy(v + 3 * t, &h[8 * t + 3], 3 * sizeof(double));
        k[t] = h[8 * t + 6];
        if (h[8 * t + 7] != 0.0) worst = (int)h[8 * t + 7];
    }
    if (worst != 0) return fail(c, RSDSFM_ERR_NUMERIC, worst == -1 ? "no real eigenvalue k for at least one hypothesis" : "singular system in the k estimation");
    return RSDSFM_OK;
}

int rsdsfm_ransac_dev(rsdsfm_ctx* ctx, const double* d_q, const double* d_u, const double* d_alpha, const double* d_alpha_k,
                      int64_t n, int use_alpha_k, int32_t iterations, double tolerance, const int32_t* samples, uint64_t seed,
                      int depth_mode, int k_sign_mode, rsdsfm_ransac_out* out) {
    if (!ctx) return RSDSFM_ERR_INVALID;
    Ctx* c = &ctx->c;
    DeviceGuard device_guard_(c);
    if (!d_q || !d_u || !d_alpha || !d_alpha_k) return fail(c, RSDSFM_ERR_INVALID, "null device pointer");
    return ransac_device(c, d_q, d_u, d_alpha, d_alpha_k, n, use_alpha_k, iterations, tolerance, samples, seed, depth_mode, k_sign_mode, out,
                         nullptr, nullptr);
}

int rsdsfm_ransac(rsdsfm_ctx* ctx, const double* q, const double* u, const double* alpha, const double* alpha_k, int64_t n,
                  int use_alpha_k, int32_t iterations, double tolerance, const int32_t* samples, uint64_t seed, int depth_mode,
                  int k_sign_mode, rsdsfm_ransac_out* out) {
    if (!ctx) return RSDSFM_ERR_INVALID;
    Ctx* c = &ctx->c;
    DeviceGuard device_guard_(c);
    if (!out || !q || !u || !alpha || !alpha_k) return fail(c, RSDSFM_ERR_INVALID, "null pointer");
    if (n < 9) return fail(c, RSDSFM_ERR_INVALID, "ransac needs at least 9 points (the reference would compute rand() % 0)");
    const size_t N = (size_t)n;
    int rc = ensure_stage(c, 2 * Arena::need(16 * N) + 6 * Arena::need(8 * N) + Arena::need(24 * N) + Arena::need(N) + 4096);
    if (rc != RSDSFM_OK) return rc;
    Arena sa(c->d_stage);
    double* d_q = sa.take<double>(2 * N);
    double* d_u = sa.take<double>(2 * N);
    double* d_a = sa.take<double>(N);
    double* d_ak = sa.take<double>(N);
    rsdsfm_ransac_out dev = *out;
    dev.inlier_idx = sa.take<int64_t>(N);
    dev.inliers = sa.take<double>(3 * N);
    dev.alpha = sa.take<double>(N);
    dev.alpha_k = sa.take<double>(N);
    dev.inv_depth = sa.take<double>(N);
    dev.mask = sa.take<uint8_t>(N);
    // (host_xfer.hip: pinned chunks filled by a few host threads while the DMA engine moves the neighbouring chunk; the kernels queue behind the
    // last chunk on the same stream)
    c->ransac_cache.tag = 0;
    const uint64_t gen = c->stage_gen;
    xfer_trace("begin");
    const XferUp up[4] = {{d_q, q, 16 * N}, {d_u, u, 16 * N}, {d_a, alpha, 8 * N}, {d_ak, alpha_k, 8 * N}};
    if ((rc = xfer_h2d_many(c, up, 4)) != RSDSFM_OK) return rc;
    xfer_trace("uploaded");
    rc = ransac_device(c, d_q, d_u, d_a, d_ak, n, use_alpha_k, iterations, tolerance, samples, seed, depth_mode, k_sign_mode, &dev, nullptr, nullptr);
    if (rc != RSDSFM_OK) return rc;
    xfer_trace("solved");
    const size_t M = (size_t)dev.num_inliers;
    XferItem items[6];
    int ni = 0;
    if (out->inliers && M) items[ni++] = {out->inliers, dev.inliers, 24 * M};
    if (out->inlier_idx && M) items[ni++] = {out->inlier_idx, dev.inlier_idx, 8 * M};
    if (out->alpha && M) items[ni++] = {out->alpha, dev.alpha, 8 * M};
    if (out->alpha_k && M) items[ni++] = {out->alpha_k, dev.alpha_k, 8 * M};
    if (out->inv_depth) items[ni++] = {out->inv_depth, dev.inv_depth, 8 * N};
    if (out->mask) items[ni++] = {out->mask, dev.mask, N};
    if ((rc = xfer_d2h_many(c, items, ni)) != RSDSFM_OK) return rc;
    RSDSFM_HIP_CHECK(c, hipStreamSynchronize(c->stream));
    xfer_trace("downloaded");
    xfer_trace_dump("rsdsfm_ransac");
    // what stays on the device for rsdsfm_refine_from_ransac (refine_host.hip): the inliers, their alpha / alpha_k / indices and u, until another
    // host-pointer call takes the staging buffer
    if (out->inliers && out->alpha && out->alpha_k && out->inlier_idx && gen == c->stage_gen) {
        Ctx::RansacCache& rcache = c->ransac_cache;
        rcache.stage_gen = gen;
        rcache.d_u = d_u, rcache.d_inl = dev.inliers, rcache.d_alpha = dev.alpha, rcache.d_alpha_k = dev.alpha_k, rcache.d_idx = dev.inlier_idx;
        rcache.n = n, rcache.m = (int64_t)M, rcache.h_u = u;
        for (int j = 0; j < 16; ++j) {
            rcache.u_probe[j] = u[(2 * N - 1) * (size_t)j / 15];
            rcache.inl_probe[j] = M ? out->inliers[(3 * M - 1) * (size_t)j / 15] : 0.0;
        }
        rcache.tag = ++c->ransac_tag_counter;
    }
    out->num_inliers = dev.num_inliers;
    out->best_trial = dev.best_trial;
    memcpy(out->w, dev.w, sizeof(dev.w));
    memcpy(out->v, dev.v, sizeof(dev.v));
    out->k = dev.k;
    out->inlier_error = dev.inlier_error;
    return RSDSFM_OK;
}

}  // extern "C"
