// dist_host.hip -- the column-tiled WHOLE solve of one frame over the GPUs of a node, driven from C++ inside the library
// (BASELINE configs[3]).  One process per GPU, one context per process; every rank calls
// rsdsfm_solve_frame_tiled_dev with ITS column slab of the flow image and all ranks return the same pose and the full depth map.
//
// The reference's solver part of evaluateSingleRun (main.cc:398-522) is sequential; what makes it tile is that
//   * the reference flattens the image column-major (main.cc:398-444), so the slabs' point lists in rank order concatenate to the
//     reference's point list and a rank only ever holds its slab;
//   * every global decision (trust-region accept / reject / converge of the per-trial depth solves, minimal.cc:209-306; of the
//     joint refinement, nonlinearRefinement.cc:183-252; the mean-z sign, main.cc:466-481) depends on a handful of sums.  Each rank
//     reduces its slab to one small row, the rows are ALL-GATHERED in rank order (an all-gather, not an all-reduce: the summation
//     order stays fixed, results are bit-reproducible and independent of the transport), and every rank runs the same decide kernel
//     -- the very kernel that reduces per-workgroup partials in the single-GPU solve, with ranks in place of workgroups -- on
//     identical data: identical decisions everywhere, nothing is ever broadcast.
// Exchanges per solve: point counts (8 B), the 9 T sampled points (all-reduce of 54 T doubles with one non-zero term per entry:
// exact), [T][22] sums per RANSAC LM round, [T][2] scores when needed, 17 / 54..70 / 13 doubles per refinement stage, one z sum,
// and ONE all-gather of the depth-map slabs (8 B x pixels) -- the only data-path collective, direct per-link transfers over xGMI.
// Everything is enqueued on the context's stream; the host synchronises 4-5 times per solve (counts, RANSAC flags + winner,
// refinement poll every 5 iterations, final header), never per RANSAC round or LM iteration in the common case.
//
// Transport: RCCL (ncclAllGather / ncclAllReduce on the context's stream), resolved at run time with dlopen so that the library
// loads on hosts without RCCL; the communicator is created here from a unique id the ranks share over any channel
// (rsdsfm_dist_unique_id / rsdsfm_dist_init), or adopted from the caller (rsdsfm_dist_adopt: e.g. an MPI / torch host that already
// owns an ncclComm_t).  rsdsfm_dist_set_transport installs caller-provided collectives instead (used by the tests to run several
// logical ranks on one GPU, and by hosts with another communication library).
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "rsdsfm_internal.hpp"

using namespace rsdsfm;

namespace rsdsfm {
int alpha_ones_launch(Ctx* c, double* d_alpha, int64_t n);
}

namespace {

// ---------------------------------------------------------------------------------------------------
// RCCL, resolved at run time
// ---------------------------------------------------------------------------------------------------
// The handful of RCCL / NCCL 2.x ABI types the driver uses, declared here so that the library BUILDS without the RCCL headers as
// well (it already loads without the library): values as in rccl.h / nccl.h (ncclSuccess 0; ncclInt8 = ncclChar 0, ncclFloat64 =
// ncclDouble 8; ncclSum 0; NCCL_UNIQUE_ID_BYTES 128; the communicator is an opaque pointer).
typedef struct ncclComm* ncclComm_t;
typedef struct {
    char internal[128];
} ncclUniqueId;
typedef int ncclResult_t;
typedef int ncclDataType_t;
typedef int ncclRedOp_t;
constexpr ncclResult_t ncclSuccess = 0;
constexpr ncclDataType_t ncclChar = 0, ncclDouble = 8;
constexpr ncclRedOp_t ncclSum = 0;

struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string error;
};

Rccl* rccl() {
    static Rccl R;
    static std::once_flag once;
    std::call_once(once, [] {
        // an RCCL that is already in the process (PyTorch-ROCm ships its own librccl.so) is preferred: one RCCL per process
        const char* env = getenv("RSDSFM_RCCL_LIB");
        struct Try {
            const char* name;
            int flags;
        } tries[] = {{env, RTLD_NOW | RTLD_LOCAL},
                     {"librccl.so", RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD},
                     {"librccl.so.1", RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD},
                     {"librccl.so.1", RTLD_NOW | RTLD_LOCAL},
                     {"librccl.so", RTLD_NOW | RTLD_LOCAL},
                     {"/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL}};
        for (const Try& t : tries) {
            if (!t.name || !*t.name) continue;
            R.handle = dlopen(t.name, t.flags);
            if (R.handle) break;
        }
        if (!R.handle) {
            R.error = "RCCL not found (librccl.so / librccl.so.1; set RSDSFM_RCCL_LIB)";
            return;
        }
        auto sym = [&](const char* n) { return dlsym(R.handle, n); };
        R.GetUniqueId = reinterpret_cast<decltype(R.GetUniqueId)>(sym("ncclGetUniqueId"));
        R.CommInitRank = reinterpret_cast<decltype(R.CommInitRank)>(sym("ncclCommInitRank"));
        R.CommDestroy = reinterpret_cast<decltype(R.CommDestroy)>(sym("ncclCommDestroy"));
        R.CommAbort = reinterpret_cast<decltype(R.CommAbort)>(sym("ncclCommAbort"));
        R.AllGather = reinterpret_cast<decltype(R.AllGather)>(sym("ncclAllGather"));
        R.AllReduce = reinterpret_cast<decltype(R.AllReduce)>(sym("ncclAllReduce"));
        R.GetErrorString = reinterpret_cast<decltype(R.GetErrorString)>(sym("ncclGetErrorString"));
        if (!R.GetUniqueId || !R.CommInitRank || !R.CommDestroy || !R.AllGather || !R.AllReduce) {
            R.error = "RCCL library lacks a required symbol";
            R.handle = nullptr;
        }
    });
    return &R;
}

// ---------------------------------------------------------------------------------------------------
// per-context distributed state
// ---------------------------------------------------------------------------------------------------

struct Dist {
    int nranks = 1, rank = 0;
    ncclComm_t comm = nullptr;
    bool own_comm = false;
    rsdsfm_all_gather_fn ag = nullptr;  // caller-provided transport (overrides RCCL)
    rsdsfm_all_reduce_sum_f64_fn ar = nullptr;
    void* user = nullptr;
    void* d_buf = nullptr;  // device arena of the small exchange buffers + (when the slab stride pads the image) the gathered depth map
    size_t bytes = 0;
    void* d_session = nullptr;  // refinement session buffers
    size_t session_bytes = 0;
    void* d_flow = nullptr;  // rank-indexed flow (quirk Q2): the gathered heads of the slabs' flow lists + this rank's columns
    size_t flow_bytes = 0;
    void* d_xchg = nullptr;  // {point count, setup status} of every rank: allocated apart from (and before) everything that can fail
    size_t xchg_bytes = 0;
    int host_syncs = 0, collectives = 0, ransac_rounds = 0;  // diagnostics of the last solve
    // what the previous solve on this communicator needed -- decides what the next one enqueues ahead of its host reads, never a
    // result (every rank derives them from the same replicated decisions, so all ranks enqueue the same collectives)
    int score_idle = kScoreIdleLimit;  // consecutive solves (saturating) that did not need the separate scoring pass behind round 0 (rsdsfm_internal.hpp)
    int refine_iters_hint = -1;  // refinement SLOTS the previous solve consumed (one exchange each; -1: none yet)
    // "warm": the previous solve on this communicator succeeded on EVERY rank with exactly this shape, so this one needs no allocation on any
    // rank and its setup cannot fail for lack of memory; with dense_hint (every slab of that solve kept all its pixels) the ranks then
    // go on with the counts a dense frame has instead of waiting for the counts exchange (checked with the RANSAC's first host read)
    bool warm = false, dense_hint = false;
    int warm_rows = 0, warm_cols = 0, warm_T = 0, warm_flags = 0;
    // round 0 of the RANSAC's LM solves and the minimal solver's SVD through the in-range function cores (as in the single-context solve):
    // a shard that met an argument out of range says so in the trailer of its rows, every rank sees it and all start the RANSAC over with
    // the standard functions, which the communicator then keeps for its next 16 solves
    int standard_math = 0;
    int64_t restarts = 0;
    // the RANSAC's depth solves on the analytic LM trajectory (ransac_lma_kernels.hip), as in the single-context solve: a run whose guards trip
    // (every rank reads the same replicated flag words) starts over iterate by iterate, and the communicator stays there for its next solves
    // (renewed by a solve that ends in a tie the analytic arithmetic could not break: noise-free data)
    int lma_hold = 0;
    // the refinement (+ depth-map stage) goes behind the SPECULATED final stage of the RANSAC, from the device-resident winner, while that
    // stage was the one that counted in one of the communicator's last two solves and -- reference flow indexing (quirk Q2) over several ranks --
    // the previous solve needed no remote flow column
    int spec_miss = 2;
    bool q2_local = false;
    bool lockstep_error = false;  // the last solve's error return came after the whole solve ran in lockstep with the peers (hints kept)
};

Dist* dist_of(Ctx* c, bool create) {
    if (!c->dist && create) c->dist = new (std::nothrow) Dist();
    return static_cast<Dist*>(c->dist);
}

int nccl_fail(Ctx* c, ncclResult_t r, const char* what) {
    Rccl* R = rccl();
    c->err = std::string(what) + ": " + (R->GetErrorString ? R->GetErrorString(r) : "RCCL error");
    return RSDSFM_ERR_HIP;
}

// all-gather of `bytes` bytes per rank (rank order) on the context's stream; send may alias recv + rank * bytes
int all_gather(Ctx* c, Dist* D, const void* d_send, void* d_recv, size_t bytes) {
    if (bytes == 0) return RSDSFM_OK;
    D->collectives += 1;
#ifdef RSDSFM_DEBUG_HOOKS  // (debug builds only: -DRSDSFM_DEBUG_HOOKS)
    if (getenv("RSDSFM_TILED_DEBUG")) fprintf(stderr, "[rank %d] collective %d: all_gather %zu bytes\n", D->rank, D->collectives, bytes);
#endif
    if (D->ag) {
        if (D->ag(D->user, d_send, d_recv, bytes, c->stream) != 0) return fail(c, RSDSFM_ERR_HIP, "caller-provided all-gather failed");
        return RSDSFM_OK;
    }
    if (D->comm) {
        ncclResult_t r = rccl()->AllGather(d_send, d_recv, bytes, ncclChar, D->comm, c->stream);
        if (r != ncclSuccess) return nccl_fail(c, r, "ncclAllGather");
        return RSDSFM_OK;
    }
    if (D->nranks != 1) return fail(c, RSDSFM_ERR_INVALID, "no communicator: call rsdsfm_dist_init / rsdsfm_dist_adopt / rsdsfm_dist_set_transport first");
    if (d_send != d_recv) RSDSFM_HIP_CHECK(c, hipMemcpyAsync(d_recv, d_send, bytes, hipMemcpyDeviceToDevice, c->stream));
    return RSDSFM_OK;
}

// in-place sum over ranks of `count` doubles.  Only used where at most one rank contributes a non-zero term per entry (exact).
int all_reduce_sum(Ctx* c, Dist* D, double* d_buf, size_t count) {
    if (count == 0) return RSDSFM_OK;
    D->collectives += 1;
#ifdef RSDSFM_DEBUG_HOOKS
    if (getenv("RSDSFM_TILED_DEBUG")) fprintf(stderr, "[rank %d] collective %d: all_reduce %zu doubles\n", D->rank, D->collectives, count);
#endif
    if (D->ar) {
        if (D->ar(D->user, d_buf, count, c->stream) != 0) return fail(c, RSDSFM_ERR_HIP, "caller-provided all-reduce failed");
        return RSDSFM_OK;
    }
    if (D->comm) {
        ncclResult_t r = rccl()->AllReduce(d_buf, d_buf, count, ncclDouble, ncclSum, D->comm, c->stream);
        if (r != ncclSuccess) return nccl_fail(c, r, "ncclAllReduce");
        return RSDSFM_OK;
    }
    if (D->nranks != 1) return fail(c, RSDSFM_ERR_INVALID, "no communicator");
    return RSDSFM_OK;
}

int ensure_dev(Ctx* c, void** p, size_t* have, size_t bytes) {
    if (bytes <= *have) return RSDSFM_OK;
    RSDSFM_HIP_CHECK(c, hipStreamSynchronize(c->stream));
    if (*p) (void)hipFree(*p);
    *p = nullptr;
    *have = 0;
    RSDSFM_HIP_CHECK(c, hipMalloc(p, bytes));
    *have = bytes;
    return RSDSFM_OK;
}

// the first exchange's buffer ({point count, setup status} of every rank) exists before any solve: allocated where the communicator or
// the transport is installed, so that no solve can fail on one rank before it has told the others
int reserve_xchg(Ctx* c, Dist* D, int nranks) { return ensure_dev(c, &D->d_xchg, &D->xchg_bytes, Arena::need(16 * (size_t)nranks + 64)); }

// the scheduling hints back to their defaults (what a solve that returned an error leaves behind: the ranks may have left at different
// points, and a hint that differs between ranks would make the next solve enqueue different collectives on different ranks)
void reset_hints(Dist* D) {
    D->score_idle = kScoreIdleLimit;
    D->refine_iters_hint = -1;
    D->warm = D->dense_hint = false;
    D->standard_math = 0;
    D->lma_hold = 0;
    D->spec_miss = 2;
    D->q2_local = false;
}

int sync(Ctx* c, Dist* D) {
    D->host_syncs += 1;
    RSDSFM_HIP_CHECK(c, hipStreamSynchronize(c->stream));
    return RSDSFM_OK;
}

// the 9 T sampled points as the minimal solver wants them (q9, u9: [9T][2]; a9, ak9: [9T]), one contiguous block of 54 T doubles:
// a rank writes the points it owns and zeros elsewhere, so the sum over ranks has exactly one non-zero term per entry
// tail (optional): out[6 * count + r] = this rank's point count for r == rank (as a double: exact), 0 for the other ranks -- the counts
// exchange rides in the same all-reduce when the ranks went ahead on the counts of a dense frame (my_count: the flatten's device word)
__global__ __launch_bounds__(256) void pack_samples_kernel(const double2* __restrict__ q, const double2* __restrict__ u,
                                                          const double* __restrict__ alpha, const double* __restrict__ alpha_k,
                                                          int64_t n_local, int64_t offset, const int32_t* __restrict__ samples,
                                                          int count, double* __restrict__ out, const int64_t* __restrict__ my_count, int rank,
                                                          int nranks) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (my_count && j < nranks) out[6 * (size_t)count + j] = j == rank ? (double)*my_count : 0.0;
    if (j >= count) return;
    const int64_t loc = (int64_t)samples[j] - offset;
    const bool mine = loc >= 0 && loc < n_local;
    double2 qq = make_double2(0.0, 0.0), uu = make_double2(0.0, 0.0);
    double a = 0.0, ak = 0.0;
    if (mine) {
        qq = q[loc];
        uu = u[loc];
        a = alpha[loc];
        ak = alpha_k[loc];
    }
    double* q9 = out;
    double* u9 = out + 2 * (size_t)count;
    double* a9 = out + 4 * (size_t)count;
    double* ak9 = out + 5 * (size_t)count;
    q9[2 * j] = qq.x, q9[2 * j + 1] = qq.y;
    u9[2 * j] = uu.x, u9[2 * j + 1] = uu.y;
    a9[j] = a;
    ak9[j] = ak;
}

// Rank-indexed flow (quirk Q2, main.cc:457 -> nonlinearRefinement.cc:209-212): the i-th inlier of the GLOBAL inlier list reads column
// i of the GLOBAL, un-compacted flow list.  This rank's inliers hold the global ranks [prefix, prefix + m); column g of the global
// list is column g - offset_r of the slab r with offset_r <= g < offset_r + cnt_r.  `heads` holds the first `lmax` columns of every
// slab's flow list in rank order (only columns below the global inlier count are ever addressed: the ranks stop there).
__global__ __launch_bounds__(256) void rank_flow_gather_kernel(const double2* __restrict__ heads, int64_t lmax, const int64_t* __restrict__ cnt_all,
                                                              int nranks, int64_t prefix, int64_t m, double2* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const int64_t g = prefix + i;
    int64_t off = 0;
    int r = 0;
    for (; r < nranks - 1; ++r) {
        const int64_t cnt = cnt_all[r];
        if (g < off + cnt) break;
        off += cnt;
    }
    out[i] = heads[(int64_t)r * lmax + (g - off)];
}

}  // namespace

namespace rsdsfm {
// rsdsfm_set_lm_arithmetic: the communicator's hold on the iterate-by-iterate kernels starts over with the switch (like the context's own)
void dist_reset_hold(Ctx* c) {
    if (c->dist) static_cast<Dist*>(c->dist)->lma_hold = 0;
}

void dist_release(Ctx* c) {
    Dist* D = static_cast<Dist*>(c->dist);
    if (!D) return;
    if (D->comm && D->own_comm && rccl()->CommDestroy) (void)rccl()->CommDestroy(D->comm);
    if (D->d_buf) (void)hipFree(D->d_buf);
    if (D->d_session) (void)hipFree(D->d_session);
    if (D->d_flow) (void)hipFree(D->d_flow);
    if (D->d_xchg) (void)hipFree(D->d_xchg);
    delete D;
    c->dist = nullptr;
}
}  // namespace rsdsfm

extern "C" {

int rsdsfm_dist_unique_id(void* id_128_bytes) {
    if (!id_128_bytes) return RSDSFM_ERR_INVALID;
    Rccl* R = rccl();
    if (!R->handle) return RSDSFM_ERR_NO_DEVICE;
    ncclUniqueId id;
    if (R->GetUniqueId(&id) != ncclSuccess) return RSDSFM_ERR_HIP;
    static_assert(sizeof(ncclUniqueId) == RSDSFM_DIST_ID_BYTES, "id size");
    memcpy(id_128_bytes, &id, sizeof(id));
    return RSDSFM_OK;
}

int rsdsfm_dist_init(rsdsfm_ctx* ctx, int32_t nranks, int32_t rank, const void* id_128_bytes) {
    if (!ctx) return RSDSFM_ERR_INVALID;
    Ctx* c = &ctx->c;
    DeviceGuard device_guard_(c);
    if (nranks < 1 || rank < 0 || rank >= nranks || !id_128_bytes) return fail(c, RSDSFM_ERR_INVALID, "bad arguments");
    Rccl* R = rccl();
    if (!R->handle) return fail(c, RSDSFM_ERR_NO_DEVICE, R->error.c_str());
    Dist* D = dist_of(c, true);
    if (!D) return fail(c, RSDSFM_ERR_INVALID, "out of host memory");
    if (D->comm && D->own_comm) (void)R->CommDestroy(D->comm);
    D->comm = nullptr;
    ncclUniqueId id;
    memcpy(&id, id_128_bytes, sizeof(id));
    D->nranks = 1, D->rank = 0, D->own_comm = false;  // (what a failed init leaves behind: a single rank without communicator)
    D->ag = nullptr;
    D->ar = nullptr;
    ncclResult_t r = R->CommInitRank(&D->comm, nranks, id, rank);  // on the context's device (made current by the guard)
    if (r != ncclSuccess) {
        D->comm = nullptr;
        return nccl_fail(c, r, "ncclCommInitRank");
    }
    D->own_comm = true;
    D->nranks = nranks;
    D->rank = rank;
    D->ag = nullptr;
    D->ar = nullptr;
    reset_hints(D);
    return reserve_xchg(c, D, nranks);
}

int rsdsfm_dist_adopt(rsdsfm_ctx* ctx, void* nccl_comm, int32_t nranks, int32_t rank) {
    if (!ctx) return RSDSFM_ERR_INVALID;
    Ctx* c = &ctx->c;
    DeviceGuard device_guard_(c);
    if (!nccl_comm || nranks < 1 || rank < 0 || rank >= nranks) return fail(c, RSDSFM_ERR_INVALID, "bad arguments");
    Rccl* R = rccl();
    if (!R->handle) return fail(c, RSDSFM_ERR_NO_DEVICE, R->error.c_str());
    Dist* D = dist_of(c, true);
    if (!D) return fail(c, RSDSFM_ERR_INVALID, "out of host memory");
    if (D->comm && D->own_comm) (void)R->CommDestroy(D->comm);
    D->comm = static_cast<ncclComm_t>(nccl_comm);
    D->own_comm = false;
    D->nranks = nranks;
    D->rank = rank;
    D->ag = nullptr;
    D->ar = nullptr;
    reset_hints(D);
    return reserve_xchg(c, D, nranks);
}

int rsdsfm_dist_set_transport(rsdsfm_ctx* ctx, int32_t nranks, int32_t rank, rsdsfm_all_gather_fn all_gather_fn,
                              rsdsfm_all_reduce_sum_f64_fn all_reduce_fn, void* user) {
    if (!ctx) return RSDSFM_ERR_INVALID;
    Ctx* c = &ctx->c;
    DeviceGuard device_guard_(c);
    if (nranks < 1 || rank < 0 || rank >= nranks || (nranks > 1 && (!all_gather_fn || !all_reduce_fn))) return fail(c, RSDSFM_ERR_INVALID, "bad arguments");
    Dist* D = dist_of(c, true);
    if (!D) return fail(c, RSDSFM_ERR_INVALID, "out of host memory");
    if (D->comm && D->own_comm && rccl()->CommDestroy) (void)rccl()->CommDestroy(D->comm);
    D->comm = nullptr;
    D->own_comm = false;
    D->nranks = nranks;
    D->rank = rank;
    D->ag = all_gather_fn;
    D->ar = all_reduce_fn;
    D->user = user;
    reset_hints(D);
    return reserve_xchg(c, D, nranks);
}

int rsdsfm_dist_finalize(rsdsfm_ctx* ctx) {
    if (!ctx) return RSDSFM_ERR_INVALID;
    Ctx* c = &ctx->c;
    DeviceGuard device_guard_(c);
    RSDSFM_HIP_CHECK(c, hipStreamSynchronize(c->stream));
    dist_release(c);
    return RSDSFM_OK;
}

int rsdsfm_tiled_slab_bounds(int32_t cols, int32_t nranks, int32_t rank, int32_t* col0, int32_t* slab_cols, int32_t* stride_cols) {
    if (cols < 0 || nranks < 1 || rank < 0 || rank >= nranks) return RSDSFM_ERR_INVALID;
    const int per = (cols + nranks - 1) / nranks;
    const int c0 = std::min(cols, rank * per), c1 = std::min(cols, (rank + 1) * per);
    if (col0) *col0 = c0;
    if (slab_cols) *slab_cols = c1 - c0;
    if (stride_cols) *stride_cols = per;
    return RSDSFM_OK;
}

static int solve_frame_tiled_impl(rsdsfm_ctx* ctx, const double* d_img_slab, int32_t rows, int32_t cols, double fx, double fy, double cx,
                                  double cy, double gamma, const rsdsfm_frame_params* prm, double* d_depth_map, double* d_R_rows9,
                                  double* d_t_rows3, rsdsfm_frame_result* res, rsdsfm_tiled_info* info);

int rsdsfm_solve_frame_tiled_dev(rsdsfm_ctx* ctx, const double* d_img_slab, int32_t rows, int32_t cols, double fx, double fy, double cx,
                                 double cy, double gamma, const rsdsfm_frame_params* prm, double* d_depth_map, double* d_R_rows9,
                                 double* d_t_rows3, rsdsfm_frame_result* res, rsdsfm_tiled_info* info) {
    const int rc = solve_frame_tiled_impl(ctx, d_img_slab, rows, cols, fx, fy, cx, cy, gamma, prm, d_depth_map, d_R_rows9, d_t_rows3, res, info);
    if (rc != RSDSFM_OK && ctx) {  // an error return: the ranks may have left at different points -- no hint survives it
        Dist* D = static_cast<Dist*>(ctx->c.dist);
        // (except an error this rank reported AFTER completing the solve in lockstep with its peers: they know nothing of it and keep their
        // hints, so this rank must keep the same ones)
        if (D && !D->lockstep_error) reset_hints(D);
        if (D) D->lockstep_error = false;
    }
    return rc;
}

static int solve_frame_tiled_impl(rsdsfm_ctx* ctx, const double* d_img_slab, int32_t rows, int32_t cols, double fx, double fy, double cx,
                                  double cy, double gamma, const rsdsfm_frame_params* prm, double* d_depth_map, double* d_R_rows9,
                                  double* d_t_rows3, rsdsfm_frame_result* res, rsdsfm_tiled_info* info) {
    if (!ctx) return RSDSFM_ERR_INVALID;
    Ctx* c = &ctx->c;
    DeviceGuard device_guard_(c);
    if (!prm || !res || rows <= 0 || cols <= 0 || !d_depth_map) return fail(c, RSDSFM_ERR_INVALID, "bad arguments");
    if (prm->flow_index_mode != RSDSFM_FLOW_COMPAT_RANK && prm->flow_index_mode != RSDSFM_FLOW_GATHERED) return fail(c, RSDSFM_ERR_INVALID, "unknown flow_index_mode");
    if (prm->struct_bytes != 0 && prm->struct_bytes != (int32_t)sizeof(rsdsfm_frame_params))
        return fail(c, RSDSFM_ERR_INVALID, "rsdsfm_frame_params: struct_bytes is neither 0 nor sizeof(rsdsfm_frame_params) -- caller built against another header (use rsdsfm_frame_params_init)");
    const int depth_mode = prm->depth_mode;
    if (depth_mode != RSDSFM_DEPTH_CLOSED_FORM && depth_mode != RSDSFM_DEPTH_CERES_LM) return fail(c, RSDSFM_ERR_INVALID, "unknown depth_mode");
    const int T = prm->ransac_trials;
    if (T < 0) return fail(c, RSDSFM_ERR_INVALID, "negative trial count");
    Dist* D = dist_of(c, true);
    if (!D) return fail(c, RSDSFM_ERR_INVALID, "out of host memory");
    D->host_syncs = D->collectives = D->ransac_rounds = 0;
    const int R = D->nranks, rank = D->rank;
    int32_t col0 = 0, sc = 0, per = 0;
    rsdsfm_tiled_slab_bounds(cols, R, rank, &col0, &sc, &per);
    const size_t Ns = (size_t)rows * (size_t)sc;      // pixels of this slab
    const size_t cap = (size_t)rows * (size_t)per;     // pixels of a full-width slab (all-gather stride)
    const size_t N1 = std::max<size_t>(Ns, 1);
    const int Tn = std::max(T, 1);
    const int batch = std::min(Tn, kRansacBatch);
    const int np = prm->use_acceleration_mode ? 7 : 6;
    const int nsr = ransac_rows_doubles();
    const bool padded = (size_t)R * cap != (size_t)rows * (size_t)cols;

    // ---- frame buffers of the slab (context arena shared with rsdsfm_solve_frame_dev) ----
    const size_t need_frame = 2 * Arena::need(16 * N1) + 5 * Arena::need(8 * N1) + 2 * Arena::need(24 * N1) + Arena::need(8 * N1) + Arena::need(N1) +
                              Arena::need(4 * N1) + 4096;
    // A failure that only THIS rank sees (an allocation, a null slab pointer) must not leave the other ranks waiting in the next
    // collective for ever: everything that can fail locally happens in this setup part, its outcome travels with the point counts in
    // the first exchange, and every rank leaves together (RSDSFM_ERR_PEER on the ranks that were fine).  The exchange buffer itself
    // exists since the communicator / transport was installed (reserve_xchg; a context that never had one is a single rank: nobody to
    // strand); buffers whose size depends on later results are sized by their upper bounds here.
    int rc = reserve_xchg(c, D, R);
    if (rc != RSDSFM_OK) return rc;
    int64_t* d_xchg = static_cast<int64_t*>(D->d_xchg);  // [R][2] = {point count, setup status}
    if (Ns > 0 && !d_img_slab) rc = fail(c, RSDSFM_ERR_INVALID, "null slab pointer");
    if (rc == RSDSFM_OK) rc = ensure_dev(c, &c->d_frame, &c->frame_bytes, need_frame);
    Arena fa(c->d_frame);
    double* d_q = fa.take<double>(2 * N1);
    double* d_u = fa.take<double>(2 * N1);
    double* d_a = fa.take<double>(N1);
    double* d_ak = fa.take<double>(N1);
    double* d_in_a = fa.take<double>(N1);
    double* d_in_ak = fa.take<double>(N1);
    double* d_rho = fa.take<double>(N1);
    double* d_inl = fa.take<double>(3 * N1);
    double* d_inl_ref = fa.take<double>(3 * N1);
    int64_t* d_idx = fa.take<int64_t>(N1);
    uint8_t* d_mask = fa.take<uint8_t>(N1);
    int32_t* d_ys = fa.take<int32_t>(N1);

    // ---- small exchange buffers ----
    const int row_max = std::max({nsr * batch + 2, ransac_lma_rows_doubles() * batch, refine_slot_row_doubles(np), refine_rf_row_doubles(np), refine_stage_row_doubles(np, 0), refine_stage_row_doubles(np, 1), refine_stage_row_doubles(np, 2), 2 * batch});
    size_t need_d = 2 * Arena::need(8 * (size_t)R + 64) + Arena::need(8 * (size_t)R * Tn) + Arena::need(4 * 9 * (size_t)Tn) + Arena::need(8 * (54 * (size_t)Tn + (size_t)R + 8)) + Arena::need(8 * 8 * (size_t)Tn) +
                    Arena::need(sizeof(LmState) * Tn) + Arena::need(4 * (size_t)Tn) + Arena::need(64) + 2 * Arena::need(8 * (size_t)Tn) +
                    2 * Arena::need(sizeof(RansacBest)) + Arena::need(8 * (size_t)row_max) + Arena::need(8 * (size_t)row_max * R) + Arena::need(64) +
                    Arena::need(8 * (size_t)R + 64) + Arena::need(64) + (padded ? Arena::need(8 * cap * R) : 0) + Arena::need(4 * (size_t)batch) +
                    Arena::need(4 * ransac_lma_list_ints(batch)) + 4096;
    if (rc == RSDSFM_OK) rc = ensure_dev(c, &D->d_buf, &D->bytes, need_d);
    Arena da(D->d_buf);
    int64_t* d_cnt_all = da.take<int64_t>((size_t)R + 8);
    int64_t* d_m_all = da.take<int64_t>((size_t)R + 8);
    double* d_cnt_rt = da.take<double>((size_t)R * Tn);  // [rank][hypothesis]: the slabs' shares of every hypothesis' inlier count
    int32_t* d_samples = da.take<int32_t>(9 * (size_t)Tn);
    double* d_pts = da.take<double>(54 * (size_t)Tn + (size_t)R + 8);  // the sampled points + (warm path) the ranks' point counts behind them
    double* d_hyp = da.take<double>(8 * (size_t)Tn);
    char* zero_begin = da.base + da.off;  // states, scored, flags: one memset
    LmState* d_states = da.take<LmState>(Tn);
    int* d_scored = da.take<int>(Tn);
    int* d_flags = da.take<int>(16);
    int* d_irr_count = da.take<int>(batch);  // analytic pass: listed pixels per hypothesis of the batch (zeroed with the states)
    const size_t zero_bytes = (size_t)((da.base + da.off) - zero_begin);
    int* d_irr_list = da.take<int>(ransac_lma_list_ints(batch));
    double* d_tcount = da.take<double>(Tn);
    double* d_terr = da.take<double>(Tn);
    RansacBest* d_best = da.take<RansacBest>(1);
    RansacBest* d_best_shard = da.take<RansacBest>(1);
    double* d_row = da.take<double>(row_max);
    double* d_rows_all = da.take<double>((size_t)row_max * R);
    double* d_zs = da.take<double>(8);
    double* d_zs_all = da.take<double>((size_t)R + 8);
    double* d_header = da.take<double>(8);
    double* d_gather = padded ? da.take<double>(cap * R) : d_depth_map;

    // ---- per-stage workspace (stream-ordered reuse) ----
    const size_t ncells = sc > 0 ? (size_t)flatten_cells(rows, sc) : 1;
    const size_t ws_need = std::max({2 * Arena::need(sizeof(int64_t) * ncells),
                                     Arena::need(sizeof(double) * std::max<size_t>((size_t)ransac_lm_partials_doubles(c, (int64_t)N1, batch), (size_t)ransac_lma_partials_doubles(c, (int64_t)N1, batch))) + 2 * Arena::need(sizeof(int64_t) * 2048),
                                     Arena::need(8 * cap) + Arena::need(8 * 1024)}) + 4096;
    if (rc == RSDSFM_OK) rc = ensure_ws(c, ws_need);
    if (rc == RSDSFM_OK)
        rc = ensure_pinned(c, sizeof(RansacBest) + 64 + 8 * (size_t)R * 2 + 64 + sizeof(RefineState) + 64 + sizeof(int32_t) * 9 * (size_t)Tn + 8 + 64 + 16 * (size_t)R + 64 + 8 * (size_t)R + 64);
    // the refinement's session (sized for every point of the slab an inlier), the rank-indexed flow exchange (quirk Q2: at most the
    // whole flow list of every slab + this slab's columns) and the depth map's claim words
    const size_t npart_cap = (size_t)std::max(refine_partials_doubles(c, (int64_t)N1), refine_slot_partials_doubles(c, (int64_t)N1));
    if (rc == RSDSFM_OK && prm->use_refinement)
        rc = ensure_dev(c, &D->d_session, &D->session_bytes, Arena::need(sizeof(RefineState) + 64) + Arena::need(32 * N1) + 4 * Arena::need(8 * N1) + Arena::need(8 * npart_cap) + 1024);
    if (rc == RSDSFM_OK && prm->use_refinement && prm->flow_index_mode == RSDSFM_FLOW_COMPAT_RANK && R > 1)
        rc = ensure_dev(c, &D->d_flow, &D->flow_bytes, Arena::need(16 * std::max<size_t>(cap, 1) * R) + Arena::need(16 * N1) + 1024);
    if (rc == RSDSFM_OK) rc = claim_map_reserve(c, 2, N1);
    if (rc == RSDSFM_OK && !c->d_core_flag) {  // the minimal solver's persistent range-flag word (as ransac_begin allocates it)
        if (hipMalloc(reinterpret_cast<void**>(&c->d_core_flag), 64) != hipSuccess || hipMemsetAsync(c->d_core_flag, 0, 64, c->stream) != hipSuccess) {
            c->d_core_flag = nullptr;
            rc = fail(c, RSDSFM_ERR_HIP, "out of device memory (range-flag word)");
        }
    }
    const int setup_rc = rc;
    char* hp = static_cast<char*>(c->h_pinned);
    RansacBest* h_best = reinterpret_cast<RansacBest*>(hp);
    int* h_flags = reinterpret_cast<int*>(hp + sizeof(RansacBest));
    static_assert(sizeof(RansacBest) % 8 == 0, "the words behind the record are 8-byte aligned");
    int64_t* h_scan = reinterpret_cast<int64_t*>(hp + sizeof(RansacBest) + 32);  // (second half of the 64 bytes of flag words: the slab's compaction total)
    int64_t* h_cnt = reinterpret_cast<int64_t*>(hp + sizeof(RansacBest) + 64);
    int64_t* h_m = h_cnt + R;
    double* h_header = reinterpret_cast<double*>(h_m + R);
    RefineState* h_state = reinterpret_cast<RefineState*>(h_header + 8);
    int* h_bad = reinterpret_cast<int*>(reinterpret_cast<char*>(h_state) + sizeof(RefineState));
    int32_t* h_samples = reinterpret_cast<int32_t*>(reinterpret_cast<char*>(h_state) + sizeof(RefineState) + 64);

    memset(res, 0, sizeof(*res));
    // ---- flatten of the slab; point counts + setup status of all slabs ----
    // WARM path (the previous solve on this communicator succeeded on every rank with this very shape, and all its slabs were dense):
    // no rank can need an allocation, so the ranks do not wait for each other's setup status; they go on with the point counts of a
    // dense frame, the real counts ride in the sampled points' all-reduce and are checked with the RANSAC's first host read.  A rank
    // that does fail here (only an argument error is left: a null slab pointer) reports a count of -1 and works on whatever its
    // buffers hold: the check fails on every rank, all start over on the COLD path below, and its status exchange ends the call.
    const int shape_flags = (prm->use_acceleration_mode ? 1 : 0) | (prm->use_refinement ? 2 : 0) | (prm->flow_index_mode << 2) | (depth_mode << 4) |
                            (prm->use_global_shutter_mode ? 64 : 0);
    const bool warm = D->warm && D->warm_rows == rows && D->warm_cols == cols && D->warm_T == T && D->warm_flags == shape_flags && T > 0 &&
                      depth_mode == RSDSFM_DEPTH_CERES_LM && T <= kRansacBatch;
    bool spec_dense = warm && D->dense_hint;
    int path_flags = spec_dense ? 1 : 0;
    D->warm = false;  // (set again by a solve that ends well)
    std::vector<int64_t> h_xchg(2 * (size_t)R + 2, 0);  // (pageable on purpose: it must exist even when the pinned block could not grow)
    // (the samples block rounded up to 8 bytes: 36 Tn bytes is only 4-byte aligned for odd T, and this is read as doubles)
    double* h_counts_tail = reinterpret_cast<double*>(reinterpret_cast<char*>(h_samples) + ((sizeof(int32_t) * 9 * (size_t)Tn + 7) & ~(size_t)7) + 64 + 16 * (size_t)R + 64);
    int64_t n_total = 0, offset = 0, n = 0;
restart_cold:
    if (!spec_dense) {
        h_xchg[1] = setup_rc;
        RSDSFM_HIP_CHECK(c, hipMemcpyAsync(d_xchg + 2 * rank, h_xchg.data(), 2 * sizeof(int64_t), hipMemcpyHostToDevice, c->stream));  // {0, status}
        if (setup_rc == RSDSFM_OK && Ns > 0) {
            Arena ws(c->d_ws);
            int64_t* d_counts = ws.take<int64_t>(ncells);
            int64_t* d_offsets = ws.take<int64_t>(ncells);
            rc = flatten_launch(c, d_img_slab, rows, sc, col0, fx, fy, cx, cy, gamma, prm->flow_threshold, d_q, d_u, d_a, d_ak, d_counts, d_offsets,
                                d_xchg + 2 * rank, nullptr);
            if (rc != RSDSFM_OK) return rc;
        }
        rc = all_gather(c, D, d_xchg + 2 * rank, d_xchg, 2 * sizeof(int64_t));
        if (rc != RSDSFM_OK) return rc;
        RSDSFM_HIP_CHECK(c, hipMemcpyAsync(h_xchg.data(), d_xchg, 2 * sizeof(int64_t) * R, hipMemcpyDeviceToHost, c->stream));
        if (setup_rc == RSDSFM_OK) RSDSFM_HIP_CHECK(c, hipMemsetAsync(zero_begin, 0, zero_bytes, c->stream));
        rc = sync(c, D);
        if (rc != RSDSFM_OK) return rc;
        for (int r = 0; r < R; ++r)
            if (h_xchg[2 * (size_t)r + 1] != 0) {
                if (setup_rc != RSDSFM_OK) return setup_rc;  // (this rank's own message is in place)
                return fail(c, RSDSFM_ERR_PEER, ("rank " + std::to_string(r) + " failed while setting up its slab (code " + std::to_string(h_xchg[2 * (size_t)r + 1]) + ")").c_str());
            }
        for (int r = 0; r < R; ++r) h_cnt[r] = h_xchg[2 * (size_t)r];
    } else {
        // the counts of a dense frame; this rank's real count lands in its exchange word (device) and travels with the sampled points
        for (int r = 0; r < R; ++r) {
            int32_t c0r = 0, scr = 0;
            rsdsfm_tiled_slab_bounds(cols, R, r, &c0r, &scr, nullptr);
            h_cnt[r] = (int64_t)rows * scr;
        }
        h_xchg[0] = setup_rc == RSDSFM_OK ? 0 : -1;  // (a failing rank: a count no dense slab has)
        RSDSFM_HIP_CHECK(c, hipMemcpyAsync(d_xchg + 2 * rank, h_xchg.data(), sizeof(int64_t), hipMemcpyHostToDevice, c->stream));
        if (setup_rc == RSDSFM_OK && Ns > 0) {
            Arena ws(c->d_ws);
            int64_t* d_counts = ws.take<int64_t>(ncells);
            int64_t* d_offsets = ws.take<int64_t>(ncells);
            rc = flatten_launch(c, d_img_slab, rows, sc, col0, fx, fy, cx, cy, gamma, prm->flow_threshold, d_q, d_u, d_a, d_ak, d_counts, d_offsets,
                                d_xchg + 2 * rank, nullptr);
            if (rc != RSDSFM_OK) return rc;
        }
        RSDSFM_HIP_CHECK(c, hipMemsetAsync(zero_begin, 0, zero_bytes, c->stream));
    }
    RSDSFM_HIP_CHECK(c, hipMemcpyAsync(d_cnt_all, h_cnt, sizeof(int64_t) * R, hipMemcpyHostToDevice, c->stream));  // (the rank-indexed flow gather reads them)
    n_total = 0, offset = 0;
    for (int r = 0; r < R; ++r) {
        if (r == rank) offset = n_total;
        n_total += h_cnt[r];
    }
    n = h_cnt[rank];
    res->n_points = n_total;
    if (n_total < 9) return fail(c, RSDSFM_ERR_INVALID, "ransac needs at least 9 points (the reference would compute rand() % 0)");
    if (n_total > (int64_t)INT32_MAX) return fail(c, RSDSFM_ERR_INVALID, "n exceeds the int32 sample index range");
    if (prm->use_global_shutter_mode) {  // main.cc:441-444
        rc = alpha_ones_launch(c, d_a, n);
        if (rc != RSDSFM_OK) return rc;
    }

    // ---- hypotheses: deterministic sampler on every rank, sampled points by one exact all-reduce, minimal solver replicated ----
    // the in-range function cores (device_math.hpp) in the minimal solver's SVD and in round 0 of the LM solves, as the single-context solve
    // runs them: one hypothesis batch, LM mode, the user's switch (rsdsfm_set_ransac_math: the same on every rank), no recent restart
    bool core = T > 0 && T <= kRansacBatch && depth_mode == RSDSFM_DEPTH_CERES_LM && c->ransac_math_mode == 0 && D->standard_math == 0;  // (replicated knowledge only -- not this rank's setup status: what is exchanged, and how many bytes, must be the same on every rank)
    if (!core && D->standard_math > 0) D->standard_math -= 1;
    int* d_core_flags = d_flags + 8;  // (inside the zeroed block)
    int m9_epoch = 0;
    if (T > 0) {
        sample_indices(n_total, T, prm->seed, h_samples);
        RSDSFM_HIP_CHECK(c, hipMemcpyAsync(d_samples, h_samples, sizeof(int32_t) * 9 * (size_t)T, hipMemcpyHostToDevice, c->stream));
        hipLaunchKernelGGL(pack_samples_kernel, dim3((9 * T + 255) / 256), dim3(256), 0, c->stream, reinterpret_cast<const double2*>(d_q),
                           reinterpret_cast<const double2*>(d_u), d_a, d_ak, n, offset, d_samples, 9 * T, d_pts,
                           spec_dense ? static_cast<const int64_t*>(d_xchg + 2 * rank) : nullptr, rank, R);
        RSDSFM_HIP_CHECK(c, hipGetLastError());
        rc = all_reduce_sum(c, D, d_pts, 54 * (size_t)T + (spec_dense ? (size_t)R : 0));
        if (rc != RSDSFM_OK) return rc;
    }
    // the depth solves of the trials on the analytic LM trajectory (lma_common.hpp), as in the single-context solve: the user's switch
    // (rsdsfm_set_lm_arithmetic: the same on every rank) and the communicator's hold -- replicated knowledge, every rank decides alike
    bool analytic = T > 0 && depth_mode == RSDSFM_DEPTH_CERES_LM && c->lm_arithmetic == 0 && D->lma_hold == 0;  // (replicated knowledge only, like `core`: a rank whose setup failed still issues the exchanges its peers issue, with the same byte counts, until the counts / status check sends everyone to restart_cold)
    const bool lma_may_return = depth_mode == RSDSFM_DEPTH_CERES_LM && c->lm_arithmetic == 0;  // (ties are reported: they renew the hold)
    bool lma_restarted = false, tie_seen = false, spec_refine_tried = false;
    const int lma_cand[2] = {2, 1};  // fused iterates: fixed, like kTiledFusedBase (every rank must fuse the same ones)
restart_ransac:
    if (T > 0) {
        Minimal9Direct dir;
        if (core && T <= c->num_cus * 2) {  // (the wave-per-hypothesis solver: the one with the cores)
            c->core_epoch = c->core_epoch >= 0x3fffffff ? 1 : c->core_epoch + 1;
            dir.core_flag = c->d_core_flag;
            dir.core_epoch = m9_epoch = c->core_epoch;
        }
        rc = minimal9_launch(c, d_pts, d_pts + 18 * (size_t)T, d_pts + 36 * (size_t)T, d_pts + 45 * (size_t)T, nullptr, T, prm->use_acceleration_mode,
                             prm->k_sign_mode, d_hyp, nullptr, 0, m9_epoch ? &dir : nullptr);
        if (rc != RSDSFM_OK) return rc;
    }

    // ---- RANSAC: LM rounds of the hypothesis batches, scores, winner, compaction ----
    Arena ws(c->d_ws);
    double* d_partials = ws.take<double>(std::max<size_t>((size_t)ransac_lm_partials_doubles(c, (int64_t)N1, batch), (size_t)ransac_lma_partials_doubles(c, (int64_t)N1, batch)));
    int64_t* d_bcounts = ws.take<int64_t>(2048);
    int64_t* d_boffs = ws.take<int64_t>(2048);
    bool spec_scored = false;
    auto final_stage = [&](bool spec) -> int {  // winner (replicated), its dense 1/depth + mask + compaction on the slab, inlier counts of all slabs
        // the winner (replicated) -- and with it the inlier counts of ALL slabs: the ranks' shares of the winner's count were in the rows the
        // scores came from (ransac_decide_kernel / ransac_reduce_scores_kernel keep them), so the counts need no exchange of their own
        // (spec: behind round 0, ahead of the host's read of the flag words -- the pick reads them itself and marks a RANSAC that is not over
        // `undecided`, so that the final stage and everything enqueued behind it on speculation leave at once)
        int rc2 = ransac_pick_launch(c, d_tcount, d_terr, T, d_hyp, d_best, nullptr, spec ? d_flags : nullptr, nullptr, spec && spec_scored ? 1 : 0, d_cnt_rt, T, R, d_m_all,
                                     !lma_may_return ? 0.0 : (analytic ? kLmaTie : -kLmaTie));
        if (rc2 != RSDSFM_OK) return rc2;
        // the compaction stores the SLAB's scan total into its record: every rank works on a copy of the (identical) winner record
        RSDSFM_HIP_CHECK(c, hipMemcpyAsync(d_best_shard, d_best, sizeof(RansacBest), hipMemcpyDeviceToDevice, c->stream));
        rc2 = ransac_final_launch(c, d_q, d_u, d_a, d_ak, n, d_best_shard, d_states, depth_mode, prm->ransac_tol, d_rho, d_mask, d_bcounts, d_boffs,
                                  d_idx, d_inl, d_in_a, d_in_ak, nullptr);
        if (rc2 != RSDSFM_OK) return rc2;
        static_assert(sizeof(d_best_shard->num_inliers_scan) == sizeof(int64_t), "count type");
        // (what the compaction found on THIS slab travels to the host beside the counts from the rows: they must agree)
        RSDSFM_HIP_CHECK(c, hipMemcpyAsync(h_scan, &d_best_shard->num_inliers_scan, sizeof(int64_t), hipMemcpyDeviceToHost, c->stream));
        RSDSFM_HIP_CHECK(c, hipMemcpyAsync(h_best, d_best, sizeof(RansacBest), hipMemcpyDeviceToHost, c->stream));
        RSDSFM_HIP_CHECK(c, hipMemcpyAsync(h_m, d_m_all, sizeof(int64_t) * R, hipMemcpyDeviceToHost, c->stream));
        return RSDSFM_OK;
    };
    // ---- the refinement and the depth-map stage as closures: they go behind the DEFINITIVE final stage, from the host-side result, or
    // -- one host wait less -- behind the SPECULATED one, from the device-resident winner (inlier count and start pose in the slab's RansacBest
    // record: RefineBuffers::m_on_device, as in the single-context frame solve), before the host has read anything of the RANSAC ----
    int64_t m_total = 0, m = 0;
    double v[3] = {0.0, 0.0, 0.0}, w[3] = {0.0, 0.0, 0.0}, k = 0.0;
    bool local_index_bad = false;
    // mean-z sign (global), depth-map slab, ONE all-gather of the slabs, pose table.  The motion comes from the host (v_host; pose table by a
    // launch of its own) or -- enqueued behind the refinement's output pass BEFORE the host has read the state, so that the wait for the state
    // covers this stage too -- from the device-resident state (v_dev = RefineState::p; the sign kernel writes the pose table).
    // m_dev: the slab's inlier count, device-resident (the host's is an upper bound then)
    auto depth_stage = [&](double* d_points, const double* d_zsums, int nz, const double* v_host, const double* w_host, double k_host,
                           const double* state_p_dev, const int64_t* m_dev) -> int {
        double* d_slab = d_gather + (size_t)rank * cap;
        if (cap > Ns) RSDSFM_HIP_CHECK(c, hipMemsetAsync(d_slab + Ns, 0, sizeof(double) * (cap - Ns), c->stream));  // columns past the image: zeros
        PoseTableOut pt;
        const bool table = d_R_rows9 && d_t_rows3;
        if (state_p_dev && table) pt.R = d_R_rows9, pt.t = d_t_rows3, pt.rows = rows, pt.gamma = gamma, pt.wk_dev = state_p_dev + 3;
        int rc2 = depth_map_slab_launch(c, d_points, m_dev ? n : m, d_zsums, nz, m_dev ? n_total : m_total, v_host, fx, fy, cx, cy, rows, col0, sc, d_slab, nullptr, d_ys,
                                        d_header, h_header, state_p_dev, m_dev, state_p_dev && table ? &pt : nullptr);
        if (rc2 != RSDSFM_OK) return rc2;
        rc2 = all_gather(c, D, d_slab, d_gather, sizeof(double) * cap);  // the one data-path collective
        if (rc2 != RSDSFM_OK) return rc2;
        if (padded)
            RSDSFM_HIP_CHECK(c, hipMemcpyAsync(d_depth_map, d_gather, sizeof(double) * (size_t)rows * (size_t)cols, hipMemcpyDeviceToDevice, c->stream));
        if (!state_p_dev && table) {
            Pose pose;
            for (int i = 0; i < 3; ++i) pose.v[i] = v_host[i], pose.w[i] = w_host[i];
            pose.k = k_host;
            rc2 = pose_table_launch(c, pose, gamma, rows, d_R_rows9, d_t_rows3, d_header + 1);  // v' (possibly flipped) from the device header
            if (rc2 != RSDSFM_OK) return rc2;
        }
        return RSDSFM_OK;
    };
    RefineBuffers RB;
    int launched = 0;
    bool depth_done = false, spec_refine = false, final_spec = false, q2_remote = false;
    // the refinement on radius-factorised Schur sums (refine_rf_kernels.hip) unless the reference's arithmetic is asked for (the same switch on
    // every rank); a guard of that path ends the solve with kTermRestartExact in the REPLICATED state: every rank runs it again iterate by iterate
    bool refine_rf = c->lm_arithmetic == 0 && c->refine_arithmetic == 0;
    const int refine_hint = D->refine_iters_hint;
    // buffers of the refinement's session; dev: the slab's inlier count is device-resident (upper bound n), no remote flow column
    auto refine_layout = [&](bool dev) -> int {
        const int64_t mm = dev ? n : m;
        const size_t M = (size_t)std::max<int64_t>(mm, 1);
        const size_t npart = (size_t)std::max(refine_partials_doubles(c, mm), refine_slot_partials_doubles(c, mm));
        Arena sa(D->d_session);  // (allocated in the setup part for the slab's upper bound)
        RB = RefineBuffers();
        RB.flow = d_u;
        RB.n_flow = n;
        RB.m = mm;
        RB.m_on_device = dev;
        RB.inl = d_inl;
        RB.alpha = d_in_a;
        RB.alpha_k = d_in_ak;
        RB.inlier_idx = d_idx;
        RB.flow_index_mode = RSDSFM_FLOW_GATHERED;
        RB.want_zsum = true;  // the rows of the refinement carry the sum of z: the sign test below needs no exchange of its own
        if (prm->flow_index_mode == RSDSFM_FLOW_COMPAT_RANK) {
            // The reference's default (quirk Q2): inlier i of the global list reads flow column i of the global list.  This rank's
            // inliers are the global ranks [prefix, prefix + m), and since a slab never holds more inliers than points those
            // columns live on slabs <= rank.  Every rank knows every count, so all ranks agree on what is exchanged: the heads of
            // the slabs' flow lists up to the global inlier count (nothing at all when every needed column is local -- e.g. every
            // pixel an inlier), all-gathered in rank order; then each rank picks its m columns.
            // (dev: enqueued on the assumption that no column is remote -- the previous solve's case --, checked once the counts are known)
            if (!dev && q2_remote) {
                int64_t prefix = 0, lmax = 0;
                int64_t pm = 0, po = 0;  // inliers / points in front of slab r
                for (int r = 0; r < R; ++r) {
                    lmax = std::max(lmax, std::min<int64_t>(h_cnt[r], std::max<int64_t>(m_total - po, 0)));
                    if (r == rank) prefix = pm;
                    pm += h_m[r];
                    po += h_cnt[r];
                }
                const size_t L = (size_t)std::max<int64_t>(lmax, 1);
                Arena fl(D->d_flow);  // (allocated in the setup part: L <= the slab stride, M <= the slab's points)
                double* d_heads = fl.take<double>(2 * L * R);
                double* d_flow_rank = fl.take<double>(2 * M);
                const int64_t mine = std::min<int64_t>(n, std::max<int64_t>(m_total - offset, 0));  // columns of this slab the ranks can reach
                if (mine > 0)
                    RSDSFM_HIP_CHECK(c, hipMemcpyAsync(d_heads + 2 * L * (size_t)rank, d_u, 16 * (size_t)mine, hipMemcpyDeviceToDevice, c->stream));
                int rc2 = all_gather(c, D, d_heads + 2 * L * (size_t)rank, d_heads, 16 * L);
                if (rc2 != RSDSFM_OK) return rc2;
                if (m > 0) {
                    hipLaunchKernelGGL(rank_flow_gather_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, c->stream,
                                       reinterpret_cast<const double2*>(d_heads), (int64_t)L, d_cnt_all, R, prefix, m,
                                       reinterpret_cast<double2*>(d_flow_rank));
                    RSDSFM_HIP_CHECK(c, hipGetLastError());
                }
                RB.flow = d_flow_rank;
                RB.n_flow = m;
            }
            // (no remote column: prefix == offset for every slab with inliers, so the global column of local inlier i is local column i)
            RB.flow_index_mode = RSDSFM_FLOW_COMPAT_RANK;
            RB.inlier_idx = nullptr;
        }
        char* state_block = sa.take<char>(sizeof(RefineState) + 64);
        RB.state = reinterpret_cast<RefineState*>(state_block);
        RB.bad_index = reinterpret_cast<int*>(state_block + sizeof(RefineState));
        RB.uu = sa.take<double>(4 * M);
        RB.beta = sa.take<double>(M);
        RB.rho_a = sa.take<double>(M);
        RB.rho_b = sa.take<double>(M);
        RB.srho = sa.take<double>(M);
        RB.partials = sa.take<double>(npart);
        return RSDSFM_OK;
    };
    // ONE exchange per LM iteration (refine_kernels.hip, slot kernels): a slot's pass carries the back-substitution of iteration i and the
    // Schur sums of iteration i + 1 speculated at the candidate for the radius an accepted step of quality >= 0.937 gets; a slot whose
    // speculation did not apply is followed by a plain Schur slot (the kernels read which kind from the replicated state).  Slots per
    // host poll: what the previous solve on this communicator consumed (6 before there is one: the first Schur slot + 5 iterations; at most
    // 28), later chunks what it still needed at that point (2 .. 8).  Every rank holds the same hint, so all ranks issue the same
    // collectives; the chunking changes when the host looks at the state, never what the kernels compute.
    // (slot j of a chunk: the pass, whose prologue is the replicated stage of slot j - 1 on the rows gathered then; the shard's row; the exchange)
    auto refine_chunk = [&](int chunk) -> int {
        int rc2 = RSDSFM_OK;
        if (refine_rf) {
            // (slot g = launched + j of the solve: its pass -- slot 0: iteration zero + the Schur sums of iteration 1 --, the shard's row
            // [sums | listed inliers], the exchange; the stage of slot g - 1 on the rows gathered then runs in the pass's prologue)
            const int64_t* mt_dev = RB.m_on_device ? &d_best_shard->num_inliers : nullptr;
            for (int j = 0; j < chunk; ++j) {
                const int g = launched + j;
                rc2 = refine_rf_pass_launch(c, RB, np, g, launched, d_rows_all, R, m_total, mt_dev);
                if (rc2 != RSDSFM_OK) return rc2;
                rc2 = refine_rf_row_launch(c, RB, np, g, d_row);
                if (rc2 != RSDSFM_OK) return rc2;
                rc2 = all_gather(c, D, d_row, d_rows_all, sizeof(double) * (size_t)refine_rf_row_doubles(np));
                if (rc2 != RSDSFM_OK) return rc2;
            }
            rc2 = refine_rf_apply_launch(c, RB, np, launched + chunk - 1, true, d_rows_all, R, m_total, mt_dev);
        } else {
            for (int j = 0; j < chunk; ++j) {
                rc2 = refine_slot_rows_launch(c, RB, np, d_row, j, d_rows_all, R);
                if (rc2 != RSDSFM_OK) return rc2;
                rc2 = all_gather(c, D, d_row, d_rows_all, sizeof(double) * (size_t)refine_slot_row_doubles(np));
                if (rc2 != RSDSFM_OK) return rc2;
            }
            rc2 = refine_slot_apply_launch(c, RB, np, d_rows_all, R, chunk);  // the stage behind the chunk's last exchange -> RB.state
        }
        if (rc2 != RSDSFM_OK) return rc2;
        launched += chunk;
        rc2 = refine_finish_launch(c, RB, d_inl_ref);  // enqueued before the poll: the common case ends within one chunk
        if (rc2 != RSDSFM_OK) return rc2;
        // ... and so is the depth-map stage, from the device-resident state, behind a chunk that can be the last one (every rank holds
        // the same hint): the poll's wait then covers it.  Behind a chunk the solve outlives it runs again; what it wrote is overwritten.
        depth_done = refine_hint < 1 || launched + 1 >= refine_hint;
        if (depth_done) {
            rc2 = depth_stage(d_inl_ref, &RB.state->zsum, 1, nullptr, nullptr, 0.0, RB.state->p, RB.m_on_device ? &RB.state->m : nullptr);
            if (rc2 != RSDSFM_OK) return rc2;
        }
        RSDSFM_HIP_CHECK(c, hipMemcpyAsync(h_state, RB.state, sizeof(RefineState) + sizeof(int), hipMemcpyDeviceToHost, c->stream));
        return RSDSFM_OK;
    };
    // start state, iteration zero (its rows all-gathered, the decision replicated), the first chunk of slots
    auto refine_start = [&]() -> int {
        int rc2 = RSDSFM_OK;
        if (RB.m_on_device) {
            rc2 = refine_state_from_best_launch(c, d_best_shard, RB, np);  // (count = the slab's compaction total, pose = the winner's; an undecided RANSAC: no inliers)
            if (rc2 != RSDSFM_OK) return rc2;
        } else {
            memset(h_state, 0, sizeof(RefineState));
            h_state->np = np;
            for (int i = 0; i < 3; ++i) h_state->p[i] = v[i], h_state->p[3 + i] = w[i];
            h_state->p[6] = k;
            h_state->termination = -1;
            h_state->radius = kInitialRadius;
            h_state->need_schur = 1;  // the first slot is the Schur pass of iteration 1
            memset(h_bad, 0, kRefineStateBlockTail);  // bad-index flag + the list counters of the radius-factorised path
            RSDSFM_HIP_CHECK(c, hipMemcpyAsync(RB.state, h_state, sizeof(RefineState) + kRefineStateBlockTail, hipMemcpyHostToDevice, c->stream));
        }
        rc2 = refine_trace_reset(c);
        if (rc2 != RSDSFM_OK) return rc2;
        if (refine_rf) {  // (no iteration zero of its own: slot 0 carries it)
            c->refine_rf_runs += 1;
            return refine_chunk(refine_hint >= 1 ? std::min(refine_hint, 28) : 6);
        }
        rc2 = refine_stage_rows_launch(c, RB, np, 0, d_row);
        if (rc2 != RSDSFM_OK) return rc2;
        rc2 = all_gather(c, D, d_row, d_rows_all, sizeof(double) * (size_t)refine_stage_row_doubles(np, 0));
        if (rc2 != RSDSFM_OK) return rc2;
        rc2 = refine_stage_apply_launch(c, RB, np, 0, d_rows_all, R, m_total, RB.m_on_device ? &d_best_shard->num_inliers : nullptr);
        if (rc2 != RSDSFM_OK) return rc2;
        return refine_chunk(refine_hint >= 1 ? std::min(refine_hint, 28) : 6);  // (k refined: two slots per LM iteration, ~27 in all)
    };
    bool final_done = false;
    for (int b0 = 0; b0 < T; b0 += batch) {
        const int B = std::min(batch, T - b0);
        bool need_score = true;
        if (depth_mode == RSDSFM_DEPTH_CERES_LM) {
            if (b0 > 0) RSDSFM_HIP_CHECK(c, hipMemsetAsync(d_flags, 0, sizeof(int) * 4, c->stream));
            for (int round = 0;; ++round) {
                if (round > 4 * kMaxIter) return fail(c, RSDSFM_ERR_NUMERIC, "LM state machines did not terminate");
                const bool core_round = core && round == 0 && !analytic;  // (core implies one batch: B == T)
                if (analytic && round == 0) {
                    // ONE pixel pass + ONE exchange of [B][kLmaRow] closed-form sums per rank, the trust-region loop replicated on the gathered
                    // rows; the minimal solver's range flag is replicated knowledge (every rank ran the same solver on the same points), so it
                    // needs no trailer.  A hypothesis whose own guards tripped comes back as still running: rounds 1, 2, ... are the
                    // iterate-by-iterate kernels', where only such hypotheses take part (replicated: every rank sees the same states)
                    if (b0 > 0) RSDSFM_HIP_CHECK(c, hipMemsetAsync(d_irr_count, 0, sizeof(int) * (size_t)batch, c->stream));
                    rc = ransac_lma_rows_launch(c, d_q, d_u, d_a, d_ak, n, d_hyp + (size_t)b0 * 8, B, d_partials, prm->ransac_tol, lma_cand, 2, d_irr_count, d_irr_list, d_row);
                    if (rc != RSDSFM_OK) return rc;
                    const size_t row_doubles = (size_t)ransac_lma_rows_doubles() * (size_t)B;
                    rc = all_gather(c, D, d_row, d_rows_all, sizeof(double) * row_doubles);
                    if (rc != RSDSFM_OK) return rc;
                    rc = ransac_lma_decide_rows_launch(c, d_rows_all, R, (int64_t)row_doubles, B, n_total, d_hyp + (size_t)b0 * 8, d_states + b0, d_flags, d_scored + b0,
                                                       d_tcount + b0, d_terr + b0, prm->ransac_tol, lma_cand, 2, nullptr, nullptr, d_cnt_rt + b0, T,
                                                       m9_epoch ? c->d_core_flag : nullptr, m9_epoch);
                    if (rc != RSDSFM_OK) return rc;
                } else {
                    rc = ransac_lm_rows_launch(c, d_q, d_u, d_a, d_ak, n, d_hyp + (size_t)b0 * 8, B, d_states + b0, d_partials, round, prm->ransac_tol, d_row,
                                               core_round ? d_core_flags : nullptr, m9_epoch ? c->d_core_flag : nullptr, m9_epoch);
                    if (rc != RSDSFM_OK) return rc;
                    rc = all_gather(c, D, d_row, d_rows_all, sizeof(double) * (size_t)ransac_rows_payload_doubles(B, core_round));
                    if (rc != RSDSFM_OK) return rc;
                    rc = ransac_decide_rows_launch(c, d_rows_all, R, B, d_states + b0, n_total, round, d_flags, d_scored + b0, d_tcount + b0, d_terr + b0, core_round,
                                                   d_cnt_rt + b0, T);
                    if (rc != RSDSFM_OK) return rc;
                }
                D->ransac_rounds += 1;
                if (round == 0 && B == T) {  // the common case is decided and scored by round 0: enqueue the final stage before reading the flags
                    if (D->score_idle < kScoreIdleLimit) {
                        // a recent solve needed the separate scoring pass (hypotheses that end at an iterate round 0 does not score):
                        // enqueue it ahead of the flags too -- it only touches hypotheses round 0 left unscored -- which saves a host
                        // round trip, a discarded final stage and its all-gather
                        rc = ransac_score_rows_launch(c, d_q, d_u, d_a, d_ak, n, d_hyp, T, d_states, depth_mode, prm->ransac_tol, d_scored, d_partials, d_row);
                        if (rc != RSDSFM_OK) return rc;
                        rc = all_gather(c, D, d_row, d_rows_all, sizeof(double) * (size_t)T * 2);
                        if (rc != RSDSFM_OK) return rc;
                        rc = ransac_score_merge_launch(c, d_rows_all, R, T, d_scored, d_tcount, d_terr, d_cnt_rt, T);
                        if (rc != RSDSFM_OK) return rc;
                        spec_scored = true;
                    }
                    rc = final_stage(true);
                    if (rc != RSDSFM_OK) return rc;
                    final_done = final_spec = true;
                    if (prm->use_refinement && !spec_refine_tried && D->spec_miss < 2 &&
                        (R == 1 || prm->flow_index_mode != RSDSFM_FLOW_COMPAT_RANK || D->q2_local)) {
                        // the refinement's start, its first chunk of slots and the depth-map stage behind this final stage, from the device-resident
                        // winner: on typical data the host's ONE wait below then covers the whole solve (every rank holds the same hints)
                        spec_refine_tried = true;
                        rc = refine_layout(true);
                        if (rc == RSDSFM_OK) rc = refine_start();
                        if (rc != RSDSFM_OK) return rc;
                        spec_refine = true;
                    }
                }
                RSDSFM_HIP_CHECK(c, hipMemcpyAsync(h_flags, d_flags, sizeof(int) * 4, hipMemcpyDeviceToHost, c->stream));
                const bool check_counts = spec_dense && round == 0 && b0 == 0;  // the warm path's first host read: were the slabs dense?
                if (check_counts)
                    RSDSFM_HIP_CHECK(c, hipMemcpyAsync(h_counts_tail, d_pts + 54 * (size_t)T, sizeof(double) * R, hipMemcpyDeviceToHost, c->stream));
                rc = sync(c, D);
                if (rc != RSDSFM_OK) return rc;
                if (check_counts) {
                    bool held = true;
                    for (int r = 0; r < R; ++r) held = held && h_counts_tail[r] == (double)h_cnt[r];
                    if (!held) {  // a slab dropped pixels (or a rank failed): everything so far ran on wrong counts -- all ranks see the same
                        spec_dense = false;  // gathered counts and start over together, through the counts / status exchange
                        D->dense_hint = false;
                        path_flags |= 2;
                        goto restart_cold;
                    }
                }
                // (the speculated pick's tie flag counts only where every trial had its score: this driver's pick does not see the flag words)
                const bool spec_tie = final_done && h_flags[0] == 0 && (h_flags[1] == 0 || spec_scored) && h_best->lma_tie != 0;
                if (analytic && ((h_flags[3] & 2) != 0 || spec_tie) && !((h_flags[3] & 1) && core)) {
                    // a guard of the analytic pass tripped, or the speculated pick met a tie it must not break (replicated: every rank reads the
                    // same words): the depth solves start over iterate by iterate on every rank, and the communicator stays there for a while
                    analytic = false;
                    lma_restarted = true;
                    D->lma_hold = 16;
                    c->lma_restarts += 1;
                    c->lma_last_guard = (h_flags[3] >> 8) | (spec_tie ? (1 << 7) : 0);
                    path_flags |= 8;
                    RSDSFM_HIP_CHECK(c, hipMemsetAsync(zero_begin, 0, zero_bytes, c->stream));
                    goto restart_ransac;  // (from the minimal solver on, like a function-core miss: one path to start over on)
                }
                if ((core_round || (analytic && core && round == 0)) && (h_flags[3] & 1) != 0) {
                    // some shard's round 0 (or the minimal solver) met an argument outside the range of the function cores: every rank read
                    // the same flag, all run the RANSAC again from the minimal solver on with the standard functions (identical results)
                    core = false;
                    m9_epoch = 0;
                    D->standard_math = 16;
                    D->restarts += 1;
                    c->ransac_restarts += 1;
                    path_flags |= 4;
                    RSDSFM_HIP_CHECK(c, hipMemsetAsync(zero_begin, 0, zero_bytes, c->stream));
                    goto restart_ransac;
                }
                if (h_flags[0] == 0) break;
                final_done = false;
            }
            need_score = h_flags[1] > 0;
            if (B == T) D->score_idle = need_score ? 0 : std::min(D->score_idle + 1, kScoreIdleLimit);
            if (need_score && final_done && spec_scored) need_score = false;  // round 0 decided everything and the pass already ran
            if (need_score) final_done = false;
        }
        if (need_score) {
            const int* sc_ptr = depth_mode == RSDSFM_DEPTH_CERES_LM ? d_scored + b0 : nullptr;
            rc = ransac_score_rows_launch(c, d_q, d_u, d_a, d_ak, n, d_hyp + (size_t)b0 * 8, B, d_states + b0, depth_mode, prm->ransac_tol, sc_ptr, d_partials, d_row);
            if (rc != RSDSFM_OK) return rc;
            rc = all_gather(c, D, d_row, d_rows_all, sizeof(double) * (size_t)B * 2);
            if (rc != RSDSFM_OK) return rc;
            rc = ransac_score_merge_launch(c, d_rows_all, R, B, sc_ptr, d_tcount + b0, d_terr + b0, d_cnt_rt + b0, T);
            if (rc != RSDSFM_OK) return rc;
        }
    }
    if (!final_done) {
        rc = final_stage(false);
        if (rc != RSDSFM_OK) return rc;
        rc = sync(c, D);
        if (rc != RSDSFM_OK) return rc;
        final_spec = spec_refine = false;  // (the speculated final stage, and whatever went behind it, saw incomplete trials)
    }
    if (analytic && h_best->lma_tie) {  // guard (d) at the definitive pick (replicated: the trial scores are)
        analytic = false;
        lma_restarted = true;
        D->lma_hold = 16;
        c->lma_restarts += 1;
        c->lma_last_guard = 1 << 7;
        path_flags |= 8;
        RSDSFM_HIP_CHECK(c, hipMemsetAsync(zero_begin, 0, zero_bytes, c->stream));
        goto restart_ransac;
    }
    tie_seen = h_best->lma_tie != 0;
    m_total = 0;
    for (int r = 0; r < R; ++r) m_total += h_m[r];
    m = h_m[rank];
    // the winner's count against the ranks' shares is replicated knowledge: every rank returns alike.  This slab's compaction total against its
    // share is NOT (only this rank knows it): a rank that left here would strand the others in the refinement's exchanges and come back to
    // the next solve with other hints than theirs -- so it stays in lockstep to the end of the solve (its slab's part of the result is not to
    // be trusted: the buffers are sized for any count), keeps its hints, and reports the error then
    if (m_total != h_best->num_inliers) return fail(c, RSDSFM_ERR_NUMERIC, "inlier count mismatch between scoring and compaction");
    const bool local_scan_bad = *h_scan != m;
    res->num_inliers = m_total;
    res->best_trial = h_best->best_trial;
    memcpy(res->ransac_w, &h_best->hyp[0], 3 * sizeof(double));
    memcpy(res->ransac_v, &h_best->hyp[3], 3 * sizeof(double));
    res->ransac_k = h_best->hyp[6];
    for (int i = 0; i < 3; ++i) v[i] = h_best->hyp[3 + i], w[i] = h_best->hyp[i];
    k = h_best->hyp[6];
    double* d_final = d_inl;
    const double* d_zsum_global = nullptr;
    {
        // reference flow indexing (quirk Q2) over several ranks: does any slab need a flow column of another one?  (replicated: the counts are)
        q2_remote = false;
        int64_t pm = 0, po = 0;  // inliers / points in front of slab r
        for (int r = 0; r < R; ++r) {
            if (h_m[r] > 0 && pm != po) q2_remote = true;
            pm += h_m[r];
            po += h_cnt[r];
        }
        q2_remote = q2_remote && prm->flow_index_mode == RSDSFM_FLOW_COMPAT_RANK;
        // what went behind the speculated final stage counts if that stage did and it assumed the right flow columns; the hints of the next solve
        if (spec_refine && (h_best->undecided || q2_remote)) spec_refine = false;
        D->spec_miss = (final_spec && !h_best->undecided) ? 0 : std::min(D->spec_miss + 1, 2);
        D->q2_local = !q2_remote;
        if (spec_refine) path_flags |= 16;
    }

    // ---- joint refinement: iteration zero, then ONE slot (pass + exchange) per LM iteration, polled in chunks ----
    if (prm->use_refinement) {
        if (!spec_refine) {
            launched = 0;
            depth_done = false;
            rc = refine_layout(false);
            if (rc == RSDSFM_OK) rc = refine_start();
            if (rc != RSDSFM_OK) return rc;
            rc = sync(c, D);
            if (rc != RSDSFM_OK) return rc;
        }
        for (;;) {  // (here: a chunk is behind us and the host has its state)
            if (*h_bad) local_index_bad = true;  // (rank-local like the count check above: reported at the end, in lockstep)
            if (refine_rf && h_state->termination == kTermRestartExact) {
                // a guard of the radius-factorised path (replicated state: every rank sees it): the refinement again, iterate by iterate, from
                // the host-side RANSAC result
                c->refine_rf_restarts += 1;
                c->refine_rf_last_guard = h_state->rf_guard;
                refine_rf = false;
                spec_refine = false;
                path_flags &= ~16;
                launched = 0;
                depth_done = false;
                rc = refine_layout(false);
                if (rc == RSDSFM_OK) rc = refine_start();
                if (rc != RSDSFM_OK) return rc;
                rc = sync(c, D);
                if (rc != RSDSFM_OK) return rc;
                continue;
            }
            if (h_state->termination >= 0) break;
            if (launched > 8 * kMaxIter + 16) return fail(c, RSDSFM_ERR_NUMERIC, "refinement did not terminate");
            rc = refine_chunk(refine_hint >= 1 ? std::min(8, std::max(2, refine_hint - launched)) : 5);
            if (rc != RSDSFM_OK) return rc;
            rc = sync(c, D);
            if (rc != RSDSFM_OK) return rc;
        }
        D->refine_iters_hint = h_state->slots;
        if (refine_rf) c->refine_rf_resolves += h_state->rf_resolves;
        path_flags |= (std::min(h_state->slots, 0xFFFF) << 8);
        d_zsum_global = &RB.state->zsum;  // every rank holds the same GLOBAL sum of z of the final state (replicated decide stages)
        for (int i = 0; i < 3; ++i) v[i] = h_state->p[i], w[i] = h_state->p[3 + i];
        k = h_state->p[6];
        res->refine_summary.num_iterations = h_state->iteration;
        res->refine_summary.num_successful_steps = h_state->num_successful;
        res->refine_summary.num_unsuccessful_steps = h_state->num_unsuccessful;
        res->refine_summary.termination = h_state->termination;
        res->refine_summary.initial_cost = h_state->initial_cost;
        res->refine_summary.final_cost = h_state->cost;
        res->refine_summary.final_radius = h_state->radius;
        d_final = d_inl_ref;
    } else {
        depth_done = false;
    }

    // ---- mean-z sign (global), depth-map slab, ONE all-gather of the slabs, pose table (unless it rode behind the refinement's last chunk) ----
    if (!depth_done) {
        Arena ws2(c->d_ws);
        double* d_zpart = ws2.take<double>(1024);
        const double* d_zsums = d_zsum_global;
        int nz = 1;
        if (!d_zsums) {  // no refinement ran: the slabs' sums of z travel in an exchange of their own
            rc = zsum_row_launch(c, d_final, m, d_zpart, d_zs);
            if (rc != RSDSFM_OK) return rc;
            rc = all_gather(c, D, d_zs, d_zs_all, sizeof(double));
            if (rc != RSDSFM_OK) return rc;
            d_zsums = d_zs_all;
            nz = R;
        }
        rc = depth_stage(d_final, d_zsums, nz, v, w, k, nullptr, nullptr);
        if (rc != RSDSFM_OK) return rc;
        rc = sync(c, D);
        if (rc != RSDSFM_OK) return rc;
    }
    {  // what the next solve on this communicator may take for granted (identical on every rank: all of it is replicated knowledge)
        bool dense_now = true;
        for (int r = 0; r < R; ++r) {
            int32_t c0r = 0, scr = 0;
            rsdsfm_tiled_slab_bounds(cols, R, r, &c0r, &scr, nullptr);
            dense_now = dense_now && h_cnt[r] == (int64_t)rows * scr;
        }
        D->dense_hint = dense_now;
        // an iterate-by-iterate solve inside a hold: the hold goes on while the data keeps showing ties the analytic arithmetic cannot break
        if (!lma_restarted && !analytic && D->lma_hold > 0) D->lma_hold = tie_seen ? 16 : D->lma_hold - 1;
        D->warm = true;
        D->warm_rows = rows, D->warm_cols = cols, D->warm_T = T, D->warm_flags = shape_flags;
    }
    res->flipped = h_header[0] != 0.0;
    res->v[0] = h_header[1], res->v[1] = h_header[2], res->v[2] = h_header[3];
    memcpy(res->w, w, sizeof(w));
    res->k = k;
    res->d_inliers = d_final;
    res->d_inlier_idx = d_idx;
    res->d_scanline = d_ys;
    if (info) {
        info->nranks = R;
        info->rank = rank;
        info->col0 = col0;
        info->slab_cols = sc;
        info->shard_points = n;
        info->shard_inliers = m;
        info->host_syncs = D->host_syncs;
        info->collectives = D->collectives;
        info->ransac_rounds = D->ransac_rounds;
        info->path_flags = path_flags;
    }
    if (local_scan_bad || local_index_bad) {
        D->lockstep_error = true;  // (the hints stay: this rank issued every collective its peers issued)
        return local_scan_bad ? fail(c, RSDSFM_ERR_NUMERIC, "inlier count mismatch between scoring and compaction on this rank's slab")
                              : fail(c, RSDSFM_ERR_INVALID, "flow index out of range (bad inlier_idx)");
    }
    return RSDSFM_OK;
}

// ---------------------------------------------------------------------------------------------------
// the row-tiled DENSE DEPTH solve (BASELINE configs[3] as literally stated: "row-tiled ... with an all-gather of the depth map"):
// minimal::estimateInverseDepths (minimal.cc:170-306) of ONE frame whose flattened point list is sharded over the ranks by
// contiguous index ranges (a row tile of the image is a contiguous range of the column-major point list of a transposed scan; any
// contiguous range works because the per-pixel solves are independent given the pose).  Ceres-LM mode: per LM launch every rank
// reduces its shard to one row of NS sums, the rows are all-gathered in rank order and every rank takes the same accept / converge
// decision; closed-form mode: no exchange at all.  Then ONE all-gather of the inverse-depth shards.  The common case (accept the
// first step, converge) costs launch 0 -> row -> all-gather -> decide -> launch 1 -> all-gather -> ONE host synchronisation.
// ---------------------------------------------------------------------------------------------------
int rsdsfm_tiled_shard_bounds(int64_t n, int32_t nranks, int32_t rank, int64_t* i0, int64_t* count, int64_t* stride) {
    if (n < 0 || nranks < 1 || rank < 0 || rank >= nranks) return RSDSFM_ERR_INVALID;
    int64_t per = (n + nranks - 1) / nranks;
    per += per & 1;  // even shard starts keep the 8-byte-per-point arrays 16-byte aligned
    const int64_t b0 = std::min<int64_t>(n, (int64_t)rank * per), b1 = std::min<int64_t>(n, ((int64_t)rank + 1) * per);
    if (i0) *i0 = b0;
    if (count) *count = b1 - b0;
    if (stride) *stride = per;
    return RSDSFM_OK;
}

int rsdsfm_estimate_inverse_depths_tiled_dev(rsdsfm_ctx* ctx, const double* d_q_shard, const double* d_u_shard, int64_t n_total,
                                             const double v[3], const double w[3], double k, const double* d_alpha_shard,
                                             const double* d_alpha_k_shard, int depth_mode, double* d_inv_depth,
                                             rsdsfm_lm_summary* summary, rsdsfm_tiled_info* info) {
    if (!ctx) return RSDSFM_ERR_INVALID;
    Ctx* c = &ctx->c;
    DeviceGuard device_guard_(c);
    if (n_total < 0 || !v || !w) return fail(c, RSDSFM_ERR_INVALID, "bad arguments");
    if (depth_mode != RSDSFM_DEPTH_CLOSED_FORM && depth_mode != RSDSFM_DEPTH_CERES_LM) return fail(c, RSDSFM_ERR_INVALID, "unknown depth_mode");
    Dist* D = dist_of(c, true);
    if (!D) return fail(c, RSDSFM_ERR_INVALID, "out of host memory");
    D->host_syncs = D->collectives = D->ransac_rounds = 0;
    const int R = D->nranks, rank = D->rank;
    int64_t i0 = 0, ns = 0, per = 0;
    rsdsfm_tiled_shard_bounds(n_total, R, rank, &i0, &ns, &per);
    if (ns > 0 && (!d_q_shard || !d_u_shard || !d_alpha_shard || !d_alpha_k_shard)) return fail(c, RSDSFM_ERR_INVALID, "null shard pointer");
    if (n_total > 0 && !d_inv_depth) return fail(c, RSDSFM_ERR_INVALID, "null output pointer");
    if (summary) memset(summary, 0, sizeof(*summary));
    Pose pose;
    memcpy(pose.v, v, sizeof(pose.v));
    memcpy(pose.w, w, sizeof(pose.w));
    pose.k = k;
    const size_t cap = (size_t)std::max<int64_t>(per, 2);
    const bool padded = (size_t)R * cap != (size_t)n_total;
    const size_t row_max = (size_t)std::max((int)NS, depth_lma_row_doubles());
    int rc = ensure_dev(c, &D->d_buf, &D->bytes, Arena::need(8 * row_max) + Arena::need(8 * row_max * R) + (padded ? Arena::need(8 * cap * R) : 0) + 4096);
    if (rc != RSDSFM_OK) return rc;
    Arena da(D->d_buf);
    double* d_row = da.take<double>(row_max);
    double* d_rows_all = da.take<double>(row_max * R);
    double* d_gather = padded ? da.take<double>(cap * R) : d_inv_depth;
    double* d_rho = d_gather + (size_t)rank * cap;  // the shard is solved in place in the all-gather buffer
    // an empty shard still takes part in every collective; its kernels run over zero points of a valid (unused) address
    const double* q = ns > 0 ? d_q_shard : d_rho;
    const double* u = ns > 0 ? d_u_shard : d_rho;
    const double* a = ns > 0 ? d_alpha_shard : d_rho;
    const double* ak = ns > 0 ? d_alpha_k_shard : d_rho;

    int launches = 0;
    if (depth_mode == RSDSFM_DEPTH_CLOSED_FORM) {
        if (ns > 0) {
            rc = depth_closed_form_launch(c, q, u, a, ak, ns, pose, d_rho);
            if (rc != RSDSFM_OK) return rc;
        }
    } else {
        // decide launch `id` from the all-gathered rows (rank order => the same sums, bit for bit, on every rank)
        auto decide = [&](int id) -> int {
            int r2 = depth_lm_reduce_launch(c, ns, d_row);
            if (r2 != RSDSFM_OK) return r2;
            r2 = all_gather(c, D, d_row, d_rows_all, sizeof(double) * NS);
            if (r2 != RSDSFM_OK) return r2;
            return depth_lm_decide_rows_launch(c, d_rows_all, R, n_total, id);
        };
        auto read_state = [&]() -> int {
            RSDSFM_HIP_CHECK(c, hipMemcpyAsync(c->h_lm, c->d_lm, sizeof(LmState), hipMemcpyDeviceToHost, c->stream));
            return sync(c, D);
        };
        // the analytic LM trajectory (depth_lma_kernels.hip; the user's switch is the same on every rank): ONE pass over the shard, ONE exchange of
        // the closed-form row, the decision replicated -- no rounds.  A guard that trips (every rank reads the same replicated state) sends every
        // rank through the iterate-by-iterate protocol below.
        bool analytic_done = false;
        if (depth_lma_allowed(c, ns) && n_total <= (int64_t)INT32_MAX) {
            rc = depth_lma_shard_launch(c, q, u, a, ak, ns, pose, d_rho, d_row);
            if (rc != RSDSFM_OK) return rc;
            rc = all_gather(c, D, d_row, d_rows_all, sizeof(double) * (size_t)depth_lma_row_doubles());
            if (rc != RSDSFM_OK) return rc;
            rc = depth_lma_shard_finish_launch(c, q, u, a, ak, ns, pose, d_rho, d_rows_all, R, n_total);
            if (rc != RSDSFM_OK) return rc;
            launches = 2;
            rc = read_state();
            if (rc != RSDSFM_OK) return rc;
            analytic_done = c->h_lm->status == 1;
            if (!analytic_done) c->lma_restarts += 1, c->lma_last_guard = 1 << std::min(std::max(c->h_lm->iteration, 0), 15);
        }
        if (analytic_done) {
            c->lm_issued_k = c->lm_issued_d = 0;
            fill_lm_summary(*c->h_lm, summary);
        } else {
        // fast path, no host round trip: speculate (launch 0) -> decide -> launch 1 (applies / continues / nothing to do)
        rc = depth_lm_launch(c, q, u, a, ak, ns, pose, d_rho, 0);
        if (rc != RSDSFM_OK) return rc;
        rc = decide(0);
        if (rc != RSDSFM_OK) return rc;
        rc = depth_lm_launch(c, q, u, a, ak, ns, pose, d_rho, 1);
        if (rc != RSDSFM_OK) return rc;
        int issued = 2;
        launches = 2;
        rc = read_state();
        if (rc != RSDSFM_OK) return rc;
        for (;;) {
            const LmState& st = *c->h_lm;
            if (st.status == 1) break;
            if (launches > 4 * kMaxIter + 8) return fail(c, RSDSFM_ERR_NUMERIC, "LM state machine did not terminate");
            if (st.status == 2) {  // result known; the designated launch writes it (whether a rank needs it depends on its own iterate predictor)
                if (st.next_launch >= issued) {
                    rc = depth_lm_launch(c, q, u, a, ak, ns, pose, d_rho, st.next_launch);
                    if (rc != RSDSFM_OK) return rc;
                    ++launches;
                }
                break;
            }
            const int id = st.next_launch;  // status 0 on every rank alike (it depends on the sums only)
            if (id >= issued) {
                rc = depth_lm_launch(c, q, u, a, ak, ns, pose, d_rho, id);
                if (rc != RSDSFM_OK) return rc;
                issued = id + 1;
                ++launches;
            }
            rc = decide(id);
            if (rc != RSDSFM_OK) return rc;
            rc = read_state();
            if (rc != RSDSFM_OK) return rc;
        }
        c->lm_issued_k = c->lm_issued_d = 0;  // nothing for rsdsfm_depth_finish_dev to continue
        fill_lm_summary(*c->h_lm, summary);
        }
    }
    // ---- ONE all-gather of the inverse-depth shards (the data-path collective: 8 B x points) ----
    if ((size_t)ns < cap) RSDSFM_HIP_CHECK(c, hipMemsetAsync(d_rho + ns, 0, sizeof(double) * (cap - (size_t)ns), c->stream));
    rc = all_gather(c, D, d_rho, d_gather, sizeof(double) * cap);
    if (rc != RSDSFM_OK) return rc;
    if (padded && n_total > 0)
        RSDSFM_HIP_CHECK(c, hipMemcpyAsync(d_inv_depth, d_gather, sizeof(double) * (size_t)n_total, hipMemcpyDeviceToDevice, c->stream));
    if (info) {
        memset(info, 0, sizeof(*info));
        info->nranks = R;
        info->rank = rank;
        info->shard_points = ns;
        info->host_syncs = D->host_syncs;
        info->collectives = D->collectives;
        info->ransac_rounds = launches;  // depth solve: LM kernel launches of this rank
    }
    if (depth_mode == RSDSFM_DEPTH_CERES_LM && c->h_lm->termination == RSDSFM_TERM_FAILURE)
        return fail(c, RSDSFM_ERR_NUMERIC, "LM failure (5 consecutive invalid steps)");
    return RSDSFM_OK;
}

}  // extern "C"
