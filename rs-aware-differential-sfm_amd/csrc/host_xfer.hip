// host_xfer.hip -- the host-pointer boundary's transfers (rsdsfm_ransac / rsdsfm_refine / rsdsfm_estimate_inverse_depths ...: what
// minimal::ransac and nonLinearRefinement in host/*.h call, reference main.cc:447-457).
//
// The reference's API hands over pageable host arrays (Eigen objects) and wants pageable arrays back; a 1280x720 pair moves ~44 MB in and
// ~67 MB out.  Uploads: hipMemcpyAsync from pageable memory stages through the runtime's own bounce buffer on ONE thread (22-28 GB/s on the box,
// tools/xfer_probe.hip -> profiles/r06_xfer_probe.txt); here a ring of pinned chunks per process and device is filled by a small pool of host
// threads while the DMA engines move the neighbouring chunks, alternating between the context's stream and a side stream (the commands' fixed
// costs overlap); the kernels enqueued next run behind the last chunk with no host synchronisation.  xfer_h2d* return once the callers' arrays
// have been READ (the DMA may still be in flight).  Downloads: the runtime's pageable path already runs at the link's rate (see xfer_d2h_many).
// Small uploads (< 256 KB in all) take the plain path.  RSDSFM_XFER_TRACE=1 prints where a host-pointer call's time went.
#include <string.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

#include "rsdsfm_internal.hpp"

namespace rsdsfm {

namespace {

// Chunk sizes (tools/xfer_probe.hip, profiles/r06_xfer_probe.txt: one copy command costs ~13 us beyond its bytes, an event behind it ~6 us more:
// 4 MiB chunks reach 45 GB/s of the link's 56.7, 8 MiB chunks alternating between two streams 54): 8 MiB chunks in rotation behind a 2 and a
// 4 MiB chunk, so that the DMA engine has work after ~45 us.
constexpr size_t kMiB = (size_t)1 << 20;
constexpr size_t kXferBig = 8 * kMiB;
constexpr int kXferBigSlots = 4;
constexpr int kXferSlots = kXferBigSlots + 2;  // [0 .. 4): 8 MiB each; 4: 2 MiB; 5: 4 MiB
constexpr size_t kXferSlotBytes[kXferSlots] = {kXferBig, kXferBig, kXferBig, kXferBig, 2 * kMiB, 4 * kMiB};
constexpr int kSlotSmall2 = 4, kSlotSmall4 = 5;
constexpr size_t kXferSmall = (size_t)256 << 10;

// a few persistent workers that copy pieces of one chunk (a fork / join per chunk: ~10 us)
class CopyPool {
public:
    explicit CopyPool(int n) : stop_(false), pending_(0), epoch_(0) {
        for (int i = 0; i < n; ++i) workers_.emplace_back([this, i, n] { run(i, n); });
    }
    ~CopyPool() {
        {
            std::lock_guard<std::mutex> g(m_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto& t : workers_) t.join();
    }
    int size() const { return (int)workers_.size(); }
    // dst[0 .. n) = src[0 .. n), split over the workers and the calling thread
    void copy(char* dst, const char* src, size_t n) {
        const int parts = size() + 1;
        if (n < ((size_t)64 << 10) || parts == 1) {
            memcpy(dst, src, n);
            return;
        }
        const size_t per = ((n + parts - 1) / parts + 4095) & ~(size_t)4095;
        {
            std::lock_guard<std::mutex> g(m_);
            dst_ = dst, src_ = src, n_ = n, per_ = per;
            pending_ = size();
            ++epoch_;
        }
        cv_.notify_all();
        const size_t a = std::min(n, per * (size_t)size());
        if (n > a) memcpy(dst + a, src + a, n - a);  // (the caller takes the last piece)
        std::unique_lock<std::mutex> g(m_);
        done_.wait(g, [this] { return pending_ == 0; });
    }

private:
    void run(int i, int n) {
        (void)n;
        unsigned long long seen = 0;
        for (;;) {
            char* d;
            const char* s;
            size_t len, per;
            {
                std::unique_lock<std::mutex> g(m_);
                cv_.wait(g, [&] { return stop_ || epoch_ != seen; });
                if (stop_) return;
                seen = epoch_;
                d = dst_, s = src_, len = n_, per = per_;
            }
            const size_t a = std::min(len, per * (size_t)i), b = std::min(len, per * (size_t)(i + 1));
            if (b > a) memcpy(d + a, s + a, b - a);
            {
                std::lock_guard<std::mutex> g(m_);
                if (--pending_ == 0) done_.notify_all();
            }
        }
    }
    std::vector<std::thread> workers_;
    std::mutex m_;
    std::condition_variable cv_, done_;
    bool stop_;
    int pending_;
    unsigned long long epoch_;
    char* dst_ = nullptr;
    const char* src_ = nullptr;
    size_t n_ = 0, per_ = 0;
};

// one ring per process and device (transfers of different contexts on a device serialise on it: they share the link anyway)
struct XferRing {
    std::mutex m;
    char* pinned[kXferSlots] = {};
    hipEvent_t ev[kXferSlots] = {};
    bool busy[kXferSlots] = {};
    hipStream_t side = nullptr;  // uploads alternate between the context's stream and this one: the commands' fixed costs overlap (54 vs 49 GB/s)
    hipEvent_t ev_order = nullptr;
    bool ok = false, tried = false;
    CopyPool* pool = nullptr;
};
XferRing g_ring[64];

// one piece of one array, in one pinned slot
struct Piece {
    int item;
    size_t off, len;
    int slot;
};

int xfer_threads() {
    static int n = -1;
    if (n < 0) {
        const char* e = getenv("RSDSFM_XFER_THREADS");  // helper threads beside the caller's (0: the caller's thread alone)
        const unsigned hw = std::thread::hardware_concurrency();
        // (3 helpers + the caller fill the ring faster than the link empties it; 2 ... 9 measured alike: tools/host_boundary_probe.py)
        n = e ? std::max(0, std::min(atoi(e), 32)) : (int)std::max(0u, std::min(3u, hw > 2 ? hw / 2 - 1 : 0u));
    }
    return n;
}

XferRing* ring_of(Ctx* c) {
    XferRing* R = &g_ring[c->device & 63];
    std::lock_guard<std::mutex> g(R->m);  // (contexts of several threads may meet here for the first time)
    if (!R->tried) {
        R->tried = true;
        bool good = true;
        for (int s = 0; s < kXferSlots && good; ++s)
            good = hipHostMalloc((void**)&R->pinned[s], kXferSlotBytes[s], hipHostMallocDefault) == hipSuccess && hipEventCreateWithFlags(&R->ev[s], hipEventDisableTiming) == hipSuccess;
        good = good && hipStreamCreateWithFlags(&R->side, hipStreamNonBlocking) == hipSuccess && hipEventCreateWithFlags(&R->ev_order, hipEventDisableTiming) == hipSuccess;
        if (good) R->pool = new (std::nothrow) CopyPool(xfer_threads());
        R->ok = good && R->pool;
    }
    return R->ok ? R : nullptr;
}

// the pieces of an upload of `count` arrays of the given sizes, in transfer order: the first two pieces are 2 and 4 MiB, everything else 8 MiB,
// slots in rotation.  A piece never spans two arrays.
void plan_pieces(const size_t* bytes, int count, std::vector<Piece>* plan) {
    plan->clear();
    int big = 0, g = 0;
    for (int i = 0; i < count; ++i) {
        size_t off = 0, left = bytes[i];
        while (left) {
            size_t len;
            int slot;
            if (g == 0)
                len = std::min(left, 2 * kMiB), slot = kSlotSmall2;
            else if (g == 1)
                len = std::min(left, 4 * kMiB), slot = kSlotSmall4;
            else
                len = std::min(left, kXferBig), slot = big, big = (big + 1) % kXferBigSlots;
            plan->push_back({i, off, len, slot});
            off += len, left -= len, ++g;
        }
    }
}

// RSDSFM_XFER_TRACE=1: where a host-pointer call's time goes (labels + microseconds to stderr at the end of the call; tools/host_boundary_probe.py).
// A diagnostic for ONE calling thread: the record buffer is not locked.
struct TraceRec {
    const char* label;
    double us;
    bool span;
};
TraceRec g_trace[256];
int g_ntrace = 0;
double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
struct Span {  // adds its lifetime to *acc when tracing
    double* acc;
    double t0;
    explicit Span(double* a) : acc(a), t0(a ? now_us() : 0.0) {}
    ~Span() {
        if (acc) *acc += now_us() - t0;
    }
};

}  // namespace

bool xfer_trace_on() {
    static int on = -1;
    if (on < 0) {
        const char* e = getenv("RSDSFM_XFER_TRACE");
        on = e && atoi(e) > 0 ? 1 : 0;
    }
    return on == 1;
}
void xfer_trace(const char* label) {
    if (xfer_trace_on() && g_ntrace < 256) g_trace[g_ntrace++] = {label, now_us(), false};
}
void xfer_trace_span(const char* label, double us) {  // (a duration, not a time stamp)
    if (xfer_trace_on() && g_ntrace < 256) g_trace[g_ntrace++] = {label, us, true};
}
void xfer_trace_dump(const char* what) {
    if (!xfer_trace_on()) return;
    double first = 0.0, prev = 0.0;
    fprintf(stderr, "[xfer] %s:", what);
    for (int i = 0; i < g_ntrace; ++i) {
        if (g_trace[i].span) {
            fprintf(stderr, " (%s %.0f)", g_trace[i].label, g_trace[i].us);
            continue;
        }
        if (first == 0.0) first = prev = g_trace[i].us;
        fprintf(stderr, " %s +%.0f", g_trace[i].label, g_trace[i].us - prev);
        prev = g_trace[i].us;
    }
    fprintf(stderr, " | total %.0f us\n", prev - first);
    g_ntrace = 0;
}

// host -> device behind what the context's stream holds; returns once the callers' arrays have been READ (the kernels enqueued next on the
// context's stream run behind the last piece)
int xfer_h2d_many(Ctx* c, const XferUp* items, int count) {
    size_t bytes[16], total = 0;
    if (count > 16) return fail(c, RSDSFM_ERR_INVALID, "xfer_h2d_many: too many arrays");
    for (int i = 0; i < count; ++i) total += bytes[i] = items[i].bytes;
    if (total == 0) return RSDSFM_OK;
    XferRing* R = total >= kXferSmall ? ring_of(c) : nullptr;
    if (!R) {
        for (int i = 0; i < count; ++i)
            if (bytes[i]) RSDSFM_HIP_CHECK(c, hipMemcpyAsync(items[i].dev, items[i].host, bytes[i], hipMemcpyHostToDevice, c->stream));
        return RSDSFM_OK;
    }
    std::lock_guard<std::mutex> g(R->m);
    const bool tr = xfer_trace_on();
    double t_wait = 0.0, t_copy = 0.0, t_enq = 0.0;
    std::vector<Piece> plan;
    plan_pieces(bytes, count, &plan);
    // the side stream starts behind what the context's stream holds now (an earlier call's kernels may still read the destination)
    const bool two = plan.size() > 1;
    if (two) {
        RSDSFM_HIP_CHECK(c, hipEventRecord(R->ev_order, c->stream));
        RSDSFM_HIP_CHECK(c, hipStreamWaitEvent(R->side, R->ev_order, 0));
    }
    int last_side_slot = -1;
    for (size_t k = 0; k < plan.size(); ++k) {
        const Piece& p = plan[k];
        if (R->busy[p.slot]) {
            Span sp(tr ? &t_wait : nullptr);
            RSDSFM_HIP_CHECK(c, hipEventSynchronize(R->ev[p.slot]));  // (the DMA that last read this slot)
        }
        {
            Span sp(tr ? &t_copy : nullptr);
            R->pool->copy(R->pinned[p.slot], static_cast<const char*>(items[p.item].host) + p.off, p.len);
        }
        Span sp(tr ? &t_enq : nullptr);
        hipStream_t st = (k & 1) ? R->side : c->stream;
        RSDSFM_HIP_CHECK(c, hipMemcpyAsync(static_cast<char*>(items[p.item].dev) + p.off, R->pinned[p.slot], p.len, hipMemcpyHostToDevice, st));
        RSDSFM_HIP_CHECK(c, hipEventRecord(R->ev[p.slot], st));
        R->busy[p.slot] = true;
        if (k & 1) last_side_slot = p.slot;
    }
    if (last_side_slot >= 0) RSDSFM_HIP_CHECK(c, hipStreamWaitEvent(c->stream, R->ev[last_side_slot], 0));
    if (tr) xfer_trace_span("h2d wait", t_wait), xfer_trace_span("copy", t_copy), xfer_trace_span("enqueue", t_enq);
    return RSDSFM_OK;
}

int xfer_h2d(Ctx* c, void* d_dst, const void* h_src, size_t bytes) {
    const XferUp one = {d_dst, h_src, bytes};
    return xfer_h2d_many(c, &one, 1);
}

// several device -> host copies behind everything enqueued on the context's stream; returns once the callers' arrays hold the data.  The
// runtime's own path for pageable destinations moves warm arrays at the link's rate (55 GB/s: it pins the caller's pages and lets the engine
// write them directly); draining the ring with the thread pool was measured beside it and is slower (40 GB/s: every piece's drain starts behind
// its own DMA, HISTORY.md), so downloads do not go through the ring.
int xfer_d2h_many(Ctx* c, const XferItem* items, int count) {
    bool any = false;
    for (int i = 0; i < count; ++i)
        if (items[i].bytes) {
            RSDSFM_HIP_CHECK(c, hipMemcpyAsync(items[i].host, items[i].dev, items[i].bytes, hipMemcpyDeviceToHost, c->stream));
            any = true;
        }
    if (any) RSDSFM_HIP_CHECK(c, hipStreamSynchronize(c->stream));
    return RSDSFM_OK;
}

int xfer_d2h(Ctx* c, void* h_dst, const void* d_src, size_t bytes) {
    const XferItem one = {h_dst, d_src, bytes};
    return xfer_d2h_many(c, &one, 1);
}

}  // namespace rsdsfm
